#!/bin/bash
# development: same-box A/B of bench.py under environment settings.   usage: ab_bench.sh model "ENV=.." "ENV=.." ...
m=$1; shift
for rep in 1 2; do
for e in "$@"; do
  env $e timeout 300 python bench.py --model $m --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernels']
        print('$m', '$e', 'frames/s %.3f  ms/step %.2f ' % (d['value'], d['ms_per_step']), {n.split('<')[0]+n[-12:]:v['ms'] for n,v in k.items() if v['ms']>1})
"
done; done
