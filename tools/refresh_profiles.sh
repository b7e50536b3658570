# Re-create the round's measurement set on the GPU box (run from the repo root under gpurun); results land in gpurun_out/r02/
# and are copied to profiles/r02_* by hand after review:  bench lines, rocprofv3 kernel-trace summaries of the SAME commands,
# the PMC collection (tools/pmc_collect.py).
set -x
R=gpurun_out/r02; mkdir -p $R
timeout 900 python bench.py --steps 20 --warmup 5 > $R/bench_default.json 2> $R/bench_default.err
timeout 400 python bench.py --workload ip --steps 20 --warmup 5 > $R/bench_ip.json 2> $R/bench_ip.err
cd /tmp && export TMPDIR=/tmp
for w in sdxl sd15; do
  rm -rf /tmp/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-also --min-seconds 0 > $GRAFT_REPO_ROOT/$R/${w}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$R/${w}_rocprof.err
  db=$(find /tmp/prof_$w -name '*.db' | head -1)
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db $GRAFT_REPO_ROOT/$R/${w}_kernel_stats.txt > /dev/null
done
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/pmc_collect.py $R/pmc.json > $R/pmc.log 2>&1
tail -5 $R/pmc.log
