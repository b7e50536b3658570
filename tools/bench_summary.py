#!/usr/bin/env python3
"""Print the headline numbers and the per-kernel table of a bench.py JSON line (file argument or stdin)."""
import json
import sys
d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
print(f"{d['value']:.3f} frames/s  {d['ms_per_step']:.2f} ms/step  ({d['config']['workload'][:40]})")
r = d.get("roofline")
if r:
    print(f"dominant {r['kernel']}: frac {r['frac']:.3f}  avg {r['avg_launch_us']:.1f} us  share {r['share_of_kernel_time']:.2f}  "
          f"stack {r['stack_tflops']:.0f} TF  kernels {r['kernel_ms_per_2steps']:.1f} ms / 2 steps")
    pl = r.get("power_limited")
    if pl:
        print(f"   power-limited dense MFMA rate of this box on random operands: {pl['peak_on_random_operands']:.0f} TF/s at {pl['clock_ghz']:.2f} GHz "
              f"(zeros: {pl['peak_on_zeros']:.0f} at {pl['clock_on_zeros_ghz']:.2f}) -> dominant kernel {pl['frac']:.3f}, stack {pl['stack_frac']:.3f} of it")
    for k, v in r["kernels"].items():
        print(f"   {k:36s} {v['ms']:8.3f} ms {v['launches']:4d} launches  {v['tflops']:7.1f} TF  (executed {v['tflops_executed']:7.1f})")
for k, v in (d.get("also") or {}).items():
    print(f"also {k}: {v['value']:.3f} frames/s  {v['ms_per_step']:.2f} ms/step")
if "cpu_baseline" in d:
    print("cpu_baseline", d["cpu_baseline"]["value"], "cores", d["cpu_baseline"]["cores"])
