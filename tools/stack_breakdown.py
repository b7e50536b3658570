#!/usr/bin/env python3
"""Per-launch-shape breakdown of one AID step + one plain step of a bench.py workload (HIP-event timing).
usage: python tools/stack_breakdown.py [sdxl|sd15|ip|seq16] [extra bench.py flags ...]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aid_amd
import bench
name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "sdxl"
sys.argv = [sys.argv[0], "--no-graph"] + [a for a in sys.argv[2:]]
if "--passes" not in sys.argv and name in ("sd15", "ip"):
    sys.argv += ["--passes", "serial"]            # the launches of the default (two-stream) form, each timed alone on the device
args = bench.parse()
dev = torch.device("cuda:0")
wl = bench.build_workload(name, args, 1, 0, dev, torch, aid_amd)
loop = wl["loop"]
lib = aid_amd._lib.load()
last = loop.num_inference_steps - 1
loop.step(0); loop.step(last); torch.cuda.synchronize()
lib.aid_profile_begin(); loop.step(0); loop.step(last)
buf = (aid_amd._lib.AidProfileEntry * 8192)(); n = lib.aid_profile_end(buf, 8192)
agg = collections.OrderedDict()
for e in buf[:n]:
    key = (e.kernel.decode(), round(e.flops / 1e9, 2), round(e.flops_executed / 1e9, 2))
    a = agg.setdefault(key, [0.0, 0]); a[0] += e.ms; a[1] += 1
tot = sum(v[0] for v in agg.values())
print(f"{wl['what']}\ntotal kernel ms (1 AID step + 1 plain step): {tot:.2f}")
for (k, gf, gfx), (ms, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:36s} {gf:9.2f} GF ({gfx:9.2f} exec) x{cnt:4d}  {ms / cnt * 1e3:8.1f} us  {gf / (ms / cnt):7.1f} / {gfx / (ms / cnt):7.1f} TF/s  {ms:7.2f} ms {100 * ms / tot:5.1f}%")
