#!/usr/bin/env python3
"""Per-launch-shape breakdown of one AID step + one plain step of the attention stack (HIP-event timing)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aid_amd
from aid_amd.loop import AidDenoiseLoop, install_sequence_processors
import bench
model = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
separate = len(sys.argv) > 2 and sys.argv[2] == "separate"
dev = torch.device("cuda:0"); dtype = bench.DTYPES[model]; early = bench.DEFAULT_EARLY[model]
unet = aid_amd.AttnStackUNet(model, dtype=dtype, device=dev)
xs, cond, uncond = bench.make_inputs(unet, 7, dtype, dev)
install_sequence_processors(unet, 7, early=early, num_inference_steps=50)
loop = AidDenoiseLoop(unet, xs, cond, uncond, num_inference_steps=50, use_graphs=False, batched_cfg=not separate)
lib = aid_amd._lib.load()
loop.step(0); loop.step(49); torch.cuda.synchronize()
lib.aid_profile_begin(); loop.step(0); loop.step(49)
buf = (aid_amd._lib.AidProfileEntry * 8192)(); n = lib.aid_profile_end(buf, 8192)
agg = collections.OrderedDict()
for e in buf[:n]:
    key = (e.kernel.decode(), round(e.flops / 1e9, 2))
    a = agg.setdefault(key, [0.0, 0]); a[0] += e.ms; a[1] += 1
tot = sum(v[0] for v in agg.values())
print(f"total kernel ms (2 steps): {tot:.2f}")
for (k, gf), (ms, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:34s} {gf:10.2f} GF/launch  x{cnt:4d}  {ms / cnt * 1e3:8.1f} us  {gf / (ms / cnt):8.1f} TF/s  {ms:7.2f} ms {100 * ms / tot:5.1f}%")
