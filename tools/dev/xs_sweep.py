#!/usr/bin/env python3
"""Development: time of the short-stream kernel over key counts (tiles per item) and item counts — separates the per-item from the
per-tile cost.  PLAIN, 14 frames, bf16."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import aid_amd  # noqa: E402
from aid_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
dt = torch.bfloat16
for s, h in ((1024, 20), (4096, 10), (2048, 20)):
    for l in (64, 77, 128, 192, 256):
        n, c = 14, h * 64
        lt = (l + 63) // 64 * 64
        q = torch.randn(n, s, c, device=dev).to(dt)
        k = torch.zeros(n, lt, c, device=dev, dtype=dt); k[:, :l] = torch.randn(n, l, c, device=dev).to(dt)
        vt = torch.zeros(n, c, lt, device=dev, dtype=dt); vt[:, :, :l] = torch.randn(n, c, l, device=dev).to(dt)
        out = torch.empty_like(q)
        kw = dict(l=l, mode="plain", out=out, kv_padded=True, q_prescaled=True)
        for _ in range(3): ops.attn_fwd(q, k, vt, h, **kw)
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.attn_fwd(q, k, vt, h, **kw)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        items = n * h * ((s + 255) // 256)
        print(f"S{s} H{h} L{l:3d} tiles/item {lt // 64}  items/WG {items / 256:5.2f}  {ops.last_attn_variant():18s} {statistics.median(ts):7.1f} us", flush=True)
