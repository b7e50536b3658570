#!/usr/bin/env python3
"""Development: the 77-key cross-attention launches of the SDXL stack (14 frames over 6 shared contexts) on the short-stream ping-pong
kernel (padded keys / values) against the program-order kernel (ATTN_V2 = 0), interleaved rounds in one process."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import aid_amd  # noqa: E402
from aid_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ROUNDS, ITERS = 7, 20
for s, h in ((1024, 20), (4096, 10)):
    n, l, dt = 7, 77, torch.bfloat16
    c = h * 64
    g = torch.Generator(device=dev).manual_seed(s)
    q = (torch.randn(2 * n, s, c, device=dev, generator=g) * 0.6).to(dt)
    k = torch.zeros(6, 128, c, device=dev, dtype=dt)
    k[:, :l] = torch.randn(6, l, c, device=dev, generator=g).to(dt)
    vt = torch.zeros(6, c, 128, device=dev, dtype=dt)
    vt[:, :, :l] = torch.randn(6, c, l, device=dev, generator=g).to(dt)
    kv_map = torch.tensor([0, 1, 1, 1, 1, 1, 2, 3, 4, 4, 4, 4, 4, 5], dtype=torch.int32, device=dev)
    cf = aid_amd.generate_beta_tensor(n, 50, 50)
    cf[0], cf[-1] = 0, 1
    coef = torch.tensor(cf.to(dt).float().tolist() + [-1.0] * n, device=dev)
    out = torch.empty_like(q)
    for mode in ("outer", "plain"):
        kw = dict(l=l, mode=mode, kv_map=kv_map, out=out, kv_padded=True, q_prescaled=True)
        if mode == "outer":
            kw.update(fused=True, coef=coef, begin=0, end=2, n_plain=n)
        res = {0: [], 1: []}
        names = {}
        for v2 in (0, -1):
            ops.set_tuning("ATTN_V2", v2)
            ops.attn_fwd(q, k, vt, h, **kw); ops.attn_fwd(q, k, vt, h, **kw)
            names[v2] = ops.last_attn_variant()
        torch.cuda.synchronize()
        for r in range(ROUNDS):
            for i, v2 in enumerate((0, -1)):
                ops.set_tuning("ATTN_V2", v2)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(ITERS):
                    ops.attn_fwd(q, k, vt, h, **kw)
                e1.record()
                torch.cuda.synchronize()
                res[i].append(e0.elapsed_time(e1) * 1e3 / ITERS)
        for i, v2 in enumerate((0, -1)):
            print(f"S{s} H{h} x77 {mode:6s} {names[v2]:34s} {statistics.median(res[i]):7.1f} us (min {min(res[i]):7.1f})", flush=True)
ops.set_tuning("ATTN_V2", -1)
print("done")
