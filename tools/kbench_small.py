#!/usr/bin/env python3
"""Small-image attention shapes (SD1.5 S=256 / S=64, d=160) at N=7 and N=14 frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aid_amd
from aid_amd import ops
sys.argv = [sys.argv[0]]
dev = torch.device("cuda:0")
lib = aid_amd._lib.load()
def timed(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    lib.aid_profile_begin()
    for _ in range(iters): fn()
    buf = (aid_amd._lib.AidProfileEntry * 4096)(); n = lib.aid_profile_end(buf, 4096)
    agg = {}
    for e in buf[:n]:
        a = agg.setdefault(e.kernel.decode(), [0.0, 0.0, 0]); a[0] += e.ms; a[1] += e.flops; a[2] += 1
    return agg
for n in (7, 14):
    for (s, l, h, d) in ((256, 256, 8, 160), (64, 64, 8, 160), (256, 77, 8, 160), (1024, 1024, 8, 80)):
        dt = torch.float16; c = h * d
        q = torch.randn(n, s, c, device=dev).to(dt); k = torch.randn(n, l, c, device=dev).to(dt)
        vt = torch.randn(n, c, (l + 7) // 8 * 8, device=dev).to(dt); out = torch.empty_like(q)
        coef = torch.rand(n, device=dev); coef[0] = 0; coef[-1] = 1
        for mode, fused in (("plain", False), ("inner", True)):
            for k_, (ms, fl, cnt) in timed(lambda: ops.attn_fwd(q, k, vt, h, l=l, mode=mode, fused=fused, coef=coef, out=out)).items():
                if "attn" in k_: print(f"N{n} S{s} L{l} d{d} {mode:6s} {k_:32s} {ms/cnt*1e3:8.1f} us {fl/ms/1e9:8.1f} TF/s", flush=True)
