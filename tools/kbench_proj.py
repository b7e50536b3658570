#!/usr/bin/env python3
"""Grouped q/k/V^T projection launch and out-projection launch at every SD1.5 / SDXL level (batched CFG: 14 frames)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aid_amd
from aid_amd import ops
dev = torch.device("cuda:0"); lib = aid_amd._lib.load()
def timed(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize(); lib.aid_profile_begin()
    for _ in range(iters): fn()
    buf = (aid_amd._lib.AidProfileEntry * 512)(); n = lib.aid_profile_end(buf, 512)
    return sum(e.ms for e in buf[:n]) / n * 1e3, sum(e.flops for e in buf[:n]) / n
for tag, dt, s, c in (("sd15 L0", torch.float16, 4096, 320), ("sd15 L1", torch.float16, 1024, 640), ("sd15 L2", torch.float16, 256, 1280),
                      ("sdxl L1", torch.bfloat16, 4096, 640), ("sdxl L2", torch.bfloat16, 1024, 1280)):
    n = 14
    x = torch.randn(n, s, c, device=dev).to(dt)
    wq, wk, wv, wo = (torch.randn(c, c, device=dev).to(dt) for _ in range(4)); bo = torch.randn(c, device=dev).to(dt)
    q = torch.empty_like(x); k = torch.empty_like(x); vt = torch.empty(n, c, s, device=dev, dtype=dt); y = torch.empty_like(x)
    def qkv():
        ops.gemm_nt([dict(a=x, b=wq, c=q, m=n * s, n=c, k=c, lda=c, ldb=c, ldc=c),
                     dict(a=x, b=wk, c=k, m=n * s, n=c, k=c, lda=c, ldb=c, ldc=c),
                     dict(a=x, b=wv, c=vt, m=n * s, n=c, k=c, lda=c, ldb=c, ldc=s, stride_c=c * s, trans_rows=s)])      # as aid_processor_fwd issues it
    us, fl = timed(qkv); print(f"{tag} qkv  M={n*s} C={c}: {us:7.1f} us {fl/us/1e6:7.1f} TF/s  ideal-HBM {(4*n*s*c*2 + 3*c*c*2)/5e6:6.1f} us")
    us, fl = timed(lambda: ops.linear(x, wo, bo, out=y)); print(f"{tag} out  M={n*s} C={c}: {us:7.1f} us {fl/us/1e6:7.1f} TF/s  ideal-HBM {(2*n*s*c*2)/5e6:6.1f} us")

# cross-attention projections (text context L = 77): one grouped launch vs q alone + k / V^T in a second launch
for tag, dt, s, c, cc in (("sd15 L0", torch.float16, 4096, 320, 768), ("sd15 L1", torch.float16, 1024, 640, 768), ("sd15 L2", torch.float16, 256, 1280, 768),
                          ("sdxl L1", torch.bfloat16, 4096, 640, 2048), ("sdxl L2", torch.bfloat16, 1024, 1280, 2048)):
    n, l, lp = 14, 77, 80
    for nctx in (14, 6):
        x = torch.randn(n, s, c, device=dev).to(dt); e = torch.randn(nctx, l, cc, device=dev).to(dt)
        wq = torch.randn(c, c, device=dev).to(dt); wk, wv = (torch.randn(c, cc, device=dev).to(dt) for _ in range(2))
        q = torch.empty_like(x); k = torch.empty(nctx, l, c, device=dev, dtype=dt); vt = torch.empty(nctx, c, lp, device=dev, dtype=dt)
        pq = dict(a=x, b=wq, c=q, m=n * s, n=c, k=c, lda=c, ldb=c, ldc=c)
        pk = dict(a=e, b=wk, c=k, m=nctx * l, n=c, k=cc, lda=cc, ldb=cc, ldc=c)
        pv = dict(a=wv, b=e, c=vt, m=c, n=l, k=cc, lda=cc, ldb=cc, ldc=lp, batch=nctx, stride_a=0, stride_b=l * cc, stride_c=c * lp)
        u1, _ = timed(lambda: ops.gemm_nt([pq, pk, pv])); v1 = ops.last_gemm_variant()
        u2, _ = timed(lambda: ops.gemm_nt([pq])); v2 = ops.last_gemm_variant()
        u3, _ = timed(lambda: ops.gemm_nt([pk, pv])); v3 = ops.last_gemm_variant()
        print(f"{tag} cross qkv nctx={nctx:2d}: grouped {u1:6.1f} us ({v1})   q alone {u2:6.1f} us ({v2})   k+vt alone {u3:6.1f} us ({v3})")
