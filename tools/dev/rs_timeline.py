#!/usr/bin/env python3
"""Phase timeline of the row-stationary GEMM kernel (development build libaid_rsvar.so, GEMM_PP=3 = VAR 9): shader-clock stamps of
workgroup 0, waves 0 (early) and 4 (late), per slice step.  usage: make -C tools/dev libaid_rsvar.so && AID_LIB_PATH=tools/dev/libaid_rsvar.so python tools/dev/rs_timeline.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import aid_amd
from aid_amd import ops
dev = torch.device("cuda:0")
m, n, k = 57344, 640, 640
a = torch.randn(m, k, device=dev).to(torch.bfloat16)
w = torch.randn(n, k, device=dev).to(torch.bfloat16)
c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
ops.set_tuning("GEMM_RS", 1)
for _ in range(3):
    ops.set_tuning("GEMM_PP", 0); ops.gemm_nt([dict(a=a, b=w, c=c, m=m, n=n, k=k, lda=k, ldb=k, ldc=n)])
ops.set_tuning("GEMM_PP", 3)
ops.gemm_nt([dict(a=a, b=w, c=c, m=m, n=n, k=k, lda=k, ldb=k, ldc=n)])
torch.cuda.synchronize()
t = c.view(-1)[:2 * 6 * 64 * 4].view(torch.int64).cpu().view(2, 64, 6)[:, :20]
t0 = int(t[:, 0, 0].min())
names = ["start", "waited", "barrier", "mfma0", "mfma1", "end"]
for wv, nm in ((0, "early wave 0"), (1, "late wave 4")):
    print(nm, "(cycles since the first stamp; per step: wait, barrier, flush+epi+dma, MFMAs, epilogue)")
    for v in range(20):
        r = [int(x) - t0 for x in t[wv, v]]
        d = [r[i + 1] - r[i] for i in range(5)]
        print(f"  step {v:2d}  start {r[0]:7d}   " + "  ".join(f"{x:6d}" for x in d) + f"   | step total {r[5] - r[0]:6d}")
