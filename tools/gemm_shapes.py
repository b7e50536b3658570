#!/usr/bin/env python3
"""Development only: time aid_gemm_nt on a list of shapes (HIP events).  AID_GEMM_VARIANT picks the tile config."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aid_amd
from aid_amd import ops
dev = torch.device("cuda:0")
shapes = [(300, 200, 64), (256, 256, 128), (1000, 520, 1280), (8192, 8192, 8192), (4096, 4096, 4096), (14336, 3840, 1280), (14336, 1280, 1280), (57344, 1920, 640),
          (57344, 640, 640), (57344, 960, 320), (14336, 640, 640), (3584, 1280, 1280)]
if os.environ.get("AID_SHAPES") == "short":
    shapes = [(70000, 520, 128), (8192, 8192, 8192), (4096, 4096, 4096), (14336, 3840, 1280), (14336, 1280, 1280)]
if os.environ.get("AID_SHAPES") == "small":
    shapes = [(57344, 320, 320), (14336, 640, 640), (3584, 1280, 1280), (3584, 3840, 1280), (14336, 1920, 640), (1078, 1280, 2048)]
if os.environ.get("AID_SHAPES") == "ksweep":
    shapes = [(4096, 4096, k) for k in (128, 640, 1280, 2560, 5120)] + [(14336, 1280, 1280), (14336, 3840, 1280)]
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.bfloat16
tag = os.environ.get("AID_GEMM_VARIANT", "default") + os.path.basename(os.environ.get("AID_LIB_PATH", ""))[6:-3][:6]
for m, n, k in shapes:
    a = torch.randn(m, k, device=dev).to(dt)
    b = torch.randn(n, k, device=dev).to(dt)
    out = torch.empty(m, n, device=dev, dtype=dt)
    for _ in range(5):
        ops.linear(a, b, None, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 30
    e0.record()
    for _ in range(it):
        ops.linear(a, b, None, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    out.zero_()
    ops.linear(a, b, None, out=out)
    ref = a.float() @ b.float().t()
    err = ((out.float() - ref).norm() / ref.norm()).item()
    mx = ((out.float() - ref).abs().max() / ref.abs().max()).item()
    err = max(err, mx)
    if tag == "torch":
        e0.record()
        for _ in range(it):
            c = a @ b.t()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / it * 1e3
    print(f"v{tag:8s} m{m:6d} n{n:5d} k{k:5d}: {us:8.1f} us  {2.0*m*n*k/us/1e6:7.1f} TF/s  rel {err:.1e}", flush=True)
