#!/usr/bin/env python3
"""Development only: hipBLASLt (torch.matmul) timing on the projection shapes, as a yardstick for aid_gemm_nt."""
import torch
dev = torch.device("cuda:0")
shapes = [(8192, 8192, 8192), (57344, 640, 640), (57344, 1920, 640), (14336, 1280, 1280), (14336, 3840, 1280),
          (57344, 320, 320), (57344, 960, 320), (14336, 640, 640), (3584, 1280, 1280)]
for dt in (torch.bfloat16, torch.float16):
    for m, n, k in shapes:
        a = torch.randn(m, k, device=dev, dtype=dt)
        b = torch.randn(n, k, device=dev, dtype=dt)
        for _ in range(5):
            c = a @ b.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 30
        e0.record()
        for _ in range(it):
            c = a @ b.t()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / it * 1e3
        print(f"{str(dt)[6:]:9s} m{m:6d} n{n:5d} k{k:5d}: {us:8.1f} us  {2.0*m*n*k/us/1e6:7.1f} TF/s", flush=True)
