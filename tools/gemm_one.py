#!/usr/bin/env python3
"""Run ONE GEMM shape a few times (target for rocprofv3 --pmc).  usage: gemm_one.py M N K [iters] [dtype]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aid_amd
from aid_amd import ops
m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dt = torch.float16 if (len(sys.argv) > 5 and sys.argv[5] == "f16") else torch.bfloat16
dev = torch.device("cuda:0")
a = torch.randn(m, k, device=dev).to(dt); b = torch.randn(n, k, device=dev).to(dt)
out = torch.empty(m, n, device=dev, dtype=dt)
for _ in range(iters):
    ops.linear(a, b, None, out=out)
torch.cuda.synchronize()
print("ok")
