#!/usr/bin/env python3
"""Run ONE attention-kernel configuration a few times (target for rocprofv3 --pmc / --kernel-trace).
usage: attn_one.py model(sd15|sdxl) S L mode(plain|inner|outer) fused(0|1) [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aid_amd
from aid_amd import ops
model, s, l, mode, fused = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda:0")
dt, h, d = (torch.float16, 8, 40) if model == "sd15" else (torch.bfloat16, 10, 64)
if len(sys.argv) > 7:
    h, d = int(sys.argv[7]), int(sys.argv[8])
n, c = 7, h * d
torch.manual_seed(0)
q = torch.randn(n, s, c, device=dev).to(dt); k = torch.randn(n, l, c, device=dev).to(dt)
vt = torch.randn(n, c, (l + 7) // 8 * 8, device=dev).to(dt)
coef = aid_amd.generate_beta_tensor(n, 50, 50).to(dev); coef[0] = 0; coef[-1] = 1
out = torch.empty_like(q)
for _ in range(iters):
    ops.attn_fwd(q, k, vt, h, l=l, mode=mode, fused=bool(fused), coef=coef, out=out)
torch.cuda.synchronize()
print("ok", ops.last_attn_variant())
