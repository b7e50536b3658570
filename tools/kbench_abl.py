#!/usr/bin/env python3
"""Timing of two attention shapes (used with AID_LIB_PATH=tools/ablate/libaid_ablN.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aid_amd
from aid_amd import ops
dev = torch.device("cuda:0"); lib = aid_amd._lib.load()
def timed(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize(); lib.aid_profile_begin()
    for _ in range(iters): fn()
    buf = (aid_amd._lib.AidProfileEntry * 256)(); n = lib.aid_profile_end(buf, 256)
    return sum(e.ms for e in buf[:n]) / n * 1e3
for tag, dt, h, d in (("sdxl d64", torch.bfloat16, 10, 64), ("sd15 d40", torch.float16, 8, 40)):
    n, s, c = 7, 4096, h * d
    q = torch.randn(n, s, c, device=dev).to(dt); k = torch.randn(n, s, c, device=dev).to(dt); vt = torch.randn(n, c, s, device=dev).to(dt)
    out = torch.empty_like(q)
    print(os.environ.get("AID_LIB_PATH", "product")[-12:], tag, "plain S4096: %.1f us" % timed(lambda: ops.attn_fwd(q, k, vt, h, l=s, mode="plain", out=out)))
