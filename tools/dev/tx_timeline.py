#!/usr/bin/env python3
"""Phase timeline of the text-key attention kernel (development build with -DAID_TX_ABL=4): shader-clock stamps of wave 0 of the middle
workgroup — kernel entry, fill done, then per 32-row tile: start (next Q requested), every segment done, output words ready, next Q
landed, stores issued.  usage: make -C tools/dev libaid_tx_abl4.so && AID_LIB_PATH=tools/dev/libaid_tx_abl4.so python tools/dev/tx_timeline.py [plain|outer] [S] [tiles per wave]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import aid_amd
from aid_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
s = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
h, l, n = 20 if s == 1024 else 10, 77, 7
c = h * 64
q = torch.randn(2 * n, s, c, device=dev).to(torch.bfloat16)
k = torch.randn(2 * n, l, c, device=dev).to(torch.bfloat16)
vt = torch.randn(2 * n, c, 80, device=dev).to(torch.bfloat16)
cf = aid_amd.generate_beta_tensor(n, 50, 50)
cf[0], cf[-1] = 0, 1
coef = torch.tensor(cf.to(torch.bfloat16).float().tolist() + [-1.0] * n, device=dev)
out = torch.empty_like(q)
ops.set_tuning("ATTN_TX_TILES", tiles)
kw = dict(l=l, mode=mode, out=out)
if mode != "plain":
    kw.update(fused=True, coef=coef, begin=0, end=n - 1, n_plain=n)
for _ in range(3):
    ops.attn_fwd(q, k, vt, h, **kw)
torch.cuda.synchronize()
t = out.view(-1)[:41 * 4].view(torch.int64).cpu().tolist()
cnt, ts = t[0], t[1:1 + t[0]]
print(ops.last_attn_variant(), "stamps", cnt)
t0 = ts[0]
prev = t0
for i, x in enumerate(ts):
    print(f"  {i:2d}  +{x - prev:7d}   at {x - t0:8d}")
    prev = x
