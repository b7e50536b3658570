"""Development: workgroup timeline of the ping-pong attention kernel (ablation build, bit 32): shader cycles from kernel entry to
[Q requested + DMA issued | first three tiles landed | first product done, loop starts | loop done]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import aid_amd
from aid_amd import ops
dev = torch.device("cuda:0")
for s_, h in ((1024, 20), (4096, 10)):
    n, d = 14, 64
    c = h * d
    q = torch.randn(n, s_, c, device=dev).to(torch.bfloat16); k = torch.randn(n, s_, c, device=dev).to(torch.bfloat16)
    vt = torch.randn(n, c, s_, device=dev).to(torch.bfloat16); out = torch.zeros_like(q)
    ops.set_tuning("ATTN_V2", 1); ops.set_tuning("ATTN_RES_CHUNKS", 132)
    for rep in range(2):
        out.zero_(); ops.attn_fwd(q, k, vt, h, l=s_, mode="plain", out=out); torch.cuda.synchronize()
    rows = out.view(n, s_ // 32, 32, h, d)[:, :, 0, :, :10].contiguous()
    f = rows.view(torch.int16).view(n, s_ // 32, h, 10).contiguous().view(torch.float32).view(n, s_ // 32, h, 5).float()
    m = f.reshape(-1, 5)[:, :4].mean(0).tolist()
    print(f"S={s_}: issue done {m[0]:8.0f} | 3 tiles landed {m[1]:8.0f} | loop starts {m[2]:8.0f} | loop done {m[3]:8.0f} cycles; tiles {s_ // 64}")
