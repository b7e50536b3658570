"""Development: slot timing of the ping-pong attention kernel (ablation build, bit 16 of the ablation word): shader cycles per
tile of [V work | barrier wait | M work | barrier wait] per wave group.  Needs the development build (make -C tools/dev):
AID_LIB_PATH=tools/dev/libaid_abl.so python tools/dev/pp_slots.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import aid_amd  # noqa: E402,F401
from aid_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for s_, h in ((4096, 10), (1024, 20)):
    n, d = 14, 64
    c = h * d
    q = torch.randn(n, s_, c, device=dev).to(torch.bfloat16)
    k = torch.randn(n, s_, c, device=dev).to(torch.bfloat16)
    vt = torch.randn(n, c, s_, device=dev).to(torch.bfloat16)
    out = torch.zeros_like(q)
    for abl, what in ((116, "full"), (117, "no VALU work"), (120, "no MFMA"), (118, "no DMA"), (124, "no max chain")):
        ops.set_tuning("ATTN_V2", 1)
        ops.set_tuning("ATTN_RES_CHUNKS", abl)
        out.zero_()
        ops.attn_fwd(q, k, vt, h, l=s_, mode="plain", out=out)
        torch.cuda.synchronize()
        rows = out.view(n, s_ // 32, 32, h, d)[:, :, 0, :, :8].contiguous()          # first 16 bytes of every wave's first row
        f = rows.view(torch.int16).view(n, s_ // 32, h, 8).contiguous().view(torch.float32).view(n, s_ // 32, h, 4).float()
        grp = (torch.arange(s_ // 32, device=dev) % 8) // 4
        for g in (0, 1):
            m = f[:, grp == g].reshape(-1, 4).mean(0).tolist()
            print(f"S={s_} {what:14s} group {g}: V work {m[0]:7.1f}  wait {m[1]:7.1f}  M work {m[2]:7.1f}  wait {m[3]:7.1f}  "
                  f"per tile {sum(m):7.1f} cycles")
