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
    for abl, what in ((116, "full"), (117, "no VALU work"), (120, "no MFMA"), (118, "no DMA"), (124, "no max chain"),
                      (180, "half LDS reads"), (182, "half LDS, no DMA")):
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

# workgroup timeline (bit 32): every item's wave 0 writes [requests issued, first wait done, prologue done, THIS item finished] in
# shader cycles since its workgroup started.  Items of a persistent workgroup finish at start-up + k x item time — without the
# slot stamps (each costs an s_memtime round trip and drains the LDS counter), so this is the undisturbed cost of each ablation.
import numpy as np  # noqa: E402
for s_, h in ((4096, 10), (1024, 20)):
    n, d = 14, 64
    c = h * d
    q = torch.randn(n, s_, c, device=dev).to(torch.bfloat16)
    k = torch.randn(n, s_, c, device=dev).to(torch.bfloat16)
    vt = torch.randn(n, c, s_, device=dev).to(torch.bfloat16)
    for abl, what in ((32, "full"), (33, "no VALU work"), (36, "no MFMA"), (34, "no DMA"), (35, "no VALU, no DMA"), (38, "no MFMA, no DMA"),
                      (37, "no VALU, no MFMA"), (39, "barriers only"), (96, "half LDS reads"), (40, "no max chain")):
        out = torch.zeros_like(q)
        ops.set_tuning("ATTN_V2", 1)
        ops.set_tuning("ATTN_RES_CHUNKS", 100 + abl)
        ops.attn_fwd(q, k, vt, h, l=s_, mode="plain", out=out)
        torch.cuda.synchronize()
        rows = out.view(n, s_ // 256, 256, h, d)[:, :, 0, :, :8].contiguous()             # wave 0's first row of every item
        f = rows.view(torch.int16).view(n, s_ // 256, h, 8).contiguous().view(torch.float32).view(-1, 4).float().cpu().numpy()
        fin = np.sort(f[:, 3])
        per_wg = len(fin) / 256.0
        steps = [np.median(fin[256 * (j + 1):256 * (j + 2)]) - np.median(fin[256 * j:256 * (j + 1)]) for j in range(int(per_wg) - 1)]
        nt = s_ // 64
        print(f"S={s_} {what:18s} prologue done {f[:, 2].mean():7.0f}  first item {np.median(fin[:256]):8.0f}  per further item {np.median(steps):8.0f} "
              f"= {np.median(steps) / nt / 2:6.0f} cycles per interval   last item done {fin[-1]:9.0f}")
