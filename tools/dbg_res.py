import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, aid_amd
from aid_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
  for d in (64, 40, 80):
    for l in (8, 24, 32, 33, 64, 77, 88, 96):
        n, s, h = 3, 64, 2
        c = h * d
        q = torch.randn(n, s, c, device=dev).to(dt); k = torch.randn(n, l, c, device=dev).to(dt)
        lp = (l + 7) // 8 * 8
        vt = torch.zeros(n, c, lp, device=dev, dtype=dt); vt[:, :, :l] = torch.randn(n, c, l, device=dev).to(dt)
        coef = torch.linspace(0, 1, n, device=dev)
        line = f"{str(dt)[6:]:8s} d{d} l{l:3d}: "
        for mode, fused in (("plain", False), ("inner", False), ("inner", True), ("outer", False), ("outer", True)):
            outs = []
            for res in ("0", "1"):
                os.environ["AID_ATTN_RES"] = res
                outs.append(ops.attn_fwd(q, k, vt, h, l=l, mode=mode, fused=fused, coef=coef).float()); torch.cuda.synchronize()
            line += f" {mode[0]}{int(fused)} {(outs[0] - outs[1]).abs().max().item():.2e}"
        print(line)
