#!/usr/bin/env python3
"""development check: resident-segment attention (AID_ATTN_RES=1) against the streaming kernel (=0) on the same inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aid_amd
from aid_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)


def ref(q, k, vt, l, h, mode, fused, coef):
    n, s, c = q.shape
    d = c // h
    qf = q.float().view(n, s, h, d).transpose(1, 2)
    kf = k.float().view(n, l, h, d).transpose(1, 2)
    vf = vt.float()[:, :, :l].transpose(1, 2).reshape(n, l, h, d).transpose(1, 2)
    att = lambda qq, kk, vv: torch.softmax(qq @ kk.transpose(-1, -2) * d ** -0.5, -1) @ vv
    out = []
    for i in range(n):
        ci = float(coef[i])
        if mode == "plain":
            o = att(qf[i], kf[i], vf[i])
        elif mode == "inner":
            km = ((1 - ci) * kf[0] + ci * kf[-1]).to(q.dtype).float() if 0 < ci < 1 else (kf[0] if ci == 0 else kf[-1])
            vm = ((1 - ci) * vf[0] + ci * vf[-1]).to(q.dtype).float() if 0 < ci < 1 else (vf[0] if ci == 0 else vf[-1])
            o = att(qf[i], torch.cat([kf[i], km], 1), torch.cat([vf[i], vm], 1)) if fused else att(qf[i], km, vm)
        else:
            cat = (lambda a_, b_: torch.cat([a_, b_], 1)) if fused else (lambda a_, b_: b_)
            o = (1 - ci) * att(qf[i], cat(kf[i], kf[0]), cat(vf[i], vf[0])) + ci * att(qf[i], cat(kf[i], kf[-1]), cat(vf[i], vf[-1]))
        out.append(o.transpose(0, 1).reshape(s, c))
    return torch.stack(out)

for dt in (torch.bfloat16, torch.float16):
    for (n, s, l, h, d) in ((5, 32, 32, 20, 64), (5, 32, 77, 20, 64), (5, 128, 77, 10, 64), (7, 256, 77, 8, 40), (7, 64, 64, 8, 80),
                            (3, 1024, 77, 20, 64), (7, 96, 96, 4, 40), (3, 40, 13, 2, 64)):
        c = h * d
        q = torch.randn(n, s, c, device=dev).to(dt); k = torch.randn(n, l, c, device=dev).to(dt)
        lp = (l + 7) // 8 * 8
        vt = torch.zeros(n, c, lp, device=dev, dtype=dt); vt[:, :, :l] = torch.randn(n, c, l, device=dev).to(dt)
        coef = torch.linspace(0, 1, n, device=dev, dtype=torch.float32).to(dt).float()
        for mode in ("plain", "inner", "outer"):
            for fused in (True, False):
                outs = []
                for res in ("0", "1"):
                    os.environ["AID_ATTN_RES"] = res
                    o = ops.attn_fwd(q, k, vt, h, l=l, mode=mode, fused=fused, coef=coef)
                    torch.cuda.synchronize()
                    outs.append(o.float())
                diff = (outs[0] - outs[1]).abs().max().item()
                r = ref(q, k, vt, l, h, mode, fused, coef)
                e0, e1 = ((o - r).norm() / r.norm() for o in outs)
                print(f"{str(dt)[6:]:9s} n{n} s{s} l{l} h{h} d{d} {mode:6s} fused={int(fused)}  max|res - stream| = {diff:.3e}   rel-L2 vs fp32: stream {e0:.2e}  res {e1:.2e}")
