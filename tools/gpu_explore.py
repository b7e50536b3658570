"""Exploratory GPU check (not a test): prints rel-L2 errors of the HIP path vs the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import aid_amd
from aid_amd import ops
from oracle import aid_oracle as O
import cases as C

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0))

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

# ---- GEMM
for dt in (torch.float16, torch.bfloat16):
    for (m, n, k) in ((300, 320, 320), (77 * 3, 640, 768), (4096, 640, 640), (128, 128, 64), (1, 8, 8)):
        g = torch.Generator().manual_seed(m + n + k)
        a = torch.randn(m, k, generator=g).to(dt).to(dev); b = (torch.randn(n, k, generator=g) / k ** .5).to(dt).to(dev)
        bias = torch.randn(n, generator=g).to(dt).to(dev)
        y = ops.linear(a, b, bias)
        ref = a.float().cpu().double() @ b.float().cpu().double().T + bias.float().cpu().double()
        print("gemm", dt, (m, n, k), "rel", rel(y.float().cpu().numpy(), ref.numpy()))
    # V^T problem
    f, l, cc, c = 3, 77, 96, 80
    e = torch.randn(f, l, cc).to(dt).to(dev); wk = torch.randn(c, cc).to(dt).to(dev) / cc ** .5; wv = torch.randn(c, cc).to(dt).to(dev) / cc ** .5
    k_, vt = ops.project_kv(e, wk, wv)
    kr = e.float().cpu() @ wk.float().cpu().T
    vr = (e.float().cpu() @ wv.float().cpu().T).transpose(1, 2)
    print("kv", dt, rel(k_.float().cpu().numpy(), kr.numpy()), rel(vt[:, :, :l].float().cpu().numpy(), vr.numpy()),
          "pad max", float(vt[:, :, l:].abs().max()))

# ---- attention core
for dt, npdt in ((torch.float16, np.float16), (torch.bfloat16, None)):
    for d in (40, 64, 80, 160):
        for (n, s, l, h) in ((3, 40, 77, 2), (7, 200, 200, 2), (3, 33, 130, 1)):
            for mode, fused in (("plain", False), ("inner", False), ("inner", True), ("outer", False), ("outer", True)):
                g = torch.Generator().manual_seed(d * 1000 + s)
                c = h * d
                q = torch.randn(n, s, c, generator=g).to(dt); k = torch.randn(n, l, c, generator=g).to(dt); v = torch.randn(n, l, c, generator=g).to(dt)
                coef = torch.from_numpy(O.beta_coefs(n, 3, 3)) if n > 3 else torch.tensor([0., .3, 1.])
                lp = (l + 7) // 8 * 8
                vt = torch.zeros(n, c, lp, dtype=dt); vt[:, :, :l] = v.transpose(1, 2)
                o = ops.attn_fwd(q.to(dev), k.to(dev), vt.to(dev), h, l=l, mode=mode, fused=fused, coef=coef.to(dev))
                ref = O.attn_core(q.float().numpy().astype(np.float64), k.float().numpy().astype(np.float64),
                                  v.float().numpy().astype(np.float64), h, d ** -0.5, mode, fused, coef.numpy())
                r = rel(o.float().cpu().numpy(), ref)
                flag = "" if r < (3e-3 if dt == torch.float16 else 2e-2) else "   <<<<<< BAD"
                print("attn", str(dt)[6:], d, (n, s, l, h), mode, fused, ops.last_attn_variant(), f"rel {r:.2e}{flag}")

# ---- processors vs goldens
TEXT = C.load_fixture("text_goldens.npz")
for dt in (torch.float16, torch.bfloat16):
    worst = 0
    for case in C.TEXT_CASES:
        inp = C.text_inputs(case)
        attn = aid_amd.AttnShim(case.c, case.heads, case.cc if case.cross else None, dtype=dt, device=dev)
        with torch.no_grad():
            for lin, key in ((attn.to_q, "wq"), (attn.to_k, "wk"), (attn.to_v, "wv"), (attn.to_out[0], "wo")):
                lin.weight.copy_(torch.from_numpy(inp[key]).to(dt))
            attn.to_out[0].bias.copy_(torch.from_numpy(inp["bo"]).to(dt))
        if case.mode == "plain":
            proc = aid_amd.HipAttnProcessor()
        else:
            cls = aid_amd.OuterInterpolatedAttnProcessor if case.mode.endswith("outer") else aid_amd.InnerInterpolatedAttnProcessor
            proc = cls(t=case.t, size=case.n, is_fused=case.mode.startswith("fused"), alpha=case.alpha, beta=case.beta)
        x = torch.from_numpy(inp["x"]).to(dt).to(dev)
        ctx = torch.from_numpy(inp["ctx"]).to(dt).to(dev) if case.cross else None
        y = proc(attn, x, encoder_hidden_states=ctx)
        r = rel(y.float().cpu().numpy(), TEXT[case.name]); worst = max(worst, r)
        print("golden", str(dt)[6:], case.name, f"rel {r:.2e}")
    print("worst", dt, worst)
torch.cuda.synchronize()
print("done")
