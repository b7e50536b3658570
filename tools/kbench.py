#!/usr/bin/env python3
"""Kernel micro-benchmark (GPU box): HIP-event timing of the GEMM and attention kernels at the layer
shapes of SD1.5 / SDXL via the library's aid_profile_begin/end.  Prints algorithmic TFLOP/s.
usage: python tools/kbench.py [gemm] [attn] [layer] [--iters 20]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import aid_amd  # noqa: E402
from aid_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ITERS = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
what = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()] or ["gemm", "attn", "layer"]
lib = aid_amd._lib.load()


def timed(fn, iters=ITERS):
    fn(); fn()
    torch.cuda.synchronize()
    lib.aid_profile_begin()
    for _ in range(iters):
        fn()
    buf = (aid_amd._lib.AidProfileEntry * 8192)()
    n = lib.aid_profile_end(buf, 8192)
    agg = {}
    for e in buf[:n]:
        a = agg.setdefault(e.kernel.decode(), [0.0, 0.0, 0])
        a[0] += e.ms; a[1] += e.flops; a[2] += 1
    return agg


def report(tag, agg):
    for k, (ms, fl, cnt) in agg.items():
        print(f"{tag:46s} {k:34s} {ms / cnt * 1e3:9.1f} us  {fl / ms / 1e9:8.1f} TF/s", flush=True)


if "gemm" in what:
    for dt in (torch.float16, torch.bfloat16):
        for (m, n, k) in ((28672, 320, 320), (7168, 640, 640), (1792, 1280, 1280), (28672, 640, 640),
                          (7168, 1280, 1280), (539, 1280, 2048), (539, 320, 768), (8192, 8192, 8192)):
            a = torch.randn(m, k, device=dev).to(dt); b = torch.randn(n, k, device=dev).to(dt)
            bias = torch.randn(n, device=dev).to(dt)
            out = torch.empty(m, n, device=dev, dtype=dt)
            report(f"gemm {str(dt)[6:]} m{m} n{n} k{k}", timed(lambda: ops.linear(a, b, bias, out=out)))

if "attn" in what:
    cases = [("sd15 S4096 d40 H8", torch.float16, 7, 4096, 4096, 8, 40, [("plain", False), ("inner", True), ("outer", True)]),
             ("sd15 S1024 d80 H8", torch.float16, 7, 1024, 1024, 8, 80, [("plain", False), ("inner", True)]),
             ("sd15 S256 d160 H8", torch.float16, 7, 256, 256, 8, 160, [("plain", False), ("inner", True)]),
             ("sd15 S4096 x77 d40", torch.float16, 7, 4096, 77, 8, 40, [("plain", False), ("inner", True)]),
             ("sdxl S4096 d64 H10", torch.bfloat16, 7, 4096, 4096, 10, 64, [("plain", False), ("inner", True), ("outer", True)]),
             ("sdxl S1024 d64 H20", torch.bfloat16, 7, 1024, 1024, 20, 64, [("plain", False), ("inner", True), ("outer", True)]),
             ("sdxl S1024 x77 d64", torch.bfloat16, 7, 1024, 77, 20, 64, [("plain", False), ("outer", True)])]
    for tag, dt, n, s, l, h, d, modes in cases:
        c = h * d
        q = torch.randn(n, s, c, device=dev).to(dt); k = torch.randn(n, l, c, device=dev).to(dt)
        lp = (l + 7) // 8 * 8
        vt = torch.randn(n, c, lp, device=dev).to(dt)
        coef = aid_amd.generate_beta_tensor(n, 50, 50).to(dev); coef[0] = 0; coef[-1] = 1
        out = torch.empty_like(q)
        for mode, fused in modes:
            report(f"attn {tag} {mode}{'+own' if fused else ''}",
                   timed(lambda: ops.attn_fwd(q, k, vt, h, l=l, mode=mode, fused=fused, coef=coef, out=out)))

if "layer" in what:
    for tag, dt, n, s, c, h, cc in (("sd15 L0", torch.float16, 7, 4096, 320, 8, 768), ("sd15 L1", torch.float16, 7, 1024, 640, 8, 768),
                                    ("sdxl L1", torch.bfloat16, 7, 4096, 640, 10, 2048), ("sdxl L2", torch.bfloat16, 7, 1024, 1280, 20, 2048)):
        for cross in (False, True):
            attn = aid_amd.AttnShim(c, h, cc if cross else None, dtype=dt, device=dev)
            x = torch.randn(n, s, c, device=dev).to(dt)
            ctx = torch.randn(n, 77, cc, device=dev).to(dt) if cross else None
            proc = aid_amd.HipAttnProcessor()
            report(f"layer {tag} {'cross' if cross else 'self'} plain", timed(lambda: proc(attn, x, ctx), iters=5))
torch.cuda.synchronize()
print("done")
