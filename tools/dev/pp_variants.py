#!/usr/bin/env python3
"""Development: same-process interleaved A/B of several BUILDS of the library (one source, different -D switches) on the d = 64
attention launches of the SDXL stack.  usage:
    python tools/dev/pp_variants.py base=attention-interpolation-diffusion_amd/libaid_hip.so v1=gpurun_out/v/libaid_v1.so ... \
        [--rounds 7] [--iters 8] [--shapes 4096,1024] [--modes plain,outer,inner] [--knob ATTN_V2=1]
Per shape / mode / build: median and minimum us per call over the rounds, algorithmic TFLOP/s, and the relative L2 distance of the
build's output from the first build's (a variant that changes results beyond re-association shows up here)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import aid_amd  # noqa: E402
from aid_amd import _lib, ops  # noqa: E402

builds = [a.split("=", 1) for a in sys.argv[1:] if "=" in a and not a.startswith("--") and a.split("=", 1)[1].endswith(".so")]
opt = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d     # noqa: E731
ROUNDS, ITERS = int(opt("--rounds", 7)), int(opt("--iters", 8))
shapes = [int(x) for x in opt("--shapes", "4096,1024").split(",")]
modes = opt("--modes", "plain,outer,inner").split(",")
knobs = [kv.split("=") for kv in opt("--knob", "").split(",") if kv]
libs = [(name, _lib.bind(os.path.join(ROOT, path) if not os.path.isabs(path) else path)) for name, path in builds]
dev = torch.device("cuda:0")


def use(lib):
    _lib._lib = lib
    for k, v in knobs:
        ops.set_tuning(k, int(v))


for s in shapes:
    h, d, n, dt = (10 if s == 4096 else 20), 64, 7, torch.bfloat16
    if "--heads" in sys.argv:
        h = int(opt("--heads", h))
    c = h * d
    g = torch.Generator(device=dev).manual_seed(s)
    q = (torch.randn(2 * n, s, c, device=dev, generator=g) * 0.6).to(dt)
    k = torch.randn(2 * n, s, c, device=dev, generator=g).to(dt)
    vt = torch.randn(2 * n, c, s, device=dev, generator=g).to(dt)
    cf = aid_amd.generate_beta_tensor(n, 50, 50)
    cf[0], cf[-1] = 0, 1
    vals = cf.to(dt).float().tolist() + [-1.0] * n
    coef = torch.tensor(vals, device=dev)
    for mode in modes:
        fused = mode != "plain"
        seg = 14 if mode == "plain" else (24 if mode == "outer" else 19)       # executed segment units of the 7 + 7 call
        flops = 4.0 * s * s * c * (14 if mode == "plain" else (28 if mode == "outer" else 21))
        outs, res, names = {}, {nm: [] for nm, _ in libs}, {}
        for nm, lib in libs:
            use(lib)
            out = torch.full_like(q, float("nan"))
            kw = dict(l=s, mode=mode, fused=fused, coef=coef if fused else None, begin=0, end=n - 1, out=out, n_plain=n if fused else 0)
            ops.attn_fwd(q, k, vt, h, **kw)
            ops.attn_fwd(q, k, vt, h, **kw)
            names[nm] = ops.last_attn_variant()
            outs[nm] = (out, kw)
        torch.cuda.synchronize()
        for r in range(ROUNDS):
            for nm, lib in libs:
                use(lib)
                out, kw = outs[nm]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(ITERS):
                    ops.attn_fwd(q, k, vt, h, **kw)
                e1.record()
                torch.cuda.synchronize()
                res[nm].append(e0.elapsed_time(e1) * 1e3 / ITERS)
        base = outs[libs[0][0]][0].float()
        for nm, _ in libs:
            us = statistics.median(res[nm])
            o = outs[nm][0].float()
            dist = float((o - base).norm() / base.norm()) if torch.isfinite(o).all() else float("nan")
            print(f"S{s} H{h} {mode:6s} {nm:10s} {names[nm]:26s} {us:8.1f} us (min {min(res[nm]):8.1f})  alg {flops / us / 1e6:7.1f} TF/s"
                  f"  exec {flops * seg / (14 if mode == 'plain' else (28 if mode == 'outer' else 21)) / us / 1e6:7.1f}  d(base) {dist:.2e}", flush=True)
print("done")
