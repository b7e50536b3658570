#!/usr/bin/env python3
"""Same-process interleaved A/B of GEMM engine variants (aid_set_tuning) at the projection shapes of the SDXL stack.
usage: python tools/gemm_ab.py "GEMM_PP=0" "GEMM_PP=1" "GEMM_PP=2" [--rounds 5] [--iters 10] [--torch] [--ksweep] [--short] [--shards] [--sd15]
Every variant string is a comma-separated list of NAME=value knobs.  Prints median us and TF/s per shape and variant."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import aid_amd  # noqa: E402
from aid_amd import ops  # noqa: E402

args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
opt = lambda k, d: int(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d     # noqa: E731
ROUNDS, ITERS = opt("--rounds", 5), opt("--iters", 10)
variants = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in args] or [{}]
dev = torch.device("cuda:0")
lib = aid_amd._lib.load()
dt = torch.float16 if "--f16" in sys.argv else torch.bfloat16

# (label, list of (m, n, k) problems of ONE grouped launch)
SHAPES = [("sdxl L2 qkv 3x(14336,1280,1280)", [(14336, 1280, 1280)] * 3),
          ("sdxl L2 out (14336,1280,1280)", [(14336, 1280, 1280)]),
          ("sdxl L1 qkv 3x(57344,640,640)", [(57344, 640, 640)] * 3),
          ("sdxl L1 out (57344,640,640)", [(57344, 640, 640)]),
          ("square 4096^3", [(4096, 4096, 4096)]),
          ("square 8192^3", [(8192, 8192, 8192)])]
if "--ksweep" in sys.argv:
    SHAPES = [(f"4096x4096x{k}", [(4096, 4096, k)]) for k in (128, 640, 1280, 2560, 5120)]
if "--short" in sys.argv:        # the short-K levels (row-stationary engine, GEMM_RS): batched-CFG call and one pass of a two-stream step
    SHAPES = [("sdxl L1 qkv 3x(57344,640,640)", [(57344, 640, 640)] * 3), ("sdxl L1 out (57344,640,640)", [(57344, 640, 640)]),
              ("sdxl L1 qkv 3x(28672,640,640)", [(28672, 640, 640)] * 3), ("sdxl L1 out (28672,640,640)", [(28672, 640, 640)]),
              ("sd15 L0 qkv 3x(57344,320,320)", [(57344, 320, 320)] * 3), ("sd15 L0 out (57344,320,320)", [(57344, 320, 320)]),
              ("sd15 L0 qkv 3x(28672,320,320)", [(28672, 320, 320)] * 3), ("sd15 L0 out (28672,320,320)", [(28672, 320, 320)])]
if "--shards" in sys.argv:       # the C = 1280 projections at the local batches of a sharded 16-frame sequence (2 x 4 / 6 / 9 frames x 1024 rows)
    SHAPES = [(f"sdxl L2 {nm} {f}f ({m},1280,1280)", [(m, 1280, 1280)] * c) for f in (4, 6, 9) for m in (2 * f * 1024,)
              for nm, c in (("qkv 3x", 3), ("out", 1))]
if "--ls" in sys.argv:          # launches of the lock-step engine (GEMM_LS): one pass (7 frames) of the SD1.5 stack, f16 in the bench (--f16)
    SHAPES = [("sd15 L0 out (28672,320,320)", [(28672, 320, 320)]), ("sd15 L1 out (7168,640,640)", [(7168, 640, 640)]),
              ("sd15 L1 qkv 3x(7168,640,640)", [(7168, 640, 640)] * 3),
              ("sd15 L2 out (1792,1280,1280)", [(1792, 1280, 1280)]), ("sd15 L2 qkv 3x(1792,1280,1280)", [(1792, 1280, 1280)] * 3),
              ("sd15 mid out (448,1280,1280)", [(448, 1280, 1280)]), ("sd15 mid qkv 3x(448,1280,1280)", [(448, 1280, 1280)] * 3),
              ("sd15 L0 text k+v 2x(1078,320,768)", [(1078, 320, 768)] * 2), ("sd15 L2 text k+v 2x(1078,1280,768)", [(1078, 1280, 768)] * 2),
              ("sd15 L1 out cfg-batched (14336,640,640)", [(14336, 640, 640)]), ("sd15 L2 out cfg-batched (3584,1280,1280)", [(3584, 1280, 1280)]),
              ("sdxl L2 text k+v 2x(1078,1280,2048)", [(1078, 1280, 2048)] * 2)]
if "--sd15" in sys.argv:
    SHAPES = [("sd15 L0 qkv 3x(57344,320,320)", [(57344, 320, 320)] * 3), ("sd15 L0 out", [(57344, 320, 320)]),
              ("sd15 L1 qkv 3x(14336,640,640)", [(14336, 640, 640)] * 3), ("sd15 L1 out", [(14336, 640, 640)]),
              ("sd15 L2 qkv 3x(3584,1280,1280)", [(3584, 1280, 1280)] * 3), ("sd15 L2 out", [(3584, 1280, 1280)])]


def timed(fn):
    lib.aid_profile_begin()
    for _ in range(ITERS):
        fn()
    buf = (aid_amd._lib.AidProfileEntry * 4096)()
    n = lib.aid_profile_end(buf, 4096)
    return sum(x.ms for x in buf[:n]) / n * 1e3


def apply(v):
    for name in ("GEMM_VARIANT", "GEMM_PP", "GEMM_TRI", "GEMM_RS", "CU_SHARE", "GEMM_LS"):
        ops.set_tuning(name, v.get(name, -1))


for label, probs in SHAPES:
    m, n, k = probs[0]
    a = torch.randn(m, k, device=dev).to(dt)
    ws = [torch.randn(n, k, device=dev).to(dt) for _ in probs]
    outs = [torch.empty(m, n, device=dev, dtype=dt) for _ in probs]
    fl = sum(2.0 * m_ * n_ * k_ for m_, n_, k_ in probs)
    ref = (a.float() @ ws[-1].float().t())

    def run():
        ops.gemm_nt([dict(a=a, b=w, c=o, m=m, n=n, k=k, lda=k, ldb=k, ldc=n) for w, o in zip(ws, outs)])

    res = {i: [] for i in range(len(variants))}
    names = {}
    for r in range(ROUNDS):
        for i, v in enumerate(variants):
            apply(v)
            if r == 0:
                outs[-1].zero_()
                run()
                torch.cuda.synchronize()
                err = ((outs[-1].float() - ref).norm() / ref.norm()).item()
                if i == 0:
                    first = outs[-1].clone()
                names[i] = (ops.last_gemm_variant() + ("" if torch.equal(first, outs[-1]) else " BITS-DIFFER"), err)
                run()
            res[i].append(timed(run))
    for i, v in enumerate(variants):
        us = statistics.median(res[i])
        print(f"{label:34s} {str(v):40s} {names[i][0]:28s} {us:8.1f} us (min {min(res[i]):7.1f})  {fl / us / 1e6:7.1f} TF/s  rel {names[i][1]:.1e}",
              flush=True)
    if "--torch" in sys.argv:
        wcat = torch.cat(ws, 0)
        for _ in range(3):
            c = a @ wcat.t()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            c = a @ wcat.t()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / ITERS * 1e3
        print(f"{label:34s} {'hipBLASLt (torch.matmul, one problem)':40s} {'':28s} {us:8.1f} us                {fl / us / 1e6:7.1f} TF/s", flush=True)
apply({})
