#!/usr/bin/env python3
"""Same-process interleaved A/B of attention-kernel variants (development knobs read per call by the library) at the
launch shapes of the two bench stacks: 7 AID frames + 7 plain riders, BetaPPF(50, 50) coefficients.
usage: python tools/kbench_attn_ab.py "AID_ATTN_QB=1" "AID_ATTN_QB=2" [--rounds 5] [--iters 6] [--shapes sdxl,sd15] [--only "S1024 x77"]
Prints per shape and variant: median us per launch over the rounds, algorithmic and executed TFLOP/s."""
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
ROUNDS, ITERS = opt("--rounds", 5), opt("--iters", 6)
which = sys.argv[sys.argv.index("--shapes") + 1].split(",") if "--shapes" in sys.argv else ["sdxl", "sd15"]
variants = [dict(kv.split("=") for kv in a.split(",") if kv) for a in args] or [{}]
dev = torch.device("cuda:0")
lib = aid_amd._lib.load()

SHAPES = {
    "sdxl": [("sdxl S4096 d64 H10", torch.bfloat16, 4096, 4096, 10, 64, ["outer", "plain"] + (["inner"] if "--inner" in sys.argv else [])),
             ("sdxl S1024 d64 H20", torch.bfloat16, 1024, 1024, 20, 64, ["outer", "plain"] + (["inner"] if "--inner" in sys.argv else [])),
             ("sdxl S1024 x77 d64", torch.bfloat16, 1024, 77, 20, 64, ["outer", "plain"]),
             ("sdxl S4096 x77 d64", torch.bfloat16, 4096, 77, 10, 64, ["outer", "plain"])],
    "sd15": [("sd15 S4096 d40 H8", torch.float16, 4096, 4096, 8, 40, ["inner", "plain"]),
             ("sd15 S1024 d80 H8", torch.float16, 1024, 1024, 8, 80, ["inner", "plain"]),
             ("sd15 S256 d160 H8", torch.float16, 256, 256, 8, 160, ["inner", "plain"]),
             ("sd15 S4096 x77 d40", torch.float16, 4096, 77, 8, 40, ["inner", "plain"]),
             ("sd15 S1024 x77 d80", torch.float16, 1024, 77, 8, 80, ["inner", "plain"])],
}


def one(fn):
    lib.aid_profile_begin()
    for _ in range(ITERS):
        fn()
    buf = (aid_amd._lib.AidProfileEntry * 4096)()
    n = lib.aid_profile_end(buf, 4096)
    e = [x for x in buf[:n] if x.kernel.decode().startswith("aid_attn")]
    calls = ITERS                                    # a call may be two launches (ping-pong kernel + program-order kernel)
    per = len(e) // calls
    return (sum(x.ms for x in e) / calls * 1e3, sum(x.flops for x in e[:per]), sum(x.flops_executed for x in e[:per]),
            "+".join(x.kernel.decode().replace("aid_attn", "") for x in e[:per]))


ONLY = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""       # substring of the shape tag
for grp in which:
    for tag, dt, s, l, h, d, modes in SHAPES[grp]:
        if ONLY not in tag:
            continue
        n, c = 7, h * d
        q = torch.randn(2 * n, s, c, device=dev).to(dt)
        k = torch.randn(2 * n, l, c, device=dev).to(dt)
        vt = torch.randn(2 * n, c, (l + 7) // 8 * 8, device=dev).to(dt)
        cf = aid_amd.generate_beta_tensor(n, 50, 50)
        cf[0], cf[-1] = 0, 1
        vals = cf.to(dt).float().tolist() + [-1.0] * n
        coef = torch.tensor(vals, device=dev)
        out = torch.empty_like(q)
        for mode in modes:
            if "--mode" in sys.argv and sys.argv[sys.argv.index("--mode") + 1] != mode:
                continue
            fused = mode != "plain"
            segx = ops.executed_segments(mode, fused, vals, 2 * n, None, 0, n - 1)
            fn = lambda: ops.attn_fwd(q, k, vt, h, l=l, mode=mode, fused=fused, coef=coef if fused else None,      # noqa: E731
                                      begin=0, end=n - 1, out=out, n_plain=n if fused else 0, seg_executed=segx)
            res = {i: [] for i in range(len(variants))}
            names = {}
            def apply(v):                                     # knobs are set through the library (aid_set_tuning), not the environment
                for name in ("ATTN_NW", "ATTN_QB", "ATTN_PIPE", "ATTN_RES", "ATTN_RES_CHUNKS", "ATTN_ORDER", "ATTN_V2", "ATTN_TX", "ATTN_TX_TILES"):
                    ops.set_tuning(name, int(v.get(name, v.get("AID_" + name, -1))))
            for i, v in enumerate(variants):                  # warm every variant (lazy attributes)
                apply(v); fn(); fn()
            torch.cuda.synchronize()
            for r in range(ROUNDS):
                for i, v in enumerate(variants):
                    apply(v)
                    us, fl, flx, nm = one(fn)
                    res[i].append(us); names[i] = (fl, flx, nm)
            for i, v in enumerate(variants):
                us = statistics.median(res[i])
                fl, flx, nm = names[i]
                print(f"{tag:20s} {mode:6s} {str(v):22s} {nm:44s} {us:9.1f} us  alg {fl / us / 1e6:7.1f}  exec {flx / us / 1e6:7.1f} TF/s"
                      f"   (min {min(res[i]):.1f})", flush=True)
print("done")
