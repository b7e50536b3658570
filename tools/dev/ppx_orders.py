"""Read-slot orders of the 288 x 256 GEMM's K loop (csrc/aid_gemm.hip mac_x<.., ORD>), development build libaid_ppxord.so:
    AID_LIB_PATH=tools/dev/libaid_ppxord.so python tools/dev/ppx_orders.py
GEMM_PP = 3: the product's order (requests behind the fragment reads), 4: requests first, 5: one request per four reads, 6: product order
+ s_setprio 1 on the MFMA slots.  Checks that every order gives the SAME BITS, then times them interleaved (library events)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import aid_amd  # noqa: E402
from aid_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
lib = aid_amd._lib.load()
ORDERS = (3, 4, 5, 6, 7)      # 7 = timing ablation (K offset frozen: all L2 hits), results are garbage
SHAPES = [("out / q-cross 14336x1280x1280", [(14336, 1280, 1280)]), ("q k Vt 3x(14336,1280,1280)", [(14336, 1280, 1280)] * 3),
          ("seq16 32768x1280x1280", [(32768, 1280, 1280)]), ("4096^3", [(4096, 4096, 4096)])]


def run(probs):
    ops.gemm_nt(probs)


def timed(probs, iters=10):
    lib.aid_profile_begin()
    for _ in range(iters):
        run(probs)
    buf = (aid_amd._lib.AidProfileEntry * 256)()
    n = lib.aid_profile_end(buf, 256)
    return sum(e.ms for e in buf[:n]) / iters * 1e3, buf[0].kernel.decode()


for tag, mnk in SHAPES:
    torch.manual_seed(0)
    probs = []
    for (m, n, k) in mnk:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        b = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
        c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        probs.append(dict(a=a, b=b, c=c, m=m, n=n, k=k, lda=k, ldb=k, ldc=n))
    ops.set_tuning("GEMM_TRI", 1)
    ref = None
    for o in ORDERS:
        ops.set_tuning("GEMM_PP", o)
        run(probs)
        torch.cuda.synchronize()
        out = [p["c"].clone() for p in probs]
        if ref is None:
            ref = out
        else:
            assert o == 7 or all(torch.equal(x, y) for x, y in zip(out, ref)), (tag, o)
    res = {o: [] for o in ORDERS}
    for _ in range(7):
        for o in ORDERS:
            ops.set_tuning("GEMM_PP", o)
            us, name = timed(probs)
            res[o].append(us)
    flops = sum(2.0 * m * n * k for m, n, k in mnk)
    print(f"{tag:34s} " + "  ".join(f"PP={o}: {statistics.median(res[o]):7.1f} us {flops / statistics.median(res[o]) / 1e6:6.0f} TF" for o in ORDERS) + f"   [{name}] bits equal", flush=True)
