"""Development: the C = 640 projection of the SDXL S = 4096 level with / without bias, scale, fresh inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import aid_amd
from aid_amd import ops
lib = aid_amd._lib.load()
dev = torch.device("cuda:0")
dt = torch.bfloat16
m, n, k = 57344, 640, 640
def timed(fn, iters=10):
    lib.aid_profile_begin()
    for _ in range(iters): fn()
    buf = (aid_amd._lib.AidProfileEntry * 4096)()
    c = lib.aid_profile_end(buf, 4096)
    return sum(x.ms for x in buf[:c]) / c * 1e3, buf[0].kernel.decode()
a = torch.randn(m, k, device=dev).to(dt); w = torch.randn(n, k, device=dev).to(dt); b = torch.randn(n, device=dev).to(dt)
out = torch.empty(m, n, device=dev, dtype=dt)
big = [torch.randn(m, k, device=dev).to(dt) for _ in range(8)]      # 8 x 73 MB: rotating A defeats the 256 MB cache
for name, kw in (("plain", {}), ("bias", dict(bias=b)), ("scale", dict(scale=0.18)), ("bias+scale", dict(bias=b, scale=0.18))):
    for v in (-1, 7, 31):
        ops.set_tuning("GEMM_VARIANT", v)
        t, kn = timed(lambda: ops.gemm_nt([dict(a=a, b=w, c=out, m=m, n=n, k=k, lda=k, ldb=k, ldc=n, **kw)]))
        i = [0]
        def rot():
            i[0] = (i[0] + 1) % 8
            ops.gemm_nt([dict(a=big[i[0]], b=w, c=out, m=m, n=n, k=k, lda=k, ldb=k, ldc=n, **kw)])
        t2, _ = timed(rot, 16)
        print(f"{name:10s} variant {v:3d} {ops.last_gemm_variant():20s} same A {t:6.1f} us   rotating A {t2:6.1f} us")
