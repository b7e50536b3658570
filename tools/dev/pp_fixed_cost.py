"""Development: per-workgroup fixed cost of the ping-pong attention kernel — S = 1024 query rows, 14 frames x 20 heads (1120 workgroups
= 4.375 per CU), key count swept; time = a + b * tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import aid_amd
from aid_amd import ops
lib = aid_amd._lib.load(); dev = torch.device("cuda:0"); dt = torch.bfloat16
def timed(fn, iters=10):
    fn(); fn()
    lib.aid_profile_begin()
    for _ in range(iters): fn()
    buf = (aid_amd._lib.AidProfileEntry * 4096)()
    c = lib.aid_profile_end(buf, 4096)
    e = [x for x in buf[:c] if x.kernel.decode().startswith("aid_attn")]
    return sum(x.ms for x in e) / iters * 1e3
n, s, h = 14, 1024, 20
q = torch.randn(n, s, h * 64, device=dev).to(dt)
out = torch.empty_like(q)
for v2 in (1, 0):
    ops.set_tuning("ATTN_V2", v2)
    pts = []
    for l in (512, 1024, 2048, 4096):
        k = torch.randn(n, l, h * 64, device=dev).to(dt); vt = torch.randn(n, h * 64, l, device=dev).to(dt)
        us = timed(lambda: ops.attn_fwd(q, k, vt, h, l=l, mode="plain", out=out))
        pts.append((l // 64, us))
        print(f"ATTN_V2={v2} L={l:5d} tiles {l // 64:3d}: {us:7.1f} us   {ops.last_attn_variant()}")
    (t0, u0), (t1, u1) = pts[1], pts[3]
    b = (u1 - u0) / (t1 - t0)
    print(f"   per tile {b:.2f} us per launch = {b / 4.375 * 1e3:.0f} ns per workgroup-tile; intercept {u0 - b * t0:.1f} us per launch = {(u0 - b * t0) / 4.375:.1f} us per workgroup")
