import sys, torch
sys.path.insert(0, '/root/repo')
import aid_amd
from aid_amd import ops
lib = aid_amd._lib.load()
for rows, c in ((14336, 1280), (57344, 640), (57344, 320)):
    x = torch.randn(rows, c, device='cuda').to(torch.bfloat16)
    for _ in range(3): ops.ln_stats(x)
    torch.cuda.synchronize()
    lib.aid_profile_begin()
    for _ in range(20): ops.ln_stats(x)
    buf = (aid_amd._lib.AidProfileEntry * 64)(); n = lib.aid_profile_end(buf, 64)
    us = sorted(e.ms for e in buf[:n])[n // 2] * 1e3
    print(rows, c, f"{us:.1f} us  {rows * c * 2 / us / 1e6:.2f} TB/s")
