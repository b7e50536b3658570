import sys, torch
sys.path.insert(0, "/root/repo")
import aid_amd
from aid_amd import ops
for (m, n, k) in [(33000, 512, 640), (33001, 504, 640), (14336, 1280, 1280), (3584, 3840, 1280), (57344, 320, 320), (1000, 1280, 2048), (14336, 3840, 1280), (57344, 1920, 640), (57344, 640, 640), (28672, 1280, 1280), (7168,1280,1280)]:
    a = torch.randn(m, k, device="cuda").bfloat16(); b = torch.randn(n, k, device="cuda").bfloat16()
    ops.linear(a, b)
    print(m, n, k, ops.last_gemm_variant())
