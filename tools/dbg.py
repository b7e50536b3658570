import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import aid_amd
from aid_amd import ops
from oracle import aid_oracle as O
from util import rel_l2, to_np64
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_hip_parity import _core_inputs, _coef
DEV = "cuda:0"
for d in (160, 80):
    for (n, s, l, h) in ((3, 40, 77, 2), (3, 40, 64, 2), (3, 40, 141, 2), (3, 300, 77, 2)):
        q, k, v, vt = _core_inputs(n, s, l, h, d, torch.float16, seed=d * 1000 + s)
        coef = _coef(n)
        for mode, fused in (("inner", False), ("inner", True), ("outer", False)):
            errs = []
            for rep in range(3):
                o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode=mode, fused=fused, coef=coef.to(DEV))
                ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, d ** -0.5, mode, fused, coef.numpy())
                errs.append([round(rel_l2(to_np64(o[i]), ref[i]), 5) for i in range(n)])
            print(d, (n, s, l, h), mode, fused, ops.last_attn_variant(), errs)
