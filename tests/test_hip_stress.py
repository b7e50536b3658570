"""GPU: repetition stress of the two kernels whose failures were SPORADIC.

* the folded-LayerNorm GEMM epilogue at 57344 x 640 x 640 — round 2 met a packed-FMA form of its correction that dropped a product
  in lanes 48-63 for ~2e-5 of the outputs, different elements every run (csrc/aid_gemm.hip, store_tile); the scalar form ships.  A
  single pass of the parity test sees ~37 M outputs once; here the launch is repeated 200 times per side and dtype and EVERY output
  of every repetition is held bit-for-bit against the first one, which itself is held against the fp64 oracle with the
  worst-element guard (a glitch rate of 1e-9 per output would still be caught with probability > 0.99);
* the ping-pong attention kernel at S = 4096 (LDS-DMA ring, counted waits, two wave groups one barrier apart, persistent walk):
  a mis-counted wait or an early fragment read shows up as rare wrong tiles that come and go with timing — 200 launches per mode
  (plain / fused outer / fused inner, 7 AID + 7 rider frames like the bench), bit-for-bit against the first, the first against the
  oracle on sampled query rows of every frame.
"""
import numpy as np
import pytest
import torch

from oracle import aid_oracle as O
from util import TOL, TOL_GEMM, WORST, rel_l2, to_np64, worst

pytestmark = pytest.mark.gpu

import aid_amd  # noqa: E402,F401
from aid_amd import ops  # noqa: E402

DEV = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16]
ids_dt = lambda d: str(d).split(".")[-1]  # noqa: E731
REPS = 200


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
def test_folded_layernorm_epilogue_200_repetitions(dtype):
    m, n, k = 57344, 640, 640
    g = torch.Generator().manual_seed(57344)
    x = (torch.randn(m, k, generator=g) * 1.5 + 2.0).to(dtype)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype)
    gamma = (1.0 + 0.3 * torch.randn(k, generator=g)).to(dtype)
    beta = (0.2 * torch.randn(k, generator=g)).to(dtype)
    bias = torch.randn(n, generator=g).to(dtype).to(DEV)
    xd, wd = x.to(DEV), w.to(DEV)
    st = ops.ln_stats(xd, 1e-5)
    wf, cs, sh = ops.ln_fold(wd, gamma.to(DEV), beta.to(DEV))
    ref = O.layer_norm(to_np64(x), to_np64(gamma), to_np64(beta), 1e-5) @ to_np64(w).T
    frames, rows = 4, m // 4

    def side1(y):
        ops.gemm_nt([dict(a=xd, b=wf, c=y, bias=bias, m=m, n=n, k=k, lda=k, ldb=k, ldc=n, scale=0.5,
                          ln_stats=st, ln_colsum=cs, ln_shift=sh, ln_side=1)])

    def side2(yt):
        ops.gemm_nt([dict(a=wf, b=xd, c=yt, m=n, n=rows, k=k, lda=k, ldb=k, ldc=rows, batch=frames, stride_a=0,
                          stride_b=rows * k, stride_c=n * rows, ln_stats=st, ln_colsum=cs, ln_shift=sh, ln_side=2,
                          stride_stats=rows)])

    for run, shape, check in ((side1, (m, n), lambda y: (to_np64(y), 0.5 * ref + to_np64(bias))),
                              (side2, (frames, n, rows), lambda y: (to_np64(y).transpose(0, 2, 1).reshape(m, n), ref))):
        first = torch.empty(shape, dtype=dtype, device=DEV)
        run(first)
        got, want = check(first)
        assert rel_l2(got, want) < TOL_GEMM[dtype] and worst(got, want) < WORST[dtype], ops.last_gemm_variant()
        y = torch.empty_like(first)
        bad = torch.zeros((), dtype=torch.int64, device=DEV)
        for _ in range(REPS):
            y.fill_(float("nan"))                       # a launch that skips an element cannot inherit the right value
            run(y)
            bad += (y.view(torch.int16) != first.view(torch.int16)).sum()
        assert int(bad) == 0, f"{int(bad)} outputs differed from the first launch over {REPS} repetitions ({ops.last_gemm_variant()})"


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("mode,fused", [("plain", False), ("outer", True), ("inner", True)], ids=["plain", "outer", "inner"])
def test_pingpong_attention_s4096_200_repetitions(dtype, mode, fused):
    n, s, h, riders = 7, 4096, 10, 7
    c = 64 * h
    g = torch.Generator().manual_seed(4096 + len(mode))
    q = (torch.randn(n + riders, s, c, generator=g) * 0.6).to(dtype)
    k = torch.randn(n + riders, s, c, generator=g).to(dtype)
    v = torch.randn(n + riders, s, c, generator=g).to(dtype)
    vt = v.transpose(1, 2).contiguous()
    coef = torch.from_numpy(O.beta_coefs(n, 50, 50)).float().to(dtype).float()
    cd = torch.cat([coef, -torch.ones(riders)]).to(DEV)
    kw = dict(l=s, mode="plain") if mode == "plain" else dict(l=s, mode=mode, fused=fused, coef=cd, begin=0, end=n - 1, n_plain=riders)
    qd, kd, vd = q.to(DEV), k.to(DEV), vt.to(DEV)
    first = ops.attn_fwd(qd, kd, vd, h, **kw)
    assert ops.last_attn_variant().startswith("aid_attn_pp<d64"), ops.last_attn_variant()
    assert torch.isfinite(first).all()
    # the first launch against the oracle: 24 sampled query rows of every frame, all heads
    rows = torch.randperm(s, generator=g)[:24].sort().values
    q64, k64, v64 = to_np64(q[:, rows]), to_np64(k), to_np64(v)
    if mode == "plain":
        ref = O.attn_core(q64, k64, v64, h, 64 ** -0.5, "plain", False, None)
    else:
        ref = np.concatenate([O.attn_core(q64[:n], k64[:n], v64[:n], h, 64 ** -0.5, mode, fused, coef.numpy()),
                              O.attn_core(q64[n:], k64[n:], v64[n:], h, 64 ** -0.5, "plain", False, None)])
    got = to_np64(first[:, rows.to(DEV)])
    for f in range(n + riders):
        assert rel_l2(got[f], ref[f]) < TOL[dtype], f
    assert worst(got, ref) < WORST[dtype]
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    for _ in range(REPS):
        o = ops.attn_fwd(qd, kd, vd, h, **kw)
        bad += (o.view(torch.int16) != first.view(torch.int16)).sum()
    assert int(bad) == 0, f"{int(bad)} outputs differed from the first launch over {REPS} repetitions"
