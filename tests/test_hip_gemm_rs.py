"""GPU parity of the row-stationary projection engine (csrc/aid_gemm_rs.hip): the short-K levels of the UNets — attn.to_q / to_k /
to_v / to_out[0] at C = 320 (SD1.5, S = 4096) and C = 640 (SDXL, S = 4096), reference interpolation.py:613, 623-624, 666 — against
fp64 and against the tile engines of aid_gemm.hip on the same operands (same products, another fp32 summation grouping: equal up to
the storage type's last place on the rare sum that sits on a rounding boundary)."""
import numpy as np
import pytest
import torch

from util import TOL_GEMM, WORST, rel_l2, to_np64, worst

pytestmark = pytest.mark.gpu

import aid_amd  # noqa: E402,F401
from aid_amd import ops  # noqa: E402

DEV = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16]
ids_dt = lambda d: str(d).split(".")[-1]  # noqa: E731


def _same(a, b, dtype):
    """Two engines, one product: identical storage values except for isolated last-place differences."""
    a, b = a.float(), b.float()
    diff = (a - b).abs()
    ulp = (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7) * torch.maximum(a.abs(), b.abs()).clamp_min(1e-3)
    assert bool((diff <= 1.01 * ulp).all()), float((diff / ulp).max())
    assert float((diff > 0).float().mean()) < 0.02, float((diff > 0).float().mean())


def _operands(m, n, k, dtype, seed, n_w=1):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(m, k, generator=g) + 0.3).to(dtype).to(DEV)
    ws = [(torch.randn(n, k, generator=g) / k ** 0.5).to(dtype).to(DEV) for _ in range(n_w)]
    return g, a, ws


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("m,n,k", [(2048, 640, 640),        # 8 row tiles: the slice range of a tile is split over workgroups
                                   (1024, 320, 320),        # K = 320: two weight rows per 1280-B virtual row
                                   (512, 1280, 640),        # more columns than K
                                   (256, 96, 320),          # three slices
                                   (57344, 640, 640),       # SDXL level 1 out projection at full size (the default rule takes it)
                                   (28672, 320, 320)])      # SD1.5 level 0, one pass of a two-stream step
def test_single_projection_bias_scale_residual(dtype, m, n, k, tuning):
    """One projection with scale, bias and the residual added after the rounding: fp64 within the GEMM tolerance, and the tile
    engines' values."""
    g, a, (w,) = _operands(m, n, k, dtype, m + n + k)
    bias = torch.randn(n, generator=g).to(dtype).to(DEV)
    res = torch.randn(m, n, generator=g).to(dtype).to(DEV)
    outs = {}
    for rs in (1, 0):
        tuning("GEMM_RS", rs)
        y = torch.full((m, n), float("nan"), dtype=dtype, device=DEV)
        ops.gemm_nt([dict(a=a, b=w, c=y, bias=bias, residual=res, m=m, n=n, k=k, lda=k, ldb=k, ldc=n, scale=0.25)])
        name = ops.last_gemm_variant()
        assert name.startswith("rowstat") == (rs == 1), name
        outs[rs] = y
    rows = torch.unique(torch.cat([torch.arange(0, m, 61), torch.tensor([m - 1])]))
    ref = 0.25 * (to_np64(a[rows]) @ to_np64(w).T) + to_np64(bias)
    ref = to_np64(torch.from_numpy(ref).to(dtype)) + to_np64(res[rows])             # rounded, then the residual
    got = to_np64(outs[1][rows])
    assert torch.isfinite(outs[1]).all()
    assert rel_l2(got, ref) < TOL_GEMM[dtype] and worst(got, ref) < WORST[dtype]
    _same(outs[1], outs[0], dtype)
    # the default rule: tall activations only
    tuning("GEMM_RS", -1)
    y = torch.empty(m, n, dtype=dtype, device=DEV)
    ops.gemm_nt([dict(a=a, b=w, c=y, m=m, n=n, k=k, lda=k, ldb=k, ldc=n)])
    assert ops.last_gemm_variant().startswith("rowstat") == (m >= (49152 if k == 320 else 16384)), ops.last_gemm_variant()


def test_default_rule_wants_whole_rounds_of_row_tiles(tuning):
    """More row tiles than workgroup slots: the engine is chosen only when the rounds are >= 80 % full — a 9-frame sequence shard (18
    batched frames x 4096 rows = 288 tiles of 256 rows on 256 CUs = 0.56 of two rounds) stays on the tile engines (625 TF/s on this
    engine against ~770), 32 frames (512 tiles: two whole rounds) take it; GEMM_RS = 1 still forces it.  Same values either way."""
    dtype, k = torch.bfloat16, 640
    for m, want in ((18 * 4096, False), (32 * 4096, True)):
        g, a, (w,) = _operands(m, k, k, dtype, m)
        y = torch.empty(m, k, dtype=dtype, device=DEV)
        tuning("GEMM_RS", -1)
        ops.gemm_nt([dict(a=a, b=w, c=y, m=m, n=k, k=k, lda=k, ldb=k, ldc=k)])
        assert ops.last_gemm_variant().startswith("rowstat") == want, (m, ops.last_gemm_variant())
        if not want:
            tuning("GEMM_RS", 1)
            y1 = torch.empty_like(y)
            ops.gemm_nt([dict(a=a, b=w, c=y1, m=m, n=k, k=k, lda=k, ldb=k, ldc=k)])
            assert ops.last_gemm_variant().startswith("rowstat")
            _same(y1, y, dtype)


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("frames,keys,c", [(2, 1024, 640),      # K = 640, slice range split six ways
                                           (3, 256, 320),       # K = 320
                                           (14, 4096, 640),     # SDXL level 1 self-attention, batched CFG call: 57344 rows
                                           (7, 4096, 320)])     # SD1.5 level 0, one pass
def test_grouped_qkv_with_transposed_values(dtype, frames, keys, c, tuning):
    """The q / k / V^T projections of a self-attention layer as the processor call issues them — one launch, x read once, the query
    pre-scaled, the values written as V^T[frame][channel][key] (AidGemmProblem.trans_rows)."""
    m = frames * keys
    g, x, (wq, wk, wv) = _operands(m, c, c, dtype, frames * 131 + keys + c, n_w=3)
    outs = {}
    for rs in (1, 0):
        tuning("GEMM_RS", rs)
        q, kk = (torch.full((m, c), float("nan"), dtype=dtype, device=DEV) for _ in range(2))
        vt = torch.full((frames, c, keys), float("nan"), dtype=dtype, device=DEV)
        ops.gemm_nt([dict(a=x, b=wq, c=q, m=m, n=c, k=c, lda=c, ldb=c, ldc=c, scale=0.18),
                     dict(a=x, b=wk, c=kk, m=m, n=c, k=c, lda=c, ldb=c, ldc=c),
                     dict(a=x, b=wv, c=vt, m=m, n=c, k=c, lda=c, ldb=c, ldc=keys, stride_c=c * keys, trans_rows=keys)])
        assert ops.last_gemm_variant().startswith("rowstat") == (rs == 1), ops.last_gemm_variant()
        outs[rs] = (q, kk, vt)
    q, kk, vt = outs[1]
    for t in (q, kk, vt):
        assert torch.isfinite(t).all()
    rows = torch.unique(torch.cat([torch.arange(0, m, 97), torch.tensor([m - 1])]))
    xr = to_np64(x[rows])
    assert rel_l2(to_np64(q[rows]), 0.18 * (xr @ to_np64(wq).T)) < TOL_GEMM[dtype]
    assert rel_l2(to_np64(kk[rows]), xr @ to_np64(wk).T) < TOL_GEMM[dtype]
    f = frames - 1
    ref_v = to_np64(wv) @ to_np64(x[f * keys:(f + 1) * keys]).T                     # V^T of the last frame, every key
    assert rel_l2(to_np64(vt[f]), ref_v) < TOL_GEMM[dtype] and worst(to_np64(vt[f]), ref_v) < WORST[dtype]
    for a_, b_ in zip(outs[1], outs[0]):
        _same(a_, b_, dtype)


def test_unsupported_groups_fall_back_to_the_tile_engines(tuning):
    """Ragged rows, unequal activations, other K: the group runs on aid_gemm.hip's engines whatever the knob says."""
    tuning("GEMM_RS", 1)
    dtype = torch.float16
    g, a, (w,) = _operands(1000, 640, 640, dtype, 5)                                # 1000 rows: not whole 256-row tiles
    y = torch.empty(1000, 640, dtype=dtype, device=DEV)
    ops.gemm_nt([dict(a=a, b=w, c=y, m=1000, n=640, k=640, lda=640, ldb=640, ldc=640)])
    assert not ops.last_gemm_variant().startswith("rowstat")
    assert rel_l2(to_np64(y), to_np64(a) @ to_np64(w).T) < TOL_GEMM[dtype]
    g, a2, (w2,) = _operands(1024, 1280, 1280, dtype, 6)                            # K = 1280: 320 registers of rows would not fit
    y2 = torch.empty(1024, 1280, dtype=dtype, device=DEV)
    ops.gemm_nt([dict(a=a2, b=w2, c=y2, m=1024, n=1280, k=1280, lda=1280, ldb=1280, ldc=1280)])
    assert not ops.last_gemm_variant().startswith("rowstat")


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
def test_repeat_stress_is_bit_stable(dtype, tuning):
    """200 launches of the SDXL level-1 grouped projection on the LDS-DMA ring: every launch bit-identical to the first (a slice read
    before its DMA landed, or a patch flushed late, shows up as a differing launch)."""
    tuning("GEMM_RS", 1)
    frames, keys, c = 4, 4096, 640
    m = frames * keys
    g, x, (wq, wk, wv) = _operands(m, c, c, dtype, 99, n_w=3)
    first = None
    for it in range(200):
        q, kk = (torch.empty(m, c, dtype=dtype, device=DEV) for _ in range(2))
        vt = torch.empty(frames, c, keys, dtype=dtype, device=DEV)
        ops.gemm_nt([dict(a=x, b=wq, c=q, m=m, n=c, k=c, lda=c, ldb=c, ldc=c),
                     dict(a=x, b=wk, c=kk, m=m, n=c, k=c, lda=c, ldb=c, ldc=c),
                     dict(a=x, b=wv, c=vt, m=m, n=c, k=c, lda=c, ldb=c, ldc=keys, stride_c=c * keys, trans_rows=keys)])
        if first is None:
            first = (q, kk, vt)
        else:
            assert torch.equal(q, first[0]) and torch.equal(kk, first[1]) and torch.equal(vt, first[2]), it
