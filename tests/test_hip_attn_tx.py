"""GPU: the text-key attention kernel (csrc/aid_attn_tx.hip: d = 64, at most 96 keys per segment resident in LDS, independent waves,
online softmax over the segment's score tiles, OUTER sides combined from the segments' maxima and row sums — the default for the
cross-attention calls of the SDXL stack, profiles/r05_attn_tx_notes.txt) against the fp64 oracle and against the program-order kernel
(ATTN_TX = 0) on the same call.  PLAIN calls,
fused and pure OUTER calls with PLAIN riders, interior rows with end-point coefficients, shared contexts (kv_map), per-frame / output
scales, a pre-scaled q, every key count 1 .. 96 (ragged and whole score tiles), ragged query counts, forced large score ranges between
the segments, the SDXL layer shapes with repetition, INNER calls (one interpolated segment), and what the kernel hands back to aid_attn_kernel
(accumulating calls, > 96 keys, other head dims)."""
import numpy as np
import pytest
import torch

from oracle import aid_oracle as O
from util import TOL, WORST, rel_l2, to_np64, worst

pytestmark = pytest.mark.gpu

import aid_amd  # noqa: E402
from aid_amd import ops  # noqa: E402

DEV = "cuda:0"


DTYPES = [torch.float16, torch.bfloat16]
ids_dt = lambda d: str(d).split(".")[-1]  # noqa: E731


def _inputs(n, nkv, s, l, h, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    c = h * 64
    return (torch.randn(n, s, c, generator=g).to(dtype), torch.randn(nkv, l, c, generator=g).to(dtype),
            torch.randn(nkv, l, c, generator=g).to(dtype))


def _compact(k, v, junk=False):
    """k [F, L, C], v [F, L, C] -> k, V^T [F, C, Lp] with Lp = round_up(L, 8); `junk`: NaN in the pad columns (the kernel must not
    let them reach the output: keys >= L carry P = 0 AND their values are zeroed at the fill)."""
    l = k.shape[1]
    lp = (l + 7) // 8 * 8
    vt = torch.full((v.shape[0], v.shape[2], lp), float("nan") if junk else 0.0, dtype=v.dtype)
    vt[:, :, :l] = v.transpose(1, 2)
    return k.to(DEV), vt.to(DEV)


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("shape", [(3, 300, 77, 3), (2, 256, 64, 2), (5, 97, 13, 1), (2, 1, 1, 1), (14, 1024, 77, 5), (2, 33, 65, 9),
                                   (3, 2100, 96, 2)], ids=lambda s: "n%d_s%d_l%d_h%d" % s)
def test_plain_call(dtype, shape, tuning):
    n, s, l, h = shape
    q, k, v = _inputs(n, n, s, l, h, dtype, seed=s + l)
    kc, vc = _compact(k, v, junk=True)
    o = ops.attn_fwd(q.to(DEV), kc, vc, h, l=l, mode="plain")
    assert ops.last_attn_variant() == "aid_attn_tx<d64,plain>"
    ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, 64 ** -0.5, "plain", False, None)
    assert torch.isfinite(o).all()
    assert rel_l2(to_np64(o), ref) < TOL[dtype] and worst(to_np64(o), ref) < WORST[dtype]
    assert torch.equal(ops.attn_fwd(q.to(DEV), kc, vc, h, l=l, mode="plain"), o)          # deterministic
    for tiles in (1, 5):                                        # any split of the rows over workgroups: the same bits
        tuning("ATTN_TX_TILES", tiles)
        assert torch.equal(ops.attn_fwd(q.to(DEV), kc, vc, h, l=l, mode="plain"), o)
    tuning("ATTN_TX_TILES", -1)
    tuning("ATTN_TX", 0)                                        # the program-order kernel on the same tensors
    kz, vz = _compact(k, v)
    o_old = ops.attn_fwd(q.to(DEV), kz, vz, h, l=l, mode="plain")
    assert "aid_attn_tx" not in ops.last_attn_variant() and rel_l2(to_np64(o), to_np64(o_old)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "pure"])
@pytest.mark.parametrize("l,s,riders,h", [(77, 300, 7, 3), (64, 200, 0, 2), (96, 96, 3, 1), (20, 520, 7, 5), (33, 70, 2, 2)],
                         ids=["l77", "l64", "l96", "l20", "l33"])
def test_outer_call_with_riders(dtype, fused, l, s, riders, h, tuning):
    """7 AID frames (one / two / three key segments: end points, interior rows with a coefficient of exactly 0 / 1, two-sided rows) and
    PLAIN riders in ONE launch; every frame against the oracle.  Spikes in the begin and in the end keys put the segments' maxima far
    apart: the combination weights 2^(m_seg - m_side) then span many orders of magnitude."""
    n = 7
    q, k, v = _inputs(n + riders, n + riders, s, l, h, dtype, seed=l + s + fused)
    k[0, min(5, l - 1)] = q[2, 7 % s] * 5.0
    k[n - 1, l // 2] = q[3, 11 % s] * 6.0
    coef = torch.from_numpy(O.beta_coefs(n, 3, 3)).float()
    coef[1], coef[5] = 0.0, 1.0
    cd = torch.cat([coef.to(dtype).float(), -torch.ones(riders)])
    kc, vc = _compact(k, v, junk=True)
    args = dict(l=l, mode="outer", fused=fused, coef=cd.to(DEV), begin=0, end=n - 1, n_plain=riders)
    o = ops.attn_fwd(q.to(DEV), kc, vc, h, **args)
    assert ops.last_attn_variant() == "aid_attn_tx<d64,outer>"
    q64, k64, v64 = to_np64(q), to_np64(k), to_np64(v)
    ref = O.attn_core(q64[:n], k64[:n], v64[:n], h, 64 ** -0.5, "outer", fused, coef.to(dtype).float().numpy())
    if riders:
        ref = np.concatenate([ref, O.attn_core(q64[n:], k64[n:], v64[n:], h, 64 ** -0.5, "plain", False, None)])
    assert torch.isfinite(o).all()
    for f in range(n + riders):
        assert rel_l2(to_np64(o[f]), ref[f]) < TOL[dtype], f
    assert worst(to_np64(o), ref) < WORST[dtype]
    assert torch.equal(ops.attn_fwd(q.to(DEV), kc, vc, h, **args), o)
    tuning("ATTN_TX", 0)
    kz, vz = _compact(k, v)
    o_old = ops.attn_fwd(q.to(DEV), kz, vz, h, **args)
    assert "aid_attn_tx" not in ops.last_attn_variant() and rel_l2(to_np64(o), to_np64(o_old)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
def test_shared_contexts_scales_and_prescaled_q(dtype, tuning):
    """The PAID layout: 14 frames over 6 distinct contexts (kv_map), begin / end are context rows; out_scale and a per-frame scale;
    q already multiplied by softmax_scale * log2(e) (what the fused q-projection hands over)."""
    n, riders, s, l, h = 7, 7, 260, 77, 4
    q, k, v = _inputs(n + riders, 6, s, l, h, dtype, seed=99)
    kv_map = torch.tensor([0, 1, 1, 1, 1, 1, 2, 3, 4, 4, 4, 4, 4, 5], dtype=torch.int32)
    coef = torch.from_numpy(O.beta_coefs(n, 50, 50)).float()
    coef[0], coef[-1] = 0, 1
    cd = torch.cat([coef.to(dtype).float(), -torch.ones(riders)])
    kc, vc = _compact(k, v)
    fs = torch.linspace(0.2, 1.5, n + riders)
    kw = dict(l=l, mode="outer", fused=True, coef=cd.to(DEV), begin=0, end=2, n_plain=riders, kv_map=kv_map.to(DEV))
    o = ops.attn_fwd(q.to(DEV), kc, vc, h, out_scale=0.7, frame_scale=fs.to(DEV), **kw)
    assert ops.last_attn_variant() == "aid_attn_tx<d64,outer>"
    q64, k64, v64 = to_np64(q), to_np64(k)[kv_map.long().numpy()], to_np64(v)[kv_map.long().numpy()]
    ref = np.concatenate([O.attn_core(q64[:n], k64[:n], v64[:n], h, 64 ** -0.5, "outer", True, coef.to(dtype).float().numpy()),
                          O.attn_core(q64[n:], k64[n:], v64[n:], h, 64 ** -0.5, "plain", False, None)])
    want = 0.7 * fs.numpy().astype(np.float64)[:, None, None] * ref
    for f in range(n + riders):
        assert rel_l2(to_np64(o[f]), want[f]) < TOL[dtype], f
    # pre-scaled q: the caller's q * (d^-0.5 log2 e), rounded to the storage type — the oracle sees the same rounded q
    qs = (q.float() * (64 ** -0.5 * 1.4426950408889634)).to(dtype)
    o2 = ops.attn_fwd(qs.to(DEV), kc, vc, h, q_prescaled=True, **kw)
    assert ops.last_attn_variant() == "aid_attn_tx<d64,outer>"
    qs64 = to_np64(qs)
    ref2 = np.concatenate([O.attn_core(qs64[:n], k64[:n], v64[:n], h, float(np.log(2.0)), "outer", True, coef.to(dtype).float().numpy()),
                           O.attn_core(qs64[n:], k64[n:], v64[n:], h, float(np.log(2.0)), "plain", False, None)])
    for f in range(n + riders):
        assert rel_l2(to_np64(o2[f]), ref2[f]) < TOL[dtype], f
    # an accumulating call (the IP-Adapter image branch) is not this kernel's: aid_attn_kernel adds in fp32 and rounds once
    base = torch.randn(n + riders, s, h * 64).to(dtype)
    o3 = ops.attn_fwd(q.to(DEV), kc, vc, h, l=l, mode="plain", kv_map=kv_map.to(DEV), out=base.to(DEV).clone(), accumulate=True,
                      out_scale=0.7, frame_scale=fs.to(DEV))
    assert "aid_attn_tx" not in ops.last_attn_variant()
    refp = O.attn_core(q64, k64, v64, h, 64 ** -0.5, "plain", False, None)
    assert rel_l2(to_np64(o3), to_np64(base) + 0.7 * fs.numpy().astype(np.float64)[:, None, None] * refp) < TOL[dtype]


@pytest.mark.parametrize("l", list(range(1, 20)) + [31, 32, 33, 48, 63, 64, 65, 76, 77, 78, 80, 88, 95, 96])
def test_every_kind_of_key_count(l):
    """Masking of the ragged last score tile and skipping of the empty ones: every remainder class, one to three tiles per segment."""
    dtype, n, s, h = torch.bfloat16, 3, 70, 2
    q, k, v = _inputs(n, n, s, l, h, dtype, seed=l)
    coef = torch.tensor([0.0, 0.4, 1.0])
    kc, vc = _compact(k, v, junk=True)
    o = ops.attn_fwd(q.to(DEV), kc, vc, h, l=l, mode="outer", fused=True, coef=coef.to(DEV), begin=0, end=2)
    assert ops.last_attn_variant() == "aid_attn_tx<d64,outer>"
    ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, 64 ** -0.5, "outer", True, coef.to(dtype).float().numpy())
    assert torch.isfinite(o).all() and rel_l2(to_np64(o), ref) < TOL[dtype] and worst(to_np64(o), ref) < WORST[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "pure"])
@pytest.mark.parametrize("l,s,riders,h", [(77, 300, 7, 3), (64, 200, 0, 2), (20, 130, 3, 5)], ids=["l77", "l64", "l20"])
def test_inner_call_with_riders(dtype, fused, l, s, riders, h, tuning):
    """INNER: ONE interpolated key segment (rows written by aid_lerp_kv; the end-point frame itself for a coefficient of exactly 0 / 1)
    beside the own keys under one softmax, PLAIN riders in the same launch; every frame against the oracle."""
    n = 7
    q, k, v = _inputs(n + riders, n + riders, s, l, h, dtype, seed=3 * l + s + fused)
    k[0, min(5, l - 1)] = q[2, 7 % s] * 5.0
    coef = torch.from_numpy(O.beta_coefs(n, 3, 3)).float()
    coef[1], coef[5] = 0.0, 1.0
    cd = torch.cat([coef.to(dtype).float(), -torch.ones(riders)])
    kc, vc = _compact(k, v)
    args = dict(l=l, mode="inner", fused=fused, coef=cd.to(DEV), begin=0, end=n - 1, n_plain=riders)
    o = ops.attn_fwd(q.to(DEV), kc, vc, h, **args)
    assert ops.last_attn_variant() == "aid_attn_tx<d64,inner>"
    q64, k64, v64 = to_np64(q), to_np64(k), to_np64(v)
    ref = O.attn_core(q64[:n], k64[:n], v64[:n], h, 64 ** -0.5, "inner", fused, coef.to(dtype).float().numpy())
    if riders:
        ref = np.concatenate([ref, O.attn_core(q64[n:], k64[n:], v64[n:], h, 64 ** -0.5, "plain", False, None)])
    assert torch.isfinite(o).all()
    for f in range(n + riders):
        assert rel_l2(to_np64(o[f]), ref[f]) < TOL[dtype], f
    assert worst(to_np64(o), ref) < WORST[dtype]
    assert torch.equal(ops.attn_fwd(q.to(DEV), kc, vc, h, **args), o)
    if riders:                      # a rider inside an INNER launch = the same frame in a PLAIN call, bit for bit
        op = ops.attn_fwd(q[n:].contiguous().to(DEV), kc[n:].contiguous(), vc[n:].contiguous(), h, l=l, mode="plain")
        assert torch.equal(o[n:], op)
    tuning("ATTN_TX", 0)
    o_old = ops.attn_fwd(q.to(DEV), kc, vc, h, **args)
    assert "aid_attn_tx" not in ops.last_attn_variant() and rel_l2(to_np64(o), to_np64(o_old)) < TOL[dtype]


def test_calls_that_stay_on_the_other_kernels():
    dtype, n, s, h = torch.bfloat16, 3, 64, 2
    q, k, v = _inputs(n, n, s, 97, h, dtype, seed=1)            # 97 keys: one more than the LDS regions hold
    kc, vc = _compact(k, v)
    ops.attn_fwd(q.to(DEV), kc, vc, h, l=97, mode="plain")
    assert "aid_attn_tx" not in ops.last_attn_variant()
    g = torch.Generator().manual_seed(3)                        # head dim 40 (SD1.5)
    q40, k40, v40 = (torch.randn(n, s, 80, generator=g).to(dtype) for _ in range(3))
    kc, vc = _compact(k40[:, :56], v40[:, :56])
    ops.attn_fwd(q40.to(DEV), kc, vc, 2, l=56, mode="plain")
    assert "aid_attn_tx" not in ops.last_attn_variant()


def test_sdxl_cross_attention_shapes_sampled_rows(tuning):
    """The two cross-attention shapes of the SDXL stack (14 frames, shared contexts): sampled rows of every frame against the oracle,
    the whole tensor against the program-order kernel, and 50 repetitions bit for bit."""
    dtype = torch.bfloat16
    for (s, h) in ((1024, 20), (4096, 10)):
        n, riders, l = 7, 7, 77
        q, k, v = _inputs(n + riders, 6, s, l, h, dtype, seed=s)
        kv_map = torch.tensor([0, 1, 1, 1, 1, 1, 2, 3, 4, 4, 4, 4, 4, 5], dtype=torch.int32)
        coef = torch.from_numpy(O.beta_coefs(n, 50, 50)).float()
        coef[0], coef[-1] = 0, 1
        cd = torch.cat([coef.to(dtype).float(), -torch.ones(riders)])
        kc, vc = _compact(k, v)
        for mode in ("outer", "plain"):
            kw = dict(l=l, mode=mode, kv_map=kv_map.to(DEV))
            if mode == "outer":
                kw.update(fused=True, coef=cd.to(DEV), begin=0, end=2, n_plain=riders)
            tuning("ATTN_TX", -1)
            o = ops.attn_fwd(q.to(DEV), kc, vc, h, **kw)
            assert "aid_attn_tx" in ops.last_attn_variant() and torch.isfinite(o).all()
            tuning("ATTN_TX", 0)
            o_old = ops.attn_fwd(q.to(DEV), kc, vc, h, **kw)
            assert "aid_attn_tx" not in ops.last_attn_variant()
            assert rel_l2(to_np64(o), to_np64(o_old)) < TOL[dtype], (s, mode)
            tuning("ATTN_TX", -1)
            rows = torch.tensor([0, 31, 32, 255, 256, 700, s - 1])
            q64 = to_np64(q[:, rows])
            k64, v64 = to_np64(k)[kv_map.long().numpy()], to_np64(v)[kv_map.long().numpy()]
            if mode == "outer":
                ref = np.concatenate([O.attn_core(q64[:n], k64[:n], v64[:n], h, 64 ** -0.5, "outer", True, coef.to(dtype).float().numpy()),
                                      O.attn_core(q64[n:], k64[n:], v64[n:], h, 64 ** -0.5, "plain", False, None)])
            else:
                ref = O.attn_core(q64, k64, v64, h, 64 ** -0.5, "plain", False, None)
            got = to_np64(o[:, rows.to(DEV)])
            for f in range(n + riders):
                assert rel_l2(got[f], ref[f]) < TOL[dtype], (s, mode, f)
            bad = torch.zeros((), dtype=torch.int64, device=DEV)
            for _ in range(50):
                bad += (ops.attn_fwd(q.to(DEV), kc, vc, h, **kw).view(torch.int16) != o.view(torch.int16)).sum()
            assert int(bad) == 0


@pytest.mark.parametrize("kind", ["outer", "plain"])
def test_processor_path(kind):
    """The processors' cross-attention call (cached text keys / values, compact layout) lands on this kernel; against the oracle."""
    from aid_amd import processors as P
    dtype, n, s, heads, l, cc = torch.bfloat16, 5, 300, 3, 77, 128
    c = heads * 64
    g = torch.Generator().manual_seed(8)
    attn = aid_amd.AttnShim(c, heads, cc, dtype=dtype, device=DEV)
    x = torch.randn(n, s, c, generator=g).to(dtype).to(DEV)
    ctx = torch.randn(n, l, cc, generator=g).to(dtype).to(DEV)
    proc = aid_amd.HipAttnProcessor() if kind == "plain" else aid_amd.OuterInterpolatedAttnProcessor(size=n, is_fused=True, alpha=3, beta=3)
    w = O.AttnWeights(to_np64(attn.to_q.weight), to_np64(attn.to_k.weight), to_np64(attn.to_v.weight),
                      to_np64(attn.to_out[0].weight), to_np64(attn.to_out[0].bias), heads)
    ref = O.plain_attention(to_np64(x), to_np64(ctx), w) if kind == "plain" else \
        O.outer_attention(to_np64(x), to_np64(ctx), w, to_np64(proc.coef.to(dtype)), True)
    P.clear_weight_caches()
    y = proc(attn, x, encoder_hidden_states=ctx)
    assert "aid_attn_tx" in ops.last_attn_variant(), ops.last_attn_variant()
    assert torch.equal(y, proc(attn, x, encoder_hidden_states=ctx)) and rel_l2(to_np64(y), ref) < TOL[dtype]
