"""GPU: the short-stream ping-pong kernel (csrc/aid_attn_xs.hip: d = 64, keys / values padded to whole 64-key tiles — the cached text
keys of cross-attention, L = 77; opt-in through the development knob ATTN_V2 = 1, it measures slower than the default kernel on the
stacks' launches: profiles/r04_attn_notes.txt) against the fp64 oracle and against the program-order kernel on the same call.  PLAIN calls, fused
and pure OUTER calls with PLAIN riders, interior rows with end-point coefficients, shared contexts (kv_map), accumulate / scales
(the IP-Adapter call shape), every kind of key count (1 .. 4 tiles, ragged and whole), ragged query counts, forced reference raises,
the SDXL layer shapes, and the processor path with the padded text K / V cache."""
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
    q = torch.randn(n, s, c, generator=g).to(dtype)
    k = torch.randn(nkv, l, c, generator=g).to(dtype)
    v = torch.randn(nkv, l, c, generator=g).to(dtype)
    return q, k, v


def _padded(k, v):
    """k [F, L, C], v [F, L, C] -> the padded layout of ops.project_kv(padded=True): k [F, Lt, C], V^T [F, C, Lt], zero beyond L."""
    f, l, c = k.shape
    lt = (l + 63) // 64 * 64
    kp = torch.zeros(f, lt, c, dtype=k.dtype)
    kp[:, :l] = k
    vp = torch.zeros(f, c, lt, dtype=v.dtype)
    vp[:, :, :l] = v.transpose(1, 2)
    return kp.to(DEV), vp.to(DEV)


def _compact(k, v):
    l = k.shape[1]
    lp = (l + 7) // 8 * 8
    vt = torch.zeros(v.shape[0], v.shape[2], lp, dtype=v.dtype)
    vt[:, :, :l] = v.transpose(1, 2)
    return k.to(DEV), vt.to(DEV)


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("shape", [(3, 300, 77, 3), (2, 256, 64, 2), (5, 97, 13, 1), (4, 512, 128, 2), (2, 1, 1, 1), (3, 700, 200, 2),
                                   (14, 1024, 77, 5), (2, 33, 65, 9)], ids=lambda s: "n%d_s%d_l%d_h%d" % s)
def test_plain_call(dtype, shape, tuning):
    tuning("ATTN_V2", 1)
    n, s, l, h = shape
    q, k, v = _inputs(n, n, s, l, h, dtype, seed=s + l)
    kp, vp = _padded(k, v)
    o = ops.attn_fwd(q.to(DEV), kp, vp, h, l=l, mode="plain", kv_padded=True)
    assert ops.last_attn_variant() == "aid_attn_xs<d64>"
    ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, 64 ** -0.5, "plain", False, None)
    assert torch.isfinite(o).all()
    assert rel_l2(to_np64(o), ref) < TOL[dtype] and worst(to_np64(o), ref) < WORST[dtype]
    assert torch.equal(ops.attn_fwd(q.to(DEV), kp, vp, h, l=l, mode="plain", kv_padded=True), o)          # deterministic
    tuning("ATTN_V2", 0)                                        # the program-order kernel on the same (padded) tensors
    o_old = ops.attn_fwd(q.to(DEV), kp, vp, h, l=l, mode="plain", kv_padded=True)
    assert "aid_attn_xs" not in ops.last_attn_variant() and rel_l2(to_np64(o), to_np64(o_old)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "pure"])
@pytest.mark.parametrize("l,s,riders,h", [(77, 300, 7, 3), (64, 200, 0, 2), (130, 96, 3, 1), (20, 520, 7, 5)],
                         ids=["l77", "l64", "l130", "l20"])
def test_outer_call_with_riders(dtype, fused, l, s, riders, h, tuning):
    """7 AID frames (one / two / three key segments: end points, interior rows with a coefficient of exactly 0 / 1, two-sided rows) and
    PLAIN riders in ONE launch of the short-stream kernel; every frame against the oracle."""
    tuning("ATTN_V2", 1)
    n = 7
    q, k, v = _inputs(n + riders, n + riders, s, l, h, dtype, seed=l + s + fused)
    k[0, min(5, l - 1)] = q[2, 7 % s] * 5.0                     # a spike in the begin keys and one in the end keys: forced raises
    k[n - 1, l // 2] = q[3, 11 % s] * 6.0
    coef = torch.from_numpy(O.beta_coefs(n, 3, 3)).float()
    coef[1], coef[5] = 0.0, 1.0
    cd = torch.cat([coef.to(dtype).float(), -torch.ones(riders)])
    kp, vp = _padded(k, v)
    args = dict(l=l, mode="outer", fused=fused, coef=cd.to(DEV), begin=0, end=n - 1, n_plain=riders, kv_padded=True)
    o = ops.attn_fwd(q.to(DEV), kp, vp, h, **args)
    assert ops.last_attn_variant() == "aid_attn_xs<d64,outer>"
    q64, k64, v64 = to_np64(q), to_np64(k), to_np64(v)
    ref = O.attn_core(q64[:n], k64[:n], v64[:n], h, 64 ** -0.5, "outer", fused, coef.to(dtype).float().numpy())
    if riders:
        ref = np.concatenate([ref, O.attn_core(q64[n:], k64[n:], v64[n:], h, 64 ** -0.5, "plain", False, None)])
    assert torch.isfinite(o).all()
    for f in range(n + riders):
        assert rel_l2(to_np64(o[f]), ref[f]) < TOL[dtype], f
    assert worst(to_np64(o), ref) < WORST[dtype]
    assert torch.equal(ops.attn_fwd(q.to(DEV), kp, vp, h, **args), o)
    tuning("ATTN_V2", 0)
    o_old = ops.attn_fwd(q.to(DEV), kp, vp, h, **args)
    assert "aid_attn_xs" not in ops.last_attn_variant() and rel_l2(to_np64(o), to_np64(o_old)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
def test_shared_contexts_accumulate_and_scales(dtype, tuning):
    """The PAID layout: 14 frames over 6 distinct contexts (kv_map), begin / end are context rows; then the IP-Adapter call shape:
    accumulate into an existing output with out_scale and a per-frame scale."""
    tuning("ATTN_V2", 1)
    n, riders, s, l, h = 7, 7, 260, 77, 4
    q, k, v = _inputs(n + riders, 6, s, l, h, dtype, seed=99)
    kv_map = torch.tensor([0, 1, 1, 1, 1, 1, 2, 3, 4, 4, 4, 4, 4, 5], dtype=torch.int32)
    coef = torch.from_numpy(O.beta_coefs(n, 50, 50)).float()
    coef[0], coef[-1] = 0, 1
    cd = torch.cat([coef.to(dtype).float(), -torch.ones(riders)])
    kp, vp = _padded(k, v)
    o = ops.attn_fwd(q.to(DEV), kp, vp, h, l=l, mode="outer", fused=True, coef=cd.to(DEV), begin=0, end=2, n_plain=riders,
                     kv_map=kv_map.to(DEV), kv_padded=True)
    assert ops.last_attn_variant() == "aid_attn_xs<d64,outer>"
    q64, k64, v64 = to_np64(q), to_np64(k)[kv_map.long().numpy()], to_np64(v)[kv_map.long().numpy()]
    ref = np.concatenate([O.attn_core(q64[:n], k64[:n], v64[:n], h, 64 ** -0.5, "outer", True, coef.to(dtype).float().numpy()),
                          O.attn_core(q64[n:], k64[n:], v64[n:], h, 64 ** -0.5, "plain", False, None)])
    for f in range(n + riders):
        assert rel_l2(to_np64(o[f]), ref[f]) < TOL[dtype], f
    base = torch.randn(n + riders, s, h * 64).to(dtype)
    fs = torch.linspace(0.2, 1.5, n + riders)
    o2 = ops.attn_fwd(q.to(DEV), kp, vp, h, l=l, mode="plain", kv_map=kv_map.to(DEV), out=base.to(DEV).clone(), accumulate=True,
                      out_scale=0.7, frame_scale=fs.to(DEV), kv_padded=True)
    assert ops.last_attn_variant() == "aid_attn_xs<d64>"
    refp = O.attn_core(q64, k64, v64, h, 64 ** -0.5, "plain", False, None)
    want = to_np64(base) + 0.7 * fs.numpy().astype(np.float64)[:, None, None] * refp
    assert rel_l2(to_np64(o2), want) < TOL[dtype]


@pytest.mark.parametrize("l", list(range(1, 20)) + [31, 32, 33, 48, 63, 64, 65, 80, 96, 127, 128, 129, 191, 192, 255, 256])
def test_every_kind_of_key_count(l, tuning):
    """Masking of the ragged last tile by 16-key groups: every remainder class, one to four tiles per segment."""
    tuning("ATTN_V2", 1)
    dtype, n, s, h = torch.bfloat16, 3, 70, 2
    q, k, v = _inputs(n, n, s, l, h, dtype, seed=l)
    coef = torch.tensor([0.0, 0.4, 1.0])
    kp, vp = _padded(k, v)
    o = ops.attn_fwd(q.to(DEV), kp, vp, h, l=l, mode="outer", fused=True, coef=coef.to(DEV), begin=0, end=2, kv_padded=True)
    assert ops.last_attn_variant() == "aid_attn_xs<d64,outer>"
    ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, 64 ** -0.5, "outer", True, coef.to(dtype).float().numpy())
    assert torch.isfinite(o).all() and rel_l2(to_np64(o), ref) < TOL[dtype] and worst(to_np64(o), ref) < WORST[dtype]


def test_sdxl_cross_attention_shapes_sampled_rows(tuning):
    """The two cross-attention shapes of the SDXL stack (14 frames, shared contexts): sampled rows of every frame against the oracle,
    the whole tensor against the program-order kernel."""
    tuning("ATTN_V2", 1)
    dtype = torch.bfloat16
    for (s, h) in ((1024, 20), (4096, 10)):
        n, riders, l = 7, 7, 77
        q, k, v = _inputs(n + riders, 6, s, l, h, dtype, seed=s)
        kv_map = torch.tensor([0, 1, 1, 1, 1, 1, 2, 3, 4, 4, 4, 4, 4, 5], dtype=torch.int32)
        coef = torch.from_numpy(O.beta_coefs(n, 50, 50)).float()
        coef[0], coef[-1] = 0, 1
        cd = torch.cat([coef.to(dtype).float(), -torch.ones(riders)])
        kp, vp = _padded(k, v)
        kc, vc = _compact(k, v)
        for mode in ("outer", "plain"):
            kw = dict(l=l, mode=mode, kv_map=kv_map.to(DEV))
            if mode == "outer":
                kw.update(fused=True, coef=cd.to(DEV), begin=0, end=2, n_plain=riders)
            o = ops.attn_fwd(q.to(DEV), kp, vp, h, kv_padded=True, **kw)
            assert "aid_attn_xs" in ops.last_attn_variant() and torch.isfinite(o).all()
            o_old = ops.attn_fwd(q.to(DEV), kc, vc, h, **kw)                   # compact layout: not eligible
            assert "aid_attn_xs" not in ops.last_attn_variant()
            assert rel_l2(to_np64(o), to_np64(o_old)) < TOL[dtype], (s, mode)
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
            for _ in range(50):                                 # repetition: rare wrong tiles of a mis-counted wait show up here
                bad += (ops.attn_fwd(q.to(DEV), kp, vp, h, kv_padded=True, **kw).view(torch.int16) != o.view(torch.int16)).sum()
            assert int(bad) == 0


def test_project_kv_padded_layout():
    dtype, f, l, cc, c = torch.bfloat16, 3, 77, 256, 192
    g = torch.Generator().manual_seed(1)
    e = torch.randn(f, l, cc, generator=g).to(dtype).to(DEV)
    wk = (torch.randn(c, cc, generator=g) / 16).to(dtype).to(DEV)
    wv = (torch.randn(c, cc, generator=g) / 16).to(dtype).to(DEV)
    k, vt = ops.project_kv(e, wk, wv)
    kp, vtp = ops.project_kv(e, wk, wv, padded=True)
    assert tuple(kp.shape) == (f, 128, c) and tuple(vtp.shape) == (f, c, 128)
    assert torch.equal(kp[:, :l], k) and torch.equal(vtp[:, :, :l], vt[:, :, :l])
    assert not kp[:, l:].any() and not vtp[:, :, l:].any()


@pytest.mark.parametrize("kind", ["outer", "plain"])
def test_processor_path_with_the_padded_text_cache(kind, tuning):
    """ATTN_V2 = 1: the processors project the step-invariant text keys / values into the tile-padded layout (AidProcessorArgs.kv_cached_lt)
    and the cross-attention call runs on the short-stream kernel; default: compact layout, program-order kernel.  Both against the oracle."""
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
    for v2, want in ((1, True), (-1, False)):
        tuning("ATTN_V2", v2)
        P.clear_weight_caches()
        y = proc(attn, x, encoder_hidden_states=ctx)
        assert ("aid_attn_xs" in ops.last_attn_variant()) == want, (v2, ops.last_attn_variant())
        y2 = proc(attn, x, encoder_hidden_states=ctx)                       # cache hit
        assert torch.equal(y, y2) and rel_l2(to_np64(y), ref) < TOL[dtype]
