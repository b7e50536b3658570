"""GPU parity tests: the HIP path (through the C ABI) vs the oracle and the committed goldens.
Run on the MI355X box with ``pytest -m gpu``."""
import numpy as np
import pytest
import torch

import cases as C
from oracle import aid_oracle as O
from util import TOL, TOL_GEMM, WORST, make_attn, rel_l2, rounded, to_np64, worst

pytestmark = pytest.mark.gpu

import aid_amd  # noqa: E402
from aid_amd.loop import AidDenoiseLoop, install_sequence_processors
from aid_amd import ops  # noqa: E402

DEV = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16]
ids_dt = lambda d: str(d).split(".")[-1]  # noqa: E731


def test_native_library_is_what_runs():
    lib = aid_amd._lib.load()
    import ctypes
    ncu, clk, arch = ctypes.c_int(), ctypes.c_int(), ctypes.create_string_buffer(32)
    assert lib.aid_device_info(ctypes.byref(ncu), ctypes.byref(clk), arch) == 0
    assert arch.value.decode().startswith("gfx950"), arch.value
    assert ncu.value == 256


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("mnk", [(1, 8, 8), (129, 320, 320), (231, 640, 768), (1000, 1280, 2048), (4096, 640, 640),
                                 (257, 324, 72)])
def test_gemm_nt_vs_fp64(dtype, mnk):
    m, n, k = mnk
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    a = torch.randn(m, k, generator=g).to(dtype)
    b = (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype)
    bias = torch.randn(n, generator=g).to(dtype)
    ref = to_np64(a) @ to_np64(b).T
    y = ops.linear(a.to(DEV), b.to(DEV))
    assert rel_l2(to_np64(y), ref) < TOL_GEMM[dtype] and worst(to_np64(y), ref) < WORST[dtype]
    yb = ops.linear(a.to(DEV), b.to(DEV), bias.to(DEV))
    assert rel_l2(to_np64(yb), ref + to_np64(bias)) < TOL_GEMM[dtype] and worst(to_np64(yb), ref + to_np64(bias)) < WORST[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("mnk,engine,tri", [((33000, 512, 640), "pingpong288", -1),           # 230 tiles of 288 rows: one round
                                            ((33000, 512, 640), "pingpong256+tail128", 0),   # 258 big tiles: 256 + 2 cut in four
                                            ((33001, 504, 640), "pingpong288", -1),           # ragged m and n edges
                                            ((33001, 504, 640), "pingpong256+tail128", 0),
                                            ((14336, 1280, 1280), "pingpong288", -1),         # SDXL out projection: 250 tiles
                                            ((14336, 1280, 1280), "pingpong256+tail128", 0),
                                            ((14336, 3840, 1280), "pingpong288", -1),         # 750 tiles = 2.93 rounds
                                            ((3584, 3840, 1280), "pingpong256", -1),          # 210 tiles: one partial round
                                            ((3584, 3840, 1280), "pingpong288", 1),           # forced: 13 row panels, the last 128 rows
                                            ((57344, 320, 320), "lockstep128", -1),           # short K loop (row-stationary engine off)
                                            ((57344, 320, 320), "rowstat320", -1),            # ... and what the launch gets by default
                                            ((57344, 640, 640), "rowstat640", -1),            # SDXL level 1 out projection
                                            ((1000, 1280, 2048), "lockstep128x4", -1),       # too few tiles for the big engines: alone on its CU, long K loop -> deep ring
                                            ((448, 1280, 768), "lockstep128", -1)])           # ... short K loop: the two-stage ring
def test_gemm_engine_selection_and_parity(dtype, mnk, engine, tri, tuning):
    """The k % 64 == 0 engines against fp64 on sampled rows (every row panel edge included), and the cost model
    sends each shape to the engine the stack measurements favour (profiles/r01_gemm_variants.txt, profiles/r03_gemm_notes.txt)."""
    tuning("GEMM_TRI", tri)
    tuning("GEMM_RS", -1 if engine.startswith("rowstat") else 0)
    m, n, k = mnk
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(dtype)
    b = (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype)
    bias = torch.randn(n, generator=g).to(dtype)
    y = ops.linear(a.to(DEV), b.to(DEV), bias.to(DEV))
    assert ops.last_gemm_variant() == engine
    rows = torch.unique(torch.cat([torch.arange(0, m, 997), torch.arange(255, m, 256), torch.arange(256, m, 256),
                                   torch.arange(287, m, 288), torch.arange(288, m, 288), torch.arange(256, m, 288),
                                   torch.tensor([m - 1])]))
    ref = to_np64(a[rows]) @ to_np64(b).T + to_np64(bias)
    assert rel_l2(to_np64(y[rows.to(DEV)]), ref) < TOL_GEMM[dtype]
    # every output element was written exactly once with a finite value of the right magnitude
    assert torch.isfinite(y).all()
    col = (y.float() - bias.to(DEV).float()).square().mean(dim=0).sqrt().cpu()      # a b^T has unit variance
    assert (col > 0.8).all() and (col < 1.25).all()


def test_grouped_gemm_randomized_shapes_both_engines(tuning):
    """Seeded sweep over grouped launches (1-3 problems, batched operands, ragged m / n, equal and different K loops, both
    engines forced in turn where they apply) against fp32 torch matmuls of the same rounded operands on the GPU —
    a second, independent reference next to the fp64 oracle tests; catches tile-mapping mistakes at sizes the
    oracle is too slow for."""
    rs = np.random.RandomState(950)
    engines = set()
    for it in range(36):
        dtype = DTYPES[it % 2]
        k_common = int(rs.choice([64, 128, 320, 640, 1280]))
        nprob = int(rs.randint(1, 4))
        same_k = bool(rs.randint(0, 2)) or nprob == 1
        big = it % 3 == 0                                   # every third launch is large enough for the 256 x 256 engine
        tuning("GEMM_TRI", [-1, 0, 1][(it // 3) % 3])       # big tiles of 256 rows / of 288 rows / the cost model's choice
        probs, refs = [], []
        for p in range(nprob):
            k = k_common if same_k else int(rs.choice([64, 192, 768, 2048]))
            batch = int(rs.choice([1, 1, 3]))
            m = int(rs.randint(2000, 40000)) if (big and batch == 1) else int(rs.randint(1, 1500))
            n = int(rs.choice([8, 72, 128, 320, 504, 640, 1280]))
            a = (torch.randn(batch, m, k, device=DEV) * 0.5).to(dtype)
            b = (torch.randn(n, k, device=DEV) / k ** 0.5).to(dtype)
            bias = torch.randn(n, device=DEV).to(dtype) if rs.randint(0, 2) else None
            c = torch.full((batch, m, n), float("nan"), device=DEV, dtype=dtype)
            probs.append(dict(a=a, b=b, c=c, bias=bias, m=m, n=n, k=k, lda=k, ldb=k, ldc=n, batch=batch,
                              stride_a=m * k, stride_b=0, stride_c=m * n))
            ref = a.float() @ b.float().t()
            refs.append(ref if bias is None else ref + bias.float())
        ops.gemm_nt(probs)
        engines.add(ops.last_gemm_variant())
        for p, ref in zip(probs, refs):
            got = p["c"].float()
            assert torch.isfinite(got).all(), (it, ops.last_gemm_variant())
            err = ((got - ref).norm() / ref.norm()).item()
            assert err < TOL_GEMM[dtype], (it, ops.last_gemm_variant(), p["m"], p["n"], p["k"], p["batch"], err)
    assert {"lockstep128", "pingpong256", "pingpong288"} <= {e.split("+")[0] for e in engines}, engines


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("shape", [(3, 77, 96, 80), (7, 200, 64, 128), (2, 5, 8, 40)])
def test_kv_projection_writes_transposed_values_and_zero_padding(dtype, shape):
    f, l, cc, c = shape
    g = torch.Generator().manual_seed(l)
    e = torch.randn(f, l, cc, generator=g).to(dtype)
    wk = (torch.randn(c, cc, generator=g) / cc ** 0.5).to(dtype)
    wv = (torch.randn(c, cc, generator=g) / cc ** 0.5).to(dtype)
    k, vt = ops.project_kv(e.to(DEV), wk.to(DEV), wv.to(DEV))
    assert vt.shape == (f, c, (l + 7) // 8 * 8)
    assert rel_l2(to_np64(k), to_np64(e) @ to_np64(wk).T) < TOL_GEMM[dtype]
    assert rel_l2(to_np64(vt[:, :, :l]), (to_np64(e) @ to_np64(wv).T).transpose(0, 2, 1)) < TOL_GEMM[dtype]
    lp4 = (l + 3) // 4 * 4
    assert torch.isfinite(vt[:, :, :l].float()).all()


# ------------------------------------------------------------------------------------------------
# attention core vs oracle
# ------------------------------------------------------------------------------------------------
def _core_inputs(n, s, l, h, d, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    c = h * d
    q = torch.randn(n, s, c, generator=g).to(dtype)
    k = torch.randn(n, l, c, generator=g).to(dtype)
    v = torch.randn(n, l, c, generator=g).to(dtype)
    lp = (l + 7) // 8 * 8
    vt = torch.zeros(n, c, lp, dtype=dtype)
    vt[:, :, :l] = v.transpose(1, 2)
    return q, k, v, vt


def _coef(n):
    return torch.tensor([0.0, 0.3, 1.0]) if n == 3 else torch.from_numpy(O.beta_coefs(n, 3, 3))


MODES = [("plain", False), ("inner", False), ("inner", True), ("outer", False), ("outer", True)]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("d", [40, 64, 80, 160])
@pytest.mark.parametrize("shape", [(3, 40, 77, 2), (7, 200, 200, 2), (3, 33, 130, 1), (5, 1, 1, 2), (3, 700, 64, 8)],
                         ids=lambda s: "n%d_s%d_l%d_h%d" % s)
def test_attention_core_all_modes(dtype, d, shape):
    n, s, l, h = shape
    q, k, v, vt = _core_inputs(n, s, l, h, d, dtype, seed=d * 1000 + s)
    coef = _coef(n)
    for mode, fused in MODES:
        o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode=mode, fused=fused, coef=coef.to(DEV))
        ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, d ** -0.5, mode, fused, coef.numpy())
        err = rel_l2(to_np64(o), ref)
        assert err < TOL[dtype] and worst(to_np64(o), ref) < WORST[dtype], (mode, fused, ops.last_attn_variant(), err)


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("d", [40, 64, 80])
@pytest.mark.parametrize("resident", ["0", "1"], ids=["streaming", "resident"])
def test_attention_every_ragged_key_count(dtype, d, resident, tuning):
    """Every number of valid keys in the last (ragged) tile, on the streaming kernel and on the resident-segment
    kernel (which serves l <= 96 by default).  16 < l % 64 <= 32 is the case a branchy ragged tile got wrong on the
    resident kernel: hipcc left too few wait states between the last MFMA of a score block and the first VALU read of it
    on the taken path (keys 22 / 30 of the tile lost their last k-step) — the tile is straight-line now."""
    tuning("ATTN_RES", int(resident))
    tuning("ATTN_TX", 0)            # (d = 64 PLAIN / OUTER calls of <= 96 keys default to the text-key kernel: tests/test_hip_attn_tx.py)
    n, s, h = 3, 48, 2
    coef = _coef(n)
    for l in (1, 7, 8, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 77, 80, 88, 95, 96, 97, 120, 128, 160):
        q, k, v, vt = _core_inputs(n, s, l, h, d, dtype, seed=l)
        for mode, fused in MODES:
            o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode=mode, fused=fused, coef=coef.to(DEV))
            name = ops.last_attn_variant()
            assert ("res" in name) == (resident == "1" and l <= 96 and d <= 80), (l, name)
            ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, d ** -0.5, mode, fused, coef.numpy())
            err = rel_l2(to_np64(o), ref)
            assert err < TOL[dtype] and worst(to_np64(o), ref) < WORST[dtype], (l, mode, fused, name, err)


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("knob,d,expect", [("AID_ATTN_NW=8", 64, "nw8"), ("AID_ATTN_NW=8", 40, "nw8"), ("AID_ATTN_NW=8", 80, "nw8"),
                                           ("AID_ATTN_PIPE=1", 40, "pipe"), ("AID_ATTN_PIPE=1", 64, "pipe"),
                                           ("AID_ATTN_QB=2", 40, "qb2"), ("AID_ATTN_QB=1", 40, "nw4"),
                                           ("AID_ATTN_ORDER=0", 64, "nw4")])
def test_attention_variants_behind_the_development_knobs(dtype, knob, d, expect, tuning):
    """The kernel variants that are built but not (or not everywhere) the default — eight-wave workgroups, the
    software-pipelined loop, 64 rows per wave, the plain XCD order — against the oracle at a shape with several full tiles, a
    ragged one and riders, so they stay correct while the defaults move (profiles/r02_attn_notes.txt has their timings)."""
    k_, v_ = knob.split("=")
    tuning(k_, int(v_))
    n, s, l, h = 5, 300, 330, 2
    q, k, v, vt = _core_inputs(2 * n, s, l, h, d, dtype, seed=d + len(knob))
    coef = torch.cat([_coef(n), -torch.ones(n)])
    seen = set()
    for mode, fused in MODES:
        o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode=mode, fused=fused,
                         coef=None if mode == "plain" else coef.to(DEV), begin=0, end=n - 1,
                         n_plain=0 if mode == "plain" else n)
        seen.add(ops.last_attn_variant())
        q64, k64, v64 = to_np64(q), to_np64(k), to_np64(v)
        if mode == "plain":
            ref = O.attn_core(q64, k64, v64, h, d ** -0.5, "plain", False, None)
        else:
            ref = np.concatenate([O.attn_core(q64[:n], k64[:n], v64[:n], h, d ** -0.5, mode, fused, coef[:n].numpy()),
                                  O.attn_core(q64[n:], k64[n:], v64[n:], h, d ** -0.5, "plain", False, None)])
        err = rel_l2(to_np64(o), ref)
        assert err < TOL[dtype] and worst(to_np64(o), ref) < WORST[dtype], (knob, mode, fused, ops.last_attn_variant(), err)
    assert any(expect in nm for nm in seen), (knob, seen)


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
def test_attention_core_sharded_endpoints_accumulate_and_maps(dtype):
    """begin/end != (0, N-1), kv_map, frame_scale, out_scale and accumulate (what the IP processors and
    the frame-sharded multi-GPU layout use)."""
    n, s, l, h, d = 4, 96, 50, 2, 64
    q, k, v, vt = _core_inputs(n, s, l, h, d, dtype, seed=5)
    coef = torch.tensor([0.2, 0.0, 1.0, 0.7])
    o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="outer", fused=True, coef=coef.to(DEV),
                     begin=1, end=2)
    ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, d ** -0.5, "outer", True, coef.numpy(), begin=1, end=2)
    assert rel_l2(to_np64(o), ref) < TOL[dtype]
    # kv_map: frames 0..3 use kv rows [2, 0, 0, 1]
    kv_map = torch.tensor([2, 0, 0, 1], dtype=torch.int32)
    o2 = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="plain", kv_map=kv_map.to(DEV))
    ref2 = O.attn_core(to_np64(q), to_np64(k)[kv_map.numpy()], to_np64(v)[kv_map.numpy()], h, d ** -0.5, "plain",
                       False, None)
    assert rel_l2(to_np64(o2), ref2) < TOL[dtype]
    # accumulate with out_scale and per-frame scale on top of an existing tensor
    base = torch.randn(n, s, h * d).to(dtype)
    fs = torch.tensor([0.5, 0.0, 1.0, 2.0])
    o3 = base.clone().to(DEV)
    ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="plain", out=o3, accumulate=True, out_scale=0.6,
                 frame_scale=fs.to(DEV))
    plain = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, d ** -0.5, "plain", False, None)
    ref3 = to_np64(base) + 0.6 * fs.numpy().reshape(-1, 1, 1) * plain
    assert rel_l2(to_np64(o3), ref3) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
def test_online_softmax_rescale_is_exercised(dtype):
    """Force the running-max update late in the key sequence (a spiked key in the last tile and one in
    the middle) — the data-dependent rescale branch never fires on bounded random data."""
    n, s, l, h, d = 3, 64, 300, 1, 64
    q, k, v, vt = _core_inputs(n, s, l, h, d, dtype, seed=11)
    k[:, 150] = q[:, 3] * 4.0          # large score for query 3 in tile 2
    k[:, 299] = q[:, 7] * 6.0          # even larger for query 7 in the last tile
    for mode, fused in MODES:
        o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode=mode, fused=fused, coef=_coef(n).to(DEV))
        ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, d ** -0.5, mode, fused, _coef(n).numpy())
        assert np.isfinite(to_np64(o)).all()
        assert rel_l2(to_np64(o), ref) < TOL[dtype], (mode, fused)


def test_outer_equals_inner_when_endpoints_coincide():
    """If K/V of the two end-point frames are identical every mode collapses to attention against
    [own;] that frame."""
    dtype = torch.float16
    n, s, l, h, d = 4, 80, 90, 2, 40
    q, k, v, vt = _core_inputs(n, s, l, h, d, dtype, seed=3)
    k[-1], v[-1], vt[-1] = k[0], v[0], vt[0]
    coef = torch.tensor([0.0, 0.25, 0.6, 1.0]).to(DEV)
    for fused in (False, True):
        oo = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="outer", fused=fused, coef=coef)
        oi = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="inner", fused=fused, coef=coef)
        assert rel_l2(to_np64(oo), to_np64(oi)) < 1.5e-3


def test_bitwise_deterministic():
    q, k, v, vt = _core_inputs(7, 512, 512, 4, 64, torch.bfloat16, seed=1)
    coef = _coef(7).to(DEV)
    a = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), 4, l=512, mode="outer", fused=True, coef=coef)
    b = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), 4, l=512, mode="outer", fused=True, coef=coef)
    assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------
# whole processor calls vs the committed goldens (reference outputs) and the oracle
# ------------------------------------------------------------------------------------------------
TEXT = C.load_fixture("text_goldens.npz")
IPG = C.load_fixture("ip_goldens.npz")


def _text_proc(case):
    if case.mode == "plain":
        return aid_amd.HipAttnProcessor()
    cls = aid_amd.OuterInterpolatedAttnProcessor if case.mode.endswith("outer") else aid_amd.InnerInterpolatedAttnProcessor
    return cls(t=case.t, size=case.n, is_fused=case.mode.startswith("fused"), alpha=case.alpha, beta=case.beta)


# float32 storage (AID_DTYPE_F32): the HIP path against the reference's OWN float32 outputs — no storage rounding in between, so the
# bound is rounding noise of two different fp32 summation orders (measured <= 2e-6), not a storage-type tolerance
TOL_F32 = 1e-5


@pytest.mark.parametrize("dtype", DTYPES + [torch.float32], ids=ids_dt)
@pytest.mark.parametrize("case", C.TEXT_CASES, ids=lambda c: c.name)
def test_text_processors_vs_reference_goldens(case, dtype):
    inp = C.text_inputs(case)
    attn = make_attn(aid_amd, inp, case.heads, case.cc if case.cross else None, dtype, DEV)
    x = torch.from_numpy(inp["x"]).to(dtype).to(DEV)
    ctx = torch.from_numpy(inp["ctx"]).to(dtype).to(DEV) if case.cross else None
    mask = torch.from_numpy(inp["mask"]).to(dtype).to(DEV) if case.mask is not None else None     # additive [N, 1 | S, L]
    y = _text_proc(case)(attn, x, encoder_hidden_states=ctx, attention_mask=mask)
    assert y.shape == x.shape and y.dtype == dtype
    if mask is not None:
        assert "bias" in ops.last_attn_variant() or dtype == torch.float32, ops.last_attn_variant()
    if dtype == torch.float32:                                  # same inputs, same storage type as the reference run that made the golden
        assert rel_l2(to_np64(y), TEXT[case.name]) < TOL_F32
        assert worst(to_np64(y), TEXT[case.name]) < 1e-4
        return
    # (1) against the reference's fp32 output on the un-rounded inputs (includes input rounding)
    assert rel_l2(to_np64(y), TEXT[case.name]) < TOL[dtype]
    # (2) against the oracle in fp64 on the SAME rounded inputs
    r = rounded(inp, dtype)
    w = O.AttnWeights(r["wq"], r["wk"], r["wv"], r["wo"], r["bo"], case.heads)
    if case.mode == "plain":
        ref = O.plain_attention(r["x"], r.get("ctx"), w, mask=r.get("mask"))
    else:
        coef = torch.from_numpy(TEXT[case.name + "__coef"]).to(dtype).float().numpy()
        fn = O.outer_attention if case.mode.endswith("outer") else O.inner_attention
        ref = fn(r["x"], r.get("ctx"), w, coef, case.mode.startswith("fused"), mask=r.get("mask"))
    assert rel_l2(to_np64(y), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES + [torch.float32], ids=ids_dt)
@pytest.mark.parametrize("d", [40, 64, 80, 160])
def test_attention_core_with_score_bias(dtype, d):
    """AidAttnArgs.bias (ABI v8) = diffusers' prepared attention_mask: every layout prepare_attention_mask / a caller can hand over
    ([N * H, 1, L], [N, 1, L], [N, H, S, L], [N, S, L]), ragged L, several key tiles, PLAIN / pure INNER / pure OUTER, riders."""
    n, s, l, h = 4, 70, 150, 2
    q, k, v, vt = _core_inputs(n, s, l, h, d, dtype, seed=77 + d)
    g = torch.Generator().manual_seed(d)
    coef = torch.tensor([0.0, 0.35, 1.0, -1.0])                 # frame 3 = PLAIN rider
    tol = 1e-5 if dtype == torch.float32 else TOL[dtype]

    def additive(*shape):
        keep = torch.rand(*shape, generator=g) > 0.4
        keep[..., 0] = True
        return ((1.0 - keep.float()) * -10000.0).to(dtype)
    for shape in ((n * h, 1, l), (n, 1, l), (n, h, s, l), (n, s, l)):
        m = additive(*shape)
        m64 = to_np64(m)
        for mode in ("plain", "inner", "outer"):
            cf = coef if mode != "plain" else None
            nr = 1 if mode != "plain" else 0
            o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode=mode, fused=False, coef=None if cf is None else cf.to(DEV),
                             begin=0, end=2, n_plain=nr, bias=m.to(DEV))
            if mode == "plain":
                ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, d ** -0.5, "plain", False, None, mask=m64)
            else:                                               # frames 0 .. 2 interpolate between 0 and 2, frame 3 rides PLAIN
                ref = np.empty((n, s, h * d))
                mm = m64.reshape(n, -1, m64.shape[-2], l)
                ref[:3] = O.attn_core(to_np64(q)[:3], to_np64(k)[:3], to_np64(v)[:3], h, d ** -0.5, mode, False, coef[:3].numpy(),
                                      mask=mm[:3])
                ref[3:] = O.attn_core(to_np64(q)[3:], to_np64(k)[3:], to_np64(v)[3:], h, d ** -0.5, "plain", False, None, mask=mm[3:])
            err = rel_l2(to_np64(o), ref)
            assert err < tol, (shape, mode, ops.last_attn_variant(), err)
    # a strided view (every other query row of a larger mask) needs no copy
    big = additive(n, 2 * s, l).to(DEV)
    o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, bias=big[:, ::2])
    ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, d ** -0.5, "plain", False, None, mask=to_np64(big[:, ::2])[:, None])
    assert rel_l2(to_np64(o), ref) < tol
    # masks written with -inf / finfo.min weigh their keys with exactly 0 (clamped to -1e30 inside the kernel)
    for low in (float("-inf"), torch.finfo(dtype).min):
        m = additive(n, 1, l)
        m[m < 0] = low
        o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, bias=m.to(DEV))
        assert torch.isfinite(o).all()
        m64 = np.where(to_np64(m) < 0, -1e30, 0.0)
        ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, d ** -0.5, "plain", False, None, mask=m64[:, None])
        assert rel_l2(to_np64(o), ref) < tol
    # fused calls are refused by the library, like the reference's broadcast (aid_hip.h)
    with pytest.raises(RuntimeError, match="invalid argument"):
        ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="outer", fused=True, coef=coef.to(DEV), end=2, n_plain=1,
                     bias=additive(n, 1, l).to(DEV))


def test_processor_attention_mask_protocol():
    """What the boundary does with ``attention_mask`` (interpolation.py:604-606, 651-656, 738-739, 787): prepared by the attention
    module, added to every segment's scores; fused processors fail like the reference (golden flag), IP processors too; the
    de-activated processor hands it to its original_attn / runs PLAIN with it; a mask of zeros changes nothing."""
    assert TEXT["fused_outer_with_mask_raises_runtime_error"][0] == 1 and TEXT["fused_inner_with_mask_raises_runtime_error"][0] == 1
    dtype = torch.float16
    case = next(c for c in C.TEXT_CASES if c.name == "mask_n3_d40_x_pure_outer")
    inp = C.text_inputs(case)
    attn = make_attn(aid_amd, inp, case.heads, case.cc, dtype, DEV)
    x = torch.from_numpy(inp["x"]).to(dtype).to(DEV)
    ctx = torch.from_numpy(inp["ctx"]).to(dtype).to(DEV)
    mask = torch.from_numpy(inp["mask"]).to(dtype).to(DEV)
    for cls in (aid_amd.OuterInterpolatedAttnProcessor, aid_amd.InnerInterpolatedAttnProcessor):
        with pytest.raises(RuntimeError, match="must match the existing size"):
            cls(t=0.3, is_fused=True)(attn, x, encoder_hidden_states=ctx, attention_mask=mask)
        with pytest.raises(RuntimeError, match="float16 tensor"):
            cls(t=0.3, is_fused=False)(attn, x, encoder_hidden_states=ctx, attention_mask=mask.float())
        proc = cls(t=0.3, is_fused=False)
        y0 = proc(attn, x, encoder_hidden_states=ctx)
        assert rel_l2(to_np64(proc(attn, x, encoder_hidden_states=ctx, attention_mask=torch.zeros_like(mask))), to_np64(y0)) < 2e-3
        # de-activated: the HIP plain processor takes the mask; the same numbers as HipAttnProcessor itself
        proc.original_attn = aid_amd.HipAttnProcessor()
        proc.deactivate()
        assert torch.equal(proc(attn, x, encoder_hidden_states=ctx, attention_mask=mask),
                           aid_amd.HipAttnProcessor()(attn, x, encoder_hidden_states=ctx, attention_mask=mask))
    ipa = aid_amd.IPAdapterShim(case.c, case.cc, num_tokens=4, dtype=dtype, device=DEV)
    ip = torch.randn(9, 1, 4, case.cc, dtype=dtype, device=DEV)
    for cls in (aid_amd.OuterInterpolatedIPAttnProcessor, aid_amd.ScaleControlIPAttnProcessor):
        with pytest.raises(RuntimeError, match="image-token scores"):
            cls(t=0.3, is_fused=False, ip_attn=ipa)(attn, x, encoder_hidden_states=(ctx, [ip]), attention_mask=mask)


@pytest.mark.parametrize("dtype", DTYPES + [torch.float32], ids=ids_dt)
@pytest.mark.parametrize("name", ["n16_d64_s_fused_outer", "n16_d64_s_fused_inner", "n16_d64_x_fused_outer"])
def test_shard_batches_of_the_16_frame_schedule_vs_reference_rows(name, dtype):
    """BASELINE configs[3]: the 16-frame Beta(50, 50) sequence split over 8 / 4 / 2 ranks by dist.frame_shard.  A rank's batch
    [frame 0 ; owned ; frame 15] with ITS rows of the coefficient schedule must reproduce the rows of the REFERENCE's 16-frame
    output (every frame depends on itself and the two end points only) — the coefficient slicing pinned to reference output."""
    from aid_amd import dist
    case = next(c for c in C.TEXT_CASES if c.name == name)
    inp = C.text_inputs(case)
    attn = make_attn(aid_amd, inp, case.heads, case.cc if case.cross else None, dtype, DEV)
    full = TEXT[name]
    coef_full = torch.from_numpy(TEXT[name + "__coef"])
    tol = TOL_F32 if dtype == torch.float32 else TOL[dtype]
    for world in (8, 4, 2):
        for rank in range(world):
            rows = list(dist.frame_shard(case.n, world, rank).index)
            proc = _text_proc(case)
            proc.coef = coef_full[rows].clone()
            x = torch.from_numpy(inp["x"][rows]).to(dtype).to(DEV)
            ctx = torch.from_numpy(inp["ctx"][rows]).to(dtype).to(DEV) if case.cross else None
            y = proc(attn, x, encoder_hidden_states=ctx)
            assert rel_l2(to_np64(y), full[rows]) < tol, (world, rank, rows)


@pytest.mark.parametrize("dtype", DTYPES + [torch.float32], ids=ids_dt)
@pytest.mark.parametrize("case", C.IP_CASES, ids=lambda c: c.name)
def test_ip_processors_vs_reference_goldens(case, dtype):
    inp = C.ip_inputs(case)
    attn = make_attn(aid_amd, inp, case.heads, case.cc, dtype, DEV)
    ipa = aid_amd.IPAdapterShim(case.c, case.cc, num_tokens=case.tokens, scale=case.ip_scale, dtype=dtype, device=DEV)
    with torch.no_grad():
        ipa.to_k_ip[0].weight.copy_(torch.from_numpy(inp["wk_ip"]).to(dtype))
        ipa.to_v_ip[0].weight.copy_(torch.from_numpy(inp["wv_ip"]).to(dtype))
    cls = {"outer_ip": aid_amd.OuterInterpolatedIPAttnProcessor, "inner_ip": aid_amd.InnerInterpolatedIPAttnProcessor,
           "scale_control": aid_amd.ScaleControlIPAttnProcessor,
           "scale_control_off": aid_amd.ScaleControlIPAttnProcessor,
           "outer_ip_off": aid_amd.OuterInterpolatedIPAttnProcessor,
           "inner_ip_off": aid_amd.InnerInterpolatedIPAttnProcessor}[case.kind]
    if case.kind in ("outer_ip_off", "inner_ip_off"):          # what load_aid_ip_adapter installs as the fallback
        ipa = aid_amd.HipIPAdapterAttnProcessor.wrap(ipa)
    proc = cls(t=case.t, is_fused=case.is_fused, ip_attn=ipa)
    if case.kind.endswith("_off"):
        proc.deactivate()
    ehs = (torch.from_numpy(inp["text"]).to(dtype).to(DEV), [torch.from_numpy(inp["ip"]).to(dtype).to(DEV)])
    y = proc(attn, torch.from_numpy(inp["x"]).to(dtype).to(DEV), encoder_hidden_states=ehs)
    assert rel_l2(to_np64(y), IPG[case.name]) < (TOL_F32 if dtype == torch.float32 else TOL[dtype])


def test_inner_ip_without_fusion_raises_like_reference():
    case = C.IPCase("x", "inner_ip", False, 4)
    inp = C.ip_inputs(case)
    dtype = torch.float16
    attn = make_attn(aid_amd, inp, case.heads, case.cc, dtype, DEV)
    ipa = aid_amd.IPAdapterShim(case.c, case.cc, num_tokens=4, dtype=dtype, device=DEV)
    proc = aid_amd.InnerInterpolatedIPAttnProcessor(t=0.3, is_fused=False, ip_attn=ipa)
    ehs = (torch.from_numpy(inp["text"]).to(dtype).to(DEV), [torch.from_numpy(inp["ip"]).to(dtype).to(DEV)])
    with pytest.raises(RuntimeError):
        proc(attn, torch.from_numpy(inp["x"]).to(dtype).to(DEV), encoder_hidden_states=ehs)


def test_error_behaviour_batch_mismatch_dtype_and_mask():
    attn = aid_amd.AttnShim(80, 2, dtype=torch.float16, device=DEV)
    proc = aid_amd.OuterInterpolatedAttnProcessor(size=7, is_fused=True)
    with pytest.raises(RuntimeError, match="must match the size"):        # reference: broadcast error at the lerp
        proc(attn, torch.randn(5, 16, 80, dtype=torch.float16, device=DEV))
    attn64 = aid_amd.AttnShim(80, 2, dtype=torch.float64, device=DEV)
    with pytest.raises(TypeError, match="float16 / bfloat16 / float32"):
        aid_amd.HipAttnProcessor()(attn64, torch.randn(3, 16, 80, device=DEV, dtype=torch.float64))
    with pytest.raises(TypeError, match="dtype mismatch"):                # fp32 hidden states through fp16 weights
        aid_amd.HipAttnProcessor()(attn, torch.randn(3, 16, 80, device=DEV))
    with pytest.raises(TypeError, match="float16 / bfloat16 storage"):   # the LayerNorm kernels are 16-bit only
        ops.layernorm(torch.randn(8, 64, device=DEV))
    with pytest.raises(RuntimeError, match="float16 tensor"):            # a float32 mask against float16 scores (reference: baddbmm's dtype check)
        aid_amd.HipAttnProcessor()(attn, torch.randn(3, 16, 80, dtype=torch.float16, device=DEV),
                                   attention_mask=torch.zeros(3, 1, 16, device=DEV))
    attn_bad = aid_amd.AttnShim(96, 2, dtype=torch.float16, device=DEV)   # head dim 48 unsupported
    with pytest.raises(RuntimeError, match="head dim"):
        aid_amd.HipAttnProcessor()(attn_bad, torch.randn(3, 16, 96, dtype=torch.float16, device=DEV))


def test_deactivated_processor_delegates_or_runs_plain():
    dtype = torch.float16
    case = C.TEXT_CASES[0]
    inp = C.text_inputs(case)
    attn = make_attn(aid_amd, inp, case.heads, None, dtype, DEV)
    x = torch.from_numpy(inp["x"]).to(dtype).to(DEV)
    plain = aid_amd.HipAttnProcessor()(attn, x)
    calls = []

    def original(attn_, hs, ehs, mask, temb):
        calls.append(1)
        return hs * 2
    p = aid_amd.InnerInterpolatedAttnProcessor(t=0.5, is_fused=True, original_attn=original)
    p.deactivate()
    assert torch.equal(p(attn, x), x * 2) and calls == [1]
    p2 = aid_amd.InnerInterpolatedAttnProcessor(t=0.5, is_fused=True)
    p2.deactivate()
    assert torch.equal(p2(attn, x), plain)
    p2.activate(0.5)
    assert not torch.equal(p2(attn, x), plain)


def test_4d_input_residual_and_rescale_branches():
    """interpolation.py:593-597 / 669-677 (dead for UNet calls, kept for parity)."""
    dtype = torch.float16
    attn = aid_amd.AttnShim(80, 2, dtype=dtype, device=DEV)
    x4 = torch.randn(3, 80, 4, 6, dtype=dtype, device=DEV)
    proc = aid_amd.OuterInterpolatedAttnProcessor(t=0.4, is_fused=True)
    y3 = proc(attn, x4.view(3, 80, 24).transpose(1, 2).contiguous())
    attn.residual_connection, attn.rescale_output_factor = True, 2.0
    y4 = proc(attn, x4)
    want = (y3.transpose(-1, -2).reshape(3, 80, 4, 6) + x4) / 2.0
    assert y4.shape == x4.shape and rel_l2(to_np64(y4), to_np64(want)) < 1e-3


# ------------------------------------------------------------------------------------------------
# BASELINE sizes: size-independent properties (the oracle would take minutes here)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,dtype", [("sd15", torch.float16), ("sdxl", torch.bfloat16)])
def test_full_size_layer_properties(model, dtype):
    """SD1.5 (S=4096, C=320, H=8, d=40) / SDXL (S=4096, C=640, H=10, d=64), N=7:
    (a) end-point frames of every fused mode equal plain attention of that frame,
    (b) a sampled set of query rows matches the fp64 oracle, (c) linearity in V."""
    s, c, h = (4096, 320, 8) if model == "sd15" else (4096, 640, 10)
    n, d = 7, c // h
    g = torch.Generator().manual_seed(1002)
    q = torch.randn(n, s, c, generator=g).to(dtype).to(DEV)
    k = torch.randn(n, s, c, generator=g).to(dtype).to(DEV)
    v = torch.randn(n, s, c, generator=g).to(dtype).to(DEV)
    vt = v.transpose(1, 2).contiguous()
    coef = torch.from_numpy(O.beta_coefs(n, 50, 50)).to(DEV)
    plain = ops.attn_fwd(q, k, vt, h, l=s, mode="plain")
    rows = torch.tensor([0, 31, 32, 1000, 2047, 4095])
    for mode in ("inner", "outer"):
        o = ops.attn_fwd(q, k, vt, h, l=s, mode=mode, fused=True, coef=coef)
        assert rel_l2(to_np64(o[[0, n - 1]]), to_np64(plain[[0, n - 1]])) < TOL[dtype]          # (a)
        assert rel_l2(to_np64(o[3]), to_np64(plain[3])) > 10 * TOL[dtype]
        qs = q[:, rows].contiguous()
        ref = O.attn_core(to_np64(qs), to_np64(k), to_np64(v), h, d ** -0.5, mode, True, coef.cpu().numpy())
        assert rel_l2(to_np64(o[:, rows]), ref) < TOL[dtype]                                      # (b)
    v2 = torch.randn(n, s, c, generator=g).to(dtype).to(DEV)
    o1 = ops.attn_fwd(q, k, vt, h, l=s, mode="outer", fused=True, coef=coef)
    o2 = ops.attn_fwd(q, k, v2.transpose(1, 2).contiguous(), h, l=s, mode="outer", fused=True, coef=coef)
    o12 = ops.attn_fwd(q, k, (v + v2).transpose(1, 2).contiguous(), h, l=s, mode="outer", fused=True, coef=coef)
    assert rel_l2(to_np64(o12), to_np64(o1) + to_np64(o2)) < 2 * TOL[dtype]                      # (c)


@pytest.mark.parametrize("cross", [False, True], ids=["self", "cross"])
def test_full_size_processor_call_sdxl_level2(cross):
    """One whole processor call at the SDXL C = 1280 level with the batched-CFG batch of 14 frames (S = 1024, 20 heads):
    the grouped q / k / V^T launch and the out projection run on the ping-pong GEMM engine here.  Frames 0, 3 (AID, fused
    outer) and 9 (plain rider) against the fp64 oracle on the bf16-rounded inputs."""
    dtype, n, s, c, heads, l, cc = torch.bfloat16, 7, 1024, 1280, 20, 77, 2048
    g = torch.Generator().manual_seed(1280)
    attn = aid_amd.AttnShim(c, heads, cc if cross else None, dtype=dtype, device=DEV)
    x = torch.randn(2 * n, s, c, generator=g).to(dtype)
    ctx = torch.randn(2 * n, l, cc, generator=g).to(dtype) if cross else None
    proc = aid_amd.OuterInterpolatedAttnProcessor(size=n, is_fused=True, alpha=50, beta=50)
    proc.plain_tail = n
    y = proc(attn, x.to(DEV), encoder_hidden_states=None if ctx is None else ctx.to(DEV))
    assert ops.last_gemm_variant().startswith("pingpong2")                      # the out projection, at least (256- or 288-row tiles)
    w = O.AttnWeights(*(to_np64(t) for t in (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
                                              attn.to_out[0].weight, attn.to_out[0].bias)), heads)
    coef = to_np64(proc.coef.to(dtype))
    sel = [0, 3, n - 1]                                                          # begin, interior, end
    xs, cs = to_np64(x[sel]), (None if ctx is None else to_np64(ctx[sel]))
    ref = O.outer_attention(xs, cs, w, coef[sel], True)
    assert rel_l2(to_np64(y[sel]), ref) < TOL[dtype]
    refp = O.plain_attention(to_np64(x[n + 2:n + 3]), None if ctx is None else to_np64(ctx[n + 2:n + 3]), w)
    assert rel_l2(to_np64(y[n + 2:n + 3]), refp) < TOL[dtype]


BENCH_LAYERS = [  # (model, dtype, S, C, heads, text width, mode, fused): every distinct attention layer of bench.py's two stacks
    ("sd15", torch.float16, 4096, 320, 8, 768, "inner", True), ("sd15", torch.float16, 1024, 640, 8, 768, "inner", True),
    ("sd15", torch.float16, 256, 1280, 8, 768, "inner", True), ("sd15", torch.float16, 64, 1280, 8, 768, "inner", True),
    ("sdxl", torch.bfloat16, 4096, 640, 10, 2048, "outer", True), ("sdxl", torch.bfloat16, 1024, 1280, 20, 2048, "outer", True),
]


# BASELINE configs[3]: 16 frames over 8 GPUs -> an interior rank runs the local batch [frame 0, 2 owned, frame 15] = 4 frames
# whose coefficients are rows of the 16-frame schedule (dist.frame_shard); same call structure as above
SHARD_LAYERS = [l_ + (4,) for l_ in BENCH_LAYERS if l_[0] == "sdxl"]


@pytest.mark.parametrize("cross", [False, True], ids=["self", "cross"])
@pytest.mark.parametrize("layer", [l_ + (7,) for l_ in BENCH_LAYERS] + SHARD_LAYERS,
                         ids=lambda t: f"{t[0]}_s{t[2]}_c{t[3]}_n{t[8]}")
def test_every_bench_layer_at_full_size_vs_oracle(layer, cross):
    """The exact calls bench.py times — n AID frames + n plain rider frames in one call, BetaPPF(50, 50) coefficients,
    SDXL cross-attention with the PAID shared contexts — against the fp64 oracle on sampled query rows of the begin,
    an interior and the end frame and of one rider frame.  (Whatever GEMM engine / attention variant the library
    picks for these shapes is what gets checked.)  n = 7: configs[1] / configs[2]; n = 4: the per-rank batch of
    configs[3] (2 owned frames + the 2 replicated end points, coefficient rows [0, 7, 8, 15] of the 16-frame schedule)."""
    model, dtype, s, c, heads, cc, mode, fused, n = layer
    l = 77
    g = torch.Generator().manual_seed(s + c + int(cross))
    attn = aid_amd.AttnShim(c, heads, cc if cross else None, dtype=dtype, device=DEV)
    x = torch.randn(2 * n, s, c, generator=g).to(dtype)
    shared = cross and model == "sdxl"
    idx = ([0] + [1] * (n - 2) + [2]) if shared else list(range(n))
    idx2 = idx + [i + max(idx) + 1 for i in idx]
    ctxd = torch.randn(max(idx2) + 1, l, cc, generator=g).to(dtype) if cross else None     # distinct contexts
    cls = aid_amd.OuterInterpolatedAttnProcessor if mode == "outer" else aid_amd.InnerInterpolatedAttnProcessor
    proc = cls(size=n, is_fused=fused, alpha=50, beta=50)
    if n == 4:
        from aid_amd.dist import frame_shard
        sh = frame_shard(16, 8, 3)
        assert sh.index == (0, 6, 7, 15) and sh.owned_local == (1, 3)
        proc.coef = torch.from_numpy(O.beta_coefs(16, 50, 50))[list(sh.index)].clone()
    proc.plain_tail = n
    y = proc(attn, x.to(DEV), encoder_hidden_states=None if ctxd is None else ctxd.to(DEV),
             **({"ctx_index": idx2} if shared else {}))
    w = O.AttnWeights(*(to_np64(t) for t in (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
                                              attn.to_out[0].weight, attn.to_out[0].bias)), heads)
    coef = to_np64(proc.coef.to(dtype))
    rows = np.unique(np.concatenate([np.arange(0, s, max(s // 24, 1)), [31, 32, s - 1]])) if s > 64 else np.arange(s)

    def oracle(frames, md, cf):
        xs = to_np64(x[frames])
        cs = None if ctxd is None else to_np64(ctxd[[idx2[f] for f in frames]])
        q, k, v = O._project(xs, cs, w)
        return O._out(O.attn_core(q[:, rows], k, v, heads, w.scale, md, fused and md != "plain", cf), w)

    sel = [0, n // 2, n - 1]
    assert rel_l2(to_np64(y[sel][:, rows]), oracle(sel, mode, coef[sel])) < TOL[dtype]
    assert rel_l2(to_np64(y[n + 2:n + 3][:, rows]), oracle([n + 2], "plain", None)) < TOL[dtype]


# ------------------------------------------------------------------------------------------------
# the step either side of the call: LayerNorm in front, residual add behind
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("rows,c", [(1000, 320), (77, 640), (4097, 1280), (5, 2048), (3, 8), (14 * 1024, 1280)])
def test_layernorm_vs_oracle(dtype, rows, c):
    g = torch.Generator().manual_seed(rows + c)
    x = (torch.randn(rows, c, generator=g) * 3.0 + 1.5).to(dtype)
    gamma = (1.0 + 0.2 * torch.randn(c, generator=g)).to(dtype)
    beta = (0.1 * torch.randn(c, generator=g)).to(dtype)
    y = ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), eps=1e-5)
    assert rel_l2(to_np64(y), O.layer_norm(to_np64(x), to_np64(gamma), to_np64(beta), 1e-5)) < TOL_GEMM[dtype]
    y0 = ops.layernorm(x.to(DEV), None, None, eps=1e-5)
    assert rel_l2(to_np64(y0), O.layer_norm(to_np64(x), None, None, 1e-5)) < TOL_GEMM[dtype]
    with pytest.raises(RuntimeError, match="shape"):
        ops.layernorm(torch.zeros(4, 4104, device=DEV, dtype=dtype))            # wider than 2048 channels


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("mnk", [(14336, 1280, 1280), (3000, 640, 320), (257, 324, 72), (1, 8, 8)])
def test_gemm_residual_equals_the_separate_add(dtype, mnk):
    """The residual goes in after the rounding of the projection, so the fused epilogue is the reference's
    ``attn_output + hidden_states`` bit for bit (both GEMM engines and the ragged-k kernel)."""
    m, n, k = mnk
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn(m, k, generator=g).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype).to(DEV)
    bias = torch.randn(n, generator=g).to(dtype).to(DEV)
    res = torch.randn(m, n, generator=g).to(dtype).to(DEV)
    assert torch.equal(ops.linear(a, w, bias, residual=res), ops.linear(a, w, bias) + res)
    assert torch.equal(ops.linear(a, w, None, residual=res), ops.linear(a, w) + res)


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("kind", ["outer", "inner", "plain"])
@pytest.mark.parametrize("cross", [False, True], ids=["self", "cross"])
@pytest.mark.parametrize("fold", ["0", "1"], ids=["ln_kernel", "ln_folded"])
def test_fused_sublayer_equals_norm_call_add(dtype, kind, cross, fold, monkeypatch):
    """h + attn(norm(h)) in one library call.  With the LayerNorm as its own pass (AID_LN_FOLD=0) the result is bit-identical
    to the three steps on the same kernels; folded into the projections (default) it is a different rounding path
    (x W'^T corrected in the epilogue instead of round(LayerNorm(x)) W^T) and is held against the fp64 oracle only."""
    monkeypatch.setenv("AID_LN_FOLD", fold)
    n, s, heads, d, l, cc = 5, 200, 2, 64, 77, 96
    c = heads * d
    g = torch.Generator().manual_seed(77 + int(cross))
    attn = aid_amd.AttnShim(c, heads, cc if cross else None, dtype=dtype, device=DEV)
    norm = torch.nn.LayerNorm(c, eps=1e-5).to(DEV, dtype)
    with torch.no_grad():
        norm.weight.copy_((1.0 + 0.2 * torch.randn(c, generator=g)).to(dtype))
        norm.bias.copy_((0.1 * torch.randn(c, generator=g)).to(dtype))
    h = (torch.randn(2 * n, s, c, generator=g) * 2.0 + 0.5).to(dtype).to(DEV)
    ctx = torch.randn(2 * n, l, cc, generator=g).to(dtype).to(DEV) if cross else None
    if kind == "plain":
        proc = aid_amd.HipAttnProcessor()
    else:
        cls = aid_amd.OuterInterpolatedAttnProcessor if kind == "outer" else aid_amd.InnerInterpolatedAttnProcessor
        proc = cls(size=n, is_fused=True, alpha=3, beta=3)
        proc.plain_tail = n
    fused = proc.fused_sublayer(attn, norm, h, ctx)
    xn = ops.layernorm(h, norm.weight, norm.bias, norm.eps)
    steps = h + proc(attn, xn, encoder_hidden_states=ctx)                    # the three steps on the same kernels
    if fold == "0":
        assert torch.equal(fused, steps)
    else:
        assert not torch.equal(fused, steps) and rel_l2(to_np64(fused), to_np64(steps)) < TOL[dtype]
    # against the oracle: h + attention(LayerNorm(h)) for the AID half and one rider frame
    w = O.AttnWeights(*(to_np64(t) for t in (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
                                              attn.to_out[0].weight, attn.to_out[0].bias)), heads)
    hn = O.layer_norm(to_np64(h), to_np64(norm.weight), to_np64(norm.bias), norm.eps)
    cn = None if ctx is None else to_np64(ctx)
    if kind == "plain":
        ref = to_np64(h[:3]) + O.plain_attention(hn[:3], None if cn is None else cn[:3], w)
        assert rel_l2(to_np64(fused[:3]), ref) < TOL[dtype]
    else:
        coef = to_np64(proc.coef.to(dtype))
        fn = O.outer_attention if kind == "outer" else O.inner_attention
        ref = to_np64(h[:n]) + fn(hn[:n], None if cn is None else cn[:n], w, coef, True)
        assert rel_l2(to_np64(fused[:n]), ref) < TOL[dtype]
        refp = to_np64(h[n:n + 1]) + O.plain_attention(hn[n:n + 1], None if cn is None else cn[n:n + 1], w)
        assert rel_l2(to_np64(fused[n:n + 1]), refp) < TOL[dtype]
        # a foreign wrapped processor cannot be fused: the three steps run instead
        proc.deactivate()
        proc.original_attn = lambda a_, x_, e_=None, m_=None, t_=None: aid_amd.HipAttnProcessor()(a_, x_, e_)
        assert torch.equal(proc.fused_sublayer(attn, norm, h, ctx), h + aid_amd.HipAttnProcessor()(attn, norm(h), ctx))


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("mnk", [(14336, 1280, 1280), (3000, 640, 320), (57344, 640, 640), (300, 200, 128), (257, 324, 72)],
                         ids=lambda t: "m%d_n%d_k%d" % t)
def test_gemm_with_folded_layernorm_vs_oracle(dtype, mnk):
    """LayerNorm(x) W^T computed as rstd (x W'^T - mean colsum) + shift in the GEMM epilogue — both operand sides (q / k:
    the activation is A; V^T = Wv x^T: it is B, batched per frame), all three engines (ping-pong, lock-step, ragged k),
    with scale and bias on top.  x has a mean of several sigma so the cancellation in the correction is exercised."""
    m, n, k = mnk
    g = torch.Generator().manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=g) * 1.5 + 2.0).to(dtype)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype)
    gamma = (1.0 + 0.3 * torch.randn(k, generator=g)).to(dtype)
    beta = (0.2 * torch.randn(k, generator=g)).to(dtype)
    bias = torch.randn(n, generator=g).to(dtype)
    eps = 1e-5
    xd, wd = x.to(DEV), w.to(DEV)
    st = ops.ln_stats(xd, eps)
    x64 = to_np64(x)
    assert np.allclose(to_np64(st[:, 0]), x64.mean(1), rtol=1e-5, atol=1e-5)
    assert np.allclose(to_np64(st[:, 1]), 1.0 / np.sqrt(x64.var(1) + eps), rtol=1e-4)
    wf, cs, sh = ops.ln_fold(wd, gamma.to(DEV), beta.to(DEV))
    assert torch.equal(wf, (wd.float() * gamma.to(DEV).float()).to(dtype))
    assert np.allclose(to_np64(cs), to_np64(wf).sum(1), rtol=1e-5, atol=1e-4)
    assert np.allclose(to_np64(sh), to_np64(w) @ to_np64(beta), rtol=1e-5, atol=1e-4)
    ref = O.layer_norm(x64, to_np64(gamma), to_np64(beta), eps) @ to_np64(w).T
    # side 1: y = 0.5 * LayerNorm(x) W^T + bias
    y = torch.empty(m, n, dtype=dtype, device=DEV)
    ops.gemm_nt([dict(a=xd, b=wf, c=y, bias=bias.to(DEV), m=m, n=n, k=k, lda=k, ldb=k, ldc=n, scale=0.5,
                      ln_stats=st, ln_colsum=cs, ln_shift=sh, ln_side=1)])
    assert rel_l2(to_np64(y), 0.5 * ref + to_np64(bias)) < TOL_GEMM[dtype], ops.last_gemm_variant()
    assert worst(to_np64(y), 0.5 * ref + to_np64(bias)) < WORST[dtype], ops.last_gemm_variant()
    # side 2: per "frame" f of rows, Y_f^T = W LayerNorm(x_f)^T  ([n, rows] per batch, padded row stride)
    frames = 4 if m % 4 == 0 else 1
    rows = m // frames
    ldc = (rows + 7) // 8 * 8
    yt = torch.zeros(frames, n, ldc, dtype=dtype, device=DEV)
    ops.gemm_nt([dict(a=wf, b=xd, c=yt, m=n, n=rows, k=k, lda=k, ldb=k, ldc=ldc, batch=frames, stride_a=0,
                      stride_b=rows * k, stride_c=n * ldc, ln_stats=st, ln_colsum=cs, ln_shift=sh, ln_side=2,
                      stride_stats=rows)])
    got = to_np64(yt)[:, :, :rows].transpose(0, 2, 1).reshape(m, n)
    assert rel_l2(got, ref) < TOL_GEMM[dtype] and worst(got, ref) < WORST[dtype], ops.last_gemm_variant()


# ------------------------------------------------------------------------------------------------
# batched classifier-free guidance: [cond frames ; uncond frames] in ONE call (plain rider frames)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("kind", ["outer", "inner"])
@pytest.mark.parametrize("cross", [False, True])
def test_batched_cfg_equals_two_separate_calls(dtype, kind, cross):
    n, s, heads, d, l, cc = 5, 150, 2, 64, 77, 96
    c = heads * d
    g = torch.Generator().manual_seed(21)
    attn = aid_amd.AttnShim(c, heads, cc if cross else None, dtype=dtype, device=DEV)
    xc = torch.randn(n, s, c, generator=g).to(dtype).to(DEV)
    xu = torch.randn(n, s, c, generator=g).to(dtype).to(DEV)
    cc_ = torch.randn(n, l, cc, generator=g).to(dtype).to(DEV) if cross else None
    cu_ = torch.randn(n, l, cc, generator=g).to(dtype).to(DEV) if cross else None
    cls = aid_amd.OuterInterpolatedAttnProcessor if kind == "outer" else aid_amd.InnerInterpolatedAttnProcessor
    proc = cls(size=n, is_fused=True, alpha=3, beta=3)
    y_cond = proc(attn, xc, encoder_hidden_states=cc_)                      # reference structure: AID pass ...
    y_unc = aid_amd.HipAttnProcessor()(attn, xu, encoder_hidden_states=cu_)  # ... then the plain pass
    proc.plain_tail = n
    both = proc(attn, torch.cat([xc, xu]), encoder_hidden_states=None if not cross else torch.cat([cc_, cu_]))
    assert both.shape[0] == 2 * n
    assert torch.equal(both[:n], y_cond)          # same kernels, same tiles -> bitwise
    assert torch.equal(both[n:], y_unc)
    proc.plain_tail = 0
    with pytest.raises(RuntimeError, match="must match the size"):
        proc(attn, torch.cat([xc, xu]))


# ------------------------------------------------------------------------------------------------
# shared text contexts (PAID guide prompt): keys / values of a distinct context are projected once
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("kind,fused", [("outer", True), ("outer", False), ("inner", True), ("inner", False), ("plain", False)])
def test_shared_contexts_equal_repeated_contexts(dtype, kind, fused):
    n, s, heads, d, l, cc = 7, 150, 2, 64, 77, 96
    c = heads * d
    g = torch.Generator().manual_seed(33)
    attn = aid_amd.AttnShim(c, heads, cc, dtype=dtype, device=DEV)
    x = torch.randn(2 * n, s, c, generator=g).to(dtype).to(DEV)
    distinct = torch.randn(6, l, cc, generator=g).to(dtype).to(DEV)       # cond: start, guide, end; uncond: same
    idx = [0] + [1] * (n - 2) + [2]
    idx2 = idx + [i + 3 for i in idx]
    full = distinct[idx2].contiguous()                                     # what the reference loop would pass
    if kind == "plain":
        proc = aid_amd.HipAttnProcessor()
        y_full = proc(attn, x, encoder_hidden_states=full)
        y_kw = proc(attn, x, encoder_hidden_states=distinct, ctx_index=idx2)
        proc.ctx_index = idx2
        y_attr = proc(attn, x, encoder_hidden_states=full)                 # repeated rows are de-duplicated
    else:
        cls = aid_amd.OuterInterpolatedAttnProcessor if kind == "outer" else aid_amd.InnerInterpolatedAttnProcessor
        proc = cls(size=n, is_fused=fused, alpha=3, beta=3)
        proc.plain_tail = n                                                # batched CFG: uncond half rides along
        y_full = proc(attn, x, encoder_hidden_states=full)
        y_kw = proc(attn, x, encoder_hidden_states=distinct, ctx_index=idx2)
        proc.ctx_index = idx2
        y_attr = proc(attn, x, encoder_hidden_states=full)
        # de-activated: the wrapped plain processor gets the map too
        proc.deactivate()
        proc.original_attn = aid_amd.HipAttnProcessor()
        assert torch.equal(proc(attn, x, encoder_hidden_states=distinct), aid_amd.HipAttnProcessor()(attn, x, full))
    assert torch.equal(y_kw, y_full) and torch.equal(y_attr, y_full)       # same kernels on the same rows: bitwise
    with pytest.raises(RuntimeError, match="ctx_index"):
        proc(attn, x, encoder_hidden_states=distinct, ctx_index=idx)       # wrong length
    with pytest.raises(RuntimeError, match="rows"):
        proc(attn, x, encoder_hidden_states=distinct[:5], ctx_index=idx2)  # too few context rows


def test_loop_with_shared_contexts_matches_repeated_contexts():
    n = 5
    unet = aid_amd.AttnStackUNet("sdxl", dtype=torch.bfloat16, device=DEV, scale_down=16)
    install_sequence_processors(unet, n, early="fused_outer", num_inference_steps=4)
    g = torch.Generator().manual_seed(3)
    xs = {(s, c): torch.randn(n, s, c, generator=g).to(torch.bfloat16).to(DEV) for (s, c) in unet.level_shapes()}
    cond3 = torch.randn(3, unet.text_len, unet.cross_dim, generator=g).to(torch.bfloat16).to(DEV)
    unc3 = torch.randn(3, unet.text_len, unet.cross_dim, generator=g).to(torch.bfloat16).to(DEV)
    idx = [0, 1, 1, 1, 2]
    outs = []
    for shared in (True, False):
        for batched in (True, False):
            loop = AidDenoiseLoop(unet, xs, cond3 if shared else cond3[idx].contiguous(),
                                  unc3 if shared else unc3[idx].contiguous(), num_inference_steps=4,
                                  use_graphs=shared, batched_cfg=batched, ctx_index=idx if shared else None)
            outs.append([loop.step(i) for i in (0, 3)])
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            for k in a:
                assert torch.equal(a[k], b[k])


def test_prior_exploration_renders_candidate_batches_on_the_hip_stack():
    """the Beta-prior exploration asks for several candidate coefficients per round; each round is ONE
    N-frame AID call sequence over [0, *ts, 1] with that (arbitrary, non-Beta) coefficient vector."""
    from aid_amd.prior import BetaPriorExplorer
    dtype = torch.float16
    unet = aid_amd.AttnStackUNet("sd15", dtype=dtype, device=DEV, scale_down=16)
    g = torch.Generator().manual_seed(8)
    ends = {lv: torch.randn(2, lv[0], lv[1], generator=g) for lv in unet.level_shapes()}
    ctx_ends = torch.randn(2, unet.text_len, unet.cross_dim, generator=g)
    unc = torch.randn(1, unet.text_len, unet.cross_dim, generator=g)
    sizes = []

    def generate(ts):
        coef = torch.tensor([0.0] + [float(t) for t in ts] + [1.0])
        n = coef.numel()
        sizes.append(n)
        install_sequence_processors(unet, n, early="fused_outer", coef=coef)
        xs = {lv: torch.stack([aid_amd.slerp(e[0:1], e[1:2], float(c))[0] for c in coef]).to(dtype).to(DEV) for lv, e in ends.items()}
        cond = torch.stack([torch.lerp(ctx_ends[0], ctx_ends[1], float(c)) for c in coef]).to(dtype).to(DEV)
        loop = AidDenoiseLoop(unet, xs, cond, unc.expand(n, -1, -1).to(dtype).to(DEV).contiguous(), num_inference_steps=2,
                              use_graphs=False, batched_cfg=True)
        out = loop.step(0)[unet.level_shapes()[0]].float()
        assert torch.isfinite(out).all()
        return [float(c) for c in coef], [out[i].mean(dim=0, keepdim=True).cpu() for i in range(n)]

    runs = []
    for _ in range(2):
        sizes.clear()
        frames, features, ds, xs, alpha, beta = BetaPriorExplorer(generate).explore(exploration_size=8, batch=3)
        runs.append((xs, [float(d) for d in ds], float(alpha), float(beta)))
        assert sizes == [3, 4, 5] and len(xs) == 8 and xs == sorted(xs) and np.allclose(frames, xs, atol=1e-6)
    assert runs[0] == runs[1]                                    # the HIP path is deterministic, so is the search


# ------------------------------------------------------------------------------------------------
# randomized shapes (seeded): every mode, ragged S / L, shard-style end points, both dtypes
# ------------------------------------------------------------------------------------------------
def test_randomized_shapes_all_modes():
    rs = np.random.RandomState(20240928)
    for it in range(40):
        d = int(rs.choice([40, 64, 80, 160]))
        h = int(rs.randint(1, 4))
        n = int(rs.randint(2, 9))
        s = int(rs.choice([1, 7, 31, 32, 33, 64, 127, 129, 200, 257]))
        l = int(rs.choice([1, 5, 63, 64, 65, 77, 128, 150, 193]))
        dtype = DTYPES[it % 2]
        mode, fused = MODES[int(rs.randint(0, len(MODES)))]
        begin, end = (0, n - 1) if rs.rand() < 0.5 else (int(rs.randint(0, n)), int(rs.randint(0, n)))
        q, k, v, vt = _core_inputs(n, s, l, h, d, dtype, seed=1000 + it)
        coef = torch.from_numpy(rs.rand(n).astype(np.float32))
        coef[begin], coef[end] = 0.0, 1.0
        if begin == end:
            coef[begin] = 0.0
        o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode=mode, fused=fused, coef=coef.to(DEV),
                         begin=begin, end=end)
        ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, d ** -0.5, mode, fused, coef.numpy(), begin=begin, end=end)
        err = rel_l2(to_np64(o), ref)
        assert err < TOL[dtype], (it, d, h, n, s, l, mode, fused, begin, end, err)


@pytest.mark.parametrize("shape", [(1024, 1280, 20), (4096, 640, 10)], ids=["s1024_c1280", "s4096_c640"])
@pytest.mark.parametrize("tokens", [4, 16])
def test_ip_processors_at_sdxl_layer_shape(tokens, shape):
    """BASELINE configs[4] layer shapes (SDXL S=1024/C=1280/H=20 and S=4096/C=640/H=10, Cc=2048, bf16), batch 3 = the
    per-rank batch of the 8-frame / 8-GPU layout: outer-IP, scale-control and the de-activated fallback vs the fp64 oracle."""
    dtype = torch.bfloat16
    (s, c, h), cc, l = shape, 2048, 77
    g = torch.Generator().manual_seed(5)
    mk = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(dtype)   # noqa: E731
    inp = dict(x=mk(3, s, c), text=mk(3, l, cc), ip=mk(9, 1, tokens, cc), wq=mk(c, c, sc=c ** -0.5),
               wk=mk(c, cc, sc=cc ** -0.5), wv=mk(c, cc, sc=cc ** -0.5), wo=mk(c, c, sc=c ** -0.5), bo=mk(c, sc=0.01),
               wk_ip=mk(c, cc, sc=cc ** -0.5), wv_ip=mk(c, cc, sc=cc ** -0.5))
    attn = aid_amd.AttnShim(c, h, cc, dtype=dtype, device=DEV)
    ipa = aid_amd.IPAdapterShim(c, cc, num_tokens=tokens, scale=0.7, dtype=dtype, device=DEV)
    with torch.no_grad():
        for lin, key in ((attn.to_q, "wq"), (attn.to_k, "wk"), (attn.to_v, "wv"), (attn.to_out[0], "wo"),
                         (ipa.to_k_ip[0], "wk_ip"), (ipa.to_v_ip[0], "wv_ip")):
            lin.weight.copy_(inp[key])
        attn.to_out[0].bias.copy_(inp["bo"])
    n64 = {k_: to_np64(v_) for k_, v_ in inp.items()}
    w = O.AttnWeights(n64["wq"], n64["wk"], n64["wv"], n64["wo"], n64["bo"], h)
    ipw = O.IPWeights(n64["wk_ip"], n64["wv_ip"], 0.7, tokens)
    ehs = (inp["text"].to(DEV), [inp["ip"].to(DEV)])
    coef = torch.tensor([0.0, 0.4, 1.0]).to(dtype).float().numpy()
    y = aid_amd.OuterInterpolatedIPAttnProcessor(t=0.4, is_fused=True, ip_attn=ipa)(attn, inp["x"].to(DEV), encoder_hidden_states=ehs)
    ref = O.outer_ip_attention(n64["x"], n64["text"], n64["ip"], w, ipw, coef, True)
    assert rel_l2(to_np64(y), ref) < TOL[dtype]
    y2 = aid_amd.ScaleControlIPAttnProcessor(t=0.4, is_fused=True, ip_attn=ipa)(attn, inp["x"].to(DEV), encoder_hidden_states=ehs)
    ref2 = O.scale_control_ip_attention(n64["x"], n64["text"], n64["ip"], w, ipw, coef, True, activated=True)
    assert rel_l2(to_np64(y2), ref2) < TOL[dtype]
    off = aid_amd.OuterInterpolatedIPAttnProcessor(t=0.4, is_fused=True, ip_attn=ipa)
    off.deactivate()                                  # bare weight holder -> HipIPAdapterAttnProcessor on its weights
    y3 = off(attn, inp["x"].to(DEV), encoder_hidden_states=ehs)
    assert rel_l2(to_np64(y3), O.ip_adapter_attention(n64["x"], n64["text"], n64["ip"], w, ipw)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("n,r", [(5, 3), (8, 1), (4, 4)])
def test_ip_processors_n_frame_generalisation(dtype, n, r):
    """N-frame IP processors (one image-embedding row group per frame; the reference hard-wires 3):
    N frames, r copies per frame, every variant + the de-activated fallback vs the oracle restated for N frames."""
    case = C.IPCase("gen", "outer_ip", True, 4, seed=900 + n)
    rs = np.random.RandomState(case.seed)
    c, cc = case.c, case.cc
    inp = dict(x=C.randn(rs, n, case.s, c), text=C.randn(rs, n, case.l, cc), ip=C.randn(rs, n * r, 1, case.tokens, cc),
               wq=C.randn(rs, c, c, scale=c ** -0.5), wk=C.randn(rs, c, cc, scale=cc ** -0.5),
               wv=C.randn(rs, c, cc, scale=cc ** -0.5), wo=C.randn(rs, c, c, scale=c ** -0.5), bo=C.randn(rs, c, scale=0.01),
               wk_ip=C.randn(rs, c, cc, scale=cc ** -0.5), wv_ip=C.randn(rs, c, cc, scale=cc ** -0.5))
    attn = make_attn(aid_amd, inp, case.heads, cc, dtype, DEV)
    ipa = aid_amd.IPAdapterShim(c, cc, num_tokens=case.tokens, scale=case.ip_scale, dtype=dtype, device=DEV)
    with torch.no_grad():
        ipa.to_k_ip[0].weight.copy_(torch.from_numpy(inp["wk_ip"]).to(dtype))
        ipa.to_v_ip[0].weight.copy_(torch.from_numpy(inp["wv_ip"]).to(dtype))
    rd = rounded(inp, dtype)
    w = O.AttnWeights(rd["wq"], rd["wk"], rd["wv"], rd["wo"], rd["bo"], case.heads)
    ipw = O.IPWeights(rd["wk_ip"], rd["wv_ip"], case.ip_scale, case.tokens)
    ehs = (torch.from_numpy(inp["text"]).to(dtype).to(DEV), [torch.from_numpy(inp["ip"]).to(dtype).to(DEV)])
    x = torch.from_numpy(inp["x"]).to(dtype).to(DEV)
    for cls, fn in ((aid_amd.OuterInterpolatedIPAttnProcessor, O.outer_ip_attention),
                    (aid_amd.InnerInterpolatedIPAttnProcessor, O.inner_ip_attention),
                    (aid_amd.ScaleControlIPAttnProcessor, O.scale_control_ip_attention)):
        proc = cls(size=n, is_fused=True, alpha=3, beta=3, ip_attn=ipa)
        coef = proc.coef.to(dtype).float().numpy()
        y = proc(attn, x, encoder_hidden_states=ehs)
        assert rel_l2(to_np64(y), fn(rd["x"], rd["text"], rd["ip"], w, ipw, coef, True)) < TOL[dtype], cls.__name__
    off = aid_amd.InnerInterpolatedIPAttnProcessor(size=n, is_fused=True, ip_attn=ipa)
    off.deactivate()
    assert rel_l2(to_np64(off(attn, x, encoder_hidden_states=ehs)),
                  O.ip_adapter_attention(rd["x"], rd["text"], rd["ip"], w, ipw)) < TOL[dtype]
    with pytest.raises(RuntimeError, match="frames"):
        aid_amd.OuterInterpolatedIPAttnProcessor(size=n + 1, is_fused=True, ip_attn=ipa)(attn, x, encoder_hidden_states=ehs)


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
def test_gemm_side_problems_ride_in_the_pingpong_launch(dtype):
    """Cross-attention projection group at SDXL size: the query projection (K = 1280) keeps the 256 x 256 ping-pong engine,
    the short text-context (and IP-Adapter image) projections with K = 2048 run as 128 x 128 side tiles of the same launch.
    Every output against fp64 on sampled rows."""
    n, s, c, cc, l, nctx, t_ip = 14, 1024, 1280, 2048, 77, 6, 4
    g = torch.Generator().manual_seed(99)
    x = torch.randn(n * s, c, generator=g).to(dtype)
    e = torch.randn(nctx, l, cc, generator=g).to(dtype)
    ip = torch.randn(nctx, t_ip, cc, generator=g).to(dtype)
    wq, wk, wv, wki, wvi = ((torch.randn(c, kk, generator=g) / kk ** 0.5).to(dtype) for kk in (c, cc, cc, cc, cc))
    d = lambda t: t.to(DEV)     # noqa: E731
    xd, ed, ipd, wqd, wkd, wvd, wkid, wvid = map(d, (x, e, ip, wq, wk, wv, wki, wvi))
    q = torch.empty(n * s, c, dtype=dtype, device=DEV)
    k = torch.empty(nctx, l, c, dtype=dtype, device=DEV)
    vt = torch.zeros(nctx, c, 80, dtype=dtype, device=DEV)
    kip = torch.empty(nctx, t_ip, c, dtype=dtype, device=DEV)
    vtip = torch.zeros(nctx, c, 8, dtype=dtype, device=DEV)
    probs = [dict(a=xd, b=wqd, c=q, m=n * s, n=c, k=c, lda=c, ldb=c, ldc=c),
             dict(a=ed, b=wkd, c=k, m=nctx * l, n=c, k=cc, lda=cc, ldb=cc, ldc=c),
             dict(a=wvd, b=ed, c=vt, m=c, n=l, k=cc, lda=cc, ldb=cc, ldc=80, batch=nctx, stride_a=0, stride_b=l * cc, stride_c=c * 80),
             dict(a=ipd, b=wkid, c=kip, m=t_ip, n=c, k=cc, lda=cc, ldb=cc, ldc=c, batch=nctx, stride_a=t_ip * cc, stride_b=0,
                  stride_c=t_ip * c),
             dict(a=wvid, b=ipd, c=vtip, m=c, n=t_ip, k=cc, lda=cc, ldb=cc, ldc=8, batch=nctx, stride_a=0, stride_b=t_ip * cc,
                  stride_c=c * 8)]
    for group in (probs[:3], probs):
        ops.gemm_nt(group)
        assert ops.last_gemm_variant().startswith("pingpong2") and ops.last_gemm_variant().endswith("side128"), \
            ops.last_gemm_variant()
        rows = torch.tensor([0, 127, 128, 255, 256, 5000, n * s - 1])
        assert rel_l2(to_np64(q[rows]), to_np64(x[rows]) @ to_np64(wq).T) < TOL_GEMM[dtype]
        assert rel_l2(to_np64(k), to_np64(e) @ to_np64(wk).T) < TOL_GEMM[dtype]
        ref_vt = np.einsum("ck,flk->fcl", to_np64(wv), to_np64(e))
        assert rel_l2(to_np64(vt[:, :, :l]), ref_vt) < TOL_GEMM[dtype] and float(vt[:, :, l:].abs().max()) == 0.0
    assert rel_l2(to_np64(kip), to_np64(ip) @ to_np64(wki).T) < TOL_GEMM[dtype]
    assert rel_l2(to_np64(vtip[:, :, :t_ip]), np.einsum("ck,ftk->fct", to_np64(wvi), to_np64(ip))) < TOL_GEMM[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("tri,frames,keys,c,k", [(1, 14, 1024, 1280, 1280),      # SDXL level 2: flat value projection on the 288-row engine
                                                 (-1, 14, 1024, 1280, 1280),     # the cost model's choice for the same launch
                                                 (0, 14, 1024, 1280, 1280),      # rewritten into the batched form, 256-row tiles
                                                 (1, 5, 1000, 640, 320),         # ragged: 5000 rows, last tile 104 rows; 2.5 column tiles
                                                 (-1, 3, 200, 128, 64)])         # small: lock-step engine, batched form
def test_gemm_transposed_per_frame_output(dtype, tri, frames, keys, c, k, tuning):
    """AidGemmProblem.trans_rows: the flat value projection E Wv^T written as V^T[frame][channel][key] — grouped with a q- and a
    k-shaped problem like the processor call issues it — equals the batched form V^T[f] = Wv E_f^T bit for bit and fp64 within
    the GEMM tolerance, whichever engine runs it; with and without the folded LayerNorm."""
    tuning("GEMM_TRI", tri)
    g = torch.Generator().manual_seed(frames * 31 + keys)
    e = (torch.randn(frames * keys, k, generator=g) + 0.7).to(dtype).to(DEV)
    wq, wk, wv = ((torch.randn(c, k, generator=g) / k ** 0.5).to(dtype).to(DEV) for _ in range(3))
    q, kk = (torch.empty(frames * keys, c, dtype=dtype, device=DEV) for _ in range(2))
    vt = torch.full((frames, c, keys), float("nan"), dtype=dtype, device=DEV)
    ops.gemm_nt([dict(a=e, b=wq, c=q, m=frames * keys, n=c, k=k, lda=k, ldb=k, ldc=c),
                 dict(a=e, b=wk, c=kk, m=frames * keys, n=c, k=k, lda=k, ldb=k, ldc=c),
                 dict(a=e, b=wv, c=vt, m=frames * keys, n=c, k=k, lda=k, ldb=k, ldc=keys, stride_c=c * keys, trans_rows=keys)])
    name = ops.last_gemm_variant()
    if tri == 1 and k >= 1280:                                        # (short K loops stay on the lock-step engine)
        assert name == "pingpong288", name
    ref = np.einsum("ck,flk->fcl", to_np64(wv), to_np64(e).reshape(frames, keys, k))
    assert torch.isfinite(vt).all(), name
    assert rel_l2(to_np64(vt), ref) < TOL_GEMM[dtype] and worst(to_np64(vt), ref) < WORST[dtype], name
    assert rel_l2(to_np64(kk), to_np64(e) @ to_np64(wk).T) < TOL_GEMM[dtype]
    vb = torch.empty_like(vt)
    tuning("GEMM_TRI", 0)
    ops.gemm_nt([dict(a=wv, b=e, c=vb, m=c, n=keys, k=k, lda=k, ldb=k, ldc=keys, batch=frames, stride_a=0, stride_b=keys * k,
                      stride_c=c * keys)])
    assert torch.equal(vt, vb), name                                  # same products, same summation order: bitwise
    # folded LayerNorm on the flat form (the activation is operand A: statistics by output row)
    if k % 64 == 0:
        tuning("GEMM_TRI", tri)
        gamma, beta = (torch.randn(k, generator=g) * 0.3 + 1).to(dtype).to(DEV), (torch.randn(k, generator=g) * 0.2).to(dtype).to(DEV)
        wf, cs, sh = ops.ln_fold(wv, gamma, beta)
        stats = ops.ln_stats(e)
        vl = torch.empty_like(vt)
        ops.gemm_nt([dict(a=e, b=wf, c=vl, m=frames * keys, n=c, k=k, lda=k, ldb=k, ldc=keys, stride_c=c * keys, trans_rows=keys,
                          ln_stats=stats, ln_colsum=cs, ln_shift=sh, ln_side=1)])
        xn = O.layer_norm(to_np64(e), to_np64(gamma), to_np64(beta), 1e-5).reshape(frames, keys, k)
        refl = np.einsum("ck,flk->fcl", to_np64(wv), xn)
        assert rel_l2(to_np64(vl), refl) < 2 * TOL_GEMM[dtype], (name, ops.last_gemm_variant())


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("mnk", [(1792, 1280, 1280), (7168, 640, 640), (1078, 320, 768), (1000, 1272, 2048), (130, 72, 64)])
def test_lockstep_ring_depths_give_the_same_bits(dtype, mnk, tuning):
    """GEMM_LS: the rings of the lock-step engine (2 stages, 2 workgroups / CU; 4 stages, 1 workgroup / CU) change how far ahead the
    loads run, not the order of the sums: bit-identical outputs, ragged m / n edges and a one-tile K loop included."""
    m, n, k = mnk
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(dtype).to(DEV)
    b = (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype).to(DEV)
    bias = torch.randn(n, generator=g).to(dtype).to(DEV)
    tuning("GEMM_VARIANT", 7)
    tuning("GEMM_RS", 0)
    outs, names = [], []
    for ls in (0, 1):
        tuning("GEMM_LS", ls)
        outs.append(ops.linear(a, b, bias).clone())
        names.append(ops.last_gemm_variant())
    assert names == ["lockstep128", "lockstep128x4"]
    ref = to_np64(a) @ to_np64(b).T + to_np64(bias)
    assert rel_l2(to_np64(outs[0]), ref) < TOL_GEMM[dtype]
    assert torch.equal(outs[0], outs[1])
