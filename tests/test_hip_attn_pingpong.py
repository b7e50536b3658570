"""GPU: the ping-pong attention kernel (csrc/aid_attn_pp.hip: d = 64, whole 64-key tiles; default for PLAIN and fused OUTER calls
from 2048 keys up, forced everywhere it is supported by the tuning knob ATTN_V2 = 1, profiles/r03_attn_notes.txt) against the fp64
oracle and against the program-order kernel.  A PLAIN call and a fused OUTER call whose key count is a multiple of 512 run on it
alone (three-segment frames: own -> begin -> end keys as one tile stream, the own-keys state parked in registers); any other
INNER / OUTER call is split — PLAIN riders and fused end-point frames here, the others on aid_attn_kernel in a second launch — and
both kernels split the frames by the same predicate on the DEVICE coefficients."""
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


def _inputs(n, s, l, h, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    c = h * 64
    q = torch.randn(n, s, c, generator=g).to(dtype)
    k = torch.randn(n, l, c, generator=g).to(dtype)
    v = torch.randn(n, l, c, generator=g).to(dtype)
    return q, k, v, v.transpose(1, 2).contiguous()


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("shape", [(3, 256, 128, 2), (2, 300, 256, 1), (3, 1, 192, 2), (2, 1024, 1024, 3), (5, 97, 640, 1)],
                         ids=lambda s: "n%d_s%d_l%d_h%d" % s)
def test_plain_call_on_the_pingpong_kernel(dtype, shape, tuning):
    tuning("ATTN_V2", 1)
    n, s, l, h = shape
    q, k, v, vt = _inputs(n, s, l, h, dtype, seed=s + l)
    o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="plain")
    assert "aid_attn_pp" in ops.last_attn_variant()
    ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, 64 ** -0.5, "plain", False, None)
    assert torch.isfinite(o).all()
    assert rel_l2(to_np64(o), ref) < TOL[dtype] and worst(to_np64(o), ref) < WORST[dtype]
    tuning("ATTN_V2", 0)
    o_old = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="plain")
    assert "aid_attn_pp" not in ops.last_attn_variant()
    assert rel_l2(to_np64(o), to_np64(o_old)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("mode,fused", [("outer", True), ("outer", False), ("inner", True), ("inner", False)])
def test_mixed_call_splits_the_frames_between_the_two_kernels(dtype, mode, fused, tuning):
    """7 AID frames + 7 PLAIN riders (batched CFG): riders (and, when fused, the two end-point frames) on the ping-pong kernel,
    interior frames on aid_attn_kernel; every frame against the oracle."""
    tuning("ATTN_V2", 1)
    n, s, l, h = 7, 200, 256, 2
    q, k, v, vt = _inputs(2 * n, s, l, h, dtype, seed=77)
    coef = torch.from_numpy(O.beta_coefs(n, 3, 3)).float()
    cd = torch.cat([coef.to(dtype).float(), -torch.ones(n)])
    o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode=mode, fused=fused, coef=cd.to(DEV), begin=0, end=n - 1, n_plain=n)
    q64, k64, v64 = to_np64(q), to_np64(k), to_np64(v)
    ref = np.concatenate([O.attn_core(q64[:n], k64[:n], v64[:n], h, 64 ** -0.5, mode, fused, coef.to(dtype).float().numpy()),
                          O.attn_core(q64[n:], k64[n:], v64[n:], h, 64 ** -0.5, "plain", False, None)])
    for f in range(2 * n):
        assert rel_l2(to_np64(o[f]), ref[f]) < TOL[dtype], (f, ops.last_attn_variant())
    assert worst(to_np64(o), ref) < WORST[dtype]
    # a WRONG rider hint must not change the result (the kernels decide on the device coefficients); only the launch choice moves
    o2 = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode=mode, fused=fused, coef=cd.to(DEV), begin=0, end=n - 1, n_plain=0)
    assert rel_l2(to_np64(o2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("l,s,riders", [(512, 200, 7), (1024, 96, 0), (512, 33, 3)], ids=["l512", "l1024", "l512_s33"])
def test_fused_outer_call_runs_alone_on_the_pingpong_kernel(dtype, l, s, riders, tuning):
    """Three-segment frames (own -> begin -> end keys, own-keys state parked and swapped back), end points, frames with a coefficient
    of exactly 0 / 1 that are NOT the end-point rows (two segments, the zero-weighted side dropped) and PLAIN riders in ONE launch."""
    tuning("ATTN_V2", 1)
    n, h = 7, 2
    q, k, v, vt = _inputs(n + riders, s, l, h, dtype, seed=l + s)
    coef = torch.from_numpy(O.beta_coefs(n, 3, 3)).float()
    coef[1], coef[5] = 0.0, 1.0                                 # interior rows with end-point coefficients
    cd = torch.cat([coef.to(dtype).float(), -torch.ones(riders)])
    args = dict(l=l, mode="outer", fused=True, coef=cd.to(DEV), begin=0, end=n - 1, n_plain=riders)
    o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, **args)
    assert ops.last_attn_variant() == "aid_attn_pp<d64,outer>"
    q64, k64, v64 = to_np64(q), to_np64(k), to_np64(v)
    ref = O.attn_core(q64[:n], k64[:n], v64[:n], h, 64 ** -0.5, "outer", True, coef.to(dtype).float().numpy())
    if riders:
        ref = np.concatenate([ref, O.attn_core(q64[n:], k64[n:], v64[n:], h, 64 ** -0.5, "plain", False, None)])
    for f in range(n + riders):
        assert rel_l2(to_np64(o[f]), ref[f]) < TOL[dtype], f
    assert torch.isfinite(o).all() and worst(to_np64(o), ref) < WORST[dtype]
    tuning("ATTN_V2", 0)                                        # the program-order kernel on the same call
    o_old = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, **args)
    assert "aid_attn_pp" not in ops.last_attn_variant() and rel_l2(to_np64(o), to_np64(o_old)) < TOL[dtype]
    tuning("ATTN_V2", 1)
    assert torch.equal(ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, **args), o)          # deterministic


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
@pytest.mark.parametrize("mode,fused", [("outer", False), ("inner", True), ("inner", False)])
def test_inner_and_pure_outer_calls_run_alone_on_the_pingpong_kernel(dtype, mode, fused, tuning):
    """INNER ([own ; mix] or mix alone, mix = the interpolated keys / values in k2 / vt2 or an end-point row for c = 0 / 1) and pure
    OUTER (begin side, then the end side from the EMPTY state) with riders and with interior rows whose coefficient is exactly 0 / 1."""
    tuning("ATTN_V2", 1)
    n, s, l, h, riders = 7, 150, 512, 2, 3
    q, k, v, vt = _inputs(n + riders, s, l, h, dtype, seed=len(mode) + 2 * fused)
    k[0, 100] = q[2, 7] * 5.0                                   # a spike in the begin keys and one in the end keys
    k[n - 1, 300] = q[3, 11] * 6.0
    vt = v.transpose(1, 2).contiguous()
    coef = torch.from_numpy(O.beta_coefs(n, 3, 3)).float()
    coef[1], coef[5] = 0.0, 1.0
    cd = torch.cat([coef.to(dtype).float(), -torch.ones(riders)])
    args = dict(l=l, mode=mode, fused=fused, coef=cd.to(DEV), begin=0, end=n - 1, n_plain=riders)
    o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, **args)
    assert ops.last_attn_variant() == f"aid_attn_pp<d64,{mode}>"
    q64, k64, v64 = to_np64(q), to_np64(k), to_np64(v)
    ref = np.concatenate([O.attn_core(q64[:n], k64[:n], v64[:n], h, 64 ** -0.5, mode, fused, coef.to(dtype).float().numpy()),
                          O.attn_core(q64[n:], k64[n:], v64[n:], h, 64 ** -0.5, "plain", False, None)])
    for f in range(n + riders):
        assert rel_l2(to_np64(o[f]), ref[f]) < TOL[dtype], f
    assert torch.isfinite(o).all() and worst(to_np64(o), ref) < WORST[dtype]
    tuning("ATTN_V2", 0)
    o_old = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, **args)
    assert "aid_attn_pp" not in ops.last_attn_variant() and rel_l2(to_np64(o), to_np64(o_old)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
def test_fused_outer_rescales_inside_the_begin_and_end_segments(dtype, tuning):
    """Spiked keys in the END-POINT frames: the begin side's row reference leaves the parked one behind (the first end tile's scores
    are shifted back by the difference), then the end side raises it again; accumulate + scales on the three-segment path."""
    tuning("ATTN_V2", 1)
    n, s, l, h = 4, 64, 512, 1
    q, k, v, vt = _inputs(n, s, l, h, dtype, seed=11)
    k[0, 70] = q[1, 5] * 5.0                                    # begin keys: spikes for rows of the interior frames
    k[0, 400] = q[2, 9] * 6.0
    k[n - 1, 3] = q[1, 5] * 4.0                                 # end keys: first tile and last tile
    k[n - 1, 511] = q[2, 40] * 7.0
    k[1, 200] = q[1, 17] * 5.0                                  # own keys
    vt = v.transpose(1, 2).contiguous()
    coef = torch.tensor([0.0, 0.3, 0.8, 1.0])
    fs = torch.tensor([0.5, 1.0, 2.0, 0.25])
    base = torch.randn(n, s, h * 64).to(dtype)
    out = base.clone().to(DEV)
    ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="outer", fused=True, coef=coef.to(dtype).float().to(DEV), begin=0,
                 end=n - 1, frame_scale=fs.to(DEV), out_scale=0.7, accumulate=True, out=out)
    assert ops.last_attn_variant() == "aid_attn_pp<d64,outer>"
    ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, 64 ** -0.5, "outer", True, coef.to(dtype).float().numpy())
    ref = to_np64(base) + 0.7 * fs.numpy()[:, None, None] * ref
    assert np.isfinite(to_np64(out)).all() and rel_l2(to_np64(out), ref) < TOL[dtype] and worst(to_np64(out), ref) < WORST[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=ids_dt)
def test_forced_rescale_on_the_pingpong_kernel(dtype, tuning):
    """Spiked keys late in the sequence push the row reference up after many tiles (the rare branch: O, its row-sum row and the
    tile's arguments are rescaled in the VALU slot, between PV(t - 1) and PV(t))."""
    tuning("ATTN_V2", 1)
    n, s, l, h = 2, 64, 512, 1
    q, k, v, vt = _inputs(n, s, l, h, dtype, seed=5)
    k[:, 130] = q[:, 3] * 4.0
    k[:, 300] = q[:, 9] * 5.0
    k[:, 511] = q[:, 7] * 6.0
    o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="plain")
    assert "aid_attn_pp" in ops.last_attn_variant()
    ref = O.attn_core(to_np64(q), to_np64(k), to_np64(v), h, 64 ** -0.5, "plain", False, None)
    assert np.isfinite(to_np64(o)).all() and rel_l2(to_np64(o), ref) < TOL[dtype] and worst(to_np64(o), ref) < WORST[dtype]


def test_accumulate_scales_kv_map_and_determinism(tuning):
    tuning("ATTN_V2", 1)
    dtype, n, s, l, h = torch.bfloat16, 4, 160, 128, 2
    q, k, v, vt = _inputs(n, s, l, h, dtype, seed=9)
    kv_map = torch.tensor([1, 1, 0, 2], dtype=torch.int32)
    fs = torch.tensor([0.5, 1.0, 2.0, 0.25])
    base = torch.randn(n, s, h * 64).to(dtype)
    out = base.clone().to(DEV)
    ops.attn_fwd(q.to(DEV), k[:3].contiguous().to(DEV), vt[:3].contiguous().to(DEV), h, l=l, mode="plain", kv_map=kv_map.to(DEV),
                 frame_scale=fs.to(DEV), out_scale=0.7, accumulate=True, out=out)
    assert "aid_attn_pp" in ops.last_attn_variant()
    idx = kv_map.long()
    ref = O.attn_core(to_np64(q), to_np64(k[idx]), to_np64(v[idx]), h, 64 ** -0.5, "plain", False, None)
    ref = to_np64(base) + 0.7 * fs.numpy()[:, None, None] * ref
    assert rel_l2(to_np64(out), ref) < TOL[dtype]
    outs = []
    for _ in range(3):
        o = ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode="plain")
        outs.append(o.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


def test_default_rule_for_the_pingpong_kernel(tuning):
    """No knob: fused OUTER and INNER calls from 1024 keys, PLAIN and pure OUTER calls from 2048 keys run on the ping-pong kernel (calls
    with several segments per frame: multiples of 512 keys); shorter ones do not."""
    dtype, h = torch.bfloat16, 1
    for l, mode, fused, want in ((2048, "plain", False, True), (1024, "plain", False, False), (2048, "outer", True, True),
                                 (1024, "outer", True, True), (512, "outer", True, False), (2048, "outer", False, True),
                                 (2048, "inner", True, True), (1024, "inner", True, True), (512, "inner", True, False), (2048, "inner", False, True),
                                 (2112, "outer", True, False), (2112, "plain", False, True)):
        q, k, v, vt = _inputs(3, 32, l, h, dtype, seed=l)
        coef = torch.tensor([0.0, 0.5, 1.0])
        ops.attn_fwd(q.to(DEV), k.to(DEV), vt.to(DEV), h, l=l, mode=mode, fused=fused,
                     coef=None if mode == "plain" else coef.to(DEV), begin=0, end=2)
        assert ("aid_attn_pp" in ops.last_attn_variant()) == want, (l, mode, fused, ops.last_attn_variant())


@pytest.mark.parametrize("h", [20, 5], ids=["h20", "h5"])
@pytest.mark.parametrize("mode,fused", [("outer", True), ("inner", True), ("inner", False), ("outer", False), ("plain", False)])
def test_persistent_workgroups_at_the_sdxl_level_shape(mode, fused, h, tuning):
    """S = 1024, 7 AID frames + 7 riders, 20 heads = 1120 items on 256 persistent workgroups (ATTN_PIPE = 1 forces the persistent walk
    for the mixed OUTER call too): the tile stream runs across item boundaries, the next item's Q rows come back from LDS, parked /
    swapped states start over per item.  Against the program-order kernel (itself held against the oracle) on the whole tensor.
    Round 4: the walk is a static balanced deal of heavy / light items per XCD (h = 5: 20 (head, q block) pairs do not divide over
    the 8 XCDs, some workgroups get no heavy item)."""
    dtype, n, s = torch.bfloat16, 7, 1024
    q, k, v, vt = _inputs(2 * n, s, s, h, dtype, seed=31)
    coef = torch.from_numpy(O.beta_coefs(n, 50, 50)).float()
    coef[0], coef[-1] = 0, 1
    cd = torch.cat([coef.to(dtype).float(), -torch.ones(n)]).to(DEV)
    args = dict(l=s, mode=mode, fused=fused, coef=None if mode == "plain" else cd, begin=0, end=n - 1, n_plain=0 if mode == "plain" else n)
    qd, kd, vd = q.to(DEV), k.to(DEV), vt.to(DEV)
    tuning("ATTN_V2", 1)
    tuning("ATTN_PIPE", 1)
    o = ops.attn_fwd(qd, kd, vd, h, **args)
    assert "aid_attn_pp" in ops.last_attn_variant() and torch.isfinite(o).all()
    o2 = ops.attn_fwd(qd, kd, vd, h, **args)
    assert torch.equal(o, o2)                                   # deterministic
    tuning("ATTN_PIPE", 0)                                      # one item per workgroup
    o1 = ops.attn_fwd(qd, kd, vd, h, **args)
    assert torch.equal(o, o1)                                   # the walk order does not change a bit
    tuning("ATTN_V2", 0)
    ref = ops.attn_fwd(qd, kd, vd, h, **args)
    assert "aid_attn_pp" not in ops.last_attn_variant()
    for f in range(2 * n):
        assert rel_l2(to_np64(o[f]), to_np64(ref[f])) < TOL[dtype], f


def test_full_size_sdxl_levels_sampled_rows(tuning):
    """The two self-attention shapes of the SDXL stack (14 frames) on sampled frames / heads / rows."""
    tuning("ATTN_V2", 1)
    dtype = torch.bfloat16
    for (s, h) in ((1024, 20), (4096, 10)):
        n = 14 if s == 1024 else 4
        g = torch.Generator().manual_seed(s)
        c = h * 64
        q = torch.randn(n, s, c, generator=g).to(dtype).to(DEV)
        k = torch.randn(n, s, c, generator=g).to(dtype).to(DEV)
        vt = torch.randn(n, c, s, generator=g).to(dtype).to(DEV)
        o = ops.attn_fwd(q, k, vt, h, l=s, mode="plain")
        assert "aid_attn_pp" in ops.last_attn_variant() and torch.isfinite(o).all()
        rows = torch.tensor([0, 31, 32, 255, 256, 700, s - 1])
        for f, hh in ((0, 0), (n - 1, h - 1), (n // 2, h // 2)):
            sl = slice(hh * 64, hh * 64 + 64)
            qs = to_np64(q[f, rows][:, sl]) * 64 ** -0.5
            sc = qs @ to_np64(k[f, :, sl]).T
            pr = np.exp(sc - sc.max(axis=1, keepdims=True))
            pr /= pr.sum(axis=1, keepdims=True)
            ref = pr @ to_np64(vt[f, sl, :]).T
            assert rel_l2(to_np64(o[f, rows][:, sl]), ref) < TOL[dtype], (s, f, hh)
