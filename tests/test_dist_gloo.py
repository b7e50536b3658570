"""Frame sharding (attention-interpolation-diffusion_amd/dist.py) on CPU: pure partition logic plus a
world_size-2 gloo run of the once-per-run collectives, with the oracle standing in for the kernels to
show that the local batches [frame 0] + owned + [frame N-1] reproduce the unsharded result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import aid_amd
from aid_amd import dist as adist
from oracle import aid_oracle as O


def test_partition_and_shard_layout():
    assert adist.partition_frames(16, 8) == [(2 * i, 2 * i + 2) for i in range(8)]
    assert adist.partition_frames(7, 2) == [(0, 4), (4, 7)]
    assert adist.partition_frames(7, 1) == [(0, 7)]
    with pytest.raises(ValueError):
        adist.partition_frames(1, 1)
    s = adist.frame_shard(16, 8, 3)
    assert s.index == (0, 6, 7, 15) and s.owned_local == (1, 3) and s.n_local == 4 and s.n_owned == 2
    s0, s7 = adist.frame_shard(16, 8, 0), adist.frame_shard(16, 8, 7)
    assert s0.index == (0, 1, 15) and s0.owned_local == (0, 2)
    assert s7.index == (0, 14, 15) and s7.owned_local == (1, 3)
    # config 5: 8 frames on 8 GPUs -> interior ranks run exactly [start, own, end] (the reference's batch 3)
    assert adist.frame_shard(8, 8, 4).index == (0, 4, 7)
    # every frame is owned exactly once, every local batch starts with frame 0 and ends with frame N-1
    for n, w in ((16, 8), (7, 2), (7, 4), (56, 8), (5, 8)):
        owned = []
        for r in range(w):
            sh = adist.frame_shard(n, w, r)
            assert sh.index[0] == 0 and sh.index[-1] == n - 1
            owned += [sh.index[i] for i in range(*sh.owned_local)]
        assert owned == list(range(n))
    assert adist.expected_speedup(16, 8) == 4.0 and adist.expected_speedup(16, 1) == 1.0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1002 + rank)                 # non-source ranks start with DIFFERENT data
        heads, d, s, l = 2, 8, 6, 5
        c = heads * d
        cond = dict(x=torch.randn(n_frames, s, c), ctx=torch.randn(n_frames, l, c),
                    coef=torch.rand(n_frames))
        if rank == 0:
            cond["coef"][0], cond["coef"][-1] = 0.0, 1.0
        adist.broadcast_conditioning(cond, src=0)
        shard = adist.frame_shard(n_frames, world, rank)
        xq = adist.shard_rows(cond["x"], shard).numpy().astype(np.float64)
        kv = adist.shard_rows(cond["ctx"], shard).numpy().astype(np.float64)
        coef = adist.shard_rows(cond["coef"], shard).numpy()
        outs = {}
        for mode, fused in (("outer", True), ("inner", True), ("outer", False), ("plain", False)):
            local = O.attn_core(xq, kv, kv, heads, d ** -0.5, mode, fused, coef)      # begin = row 0, end = last row
            full = adist.gather_owned(torch.from_numpy(local), shard)
            outs[f"{mode}{int(fused)}"] = full.numpy()
        q.put((rank, {k: v.numpy() for k, v in cond.items()}, outs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [7, 4])
def test_world_size_2_gloo_sharded_equals_unsharded(n_frames):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, cond0, out0), (_, cond1, out1) = res
    for k in cond0:                                      # broadcast delivered rank 0's tensors
        np.testing.assert_array_equal(cond0[k], cond1[k])
    heads, d = 2, 8
    x, kv, coef = (cond0[k].astype(np.float64) for k in ("x", "ctx", "coef"))
    for key, full in out0.items():
        mode, fused = key[:-1], bool(int(key[-1]))
        ref = O.attn_core(x, kv, kv, heads, d ** -0.5, mode, fused, coef.astype(np.float32))
        np.testing.assert_allclose(full, ref, atol=1e-12)   # identical maths, only the batch composition differs
        np.testing.assert_array_equal(full, out1[key])     # all_gather gives every rank the same tensor


def _worker_w4(rank, world, port, n_frames, q):
    """World-size 4: the rank program of a sharded run on the oracle — broadcast from rank 0 into NaN-filled buffers, local batch
    [frame 0 ; owned ; frame N-1] with the rank's rows of the schedule, all_gather of the owned rows.  N = 16 takes its inputs from
    the golden case `n16_d64_s_fused_outer` (the REFERENCE's 16-frame output is the expected result); N = 3 leaves rank 3 without a
    frame of its own (it still runs the two end points and takes part in every collective)."""
    import cases as C
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        if n_frames == 16:
            case = next(c for c in C.TEXT_CASES if c.name == "n16_d64_s_fused_outer")
            inp = C.text_inputs(case)
            heads = case.heads
            x0 = torch.from_numpy(inp["x"])
            coef0 = torch.from_numpy(O.beta_coefs(16, 50, 50))
        else:
            g = torch.Generator().manual_seed(11)
            heads = 2
            x0 = torch.randn(n_frames, 6, 16, generator=g)
            inp = {k: (torch.randn(16, 16, generator=g) / 4).numpy() for k in ("wq", "wk", "wv", "wo")}
            inp["bo"] = torch.zeros(16).numpy()
            coef0 = torch.linspace(0, 1, n_frames)
        # non-source ranks start from NaN: whatever they compute with came through the broadcast
        cond = dict(x=x0.clone() if rank == 0 else torch.full_like(x0, float("nan")),
                    coef=coef0.clone() if rank == 0 else torch.full_like(coef0, float("nan")))
        adist.broadcast_conditioning(cond, src=0)
        shard = adist.frame_shard(n_frames, world, rank)
        xl = adist.shard_rows(cond["x"], shard).numpy()
        cl = adist.shard_rows(cond["coef"], shard).numpy()
        w = O.AttnWeights(inp["wq"], inp["wk"], inp["wv"], inp["wo"], inp["bo"], heads)
        local = O.outer_attention(xl, None, w, cl, True)
        full = adist.gather_owned(torch.from_numpy(local), shard)
        q.put((rank, shard.n_owned, shard.n_local, full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [16, 5, 3])
def test_world_size_4_gloo_rank_program(n_frames):
    import cases as C
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_w4, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = [r[1] for r in res]
    assert sum(owned) == n_frames and all(r[2] <= max(owned) + 2 for r in res)
    if n_frames == 3:
        assert owned == [1, 1, 1, 0] and res[3][2] == 2            # rank 3 owns nothing and runs the two end points
    for r in res[1:]:
        np.testing.assert_array_equal(r[3], res[0][3])             # all_gather: the same sequence on every rank
    assert res[0][3].shape[0] == n_frames and np.isfinite(res[0][3]).all()
    if n_frames == 16:                                             # ... and it is the REFERENCE's 16-frame output
        gold = C.load_fixture("text_goldens.npz")["n16_d64_s_fused_outer"]
        np.testing.assert_allclose(res[0][3], gold, rtol=0, atol=2e-6)


def test_gather_owned_single_process():
    sh = adist.frame_shard(5, 1, 0)
    t = torch.arange(5.0).reshape(5, 1)
    assert torch.equal(adist.gather_owned(t, sh), t)


# ---- end-point K/V exchange instead of replicated end points, sharded VAE decode ----------------------
def _worker_f4(rank, world, port, n_frames, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        kf = torch.randn(n_frames, 6, 8, generator=g)               # "projected keys" of every frame (same on all ranks)
        vf = torch.randn(n_frames, 8, 8, generator=g)
        sh = adist.owned_shard(n_frames, world, rank)
        n = sh.n_local
        ex = adist.EndpointExchange(n_frames, world, rank)
        k = torch.cat([kf[list(sh.index)], torch.full((2, 6, 8), float("nan"))])
        vt = torch.cat([vf[list(sh.index)], torch.full((2, 8, 8), float("nan"))])
        pend = ex.exchange_async(k, vt, n)                          # host tensors: complete on return, no event
        b, e = pend.wait()
        ok = pend.event is None and (b, e) == (n, n + 1) and torch.equal(k[b], kf[0]) and torch.equal(k[e], kf[-1]) \
            and torch.equal(vt[b], vf[0]) and torch.equal(vt[e], vf[-1]) and torch.equal(k[:n], kf[list(sh.index)])
        lat = torch.arange(n_frames, dtype=torch.float32).view(-1, 1, 1, 1).expand(n_frames, 4, 2, 2)[list(sh.index)]
        imgs = adist.decode_sharded(lambda z: z.repeat(1, 1, 2, 2)[:, :3] * 2.0, lat, sh)
        want = torch.arange(n_frames, dtype=torch.float32).view(-1, 1, 1, 1).expand(n_frames, 3, 4, 4) * 2.0
        ok = ok and imgs.shape == (n_frames, 3, 4, 4) and torch.equal(imgs, want)
        q.put((rank, bool(ok), ex.owner_begin, ex.owner_end))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 16])
def test_endpoint_exchange_and_sharded_decode_world_size_2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_f4, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [True, True], res
    assert res[0][2:] == (0, 1)
