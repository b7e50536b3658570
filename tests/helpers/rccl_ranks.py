"""Rank program of tests/test_hip_rccl.py::test_rccl_real_ranks_* — one process per GPU over RCCL (backend "nccl"), launched by
torch.distributed.run.  Checks, on REAL ranks: broadcast_conditioning (src 0 -> every rank), gather_owned of a frame-sharded batch,
and the per-layer end-point hand-over (EndpointExchange) inside a self-attention processor call — the sharded outputs of every rank
against the single-rank replicated layout computed locally from the same inputs.  Prints one "OK <rank>" line per rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import aid_amd
from aid_amd import dist as adist

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
one_dev = os.environ.get("AID_RANKS_ONE_DEVICE") == "1"       # development: every rank on cuda:0, gloo collectives (1-GPU boxes)
local = 0 if one_dev else local
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if one_dev:
    dist.init_process_group("gloo")
else:
    dist.init_process_group("nccl", device_id=dev)
try:
    dtype, n, s, heads, d = torch.bfloat16, 2 * world + 2, 256, 4, 64
    c = heads * d
    g = torch.Generator().manual_seed(7)                       # every rank draws the same reference tensors ...
    x_all = torch.randn(n, s, c, generator=g).to(dtype).to(dev)
    cond_all = torch.randn(n, 77, 96, generator=g).to(dtype).to(dev)
    # ... but only rank 0's copies count: the others start from garbage and must receive the broadcast
    named = {"x": x_all.clone() if rank == 0 else torch.full_like(x_all, float("nan")),
             "cond": cond_all.clone() if rank == 0 else torch.full_like(cond_all, float("nan"))}
    out = adist.broadcast_conditioning(named, src=0)
    torch.cuda.synchronize()
    assert torch.equal(out["x"], x_all) and torch.equal(out["cond"], cond_all), "broadcast_conditioning"

    torch.manual_seed(1234)                                    # the same layer (to_out bias included) on every rank
    attn = aid_amd.AttnShim(c, heads, dtype=dtype, device=dev)
    with torch.no_grad():
        gw = torch.Generator().manual_seed(8)
        for lin in (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0]):
            lin.weight.copy_((torch.randn(lin.weight.shape, generator=gw) / c ** 0.5).to(dtype))
    full = aid_amd.OuterInterpolatedAttnProcessor(size=n, is_fused=True, alpha=6, beta=6)
    want = full(attn, x_all)                                   # the whole sequence on one device: the reference layout

    # replicated end points: every rank runs [frame 0] + owned + [frame N-1]; gather_owned re-assembles the sequence
    shard = adist.frame_shard(n, world, rank)
    rep = aid_amd.OuterInterpolatedAttnProcessor(size=shard.n_local, is_fused=True)
    rep.coef = full.coef[list(shard.index)].clone()
    y_loc = rep(attn, adist.shard_rows(x_all, shard))
    got = adist.gather_owned(y_loc, shard)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())       # noqa: E731
    assert got.shape == want.shape and torch.equal(got, want), ("gather_owned (replicated end points)", rel(got, want))

    # end-point exchange: owned frames only, the owners of frames 0 / N-1 broadcast their keys / values per layer
    own = adist.owned_shard(n, world, rank)
    ex = aid_amd.OuterInterpolatedAttnProcessor(size=own.n_local, is_fused=True)
    ex.coef = full.coef[list(own.index)].clone()
    ex.endpoint_exchange = adist.EndpointExchange(n, world, rank)
    y_ex = ex(attn, adist.shard_rows(x_all, own))
    got2 = adist.gather_owned(y_ex, own)
    torch.cuda.synchronize()
    assert got2.shape == want.shape and rel(got2, want) < 8e-3, ("endpoint exchange", rel(got2, want))
    sys.stdout.write(f"OK {rank}\n")
    sys.stdout.flush()
finally:
    dist.destroy_process_group()
