"""GPU: the RCCL branch of dist.py (torch.distributed backend "nccl" IS RCCL on ROCm) on device tensors — the
non-host-hop path of broadcast_conditioning / gather_owned — with a 1-rank process group on cuda:0.  Multi-rank
behaviour is covered on CPU by tests/test_dist_gloo.py (world_size 2, gloo); the 8-GPU run is the driver's."""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

import aid_amd  # noqa: E402
from aid_amd import dist as adist  # noqa: E402


def test_rccl_world_size_1_collectives_on_device_tensors():
    assert dist.is_nccl_available()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        g = torch.Generator().manual_seed(1)
        cond = torch.randn(5, 77, 64, generator=g).to(torch.bfloat16).cuda()
        lat = torch.randn(5, 4, 16, 16, generator=g).to(torch.float16).cuda()
        want_c, want_l = cond.clone(), lat.clone()
        assert not adist._needs_host_hop(cond)                           # device tensors go straight to RCCL
        out = adist.broadcast_conditioning({"cond": cond, "lat": lat}, src=0)
        torch.cuda.synchronize()
        assert torch.equal(out["cond"], want_c) and torch.equal(out["lat"], want_l)
        shard = adist.frame_shard(5, 1, 0)
        full = adist.gather_owned(lat, shard)                            # all_gather over RCCL, one rank
        torch.cuda.synchronize()
        assert full.shape == lat.shape and torch.equal(full, lat)
        # an all_reduce like bench.py's max-over-ranks timing
        t = torch.tensor([3.5], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 3.5
        with pytest.raises(ValueError, match="ranks"):
            adist.gather_owned(lat, adist.frame_shard(5, 2, 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["outer", "inner"])
@pytest.mark.parametrize("cross", [False, True], ids=["self", "cross"])
def test_endpoint_exchange_layout_equals_replicated_layout(kind, cross):
    """SURVEY.md §8f.4 on one rank: the batch without replicated end points + EndpointExchange (self-attention K/V rows
    handed over per layer, cross-attention from `endpoint_ctx`) gives the frames the same outputs as the standard
    layout, against the fp64 oracle.  Interior-only batch: frames 1..N-2 of an N-frame sequence, as an interior rank
    of a sharded run would hold them — the end-point rows come from a second 'owner' exchange object."""
    import numpy as np
    from oracle import aid_oracle as O
    from util import TOL, rel_l2, to_np64
    dtype, n, s, heads, d, l, cc = torch.float16, 5, 96, 2, 40, 77, 64
    c = heads * d
    g = torch.Generator().manual_seed(21)
    attn = aid_amd.AttnShim(c, heads, cc if cross else None, dtype=dtype, device="cuda:0")
    x = torch.randn(n, s, c, generator=g).to(dtype).cuda()
    ctx = torch.randn(n, l, cc, generator=g).to(dtype).cuda() if cross else None
    cls = aid_amd.OuterInterpolatedAttnProcessor if kind == "outer" else aid_amd.InnerInterpolatedAttnProcessor
    full = cls(size=n, is_fused=True, alpha=4, beta=4)
    w = O.AttnWeights(*(to_np64(t) for t in (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
                                              attn.to_out[0].weight, attn.to_out[0].bias)), heads)
    coef = full.coef.to(dtype).float().numpy()
    fn = O.outer_attention if kind == "outer" else O.inner_attention
    ref = fn(to_np64(x), None if ctx is None else to_np64(ctx), w, coef, True)

    # one rank owning everything: rows 0 / N-1 are copied into the two extra rows
    own = cls(size=n, is_fused=True, alpha=4, beta=4)
    own.endpoint_exchange = adist.EndpointExchange(n, 1, 0)
    if cross:
        own.endpoint_ctx = torch.stack([ctx[0], ctx[-1]])
    y = own(attn, x, encoder_hidden_states=ctx)
    assert rel_l2(to_np64(y), ref) < TOL[dtype]
    assert own.endpoint_exchange.calls == (0 if cross else 1)
    if not cross:
        # the hand-off runs on the exchange's side stream behind the k / v projection, the q projection overlaps it, one event
        # orders the attention launch behind it: same bits as the in-line (current-stream) exchange, call after call
        own.endpoint_exchange.overlap = False
        y0 = own(attn, x, encoder_hidden_states=None)
        own.endpoint_exchange.overlap = True
        for _ in range(5):
            assert torch.equal(own(attn, x, encoder_hidden_states=None), y0)
        assert len(own.endpoint_exchange._side) == 1

    # an interior "rank": only frames 1..N-2 in the batch; its exchange object is fed by a stand-in for the owners
    class FromOwners(adist.EndpointExchange):
        def exchange_async(self, k, vt, m):
            kk, vv = aid_amd.ops.project_kv(x[[0, n - 1]].contiguous(), attn.to_k.weight, attn.to_v.weight)
            k[m], k[m + 1], vt[m], vt[m + 1] = kk[0], kk[1], vv[0], vv[1]
            return adist.PendingExchange(m, m + 1)
    inner = cls(size=n - 2, is_fused=True)
    inner.coef = full.coef[1:-1].clone()
    inner.endpoint_exchange = FromOwners(n, 1, 0)
    if cross:
        inner.endpoint_ctx = torch.stack([ctx[0], ctx[-1]])
    yi = inner(attn, x[1:-1].contiguous(), encoder_hidden_states=None if ctx is None else ctx[1:-1].contiguous())
    assert rel_l2(to_np64(yi), ref[1:-1]) < TOL[dtype]


@pytest.mark.parametrize("endpoints", ["replicate", "exchange"])
@pytest.mark.parametrize("workload", ["sd15", "seq16"])
def test_bench_two_ranks_on_one_device(endpoints, workload):
    """`bench.py --gpus 2 --endpoints replicate|exchange` end to end: two rank processes on
    cuda:0 (AID_BENCH_ONE_DEVICE=1: gloo collectives, the device code path unchanged) — frame sharding, the conditioning
    broadcast, the per-layer end-point hand-over of the exchange layout, the all_gather of the owned frames and the
    max-over-ranks timing all run, and rank 0 prints ONE JSON line carrying the contract's keys.  The 8-GPU RCCL run
    itself is the driver's; this keeps the N > 1 bench path from rotting between rounds."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AID_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2",
           "--min-seconds", "0", "--workload", workload, "--endpoints", endpoints,
           "--no-cpu-baseline", "--no-roofline", "--no-also"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["value"] > 0 and rec["ms_per_step"] > 0
    assert rec["metric"] == "interpolation-frames/sec (50-step)" and rec["unit"] == "frames/s"
    assert rec["scaling"] == "strong" and rec["higher_is_better"] is True and rec["vs_baseline"] is None
    cfg = rec["config"]
    assert cfg["ranks"] == 2 and cfg["backend"] == "gloo" and "parity_tolerance" in cfg
    assert ("replicated end points" in cfg["parallelism"]) == (endpoints == "replicate")
    n = cfg["frames"]
    assert n == (16 if workload == "seq16" else 7)
    # replicate: ceil(interior / 2) owned + the 2 end points; exchange: owned frames only
    assert cfg["max_local_batch"] == ((n - 2 + 1) // 2 + 2 if endpoints == "replicate" else (n + 1) // 2)


def _run_ranks(n, one_device):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **({"AID_RANKS_ONE_DEVICE": "1"} if one_device else {}))
    msg = ""
    for attempt in range(3):          # the rendezvous on a just-released port is the one thing here that can fail for no reason of ours
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "tests", "helpers", "rccl_ranks.py")]
        p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        if p.returncode == 0:
            break
        tb = "\n".join(ln for ln in p.stderr.splitlines() if ln.startswith("[rank"))          # the ranks' own tracebacks
        msg = tb or p.stderr[-3000:]
        if "AssertionError" in msg:   # a rank's own check failed: not a rendezvous problem, do not retry
            break
    assert p.returncode == 0, msg
    import re
    # (two processes write to one pipe: "OK 1OK 0\n\n" is a legal interleaving)
    assert sorted(int(r) for r in re.findall(r"OK (\d+)", p.stdout)) == list(range(n)), p.stdout[-2000:]


def test_rank_program_two_ranks_on_one_device():
    """The rank program of the multi-GPU test below, two rank processes on cuda:0 over gloo: keeps it correct on the 1-GPU boxes."""
    _run_ranks(2, one_device=True)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the first multi-GPU lease runs it unattended")
def test_rccl_real_ranks_broadcast_gather_and_endpoint_exchange():
    """RCCL with MORE than one rank — broadcast_conditioning, gather_owned of both shard layouts and the per-layer
    end-point hand-over (dist.EndpointExchange) on real rank processes, one per visible GPU (up to 8), launched the way the driver
    launches bench.py.  Skipped on the 1-GPU boxes this repository has been developed on; tests/helpers/rccl_ranks.py is the rank program."""
    _run_ranks(min(torch.cuda.device_count(), 8), one_device=False)
