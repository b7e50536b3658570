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
