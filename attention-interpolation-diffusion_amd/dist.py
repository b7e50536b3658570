"""Frame sharding of one interpolation sequence over the GPUs of a node.

The reference is single-process / single-device; this is new work defined by north_star.
Frame i's interpolated attention needs only its own tensors plus K/V of the two END-POINT
frames at the same layer and step, and an end-point frame (coefficient 0 / 1) is self-contained.
So every rank runs the local batch

    [frame 0] + owned interior frames + [frame N-1]

(de-duplicated on the ranks that own an end point) with ZERO per-layer communication: the two
end-point frames are recomputed bit-identically on every rank from the same inputs.  The
local batch keeps the reference's layout convention (begin = row 0, end = last row), so the
processors need no index plumbing.  Collectives (RCCL via torch.distributed "nccl", gloo on
CPU in the tests) happen once per run: ``broadcast`` of the conditioning / initial latents /
coefficients from rank 0 and ``all_gather`` of the final latents of the owned frames.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class FrameShard:
    n_frames: int
    world_size: int
    rank: int
    owned: Tuple[int, int]            # [start, stop) global frame indices this rank owns
    index: Tuple[int, ...]            # global frame index of every local-batch row
    owned_local: Tuple[int, int]      # [start, stop) rows of the local batch that are owned

    @property
    def n_local(self) -> int:
        return len(self.index)

    @property
    def n_owned(self) -> int:
        return self.owned[1] - self.owned[0]


def partition_frames(n_frames: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced ownership ranges (the first ``n_frames % world_size`` ranks own one more)."""
    if n_frames < 2:
        raise ValueError("an interpolation sequence has at least the two end-point frames")
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    q, r = divmod(n_frames, world_size)
    out, start = [], 0
    for k in range(world_size):
        cnt = q + (1 if k < r else 0)
        out.append((start, start + cnt))
        start += cnt
    return out


def frame_shard(n_frames: int, world_size: int, rank: int) -> FrameShard:
    start, stop = partition_frames(n_frames, world_size)[rank]
    if stop == start:                         # more ranks than frames: this rank only replicates the end points
        return FrameShard(n_frames, world_size, rank, (start, stop), (0, n_frames - 1), (0, 0))
    index: List[int] = []
    if start > 0:
        index.append(0)                       # replica of the begin frame
    own_lo = len(index)
    index.extend(range(start, stop))
    own_hi = len(index)
    if stop < n_frames:
        index.append(n_frames - 1)            # replica of the end frame
    if stop == start:                         # a rank that owns nothing still runs the two end points
        own_lo = own_hi = 0
    return FrameShard(n_frames, world_size, rank, (start, stop), tuple(index), (own_lo, own_hi))


def _needs_host_hop(t: torch.Tensor, group=None) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def shard_rows(t: torch.Tensor, shard: FrameShard) -> torch.Tensor:
    """Rows of a per-frame tensor [N, ...] that make up this rank's local batch."""
    idx = torch.as_tensor(shard.index, device=t.device, dtype=torch.long)
    return t.index_select(0, idx).contiguous()


def broadcast_conditioning(tensors: Dict[str, torch.Tensor], src: int = 0, group=None) -> Dict[str, torch.Tensor]:
    """Once-per-run broadcast of the conditioning (prompt embeddings, pooled embeds / time ids, image
    embeds, initial latents, coefficients) from ``src``.  Every rank passes tensors of the right shape
    and dtype; they are filled in place."""
    if dist.is_available() and dist.is_initialized():        # also with one rank: the collective path is the same
        for name in sorted(tensors):
            t = tensors[name]
            if _needs_host_hop(t, group):          # gloo with device tensors (1-GPU development mode)
                h = t.cpu()
                dist.broadcast(h, src=src, group=group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src, group=group)
    return tensors


def gather_owned(local: torch.Tensor, shard: FrameShard, group=None) -> torch.Tensor:
    """all_gather of the OWNED rows of a local per-frame tensor into the full [N, ...] tensor
    (same result on every rank).  Owned counts differ by at most one, so rows are padded to the max."""
    lo, hi = shard.owned_local
    own = local[lo:hi]
    if not (dist.is_available() and dist.is_initialized()):
        assert shard.world_size == 1 and shard.n_owned == shard.n_frames
        return own.contiguous()
    if dist.get_world_size(group) != shard.world_size:
        raise ValueError(f"shard was made for {shard.world_size} ranks, the process group has {dist.get_world_size(group)}")
    parts = partition_frames(shard.n_frames, shard.world_size)
    mx = max(b - a for a, b in parts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: own.shape[0]] = own
    if _needs_host_hop(pad, group):
        hp = pad.cpu()
        hout = [torch.empty_like(hp) for _ in range(shard.world_size)]
        dist.all_gather(hout, hp, group=group)
        out = [o.to(pad.device) for o in hout]
    else:
        out = [torch.empty_like(pad) for _ in range(shard.world_size)]
        dist.all_gather(out, pad, group=group)
    return torch.cat([o[: b - a] for o, (a, b) in zip(out, parts)], dim=0)


def expected_speedup(n_frames: int, world_size: int) -> float:
    """Ideal speed-up of the sharded run over one GPU: N / max local batch (replicated end points cap it;
    e.g. 16 frames on 8 GPUs -> 16 / 4 = 4x)."""
    mx = max(frame_shard(n_frames, world_size, r).n_local for r in range(world_size))
    return n_frames / mx
