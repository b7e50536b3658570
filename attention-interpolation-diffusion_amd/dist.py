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

import contextlib
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


# ---------------------------------------------------------------------------------------------
# SURVEY.md §8e "alternative" / §8f.4: no replicated end points — every rank runs ONLY the frames it owns and the two
# owner ranks broadcast the projected keys / values of frames 0 and N-1 once per self-attention layer.  (Cross-attention
# needs no exchange: the end-point frames' TEXT contexts are part of the once-per-run conditioning broadcast, every rank
# projects them itself — 77 rows.)  Removes the 2x recompute of the 16-frames-on-8-GPUs layout at the price of one small
# collective per self-attention layer (SDXL: 70 per UNet pass, 2.6 - 5.2 MB per tensor).
# ---------------------------------------------------------------------------------------------
def owned_shard(n_frames: int, world_size: int, rank: int) -> FrameShard:
    """Frame shard WITHOUT replicated end points: the local batch is exactly the owned frames."""
    start, stop = partition_frames(n_frames, world_size)[rank]
    return FrameShard(n_frames, world_size, rank, (start, stop), tuple(range(start, stop)), (0, stop - start))


class EndpointExchange:
    """Per-layer hand-off of the end-point frames' projected keys / values (reference semantics: every frame attends
    with K/V of frames 0 and N-1, interpolation.py:626-635 / 760-775).  ``exchange(k, vt, n)`` takes the rank's projected
    ``k [n + 2, L, C]`` / ``vt [n + 2, C, Lp]`` (rows ``n`` and ``n + 1`` are free, see ``ops.project_kv(extra_rows=2)``),
    fills row ``n`` with frame 0's and row ``n + 1`` with frame N-1's keys / values — broadcast from the ranks that own
    them (RCCL "nccl" backend on device tensors; gloo hops through the host in the 1-GPU development mode) — and returns
    the rows to pass as ``begin`` / ``end`` to the attention call."""

    def __init__(self, n_frames: int, world_size: int, rank: int, group=None):
        parts = partition_frames(n_frames, world_size)
        self.n_frames, self.world_size, self.rank, self.group = n_frames, world_size, rank, group
        self.owner_begin = next(r for r, (a, b) in enumerate(parts) if a <= 0 < b)
        self.owner_end = next(r for r, (a, b) in enumerate(parts) if a <= n_frames - 1 < b)
        a, b = parts[rank]
        self.local_begin = 0 if rank == self.owner_begin else None              # local row of frame 0 on its owner
        self.local_end = (b - a - 1) if rank == self.owner_end else None         # local row of frame N-1 on its owner
        self.calls = 0
        self.overlap = True                 # device tensors: broadcasts on a side stream, one event per layer (exchange_async)
        self._side = {}

    def _bcast(self, t: torch.Tensor, src: int) -> None:
        if not (dist.is_available() and dist.is_initialized()):
            if self.world_size != 1:
                raise RuntimeError("EndpointExchange over several ranks needs an initialised process group")
            return
        if _needs_host_hop(t, self.group):
            h = t.cpu()
            dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src, group=self.group)

    def _side_stream(self, device: torch.device):
        s = self._side.get(device)
        if s is None:
            s = self._side[device] = torch.cuda.Stream(device=device)
        return s

    def exchange_async(self, k: torch.Tensor, vt: torch.Tensor, n: int) -> "PendingExchange":
        """Start the hand-off and return at once: on device tensors the copies and the four broadcasts are enqueued on a
        SIDE stream behind the work already on the caller's stream (the k / v projection), with one event recorded behind
        them; the caller keeps its own stream busy (the layer's q projection) and calls ``.wait()`` on the result right
        before the attention launch, which makes its stream wait for that event — no host synchronisation anywhere.
        On host tensors (gloo development mode) the exchange completes inside this call."""
        if k.shape[0] != n + 2 or vt.shape[0] != n + 2:
            raise ValueError("k / vt need two free rows behind the local frames (ops.project_kv(extra_rows=2))")
        side = event = None
        if k.is_cuda and self.overlap:
            cur = torch.cuda.current_stream(k.device)
            side = self._side_stream(k.device)
            side.wait_stream(cur)                                   # rows 0 .. n - 1 are written by work queued on `cur`
            k.record_stream(side)
            vt.record_stream(side)
        ctx = torch.cuda.stream(side) if side is not None else contextlib.nullcontext()
        with ctx:
            for row, local, src in ((n, self.local_begin, self.owner_begin), (n + 1, self.local_end, self.owner_end)):
                if local is not None:
                    k[row].copy_(k[local])
                    vt[row].copy_(vt[local])
                self._bcast(k[row], src)
                self._bcast(vt[row], src)
            if side is not None:
                event = torch.cuda.Event()
                event.record(side)
        self.calls += 1
        return PendingExchange(n, n + 1, event)

    def exchange(self, k: torch.Tensor, vt: torch.Tensor, n: int) -> Tuple[int, int]:
        return self.exchange_async(k, vt, n).wait()


class PendingExchange:
    """Result of ``EndpointExchange.exchange_async``: ``wait()`` orders the caller's CURRENT stream behind the exchange
    (a device-side event wait) and returns the (begin, end) rows for the attention call."""

    def __init__(self, begin: int, end: int, event=None):
        self.begin, self.end, self.event = begin, end, event

    def wait(self) -> Tuple[int, int]:
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
            self.event = None
        return self.begin, self.end


def decode_sharded(decode, latents_local: torch.Tensor, shard: FrameShard, group=None) -> torch.Tensor:
    """every rank decodes the latents of the frames it OWNS (``decode``: the VAE decoder, third-party)
    and the decoded images — not the latents — are all-gathered: the VAE work is sharded like the denoising, and the
    full image sequence ``[N, 3, H, W]`` ends up on every rank."""
    lo, hi = shard.owned_local
    images = decode(latents_local[lo:hi]) if hi > lo else None
    if images is None:                      # a rank that owns nothing still takes part in the collective
        probe = decode(latents_local[:1])
        images = probe[:0]
    pseudo = FrameShard(shard.n_frames, shard.world_size, shard.rank, shard.owned, tuple(range(*shard.owned)),
                        (0, shard.n_owned))
    return gather_owned(images, pseudo, group=group)


def expected_speedup(n_frames: int, world_size: int) -> float:
    """Ideal speed-up of the sharded run over one GPU: N / max local batch (replicated end points cap it;
    e.g. 16 frames on 8 GPUs -> 16 / 4 = 4x)."""
    mx = max(frame_shard(n_frames, world_size, r).n_local for r in range(world_size))
    return n_frames / mx
