"""Denoising-loop harness around the AID processors (host orchestration rows of SURVEY.md §2).

Mirrors the control flow of the reference loops:
  * N-frame batch, one frame per coefficient — gradio ``interpolate``
    (gradio_src/pipeline_interpolated_stable_diffusion.py:163-304);
  * warm-up rule of the root pipelines — AID on for the CONDITIONAL pass of steps
    ``i < int(num_inference_steps * warmup_ratio)`` (0-based), de-activated otherwise, and always
    de-activated for the UNCONDITIONAL pass (pipeline_interpolated_sd.py:1831, 1845-1848, 1870;
    SURVEY.md App. D1: the build follows root, not gradio's off-by-one);
  * classifier-free guidance ``uncond + gs * (text - uncond)`` (pipeline_interpolated_sd.py:1892).
The UNet is anything exposing diffusers' ``attn_processors`` / ``set_attn_processor`` and being
callable as ``unet(sample, encoder_hidden_states)`` — a real diffusers UNet2DConditionModel wrapped
by the caller, or :class:`AttnStackUNet` (attention calls only) in the benchmark.

MI355X-first: the three distinct passes of the loop (conditional+AID, conditional plain,
unconditional plain) are captured once into hipGraphs and replayed per step, so the 32 (SD1.5) /
140 (SDXL) x 3 kernel launches of a pass cost one graph launch on the host.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import torch

from .processors import (HipAttnProcessor, InnerInterpolatedAttnProcessor, InterpolatedAttnProcessor,
                         OuterInterpolatedAttnProcessor, cache_generation, clear_weight_caches)

EARLY_MODES = ("pure_inner", "fused_inner", "pure_outer", "fused_outer")


def install_sequence_processors(unet, size: int, early: str = "fused_outer", alpha: Optional[float] = None,
                                beta: Optional[float] = None, num_inference_steps: int = 50,
                                coef: Optional[torch.Tensor] = None) -> None:
    """One AID processor per attention layer for an N-frame sequence; Beta(alpha, beta) coefficients with
    alpha = beta = num_inference_steps by default (gradio_src/...stable_diffusion.py:203-206, 237-260).
    ``coef`` overrides the schedule (used by the frame-sharded layout: the local rows of the global
    schedule)."""
    if early not in EARLY_MODES:
        raise ValueError(f"early must be one of {EARLY_MODES}")
    alpha = num_inference_steps if alpha is None else alpha
    beta = num_inference_steps if beta is None else beta
    cls = OuterInterpolatedAttnProcessor if early.endswith("outer") else InnerInterpolatedAttnProcessor
    clear_weight_caches()                    # (see processors.py: in-place weight edits are invisible to the cache keys)
    procs = {}
    for name in unet.attn_processors.keys():
        p = cls(size=size, is_fused=early.startswith("fused"), alpha=alpha, beta=beta,
                original_attn=HipAttnProcessor())
        if coef is not None:
            assert coef.numel() == size
            p.coef = coef.detach().to(torch.float32).cpu().clone()
        procs[name] = p
    unet.set_attn_processor(procs)


def set_aid_active(unet, active: bool, plain_tail: int = 0) -> None:
    """N-frame analogue of activate_aid / deactivate_aid that keeps the coefficient schedule
    (the reference's activate(t) resets coef to [0, t, 1] for its batch-3 loop).  ``plain_tail`` frames
    appended after the interpolated ones run plain attention in the same call (batched CFG)."""
    for proc in unet.attn_processors.values():
        if isinstance(proc, InterpolatedAttnProcessor):
            proc.activated = bool(active)
            proc.plain_tail = int(plain_tail)


def set_ctx_index(unet, ctx_index: Optional[Sequence[int]]) -> None:
    """Tell every processor which row of ``encoder_hidden_states`` each frame of the next UNet call uses
    (None = one context per frame).  See InterpolatedAttnProcessor.ctx_index."""
    idx = None if ctx_index is None else [int(i) for i in ctx_index]
    for proc in unet.attn_processors.values():
        if hasattr(proc, "ctx_index"):
            proc.ctx_index = idx
        inner = getattr(proc, "original_attn", None)
        if inner is not None and hasattr(inner, "ctx_index"):
            inner.ctx_index = idx


def _record_stream(out, stream) -> None:
    if torch.is_tensor(out):
        out.record_stream(stream)
    elif isinstance(out, dict):
        for v in out.values():
            _record_stream(v, stream)
    elif isinstance(out, (tuple, list)):
        for v in out:
            _record_stream(v, stream)


def fork_join(first: Callable, second: Callable, side: "torch.cuda.Stream"):
    """``first()`` on the current stream and ``second()`` on ``side``, concurrently; both have finished (for the current stream) on
    return.  Inside a stream capture the fork and the join become edges of the graph.  The two callables run one after the other on
    the host, so Python-side state they toggle (``set_aid_active``) is seen in program order; what ``second`` returns was allocated on
    ``side`` and is handed to the current stream (``record_stream``; captured allocations live in the graph's pool)."""
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    gen = cache_generation()
    a = first()
    if cache_generation() != gen:
        # ``first`` (re)built a lazily shared tensor on the current stream — a text K / V projection, folded LayerNorm weights (cold
        # caches, new prompt tensors on a reused loop, a clear_weight_caches() in between): ``second`` reads it on the side stream, so
        # the side stream is ordered behind everything ``first`` has enqueued.  One step loses its overlap; no step races.
        side.wait_stream(cur)
    with torch.cuda.stream(side):
        b = second()
    cur.wait_stream(side)
    if not torch.cuda.is_current_stream_capturing():
        _record_stream(b, cur)
    return a, b


class AidDenoiseLoop:
    """Replays the per-step attention work of an interpolation run.

    step(i): conditional pass (AID on while i < warmup_steps) + unconditional pass (plain) + CFG
    combine of the two pass outputs.  ``use_graphs`` captures the passes into hipGraphs.
    """

    def __init__(self, unet, sample, cond, uncond, num_inference_steps: int = 50, warmup_ratio: float = 0.5,
                 guidance_scale: float = 7.5, use_graphs: bool = True, combine: Optional[Callable] = None,
                 batched_cfg: bool = False, ctx_index: Optional[Sequence[int]] = None, concurrent_cfg: bool = False):
        """``batched_cfg``: run the conditional and the unconditional pass of a step as ONE UNet call over the
        batch [cond frames ; uncond frames] (how stock diffusers pipelines do classifier-free guidance).  The
        AID processors treat the second half as plain riders (negative coefficients), so the result equals the
        reference's two separate calls while every GEMM sees twice the rows and the launch count halves."""
        """``ctx_index`` (frame -> row of ``cond`` / ``uncond``): the sequence shares text contexts (PAID guide
        prompt, sequence.py); ``cond`` / ``uncond`` then hold the DISTINCT contexts only."""
        """``cond`` / ``uncond`` may be ``(text, [image_embeds])`` tuples — the ``encoder_hidden_states`` an IP-Adapter UNet
        hands its attention layers (interpolation.py:259-266); the two passes then run as two UNet calls."""
        """``concurrent_cfg``: when the two passes of a step cannot share one UNet call (IP-Adapter contexts), run them as two UNet
        calls on two STREAMS — forked and joined inside one hipGraph when ``use_graphs`` — instead of back to back: the passes are
        independent, every kernel of one overlaps the other's (half the rows per kernel, both halves in flight).  Same results."""
        self.unet, self.sample, self.cond, self.uncond = unet, sample, cond, uncond
        self.batched_cfg = batched_cfg
        self.concurrent_cfg = concurrent_cfg and not batched_cfg
        self._side = None
        first = next(iter(sample.values())) if isinstance(sample, dict) else sample
        self.n_frames = first.shape[0]
        self.ctx_index = None if ctx_index is None else [int(i) for i in ctx_index]
        if self.ctx_index is not None and len(self.ctx_index) != self.n_frames:
            raise ValueError("ctx_index needs one entry per frame")
        if isinstance(cond, tuple):
            if batched_cfg or ctx_index is not None:
                raise ValueError("(text, [image_embeds]) contexts run as two separate passes without ctx_index")
        elif self.ctx_index is None and cond.shape[0] != self.n_frames:
            raise ValueError("one context per frame expected (or pass ctx_index)")
        self.ctx_index2 = None
        if batched_cfg:
            dup = (lambda t: torch.cat([t, t], dim=0))
            self.sample2 = {k: dup(v) for k, v in sample.items()} if isinstance(sample, dict) else dup(sample)
            self.ctx2 = torch.cat([cond, uncond], dim=0)
            if self.ctx_index is not None:
                self.ctx_index2 = self.ctx_index + [i + cond.shape[0] for i in self.ctx_index]
        self.num_inference_steps = num_inference_steps
        self.warmup_steps = int(num_inference_steps * warmup_ratio)        # pipeline_interpolated_sd.py:1831
        self.guidance_scale = guidance_scale
        self.use_graphs = use_graphs
        self.combine = combine or self._cfg
        self._graphs: Dict[str, torch.cuda.CUDAGraph] = {}
        self.fallback_reason: Optional[str] = None        # set when a capture failed and the loop went eager
        from . import ops
        self._ws = ops.WorkspaceOwner()          # the captures' workspaces are dropped with this loop (and its graphs)
        self._warmed: set = set()
        self._cap = None
        self._outs: Dict[str, object] = {}

    def _cfg(self, text, uncond):
        if isinstance(text, dict):
            return {k: uncond[k] + self.guidance_scale * (text[k] - uncond[k]) for k in text}
        return uncond + self.guidance_scale * (text - uncond)

    # -- the three distinct passes ---------------------------------------------------------------
    def _pass(self, which: str):
        set_ctx_index(self.unet, self.ctx_index2 if which.startswith("both") else self.ctx_index)
        if which == "both_aid":
            set_aid_active(self.unet, True, plain_tail=self.n_frames)
            return self.unet(self.sample2, self.ctx2)
        if which == "both_plain":
            set_aid_active(self.unet, False)
            return self.unet(self.sample2, self.ctx2)
        if which.startswith("pair"):                    # cond pass on the current stream, uncond pass on the side stream
            if self._side is None:
                self._side = torch.cuda.Stream()

            def cond_pass():
                set_aid_active(self.unet, which == "pair_aid", plain_tail=0)
                return self.unet(self.sample, self.cond)

            def uncond_pass():
                set_aid_active(self.unet, False)
                return self.unet(self.sample, self.uncond)
            from . import ops
            self._warm_pair(which, cond_pass, uncond_pass)
            with ops.cu_share(2):                       # two launch streams share the device: every call of both passes plans with half the CUs
                return fork_join(cond_pass, uncond_pass, self._side)
        if which == "cond_aid":
            set_aid_active(self.unet, True, plain_tail=0)
            return self.unet(self.sample, self.cond)
        set_aid_active(self.unet, False)
        return self.unet(self.sample, self.cond if which == "cond_plain" else self.uncond)

    def _warm_pair(self, which: str, cond_pass: Callable, uncond_pass: Callable) -> None:
        """The first forked step of a kind on cold caches: ``cond_pass`` fills lazily built shared state on the current
        stream — the folded LayerNorm weights, the text K / V cache — that ``uncond_pass`` reads on the side stream, and ``fork_join``
        orders the side stream behind the fork point only.  So the pair runs ONCE back to back on the current stream first (outputs
        dropped); every forked step after that finds the caches filled.  (Graph mode warms up the same way before its capture.)"""
        if torch.cuda.is_current_stream_capturing():
            return                                      # the eager warm-up of _run() has been here already
        key = (which, cache_generation())               # caches cleared / refilled since: warm again (fork_join guards the rest)
        if key in self._warmed:
            return
        cond_pass()
        uncond_pass()
        self._warmed = {(which, cache_generation())}

    def _run(self, which: str):
        if not self.use_graphs:
            return self._pass(which)
        g = self._graphs.get(which)
        if g is None:
            cur = torch.cuda.current_stream()
            if self._cap is None:
                self._cap = torch.cuda.Stream()         # warm-up AND capture stream of this loop: the capture adopts the warm-up's workspace
            cap = self._cap
            cap.wait_stream(cur)
            with torch.cuda.stream(cap):                # warm-up: lazy kernel attributes, coef caches, workspaces (ops.workspace)
                self._pass(which)
            cur.wait_stream(cap)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with self._ws, torch.cuda.graph(g, stream=cap):
                    self._outs[which] = self._pass(which)
            except RuntimeError as e:
                # a pass that cannot be captured (a collective of the end-point-exchange layout on a stack whose RCCL does not record
                # into graphs, a foreign UNet that synchronises): the loop goes on eagerly, loudly (pipelines._PassGraphs does the same)
                import warnings
                self.use_graphs = False
                self.fallback_reason = f"{type(e).__name__}: {e}"
                self._graphs.clear()
                self._outs.clear()
                self._ws.release()
                warnings.warn("hipGraph capture of a pass failed, the loop continues eagerly: " + self.fallback_reason, RuntimeWarning,
                              stacklevel=3)
                try:
                    torch.cuda.synchronize()
                except RuntimeError as e2:
                    raise RuntimeError(f"device unusable after a failed stream capture ({self.fallback_reason})") from e2
                return self._pass(which)
            self._graphs[which] = g
        g.replay()
        return self._outs[which]

    def aid_on(self, i: int) -> bool:
        return i < self.warmup_steps

    def step(self, i: int):
        if self.batched_cfg:
            both = self._run("both_aid" if self.aid_on(i) else "both_plain")
            n = self.n_frames
            if isinstance(both, dict):
                return self.combine({k: v[:n] for k, v in both.items()}, {k: v[n:] for k, v in both.items()})
            return self.combine(both[:n], both[n:])
        if self.concurrent_cfg:
            text, unc = self._run("pair_aid" if self.aid_on(i) else "pair_plain")
            return self.combine(text, unc)
        text = self._run("cond_aid" if self.aid_on(i) else "cond_plain")
        unc = self._run("uncond")
        return self.combine(text, unc)
