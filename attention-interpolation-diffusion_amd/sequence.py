"""Batch assembly of an interpolation run: latents, conditional / unconditional embeddings and the
coefficient schedule, laid out ``[start, interior ..., end]`` exactly like the reference's two loops.

* :func:`prepare_single` — the root pipelines' ``interpolate_single`` (batch 3, one interior frame at ``it``):
  ``latent_target = slerp(latent_start, latent_end, it)``; embeddings of the interior frame are the guide
  prompt's, or ``torch.lerp`` (``init == "linear"``) / ``slerp`` of the end points
  (pipeline_interpolated_sd.py:1653-1747).
* :func:`prepare_sequence` — the gradio N-frame ``interpolate``: spherical latents at UNIFORM t, interior
  embeddings = guide prompt or linear interpolation at uniform t, while the attention coefficients follow
  BetaPPF(alpha = beta = num_inference_steps) (gradio_src/pipeline_interpolated_stable_diffusion.py:203-260;
  SURVEY.md App. D9).

Both return a :class:`SequenceBatch`.  With a guide prompt the interior frames share one context; that is
exposed as ``ctx_index`` (frame -> distinct context row) so a caller can project each distinct text context
once (``ctx_map`` of the C ABI) instead of N times.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .interp import generate_beta_tensor, linear_interpolation, slerp, spherical_interpolation


@dataclass
class SequenceBatch:
    latents: torch.Tensor            # [N, C, h, w]
    cond: torch.Tensor               # [N, L, Cc]
    uncond: torch.Tensor             # [N, L, Cc]
    coef: torch.Tensor               # [N] fp32, end points 0 / 1
    ctx_index: torch.Tensor          # [N] int32: frame -> row of the distinct-context list
    n_distinct_ctx: int


def distinct_contexts(batch: "SequenceBatch"):
    """(cond [n_distinct, L, Cc], uncond [n_distinct, L, Cc], ctx_index list) — what AidDenoiseLoop(ctx_index=...)
    and the processors' ``ctx_index`` take: each distinct text context once, in order of first use."""
    idx = [int(i) for i in batch.ctx_index.tolist()]
    first = torch.tensor([idx.index(r) for r in range(batch.n_distinct_ctx)], dtype=torch.long)
    return batch.cond.index_select(0, first), batch.uncond.index_select(0, first), idx


def _ctx_index(n: int, guided: bool) -> torch.Tensor:
    if guided:                       # [start, guide x (n-2), end] -> 3 distinct contexts
        idx = [0] + [1] * (n - 2) + [2]
    else:
        idx = list(range(n))
    return torch.tensor(idx, dtype=torch.int32)


def prepare_single(it: float, latent_start: torch.Tensor, latent_end: torch.Tensor,
                   emb_start: torch.Tensor, emb_end: torch.Tensor,
                   uncond_start: torch.Tensor, uncond_end: torch.Tensor,
                   guide_emb: Optional[torch.Tensor] = None, uncond_guide: Optional[torch.Tensor] = None,
                   init: str = "linear") -> SequenceBatch:
    """Batch 3 = [start, target(it), end] (pipeline_interpolated_sd.py:1690-1747)."""
    assert 0 < it < 1, "t must be between 0 and 1"
    if guide_emb is not None:
        emb_t, unc_t = guide_emb, (uncond_guide if uncond_guide is not None else uncond_start)
    elif init == "linear":
        emb_t, unc_t = torch.lerp(emb_start, emb_end, it), torch.lerp(uncond_start, uncond_end, it)
    else:
        emb_t, unc_t = slerp(emb_start, emb_end, it), slerp(uncond_start, uncond_end, it)
    latents = torch.cat([latent_start, slerp(latent_start, latent_end, it), latent_end], dim=0)
    return SequenceBatch(latents, torch.cat([emb_start, emb_t, emb_end], dim=0),
                         torch.cat([uncond_start, unc_t, uncond_end], dim=0),
                         torch.tensor([0.0, it, 1.0]), _ctx_index(3, guide_emb is not None),
                         3)


def prepare_sequence(latent_start: torch.Tensor, latent_end: torch.Tensor,
                     emb_start: torch.Tensor, emb_end: torch.Tensor,
                     uncond_start: torch.Tensor, uncond_end: torch.Tensor, size: int = 7,
                     guide_emb: Optional[torch.Tensor] = None, uncond_guide: Optional[torch.Tensor] = None,
                     num_inference_steps: int = 25, alpha: Optional[float] = None,
                     beta: Optional[float] = None) -> SequenceBatch:
    """N-frame batch of the gradio ``interpolate`` loop."""
    alpha = num_inference_steps if alpha is None else alpha
    beta = num_inference_steps if beta is None else beta
    latents = spherical_interpolation(latent_start, latent_end, size)                      # :212
    if guide_emb is not None:                                                               # :221-229
        ug = uncond_guide if uncond_guide is not None else uncond_start
        cond = torch.cat([emb_start] + [guide_emb] * (size - 2) + [emb_end], dim=0)
        uncond = torch.cat([uncond_start] + [ug] * (size - 2) + [uncond_end], dim=0)
    else:                                                                                   # :231-234
        cond = linear_interpolation(emb_start, emb_end, size=size)
        uncond = linear_interpolation(uncond_start, uncond_end, size=size)
    coef = generate_beta_tensor(size, alpha=alpha, beta=beta)                               # :237-260
    coef[0], coef[-1] = 0, 1                                                                # interpolation.py:22
    guided = guide_emb is not None
    return SequenceBatch(latents, cond, uncond, coef, _ctx_index(size, guided), 3 if guided else size)
