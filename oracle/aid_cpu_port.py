"""TEST / MEASUREMENT INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.

Multi-threaded CPU port of the reference's processor calls, used ONLY as ``bench.py``'s ``cpu_baseline`` leg
(``kind: "port"``): the same arithmetic as ``oracle/aid_oracle.py`` (which is pinned to the reference's own outputs by
``tests/golden``), written with torch CPU ops so that every stage — projections, scores, softmax, PV, lerp — runs on
``torch.get_num_threads()`` host cores the way the reference's eager CPU path does ("the build's CPU
restatement with torch.set_num_threads(all cores)").  ``tests/test_oracle_golden.py`` checks it against the numpy oracle.

Like the reference it materialises scores and probabilities; unlike it (interpolation.py:651-659 builds all
``[N*H, S, 2L]`` at once) it walks the heads one at a time so the S = 4096 layers of SDXL fit in host memory — the same
flops, smaller temporaries.  Follows: outer interpolation.py:626-664, inner :760-790, plain fallback :581-584.
"""
from __future__ import annotations

from typing import Optional

import torch


def _heads(t: torch.Tensor, heads: int) -> torch.Tensor:           # [B, L, H*d] -> [B, H, L, d]
    b, l, c = t.shape
    return t.view(b, l, heads, c // heads).permute(0, 2, 1, 3)


def _attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(q k^T * scale) v for [B, L, d] operands of ONE head (get_attention_scores + bmm)."""
    p = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype), q, k.transpose(1, 2),
                      beta=0, alpha=scale).softmax(dim=-1)
    return torch.bmm(p, v)


@torch.no_grad()
def processor_call(x: torch.Tensor, ctx: Optional[torch.Tensor], wq, wk, wv, wo, bo, heads: int, mode: str,
                   fused: bool, coef: Optional[torch.Tensor], only_heads: Optional[int] = None,
                   only_rows: Optional[int] = None) -> torch.Tensor:
    """One attention-processor call, fp32 on the CPU.  mode: plain | outer | inner.
    ``only_heads`` / ``only_rows``: timing aids — run the attention core for the first k heads / the first r query rows
    only (the rest of the output stays uninitialised); the projections always run in full.  bench.py times k = 1 and
    k = 2 heads on r rows and extrapolates linearly in both."""
    e = x if ctx is None else ctx
    q, k, v = x @ wq.T, e @ wk.T, e @ wv.T
    n, s, c = q.shape
    scale = float((c // heads) ** -0.5)
    qh, kh, vh = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    out = torch.empty(n, heads, s, c // heads, dtype=x.dtype)
    cf = None if coef is None else coef.to(x.dtype).view(-1, 1, 1)
    for h in range(heads if only_heads is None else min(only_heads, heads)):
        qi, ki, vi = qh[:, h], kh[:, h], vh[:, h]
        dst = out[:, h]
        if only_rows is not None and only_rows < s:
            qi, dst = qi[:, :only_rows], out[:, h, :only_rows]
        if mode == "plain":
            dst.copy_(_attend(qi, ki, vi, scale))
            continue
        kb, ke = ki[0:1].expand_as(ki), ki[-1:].expand_as(ki)
        vb, ve = vi[0:1].expand_as(vi), vi[-1:].expand_as(vi)
        if mode == "outer":
            if fused:
                kb, ke = torch.cat([ki, kb], 1), torch.cat([ki, ke], 1)
                vb, ve = torch.cat([vi, vb], 1), torch.cat([vi, ve], 1)
            o_e = _attend(qi, ke, ve, scale)
            o_b = _attend(qi, kb, vb, scale)
            dst.copy_((1 - cf) * o_b + cf * o_e)
        else:
            kc, vc = (1 - cf) * kb + cf * ke, (1 - cf) * vb + cf * ve
            if fused:
                kc, vc = torch.cat([ki, kc], 1), torch.cat([vi, vc], 1)
            dst.copy_(_attend(qi, kc, vc, scale))
    o = out.permute(0, 2, 1, 3).reshape(n, s, c)
    return o @ wo.T + bo
