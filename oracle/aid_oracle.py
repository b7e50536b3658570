"""TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.

CPU restatement (numpy) of the reference's interpolated-attention hot path
(QY-H00/attention-interpolation-diffusion @ 2024-10-20).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the shipped package never does (it fails loudly when the HIP
library is missing instead of falling back to this code).

Pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference
itself: ``tests/golden/make_goldens.py`` imports ``/root/reference/interpolation.py``
in the build container, runs its processors on seeded inputs and commits the
resulting vectors under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function below against them (fp32, <=2e-6 abs).

Written from the mathematics of the path, every function cites the reference
lines it follows.  All arithmetic is done in ``dtype`` (float32 mirrors the
reference's CPU path; float64 is used as "truth" for tolerance studies).

Notation: N frames, S query tokens, L key tokens, C = H*d channels,
Cc context width.  Weights use torch ``Linear.weight`` layout ``[out, in]``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

__all__ = [
    "AttnWeights", "IPWeights", "beta_coefs", "coef_for_t",
    "linear", "split_heads", "merge_heads", "softmax_attention",
    "plain_attention", "outer_attention", "inner_attention",
    "outer_ip_attention", "inner_ip_attention", "scale_control_ip_attention", "ip_adapter_attention",
    "attn_core", "slerp", "linear_interpolation", "spherical_interpolation",
    "next_exploration_t",
    "layer_norm",
]


# --------------------------------------------------------------------------
# weights containers (what diffusers' ``Attention`` owns)
# --------------------------------------------------------------------------
@dataclass
class AttnWeights:
    """to_q / to_k / to_v (no bias) and to_out[0] (bias) of one attention layer."""
    wq: np.ndarray            # [C, C]
    wk: np.ndarray            # [C, Cc]
    wv: np.ndarray            # [C, Cc]
    wo: np.ndarray            # [C, C]
    bo: np.ndarray            # [C]
    heads: int

    @property
    def scale(self) -> float:  # diffusers Attention.scale = dim_head ** -0.5
        return float((self.wq.shape[0] // self.heads) ** -0.5)

    def astype(self, dt) -> "AttnWeights":
        return AttnWeights(*(np.asarray(a, dtype=dt) for a in
                             (self.wq, self.wk, self.wv, self.wo, self.bo)), heads=self.heads)


@dataclass
class IPWeights:
    """IPAdapterAttnProcessor2_0 state shared by the IP processors
    (interpolation.py:70-74): to_k_ip[0], to_v_ip[0], scale[0], num_tokens[0]."""
    wk_ip: np.ndarray         # [C, Cc]
    wv_ip: np.ndarray         # [C, Cc]
    scale: float
    num_tokens: int


# --------------------------------------------------------------------------
# coefficients
# --------------------------------------------------------------------------
def beta_coefs(size: int, alpha: float = 1.0, beta: float = 1.0) -> np.ndarray:
    """prior.py:481-502 (BetaPPF(i/(n-1))) followed by interpolation.py:21-22
    (end points forced to 0 / 1).  float32 like the reference tensor."""
    from scipy.stats import beta as beta_distribution
    probs = [i / (size - 1) for i in range(size)]
    ts = np.asarray(beta_distribution.ppf(probs, alpha, beta), dtype=np.float32)
    ts[0], ts[-1] = 0.0, 1.0
    return ts


def coef_for_t(t: float) -> np.ndarray:
    """interpolation.py:24-27 / 37-42: a given t means batch [0, t, 1]."""
    assert 0 < t < 1, "t must be between 0 and 1"
    return np.asarray([0.0, t, 1.0], dtype=np.float32)


def next_exploration_t(alpha: float, beta: float, a: float, b: float) -> float:
    """prior.py:75-85: next t = BetaPPF((F(a)+F(b))/2)."""
    from scipy.stats import beta as beta_distribution
    fa, fb = beta_distribution.cdf(a, alpha, beta), beta_distribution.cdf(b, alpha, beta)
    return float(beta_distribution.ppf((fa + fb) / 2, alpha, beta))


# --------------------------------------------------------------------------
# building blocks (un-vendored diffusers Attention)
# --------------------------------------------------------------------------
def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    y = x @ w.T
    return y if b is None else y + b


def split_heads(t: np.ndarray, heads: int) -> np.ndarray:
    """head_to_batch_dim: [B, L, H*d] -> [B, H, L, d]; a 4-D [B, E, L, D] input
    first folds E into L (interpolation.py:334-341 relies on it)."""
    if t.ndim == 4:
        b, e, l, dim = t.shape
        t = t.reshape(b, e * l, dim)
    b, l, dim = t.shape
    return t.reshape(b, l, heads, dim // heads).transpose(0, 2, 1, 3)


def merge_heads(t: np.ndarray) -> np.ndarray:
    """batch_to_head_dim: [B, H, L, d] -> [B, L, H*d]."""
    b, h, l, d = t.shape
    return t.transpose(0, 2, 1, 3).reshape(b, l, h * d)


def softmax_attention(q: np.ndarray, k: np.ndarray, v: np.ndarray, scale: float,
                      mask: Optional[np.ndarray] = None) -> np.ndarray:
    """get_attention_scores + bmm (interpolation.py:651-652): softmax(q k^T * scale [+ mask]) v,
    all operands already head-split [B, H, *, d]; ``mask`` = the prepared attention_mask as [B, H | 1, S | 1, L],
    added to the scaled scores (diffusers' ``baddbmm(attention_mask, q, k^T, beta=1, alpha=scale)``, App. A)."""
    s = (q @ k.transpose(0, 1, 3, 2)) * q.dtype.type(scale)
    if mask is not None:
        s = s + mask.astype(s.dtype)
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(axis=-1, keepdims=True)
    return p @ v


def _rep(t: np.ndarray, idx: int, n: int) -> np.ndarray:
    """key[idx:idx+1] replicated n times along the batch axis
    (interpolation.py:627-635)."""
    return np.broadcast_to(t[idx:idx + 1], (n,) + t.shape[1:])


def _cvec(coef: np.ndarray, dt) -> np.ndarray:
    return np.asarray(coef, dtype=dt).reshape(-1, 1, 1)


# --------------------------------------------------------------------------
# attention cores on already-projected q/k/v  ([N, *, C] layout)
# --------------------------------------------------------------------------
def attn_core(q: np.ndarray, k: np.ndarray, v: np.ndarray, heads: int, scale: float,
              mode: str, is_fused: bool, coef: Optional[np.ndarray],
              begin: int = 0, end: int = -1, mask: Optional[np.ndarray] = None) -> np.ndarray:
    """Interpolated attention on projected tensors.  mode in {plain, outer, inner}.
    ``mask``: the prepared attention_mask, [N * H | N, 1 | S, L] (``prepare_attention_mask``) or [N, H | 1, S | 1, L]; the
    reference adds the SAME mask to the begin and the end side (interpolation.py:651-656) and to the interpolated keys (:787);
    with ``is_fused`` its L-wide mask meets 2 L scores and the broadcast fails — so does this function.

    outer: interpolation.py:626-664;  inner: interpolation.py:760-790;
    plain: the AttnProcessor2_0 fallback (interpolation.py:581-584).
    ``begin``/``end`` select the end-point frames (the reference always uses
    0 and -1; the sharded multi-GPU layout keeps that convention).
    """
    n = q.shape[0]
    dt = q.dtype
    end = end % k.shape[0]
    qh = split_heads(q, heads)
    if mask is not None:
        mask = np.asarray(mask)
        if mask.ndim == 3:                              # [N * H, R, L] -> [N, H, R, L];  [N, R, L] -> [N, 1, R, L]
            mask = mask.reshape(n, -1, mask.shape[1], mask.shape[2])
        if is_fused and mode != "plain":
            raise RuntimeError(f"The expanded size of the tensor ({2 * k.shape[1]}) must match the existing size "
                               f"({mask.shape[-1]}) at non-singleton dimension 2")
    if mode == "plain":
        return merge_heads(softmax_attention(qh, split_heads(k, heads), split_heads(v, heads), scale, mask))
    c = _cvec(coef, dt)
    kb, ke = _rep(k, begin, n), _rep(k, end, n)
    vb, ve = _rep(v, begin, n), _rep(v, end, n)
    if mode == "outer":
        kbh, keh = split_heads(kb, heads), split_heads(ke, heads)
        vbh, veh = split_heads(vb, heads), split_heads(ve, heads)
        if is_fused:                                   # interpolation.py:643-649
            kh, vh = split_heads(k, heads), split_heads(v, heads)
            keh = np.concatenate([kh, keh], axis=-2)
            veh = np.concatenate([vh, veh], axis=-2)
            kbh = np.concatenate([kh, kbh], axis=-2)
            vbh = np.concatenate([vh, vbh], axis=-2)
        o_end = merge_heads(softmax_attention(qh, keh, veh, scale, mask))
        o_beg = merge_heads(softmax_attention(qh, kbh, vbh, scale, mask))
        return (1 - c) * o_beg + c * o_end              # interpolation.py:662-664
    if mode == "inner":
        kc = (1 - c) * kb + c * ke                      # interpolation.py:772-775
        vc = (1 - c) * vb + c * ve
        kch, vch = split_heads(kc, heads), split_heads(vc, heads)
        if is_fused:                                   # interpolation.py:781-785
            kch = np.concatenate([split_heads(k, heads), kch], axis=-2)
            vch = np.concatenate([split_heads(v, heads), vch], axis=-2)
        return merge_heads(softmax_attention(qh, kch, vch, scale, mask))
    raise ValueError(mode)


# --------------------------------------------------------------------------
# full processor calls (projection -> core -> out-proj)
# --------------------------------------------------------------------------
def _project(x, ctx, w: AttnWeights):
    e = x if ctx is None else ctx                       # interpolation.py:616-617
    return linear(x, w.wq), linear(e, w.wk), linear(e, w.wv)


def _out(o, w: AttnWeights):
    return linear(o, w.wo, w.bo)                        # interpolation.py:666-667 (dropout p=0)


def layer_norm(x: np.ndarray, gamma: Optional[np.ndarray] = None, beta: Optional[np.ndarray] = None,
               eps: float = 1e-5) -> np.ndarray:
    """torch.nn.LayerNorm over the last dimension (biased variance) — the norm1 / norm2 of diffusers'
    BasicTransformerBlock in front of attn1 / attn2 (third-party; the step before the path)."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    y = (x - mu) / np.sqrt(var + eps)
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    return y


def plain_attention(x, ctx, w: AttnWeights, mask=None) -> np.ndarray:
    """De-activated processors: original AttnProcessor2_0 (interpolation.py:581-584)."""
    q, k, v = _project(x, ctx, w)
    return _out(attn_core(q, k, v, w.heads, w.scale, "plain", False, None, mask=mask), w)


def outer_attention(x, ctx, w: AttnWeights, coef, is_fused: bool, mask=None) -> np.ndarray:
    """OuterInterpolatedAttnProcessor.__call__, interpolation.py:573-679 (``mask``: the prepared attention_mask, :604-606)."""
    q, k, v = _project(x, ctx, w)
    return _out(attn_core(q, k, v, w.heads, w.scale, "outer", is_fused, coef, mask=mask), w)


def inner_attention(x, ctx, w: AttnWeights, coef, is_fused: bool, mask=None) -> np.ndarray:
    """InnerInterpolatedAttnProcessor.__call__, interpolation.py:707-804 (``mask``: :738-739, 787)."""
    q, k, v = _project(x, ctx, w)
    return _out(attn_core(q, k, v, w.heads, w.scale, "inner", is_fused, coef, mask=mask), w)


# ---- IP-Adapter variants.  The reference hard-wires a batch of 3: ``expand(3, ...)``, ``[::3]``,
# ``[6:9]`` on a [9, 1, T, Cc] image-embedding tensor.  Restated for N = x.shape[0] frames and r = R / N copies per
# frame (``[::r]``, rows of the last frame); for N = 3, R = 9 this IS the reference's arithmetic (pinned by the goldens).
def _ip_kv(ip_rows: np.ndarray, ipw: IPWeights):
    """to_k_ip[0] / to_v_ip[0] on [3, 1, T, Cc] rows -> [3, T, C] (the 4-D
    head_to_batch_dim fold, interpolation.py:330-341)."""
    k = linear(ip_rows, ipw.wk_ip)
    v = linear(ip_rows, ipw.wv_ip)
    b, e, t, c = k.shape
    return k.reshape(b, e * t, c), v.reshape(b, e * t, c)


def outer_ip_attention(x, text, ip, w: AttnWeights, ipw: IPWeights, coef, is_fused: bool):
    """OuterInterpolatedIPAttnProcessor.__call__, interpolation.py:240-387.
    ``ip`` is ip_hidden_states[0] with shape [9, 1, T, Cc]; rows [::3] are used."""
    dt = x.dtype
    q, k, v = _project(x, text, w)
    c = _cvec(coef, dt)
    qh = split_heads(q, w.heads)

    def two_sided(k_, v_):
        n = x.shape[0]                                   # literal 3 in the reference, interpolation.py:300-303
        kbh, keh = split_heads(_rep(k_, 0, n), w.heads), split_heads(_rep(k_, -1 % k_.shape[0], n), w.heads)
        vbh, veh = split_heads(_rep(v_, 0, n), w.heads), split_heads(_rep(v_, -1 % v_.shape[0], n), w.heads)
        if is_fused:
            kh, vh = split_heads(k_, w.heads), split_heads(v_, w.heads)
            keh, veh = np.concatenate([kh, keh], -2), np.concatenate([vh, veh], -2)
            kbh, vbh = np.concatenate([kh, kbh], -2), np.concatenate([vh, vbh], -2)
        o_e = merge_heads(softmax_attention(qh, keh, veh, w.scale))
        o_b = merge_heads(softmax_attention(qh, kbh, vbh, w.scale))
        return o_b, o_e

    o_b, o_e = two_sided(k, v)
    if ip is not None:
        kip, vip = _ip_kv(ip[::ip.shape[0] // x.shape[0]], ipw)     # [::3], interpolation.py:330-331
        ip_b, ip_e = two_sided(kip, vip)
        s = dt.type(ipw.scale)
        o_b = o_b + s * ip_b                            # interpolation.py:364-367
        o_e = o_e + s * ip_e
    return _out((1 - c) * o_b + c * o_e, w)             # interpolation.py:370-375


def inner_ip_attention(x, text, ip, w: AttnWeights, ipw: IPWeights, coef, is_fused: bool):
    """InnerInterpolatedIPAttnProcessor.__call__, interpolation.py:417-545.
    The IP branch attends with the frame's OWN image keys (the interpolated
    ones are computed and dropped, interpolation.py:512-527) and is only
    shape-valid when ``is_fused`` (keys are head-split at :520-521)."""
    q, k, v = _project(x, text, w)
    o = attn_core(q, k, v, w.heads, w.scale, "inner", is_fused, coef)
    if ip is not None:
        if not is_fused:
            raise RuntimeError("inner IP branch is shape-invalid without is_fused "
                               "(reference: bmm shape mismatch, interpolation.py:525)")
        kip, vip = _ip_kv(ip[::ip.shape[0] // x.shape[0]], ipw)     # [::3], interpolation.py:502-505
        o_ip = attn_core(q, kip, vip, w.heads, w.scale, "plain", False, None)
        o = o + x.dtype.type(ipw.scale) * o_ip          # interpolation.py:530
    return _out(o, w)


def ip_adapter_attention(x, text, ip, w: AttnWeights, ipw: IPWeights):
    """What the de-activated Outer / Inner IP processors compute: they return ``self.ip_attn(...)``
    (interpolation.py:248-251, 425-428), diffusers' IPAdapterAttnProcessor2_0 (third-party):
    plain text attention + scale[0] x plain attention over the image tokens, the [R, 1, T, Cc] tensor folded into
    the batch by ``view(batch, -1, heads, head_dim)`` (frame i gets the tokens of rows [i R/B, (i+1) R/B))."""
    q, k, v = _project(x, text, w)
    o = attn_core(q, k, v, w.heads, w.scale, "plain", False, None)
    if ip is not None:
        b = x.shape[0]
        kip = linear(ip, ipw.wk_ip).reshape(b, -1, q.shape[-1])
        vip = linear(ip, ipw.wv_ip).reshape(b, -1, q.shape[-1])
        o = o + x.dtype.type(ipw.scale) * attn_core(q, kip, vip, w.heads, w.scale, "plain", False, None)
    return _out(o, w)


def scale_control_ip_attention(x, text, ip, w: AttnWeights, ipw: IPWeights, coef,
                               is_fused: bool, activated: bool = True):
    """ScaleControlIPAttnProcessor.__call__, interpolation.py:76-211: text
    attention is outer-interpolated (activated) or plain (de-activated); the
    image attention of rows [6:9] is added with the per-frame coefficient."""
    q, k, v = _project(x, text, w)
    if activated:
        o = attn_core(q, k, v, w.heads, w.scale, "outer", is_fused, coef)   # :152-183
    else:
        o = attn_core(q, k, v, w.heads, w.scale, "plain", False, None)      # :130-135
    if ip is not None:
        n, r = x.shape[0], ip.shape[0] // x.shape[0]
        last = ip[r * (n - 1):] if r == n else np.repeat(ip[r * (n - 1): r * (n - 1) + 1], n, axis=0)
        kip, vip = _ip_kv(last, ipw)                    # [6:9] (the END frame's rows), :137-138 / :187-188
        o_ip = attn_core(q, kip, vip, w.heads, w.scale, "plain", False, None)
        o = o + _cvec(coef, x.dtype) * o_ip             # :146-150 / :196
    return _out(o, w)


# --------------------------------------------------------------------------
# latent / embedding initialisation (interpolation.py:807-918)
# --------------------------------------------------------------------------
def slerp(v0: np.ndarray, v1: np.ndarray, t: float, threshold: float = 0.9995) -> np.ndarray:
    """interpolation.py:861-918: row-wise (last dim) spherical interpolation,
    lerp where the rows are (anti-)colinear or a row is all zero (NaN dot)."""
    assert v0.shape == v1.shape
    dt = v0.dtype
    with np.errstate(invalid="ignore", divide="ignore"):
        n0 = np.linalg.norm(v0, axis=-1, keepdims=True)
        n1 = np.linalg.norm(v1, axis=-1, keepdims=True)
        dot = ((v0 / n0) * (v1 / n1)).sum(-1)
        mag = np.abs(dot)
        gotta_lerp = np.isnan(mag) | (mag > threshold)
        t = dt.type(t)
        lerped = v0 + t * (v1 - v0)
        theta0 = np.arccos(dot)[..., None]
        sin0 = np.sin(theta0)
        theta_t = theta0 * t
        s0 = np.sin(theta0 - theta_t) / sin0
        s1 = np.sin(theta_t) / sin0
        slerped = s0 * v0 + s1 * v1
    return np.where(gotta_lerp[..., None], lerped, slerped).astype(dt)


def linear_interpolation(l1: np.ndarray, l2: np.ndarray, ts: Optional[Sequence[float]] = None,
                         size: int = 5) -> np.ndarray:
    """interpolation.py:807-835."""
    assert l1.shape == l2.shape
    if ts is None:
        ts = [i / (size - 1) for i in range(size)]
    return np.concatenate([l1 + l1.dtype.type(t) * (l2 - l1) for t in ts], axis=0)


def spherical_interpolation(l1: np.ndarray, l2: np.ndarray, size: int = 5) -> np.ndarray:
    """interpolation.py:838-858."""
    return np.concatenate([slerp(l1, l2, i / (size - 1)) for i in range(size)], axis=0)
