"""AID / PAID attention processors — the drop-in boundary.

Same class names, constructor arguments, attributes (``size, coef, is_fused, activated``) and
methods (``activate(t)``, ``deactivate()``, ``load_end_point``) as the reference's
``interpolation.py``; same diffusers *AttnProcessor protocol*::

    proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None) -> Tensor

The arithmetic of every call — q/k/v projection, end-point K/V interpolation, the fused
softmax(QK^T/sqrt(d))V with own-key fusion, the outer/inner lerp and the output projection —
runs in ``libaid_hip.so`` (hand-written gfx950 kernels).  There is no eager / CPU fallback:
CPU tensors, fp32 tensors or a missing library raise.

Differences from the reference that are deliberate and documented:
  * de-activated processors with ``original_attn=None`` run plain attention on the same HIP
    kernel (the reference requires a wrapped diffusers processor);
  * the per-call host->device copy of ``coef`` (interpolation.py:663) is cached on the device;
  * optional ``ctx_index`` (attribute or keyword of ``__call__``, e.g. through ``cross_attention_kwargs``): frames
    that share a text context (PAID guide prompt) have its keys / values projected once.  Results are bit-identical
    to passing the repeated contexts.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import nn

from . import ops
from .interp import generate_beta_tensor


class InterpolatedAttnProcessor(nn.Module):
    """Base class: coefficient schedule and activation state (interpolation.py:10-48)."""

    def __init__(self, t: Optional[float] = None, size: int = 7, is_fused: bool = False,
                 alpha: float = 1, beta: float = 1):
        super().__init__()
        if t is None:
            ts = generate_beta_tensor(size, alpha=alpha, beta=beta)
            ts[0], ts[-1] = 0, 1
        else:
            assert t > 0 and t < 1, "t must be between 0 and 1"
            ts = torch.tensor([0, t, 1])
            size = 3
        self.size = size
        self.coef = ts
        self.is_fused = is_fused
        self.activated = True
        # build-specific: number of extra frames appended AFTER the `size` interpolated frames that run plain
        # attention in the same call (the unconditional half of a batched classifier-free-guidance step)
        self.plain_tail = 0
        # build-specific: frame -> row of ``encoder_hidden_states`` when several frames share one text context
        # (PAID guide prompt: [start, guide x (N-2), end] = 3 distinct contexts, sequence.py).  The keys / values of
        # a shared context are then projected once instead of once per frame.  None = one context per frame.
        self.ctx_index: Optional[Sequence[int]] = None
        self._coef_cache: Dict[Tuple, torch.Tensor] = {}
        self._ctx_cache: Dict[Tuple, Tuple] = {}

    def deactivate(self):
        self.activated = False

    def activate(self, t):
        self.activated = True
        assert t > 0 and t < 1, "t must be between 0 and 1"
        self.coef = torch.tensor([0, t, 1])

    def load_end_point(self, key_begin, value_begin, key_end, value_end):
        self.key_begin = key_begin
        self.value_begin = value_begin
        self.key_end = key_end
        self.value_end = value_end

    # ---- build-specific: the step either side of the call ----------------------
    _mode: Optional[str] = None          # "outer" / "inner" in the text subclasses

    def fused_sublayer(self, attn, norm, hidden_states, encoder_hidden_states=None, ctx_index=None):
        """``hidden_states + attn(norm(hidden_states), encoder_hidden_states)`` — what diffusers' BasicTransformerBlock
        computes around attn1 / attn2 (LayerNorm, processor call, residual add) — in ONE library call: the LayerNorm
        runs as a HIP kernel in front of the projections and the residual is added in the epilogue of the out
        projection (after its rounding, so the result is bit-identical to the three separate steps on the same
        kernels).  Falls back to the three steps where the one-call form does not apply (a wrapped foreign
        ``original_attn``, 4-D inputs, Attention extras)."""
        if self._mode is None:
            raise NotImplementedError("fused_sublayer is implemented for the text processors (outer / inner)")
        foreign = (not self.activated) and getattr(self, "original_attn", None) is not None \
            and not isinstance(self.original_attn, HipAttnProcessor)
        if foreign or not _plain_sublayer_ok(attn, hidden_states):
            return hidden_states + self(attn, norm(hidden_states), encoder_hidden_states, ctx_index=ctx_index)
        ctx_index = self.ctx_index if ctx_index is None else ctx_index
        hidden_states = hidden_states.contiguous()
        return _run_text(self, attn, hidden_states, encoder_hidden_states, None, None,
                         self._mode if self.activated else "plain", ctx_index, ln=_ln_of(norm), add_to=hidden_states)

    # ---- build-specific helpers ------------------------------------------------------------
    def _coef_device(self, device: torch.device, dtype: torch.dtype, batch: int) -> torch.Tensor:
        """``coef.to(key.device, key.dtype)`` (interpolation.py:663: coefficients are rounded to
        the compute dtype) kept resident on the device as fp32 for the kernel."""
        coef = self.coef
        if coef.numel() + self.plain_tail != batch:
            # the reference fails at the broadcast of the lerp (interpolation.py:664 / 774)
            raise RuntimeError(f"The size of tensor a ({coef.numel() + self.plain_tail}) must match the size of "
                               f"tensor b ({batch}) at non-singleton dimension 0")
        key = (id(coef), coef._version, device, dtype, self.plain_tail)
        hit = self._coef_cache.get(key)
        if hit is None:
            # entries are never dropped while they may be referenced: a captured hipGraph keeps reading the device
            # tensor it was captured with (the loop alternates between a few (schedule, plain_tail) combinations)
            if len(self._coef_cache) >= _CACHE_ENTRIES:
                self._coef_cache.pop(next(iter(self._coef_cache)))
            hit = coef.detach().to(torch.float32).to(dtype).to(torch.float32)
            if self.plain_tail:                       # negative coefficient = PLAIN rider frame (aid_hip.h)
                hit = torch.cat([hit, -torch.ones(self.plain_tail)])
            hit = hit.to(device).contiguous()
            self._coef_cache[key] = hit
        return hit


_CACHE_ENTRIES = 16      # distinct (schedule, plain_tail) / context maps a processor keeps resident on the device


def _shared_context(cache: Dict, ctx_index, ctx: torch.Tensor, batch: int):
    """(distinct contexts [n_distinct, L, Cc], device int32 map [batch], host list) for a frame -> context map."""
    idx = [int(i) for i in (ctx_index.tolist() if torch.is_tensor(ctx_index) else ctx_index)]
    if len(idx) != batch:
        raise RuntimeError(f"ctx_index has {len(idx)} entries for a batch of {batch} frames")
    n_distinct = max(idx) + 1
    if min(idx) < 0 or sorted(set(idx)) != list(range(n_distinct)):
        raise RuntimeError("ctx_index must use every context row 0 .. n_distinct-1")
    key = (tuple(idx), ctx.device)
    hit = cache.get(key)
    if hit is None:
        if len(cache) >= _CACHE_ENTRIES:          # see _coef_device: captured graphs hold on to these tensors
            cache.pop(next(iter(cache)))
        first = [idx.index(r) for r in range(n_distinct)]
        hit = (torch.tensor(idx, dtype=torch.int32, device=ctx.device),
               torch.tensor(first, dtype=torch.long, device=ctx.device))
        cache[key] = hit
    dev_map, first = hit
    if ctx.shape[0] == batch and n_distinct != batch:
        ctx = ctx.index_select(0, first)          # caller passed one (repeated) context per frame
    elif ctx.shape[0] != n_distinct:
        raise RuntimeError(f"encoder_hidden_states has {ctx.shape[0]} rows; ctx_index needs {n_distinct} (or {batch})")
    return ctx.contiguous(), dev_map, idx


# ---------------------------------------------------------------------------------------------
# shared pre/post-processing of a processor call (the non-attention lines of the reference bodies)
# ---------------------------------------------------------------------------------------------
def _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb):
    """interpolation.py:586-611 / 616-621.  Returns (residual, x[N,S,C], ctx or None, restore-4d info)."""
    if attention_mask is not None:
        raise NotImplementedError("attention_mask is not supported by the HIP path "
                                  "(UNet attention of SD / SDXL never passes one)")
    residual = hidden_states
    if getattr(attn, "spatial_norm", None) is not None:
        hidden_states = attn.spatial_norm(hidden_states, temb)
    shape4 = None
    if hidden_states.ndim == 4:
        b, ch, hh, ww = hidden_states.shape
        shape4 = (b, ch, hh, ww)
        hidden_states = hidden_states.view(b, ch, hh * ww).transpose(1, 2)
    if getattr(attn, "group_norm", None) is not None:
        hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
    if encoder_hidden_states is not None and getattr(attn, "norm_cross", None):
        encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
    return residual, hidden_states.contiguous(), encoder_hidden_states, shape4


def _epilogue(attn, hidden_states, residual, shape4):
    """interpolation.py:669-679."""
    if shape4 is not None:
        b, ch, hh, ww = shape4
        hidden_states = hidden_states.transpose(-1, -2).reshape(b, ch, hh, ww)
    if getattr(attn, "residual_connection", False):
        hidden_states = hidden_states + residual
    rof = getattr(attn, "rescale_output_factor", 1.0)
    if rof != 1.0:
        hidden_states = hidden_states / rof
    return hidden_states


def _weights(attn):
    wo = attn.to_out[0]
    return attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, wo.weight, wo.bias


def _ln_of(norm) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], float]:
    """(gamma, beta, eps) of the torch LayerNorm that sits in front of an attention layer."""
    if not isinstance(norm, nn.LayerNorm) or len(norm.normalized_shape) != 1:
        raise TypeError("fused_sublayer needs the block's nn.LayerNorm over the channel dimension")
    return norm.weight, norm.bias, float(norm.eps)


def _plain_sublayer_ok(attn, hidden_states) -> bool:
    """The one-call form  h + attn(norm(h))  covers the transformer-block attention of SD / SDXL: 3-D input and none of
    the Attention extras (spatial / group norm, own residual connection, output rescale, cross-attention norm)."""
    return (hidden_states.ndim == 3 and getattr(attn, "spatial_norm", None) is None
            and getattr(attn, "group_norm", None) is None and not getattr(attn, "residual_connection", False)
            and getattr(attn, "rescale_output_factor", 1.0) == 1.0 and not getattr(attn, "norm_cross", None))


def _run_text(proc: InterpolatedAttnProcessor, attn, hidden_states, encoder_hidden_states,
              attention_mask, temb, mode: str, ctx_index=None, ln=None, add_to=None):
    residual, x, ctx, shape4 = _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
    wq, wk, wv, wo, bo = _weights(attn)
    coef = None
    if mode != "plain":
        coef = proc._coef_device(x.device, x.dtype, x.shape[0])
    n_aid = proc.coef.numel()
    begin, end = 0, (n_aid - 1) if mode != "plain" else -1
    ctx_map = None
    ctx_index = proc.ctx_index if ctx_index is None else ctx_index
    if ctx is not None and ctx_index is not None:
        ctx, ctx_map, idx = _shared_context(proc._ctx_cache, ctx_index, ctx, x.shape[0])
        if mode != "plain":
            begin, end = idx[0], idx[n_aid - 1]       # end-point rows of the key / value tensors
    elif ctx is not None:
        ctx = ctx.contiguous()
    y = ops.processor_fwd(x, ctx, wq, wk, wv, wo, bo, attn.heads, mode=mode,
                          fused=proc.is_fused if mode != "plain" else False, coef=coef,
                          begin=begin, end=end, ctx_map=ctx_map,
                          n_plain=proc.plain_tail if mode != "plain" else 0, ln=ln, residual=add_to)
    return _epilogue(attn, y, residual, shape4)


class HipAttnProcessor:
    """Plain attention (what diffusers' AttnProcessor2_0 computes) on the HIP kernel — the
    ``original_attn`` this package installs so the de-activated passes of the denoising loop
    (every unconditional pass and every post-warm-up step) stay on one code path."""

    def __init__(self):
        self.ctx_index: Optional[Sequence[int]] = None      # see InterpolatedAttnProcessor.ctx_index
        self._ctx_cache: Dict[Tuple, Tuple] = {}

    def fused_sublayer(self, attn, norm, hidden_states, encoder_hidden_states=None, ctx_index=None):
        """``hidden_states + attn(norm(hidden_states), ...)`` in one library call (see InterpolatedAttnProcessor)."""
        if not _plain_sublayer_ok(attn, hidden_states):
            return hidden_states + self(attn, norm(hidden_states), encoder_hidden_states, ctx_index=ctx_index)
        x = hidden_states.contiguous()
        wq, wk, wv, wo, bo = _weights(attn)
        ctx, ctx_map = encoder_hidden_states, None
        ctx_index = self.ctx_index if ctx_index is None else ctx_index
        if ctx is not None and ctx_index is not None:
            ctx, ctx_map, _ = _shared_context(self._ctx_cache, ctx_index, ctx, x.shape[0])
        elif ctx is not None:
            ctx = ctx.contiguous()
        return ops.processor_fwd(x, ctx, wq, wk, wv, wo, bo, attn.heads, mode="plain", ctx_map=ctx_map,
                                 ln=_ln_of(norm), residual=x)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 *args, ctx_index=None, **kwargs):
        residual, x, ctx, shape4 = _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        wq, wk, wv, wo, bo = _weights(attn)
        ctx_map = None
        ctx_index = self.ctx_index if ctx_index is None else ctx_index
        if ctx is not None and ctx_index is not None:
            ctx, ctx_map, _ = _shared_context(self._ctx_cache, ctx_index, ctx, x.shape[0])
        elif ctx is not None:
            ctx = ctx.contiguous()
        y = ops.processor_fwd(x, ctx, wq, wk, wv, wo, bo, attn.heads, mode="plain", ctx_map=ctx_map)
        return _epilogue(attn, y, residual, shape4)


class OuterInterpolatedAttnProcessor(InterpolatedAttnProcessor):
    r"""Outer attention interpolation (interpolation.py:548-679):
    (1 - t) * A(Q_t, K_1, V_1) + t * A(Q_t, K_m, V_m); fused with self-attention:
    (1 - t) * A(Q_t, [K_t, K_1], [V_t, V_1]) + t * A(Q_t, [K_t, K_m], [V_t, V_m])."""
    _mode = "outer"

    def __init__(self, t: Optional[float] = None, size: int = 7, is_fused: bool = False,
                 alpha: float = 1, beta: float = 1, original_attn=None):
        super().__init__(t=t, size=size, is_fused=is_fused, alpha=alpha, beta=beta)
        self.original_attn = original_attn

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 ctx_index=None):
        ctx_index = self.ctx_index if ctx_index is None else ctx_index
        if not self.activated:
            if self.original_attn is not None:
                if ctx_index is not None:           # only a processor that knows the keyword can take the map
                    return self.original_attn(attn, hidden_states, encoder_hidden_states, attention_mask, temb,
                                              ctx_index=ctx_index)
                return self.original_attn(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
            return _run_text(self, attn, hidden_states, encoder_hidden_states, attention_mask, temb, "plain", ctx_index)
        return _run_text(self, attn, hidden_states, encoder_hidden_states, attention_mask, temb, "outer", ctx_index)


class InnerInterpolatedAttnProcessor(InterpolatedAttnProcessor):
    r"""Inner attention interpolation (interpolation.py:682-804): keys / values are interpolated
    between the end-point frames, A(Q_t, [K_t,] (1-t) K_1 + t K_m, [V_t,] (1-t) V_1 + t V_m)."""
    _mode = "inner"

    def __init__(self, t: Optional[float] = None, size: int = 7, is_fused: bool = False,
                 alpha: float = 1, beta: float = 1, original_attn=None):
        super().__init__(t=t, size=size, is_fused=is_fused, alpha=alpha, beta=beta)
        self.original_attn = original_attn

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 ctx_index=None):
        ctx_index = self.ctx_index if ctx_index is None else ctx_index
        if not self.activated:
            if self.original_attn is not None:
                if ctx_index is not None:           # only a processor that knows the keyword can take the map
                    return self.original_attn(attn, hidden_states, encoder_hidden_states, attention_mask, temb,
                                              ctx_index=ctx_index)
                return self.original_attn(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
            return _run_text(self, attn, hidden_states, encoder_hidden_states, attention_mask, temb, "plain", ctx_index)
        return _run_text(self, attn, hidden_states, encoder_hidden_states, attention_mask, temb, "inner", ctx_index)


# ---------------------------------------------------------------------------------------------
# IP-Adapter variants (batch hard-wired to 3 by the reference)
# ---------------------------------------------------------------------------------------------
class _IPBase(InterpolatedAttnProcessor):
    def __init__(self, t=None, size=7, is_fused=False, alpha=1, beta=1, ip_attn=None):
        super().__init__(t=t, size=size, is_fused=is_fused, alpha=alpha, beta=beta)
        self.num_tokens = ip_attn.num_tokens if hasattr(ip_attn, "num_tokens") else (16,)
        self.scale = ip_attn.scale if hasattr(ip_attn, "scale") else None
        self.ip_attn = ip_attn

    def _split(self, encoder_hidden_states):
        """interpolation.py:255-266: (text, [ip]) tuple or a concatenated tensor."""
        if encoder_hidden_states is None:
            return None, None
        if isinstance(encoder_hidden_states, tuple):
            return encoder_hidden_states
        end_pos = encoder_hidden_states.shape[1] - self.num_tokens[0]
        return encoder_hidden_states[:, :end_pos, :], [encoder_hidden_states[:, end_pos:, :]]

    def _ip_kv(self, rows: torch.Tensor):
        """to_k_ip[0] / to_v_ip[0] on the selected image-embedding rows; a 4-D [B, E, T, Cc] input
        folds E into the token axis (head_to_batch_dim on 4-D, interpolation.py:334-341)."""
        if rows.ndim == 4:
            rows = rows.reshape(rows.shape[0], rows.shape[1] * rows.shape[2], rows.shape[3])
        rows = rows.contiguous()
        return ops.project_kv(rows, self.ip_attn.to_k_ip[0].weight, self.ip_attn.to_v_ip[0].weight) + (rows.shape[1],)

    def _text_qkv(self, attn, x, text):
        wq, wk, wv, _, _ = _weights(attn)
        e = x if text is None else text.contiguous()
        q = ops.linear(x, wq)
        k, vt = ops.project_kv(e, wk, wv)
        return q, k, vt, e.shape[1]

    def _finish(self, attn, o, residual, shape4):
        _, _, _, wo, bo = _weights(attn)
        return _epilogue(attn, ops.linear(o, wo, bo), residual, shape4)


class OuterInterpolatedIPAttnProcessor(_IPBase):
    r"""Outer interpolation combined with the IP-Adapter image attention (interpolation.py:214-387):
    O = (1-c) [A_text_begin + s A_ip_begin] + c [A_text_end + s A_ip_end]."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        if not self.activated:
            return self.ip_attn(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        text, ip = self._split(encoder_hidden_states)
        residual, x, text, shape4 = _prologue(attn, hidden_states, text, attention_mask, temb)
        coef = self._coef_device(x.device, x.dtype, x.shape[0])
        if x.shape[0] != 3:
            raise RuntimeError("the IP processors are defined for a batch of 3 [start, target, end] "
                               "(interpolation.py:300-303)")
        q, k, vt, l = self._text_qkv(attn, x, text)
        o = ops.attn_fwd(q, k, vt, attn.heads, l=l, mode="outer", fused=self.is_fused, coef=coef)
        if ip is not None:
            kip, vtip, t_ip = self._ip_kv(ip[0][::3])                       # interpolation.py:330-331
            ops.attn_fwd(q, kip, vtip, attn.heads, l=t_ip, mode="outer", fused=self.is_fused, coef=coef,
                         out=o, accumulate=True, out_scale=float(self.scale[0]))   # :364-372 (linear in O)
        return self._finish(attn, o, residual, shape4)


class InnerInterpolatedIPAttnProcessor(_IPBase):
    r"""Inner interpolation combined with the IP-Adapter image attention (interpolation.py:390-545).
    As in the reference the image branch attends with each frame's OWN image keys and is only
    shape-valid with ``is_fused=True`` (interpolation.py:512-527)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        if not self.activated:
            return self.ip_attn(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        text, ip = self._split(encoder_hidden_states)
        residual, x, text, shape4 = _prologue(attn, hidden_states, text, attention_mask, temb)
        coef = self._coef_device(x.device, x.dtype, x.shape[0])
        if x.shape[0] != 3:
            raise RuntimeError("the IP processors are defined for a batch of 3 [start, target, end] "
                               "(interpolation.py:477-480)")
        if ip is not None and not self.is_fused:
            raise RuntimeError("InnerInterpolatedIPAttnProcessor needs is_fused=True when image embeddings are "
                               "passed: the reference's image branch multiplies un-split keys "
                               "(batch1 dim mismatch in bmm, interpolation.py:525)")
        q, k, vt, l = self._text_qkv(attn, x, text)
        o = ops.attn_fwd(q, k, vt, attn.heads, l=l, mode="inner", fused=self.is_fused, coef=coef)
        if ip is not None:
            kip, vtip, t_ip = self._ip_kv(ip[0][::3])                       # interpolation.py:502-505
            ops.attn_fwd(q, kip, vtip, attn.heads, l=t_ip, mode="plain", out=o, accumulate=True,
                         out_scale=float(self.scale[0]))                    # :525-530
        return self._finish(attn, o, residual, shape4)


class ScaleControlIPAttnProcessor(_IPBase):
    r"""Image-prompt scale control (interpolation.py:51-211): text attention is outer-interpolated
    (activated) or plain (de-activated); the image attention of rows [6:9] is added with the
    per-frame coefficient."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        text, ip = self._split(encoder_hidden_states)
        residual, x, text, shape4 = _prologue(attn, hidden_states, text, attention_mask, temb)
        coef = self._coef_device(x.device, x.dtype, x.shape[0])
        q, k, vt, l = self._text_qkv(attn, x, text)
        if self.activated:
            if x.shape[0] != 3:
                raise RuntimeError("the IP processors are defined for a batch of 3 (interpolation.py:152-155)")
            o = ops.attn_fwd(q, k, vt, attn.heads, l=l, mode="outer", fused=self.is_fused, coef=coef)
        else:
            o = ops.attn_fwd(q, k, vt, attn.heads, l=l, mode="plain")
        if ip is not None:
            kip, vtip, t_ip = self._ip_kv(ip[0][6:9])                       # interpolation.py:137-138 / 187-188
            ops.attn_fwd(q, kip, vtip, attn.heads, l=t_ip, mode="plain", out=o, accumulate=True,
                         frame_scale=coef)                                  # :146-150 / :196
        return self._finish(attn, o, residual, shape4)


# ---------------------------------------------------------------------------------------------
# installation over anything that exposes diffusers' attn_processors / set_attn_processor
# (pipeline_interpolated_sd.py:950-1020)
# ---------------------------------------------------------------------------------------------
def load_aid(unet, t: Optional[float] = 0.5, is_fused: bool = True, atype: str = "fused_outer",
             size: int = 7, alpha: float = 1, beta: float = 1, keep_original: bool = False) -> None:
    """Wrap EVERY attention layer of ``unet`` (attn1 and attn2) with an AID processor
    (pipeline_interpolated_sd.py:950-970).  ``keep_original=True`` keeps the processor that was
    installed before as ``original_attn`` exactly like the reference; the default routes the
    de-activated passes through the HIP plain-attention kernel instead."""
    procs = {}
    current = unet.attn_processors
    for name in current.keys():
        if name.startswith("encoder"):
            procs[name] = current[name]
            continue
        orig = current[name] if keep_original else HipAttnProcessor()
        if atype == "fused_outer":
            procs[name] = OuterInterpolatedAttnProcessor(t=t, size=size, is_fused=is_fused, alpha=alpha, beta=beta,
                                                         original_attn=orig)
        elif atype == "fused_inner":
            procs[name] = InnerInterpolatedAttnProcessor(t=t, size=size, is_fused=is_fused, alpha=alpha, beta=beta,
                                                         original_attn=orig)
        else:
            raise ValueError(f"atype must be 'fused_outer' or 'fused_inner', got {atype!r}")
    unet.set_attn_processor(procs)


def activate_aid(unet, it: float) -> None:
    """pipeline_interpolated_sd.py:1008-1012."""
    for name, proc in unet.attn_processors.items():
        if not name.startswith("encoder"):
            proc.activate(it)


def deactivate_aid(unet) -> None:
    """pipeline_interpolated_sd.py:1013-1017."""
    for name, proc in unet.attn_processors.items():
        if not name.startswith("encoder"):
            proc.deactivate()
