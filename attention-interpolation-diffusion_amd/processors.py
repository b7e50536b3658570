"""AID / PAID attention processors — the drop-in boundary.

Same class names, constructor arguments, attributes (``size, coef, is_fused, activated``) and
methods (``activate(t)``, ``deactivate()``, ``load_end_point``) as the reference's
``interpolation.py``; same diffusers *AttnProcessor protocol*::

    proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None) -> Tensor

The arithmetic of every call — q/k/v projection, end-point K/V interpolation, the fused
softmax(QK^T/sqrt(d))V with own-key fusion, the outer/inner lerp and the output projection —
runs in ``libaid_hip.so`` (hand-written gfx950 kernels; fp16 / bf16 storage on the fast kernels, fp32 storage — the reference's
SD1.x default — on correctness-first fp32 kernels).  There is no eager / CPU fallback: CPU tensors or a missing library raise.

Differences from the reference that are deliberate and documented:
  * de-activated processors with ``original_attn=None`` run plain attention on the same HIP
    kernel (the reference requires a wrapped diffusers processor);
  * the per-call host->device copy of ``coef`` (interpolation.py:663) is cached on the device;
  * optional ``ctx_index`` (attribute or keyword of ``__call__``, e.g. through ``cross_attention_kwargs``): frames
    that share a text context (PAID guide prompt) have its keys / values projected once.  Results are bit-identical
    to passing the repeated contexts.
"""
from __future__ import annotations

import os
import weakref

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
        # device-side coefficient buffers, one per (device, dtype, number of coefficients, plain_tail) LAYOUT.  A buffer
        # is allocated once and from then on only rewritten in place, so a captured hipGraph that holds its address
        # keeps reading live values; ``_coef_dev[key] = [device tensor, values it holds, coef._version they came from]``
        self._coef_dev: Dict[Tuple, list] = {}
        self._ctx_cache: Dict[Tuple, Tuple] = {}
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
        # build-specific (SURVEY.md §8e alternative / §8f.4): frame-sharded run WITHOUT replicated end points.  The batch
        # holds only the rank's own frames; ``endpoint_exchange`` (dist.EndpointExchange) fetches the projected keys /
        # values of frames 0 and N-1 from their owner ranks in every self-attention call, and ``endpoint_ctx``
        # ([2, L, Cc]: text contexts of frames 0 and N-1, known to every rank) serves the cross-attention calls.
        self.endpoint_exchange = None
        self.endpoint_ctx: Optional[torch.Tensor] = None

    # ``coef`` stays a plain CPU tensor attribute like the reference's (interpolation.py:21-27, 42); assigning it
    # (``activate(t)``, ``proc.coef = ...``) refreshes the device copies at once, so a replayed graph never sees the
    # previous schedule.
    @property
    def coef(self) -> torch.Tensor:
        return self._coef

    @coef.setter
    def coef(self, value) -> None:
        self._coef = value if torch.is_tensor(value) else torch.as_tensor(value, dtype=torch.float32)
        self._refresh_coef_buffers()

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
        is folded into the projections (only its row statistics are computed; LayerNorm(x) is never written — or, for
        widths the fold does not cover, it runs as a HIP kernel in front of them) and the residual is added in the
        epilogue of the out projection, after its rounding like the block's separate add.  Falls back to the three steps where the one-call form does not apply (a wrapped foreign
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
                         self._mode if self.activated else "plain", ctx_index, ln=_ln_of(norm), add_to=hidden_states,
                         ln_folded=_ln_folded(attn, norm, encoder_hidden_states is not None))

    # ---- build-specific helpers ------------------------------------------------------------
    def _coef_device(self, device: torch.device, dtype: torch.dtype, batch: int) -> torch.Tensor:
        return self._coef_state(device, dtype, batch)[0]

    def _coef_state(self, device: torch.device, dtype: torch.dtype, batch: int) -> Tuple[torch.Tensor, Tuple[float, ...]]:
        """``coef.to(key.device, key.dtype)`` (interpolation.py:663: coefficients are rounded to
        the compute dtype) kept resident on the device as fp32 for the kernel; returns (device tensor, host values)."""
        coef = self.coef
        if coef.numel() + self.plain_tail != batch:
            # the reference fails at the broadcast of the lerp (interpolation.py:664 / 774)
            raise RuntimeError(f"The size of tensor a ({coef.numel() + self.plain_tail}) must match the size of "
                               f"tensor b ({batch}) at non-singleton dimension 0")
        key = (device, dtype, coef.numel(), self.plain_tail)
        ent = self._coef_dev.get(key)
        ver = _version_of(coef)                       # None: no version counter (inference-mode tensor) -> compare values
        if ent is None:
            vals = _coef_values(coef, dtype, self.plain_tail)
            ent = [torch.tensor(vals, dtype=torch.float32).to(device), vals, ver]
            self._coef_dev[key] = ent
        elif ver is None or ent[2] != ver:            # the tensor was (or may have been) mutated in place (proc.coef[1] = t)
            _rewrite(ent, _coef_values(coef, dtype, self.plain_tail), ver)
        return ent[0], ent[1]

    def _refresh_coef_buffers(self) -> None:
        """After ``coef`` was re-assigned: bring every device buffer of the matching layout up to date in place."""
        coef = self._coef
        for (device, dtype, n, tail), ent in self._coef_dev.items():
            if n == coef.numel():
                _rewrite(ent, _coef_values(coef, dtype, tail), _version_of(coef))


def _version_of(t: torch.Tensor) -> Optional[int]:
    """In-place edit counter of a tensor, or None where it cannot be read (tensors created under torch.inference_mode()
    raise on ``_version``): callers then fall back to comparing values (the small ``coef``) or to the address alone."""
    try:
        return t._version
    except RuntimeError:
        return None


def _coef_values(coef: torch.Tensor, dtype: torch.dtype, plain_tail: int) -> Tuple[float, ...]:
    """Host values of a device coefficient buffer: ``coef`` rounded to the compute dtype (interpolation.py:663), then
    ``plain_tail`` negative entries (negative coefficient = PLAIN rider frame, aid_hip.h)."""
    vals = coef.detach().to(torch.float32).cpu().to(dtype).to(torch.float32).tolist()
    return tuple(vals) + (-1.0,) * plain_tail


def _rewrite(ent: list, vals: Tuple[float, ...], version: int) -> None:
    """Update a device coefficient buffer IN PLACE (its address may be baked into captured graphs)."""
    if vals != ent[1]:
        if torch.cuda.is_available() and ent[0].is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the coefficient schedule changed inside a stream capture; call the processor once "
                               "eagerly (or assign `coef` / activate(t)) before capturing")
        ent[0].copy_(torch.tensor(vals, dtype=torch.float32))
        ent[1] = vals
    ent[2] = version


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
        # keyed by the map's VALUES and never dropped: a captured hipGraph keeps reading the device tensors it was
        # captured with, and a run uses a handful of distinct maps (a few hundred bytes each)
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
    """interpolation.py:586-611 / 616-621.  Returns (residual, x[N,S,C], ctx or None, restore-4d info, prepared mask or None)."""
    residual = hidden_states
    if getattr(attn, "spatial_norm", None) is not None:
        hidden_states = attn.spatial_norm(hidden_states, temb)
    shape4 = None
    if hidden_states.ndim == 4:
        b, ch, hh, ww = hidden_states.shape
        shape4 = (b, ch, hh, ww)
        hidden_states = hidden_states.view(b, ch, hh * ww).transpose(1, 2)
    if attention_mask is not None:
        # interpolation.py:598-606: the mask is prepared for the KEY sequence (the context's length, or the hidden states')
        batch_size, sequence_length, _ = (hidden_states.shape if encoder_hidden_states is None
                                          else encoder_hidden_states.shape)
        attention_mask = attn.prepare_attention_mask(attention_mask, sequence_length, batch_size)
    if getattr(attn, "group_norm", None) is not None:
        hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
    if encoder_hidden_states is not None and getattr(attn, "norm_cross", None):
        encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
    return residual, hidden_states.contiguous(), encoder_hidden_states, shape4, attention_mask


def _score_bias(mask, x: torch.Tensor, l: int, fused: bool, what: str):
    """The prepared attention_mask as the library's additive score bias (AidAttnArgs.bias).  The reference adds it to the scores of
    every key segment through ``baddbmm(attention_mask, q, k^T, beta=1, alpha=scale)`` (App. A; interpolation.py:651-656, 787): a
    FUSED call — keys ``[own ; end-point]``, 2 L wide — fails there at the broadcast of the L-wide mask, and so does this one."""
    if mask is None:
        return None
    if not torch.is_tensor(mask) or mask.dtype != x.dtype:
        raise RuntimeError(f"attention_mask must be a {x.dtype} tensor like the hidden states (the reference adds it with baddbmm), "
                           f"got {getattr(mask, 'dtype', type(mask))}")
    if fused:
        raise RuntimeError(f"The expanded size of the tensor ({2 * l}) must match the existing size ({mask.shape[-1]}) at "
                           f"non-singleton dimension 2 ({what}: an attention_mask covers one key segment; the fused "
                           "[own ; end-point] keys are twice as long — the reference fails at the same broadcast)")
    if mask.shape[-1] != l:
        raise RuntimeError(f"The expanded size of the tensor ({l}) must match the existing size ({mask.shape[-1]}) at "
                           "non-singleton dimension 2 (attention_mask vs keys)")
    return mask if mask.stride(-1) == 1 or mask.shape[-1] == 1 else mask.contiguous()


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


# Folded LayerNorm (include/aid_hip.h "aid_ln_fold"): per attention module, the projection weights
# pre-multiplied by the LayerNorm's gamma plus the two fp32 constant vectors per projection.  Rebuilt when a weight, gamma
# or beta tensor is replaced or edited in place (data_ptr + _version).  AID_LN_FOLD=0 keeps the LayerNorm as its own pass.
_FOLD_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _ln_folded(attn, norm, cross: bool):
    """(wq', wk', wv', const [6, C]) for ops.processor_fwd(ln_folded=...), or None where folding does not apply."""
    if os.environ.get("AID_LN_FOLD", "1") == "0":
        return None
    wq, wk, wv, _, _ = _weights(attn)
    c = wq.shape[1]
    if c % 64 or (not cross and (wk.shape[1] != c or wv.shape[1] != c)):
        return None
    srcs = (wq, norm.weight, norm.bias) if cross else (wq, wk, wv, norm.weight, norm.bias)
    # (an inference-mode weight has no version counter: its address alone is the key, and graphs captured with folded
    #  weights must be re-captured after any weight edit — the rebuilt tensors live at new addresses)
    key = (cross,) + tuple((t.data_ptr(), _version_of(t)) if t is not None else None for t in srcs)
    ent = _FOLD_CACHE.get(attn)
    if ent is None or ent[0] != key:
        _CACHE_GEN[0] += 1
        const = torch.zeros(6, wq.shape[0], dtype=torch.float32, device=wq.device)
        fq, const[0], const[1] = ops.ln_fold(wq, norm.weight, norm.bias)
        fk = fv = None
        if not cross:
            fk, const[2], const[3] = ops.ln_fold(wk, norm.weight, norm.bias)
            fv, const[4], const[5] = ops.ln_fold(wv, norm.weight, norm.bias)
        ent = (key, (fq, fk, fv, const))
        _FOLD_CACHE[attn] = ent
    return ent[1]


# Step-invariant text keys / values (VERDICT r2 missing #5).  The reference hands the SAME prompt_embeds tensor to every
# denoising step (pipeline_interpolated_sd.py:1859-1867), so K = to_k(ctx) and V^T = Wv ctx^T of a cross-attention layer
# are the same numbers in all 50 steps; the reference recomputes them 50 times.  Here they are projected once per
# (attention module, context tensor, frame -> context map, to_k / to_v weights) and handed to every later call
# (AidProcessorArgs.k_cached / vt_cached): the grouped launch of a cross-attention call then holds the query projection alone.
# An entry is keyed by (data_ptr, _version) of the context tensor the CALLER passed and of the two weights — an in-place edit
# bumps _version and misses — and it disappears when that context tensor is garbage-collected (weakref), so a recycled
# address can never hit.  Tensors whose version counter cannot be read (inference mode) are not cached.
# Captured hipGraphs hold the cached buffers' addresses: they stay valid as long as the context tensor they were captured
# with is alive; re-capture after replacing weights.  ``TEXT_KV_CACHE = False`` switches the cache off.
#
# What the keys CANNOT see: an edit through ``.data`` (``w.data += delta`` — peft's LoRA ``merge`` — or ``w.data.copy_()``)
# changes neither the address nor ``_version``.  Therefore the caches are scoped to a RUN: ``clear_weight_caches()`` is called by
# ``load_aid`` / ``load_aid_ip_adapter`` / ``install_sequence_processors`` and at the start of every ``interpolate`` /
# ``interpolate_single``; a caller that edits weights in place BETWEEN processor calls of its own loop must call it too
# (INTEGRATION.md §3).
TEXT_KV_CACHE = True
_KV_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
# Bumped whenever a lazily built shared tensor is (re)built or dropped — a text K / V projection, a folded LayerNorm weight set, a
# clear_*() call.  ``loop.fork_join`` compares it around the pass it runs first: a fill on one stream must be ordered before the
# other stream reads it.
_CACHE_GEN = [0]


def cache_generation() -> int:
    return _CACHE_GEN[0]


def clear_text_kv_cache() -> None:
    """Drop every cached text key / value projection (they are re-projected by the next cross-attention call)."""
    _KV_CACHE.clear()
    _CACHE_GEN[0] += 1


def clear_weight_caches() -> None:
    """Drop everything derived from attention WEIGHTS: the text K / V cache and the folded LayerNorm weights.  Call after an
    in-place weight edit that leaves ``data_ptr`` and ``_version`` unchanged (``w.data += ...``, a merged LoRA); graphs captured
    before the call still read the old derived tensors and must be re-captured."""
    _KV_CACHE.clear()
    _FOLD_CACHE.clear()
    _CACHE_GEN[0] += 1


def _vkey(t: torch.Tensor):
    v = _version_of(t)                    # inference-mode tensor: in-place edits cannot be detected -> not cached
    return None if v is None else (t.data_ptr(), v)


def _text_kv(attn, ehs, ctx: torch.Tensor, idx, wk: torch.Tensor, wv: torch.Tensor):
    """(k [n_ctx, L, C], vt [n_ctx, C, Lp]) of the distinct contexts ``ctx`` derived from the caller's tensor ``ehs``
    (``idx`` = frame -> context map or None), from the cache or projected now; None where caching does not apply."""
    if not TEXT_KV_CACHE or ehs is None or not torch.is_tensor(ehs):
        return None
    ks = (_vkey(ehs), _vkey(wk), _vkey(wv))
    if None in ks:
        return None
    key = ks + (tuple(ehs.shape), ehs.dtype, tuple(idx) if idx is not None else None)
    try:
        per = _KV_CACHE.get(attn)
        if per is None:
            per = {}
            _KV_CACHE[attn] = per
    except TypeError:                     # an attention object that cannot be weakly referenced
        return None
    hit = per.get(key)
    if hit is None:
        _CACHE_GEN[0] += 1
        k, vt = ops.project_kv(ctx, wk, wv)
        for old in [kk for kk in per if kk[0][0] == key[0][0] and kk[3:] == key[3:]]:
            per.pop(old, None)            # the same tensor at an older version (or with replaced weights)

        def _drop(_ref, per=per, key=key):
            per.pop(key, None)
        hit = (k, vt, weakref.ref(ehs, _drop))
        per[key] = hit
    return hit[0], hit[1]


def _plain_sublayer_ok(attn, hidden_states) -> bool:
    """The one-call form  h + attn(norm(h))  covers the transformer-block attention of SD / SDXL: 3-D input and none of
    the Attention extras (spatial / group norm, own residual connection, output rescale, cross-attention norm)."""
    return (hidden_states.ndim == 3 and hidden_states.dtype != torch.float32       # (fp32 storage: norm, call, add as three steps)
            and getattr(attn, "spatial_norm", None) is None
            and getattr(attn, "group_norm", None) is None and not getattr(attn, "residual_connection", False)
            and getattr(attn, "rescale_output_factor", 1.0) == 1.0 and not getattr(attn, "norm_cross", None))


def _run_text(proc: InterpolatedAttnProcessor, attn, hidden_states, encoder_hidden_states,
              attention_mask, temb, mode: str, ctx_index=None, ln=None, add_to=None, ln_folded=None):
    residual, x, ctx, shape4, mask = _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
    ehs = ctx                                         # the tensor the caller holds on to across steps (cache key)
    wq, wk, wv, wo, bo = _weights(attn)
    bias = _score_bias(mask, x, x.shape[1] if ctx is None else ctx.shape[1], mode != "plain" and proc.is_fused,
                       type(proc).__name__)
    coef = vals = None
    if mode != "plain":
        coef, vals = proc._coef_state(x.device, x.dtype, x.shape[0])
    n_aid = proc.coef.numel()
    begin, end = 0, (n_aid - 1) if mode != "plain" else -1
    ctx_map = idx = None
    ctx_index = proc.ctx_index if ctx_index is None else ctx_index
    exchange = getattr(proc, "endpoint_exchange", None)
    if exchange is not None and mode != "plain":
        if bias is not None:
            raise NotImplementedError("the end-point exchange layout takes no attention_mask")
        if ln is not None or add_to is not None:
            raise NotImplementedError("the end-point exchange layout runs the plain processor call (no sublayer fusion)")
        n = x.shape[0]
        if ctx is None:                               # self-attention: keys / values of frames 0 / N-1 come from their owners
            # the four broadcasts run on the exchange's side stream while this stream does the q projection; one event per layer
            k, vt = ops.project_kv(x, wk, wv, extra_rows=2)
            pending = exchange.exchange_async(k, vt, n)
            q = ops.linear(x, wq)
            begin, end = pending.wait()
            o = ops.attn_fwd(q, k, vt, attn.heads, l=x.shape[1], mode=mode, fused=proc.is_fused, coef=coef,
                             begin=begin, end=end, n_plain=proc.plain_tail)
            return _epilogue(attn, ops.linear(o, wo, bo), residual, shape4)
        if proc.endpoint_ctx is None:
            raise RuntimeError("endpoint_exchange is set: cross-attention needs `endpoint_ctx` (text contexts of the two "
                               "end-point frames)")
        # local frames keep their (possibly shared) context rows; the two end-point contexts are appended behind them
        base = [int(i) for i in ctx_index] if ctx_index is not None else list(range(n))
        nctx = max(base) + 1
        if len(base) != n or min(base) < 0 or sorted(set(base)) != list(range(nctx)):
            raise RuntimeError("ctx_index must have one entry per local frame and use every context row 0 .. n_distinct-1")
        # the concatenated tensor [distinct local contexts ; two end-point contexts] is loop-invariant: built ONCE per
        # (caller's context tensor, map) — no per-call host -> device copy in front of the side-stream exchange —
        # and it is the tensor the text K / V cache keys on
        ectx = proc.endpoint_ctx
        ck = ("exchange", tuple(base), ctx.data_ptr(), _version_of(ctx), tuple(ctx.shape), ctx.dtype, ctx.device,
              ectx.data_ptr(), _version_of(ectx), tuple(ectx.shape), ectx.dtype, ectx.device)
        hit = proc._ctx_cache.get(ck)
        # an entry is only as good as the two tensors it was built from: a fresh prompt-embedding tensor at a recycled address
        # (same shape, version 0) must not find the previous prompt's contexts — the weak references say whether the
        # objects are still THE objects, and their callbacks drop the entry when either tensor dies
        if hit is not None and (hit[1]() is not ctx or hit[2]() is not ectx):
            proc._ctx_cache.pop(ck, None)
            hit = None
        if hit is None:
            own = ctx
            if own.shape[0] == n and nctx != n:
                own = own.index_select(0, torch.tensor([base.index(r) for r in range(nctx)], device=ctx.device))
            elif own.shape[0] != nctx:
                raise RuntimeError(f"encoder_hidden_states has {ctx.shape[0]} rows; the frame -> context map needs {nctx}")
            for old_key in [k_ for k_ in proc._ctx_cache if k_ and k_[0] == "exchange"]:
                proc._ctx_cache.pop(old_key, None)                # one live entry per processor (per run)
            def _drop(_ref, cache=proc._ctx_cache, key=ck):
                cache.pop(key, None)
            hit = (torch.cat([own, ectx.to(ctx.dtype)], dim=0).contiguous(), weakref.ref(ctx, _drop), weakref.ref(ectx, _drop))
            proc._ctx_cache[ck] = hit
        full = hit[0]
        full_map = base + [nctx, nctx + 1]
        ctx2, ctx_map, idx2 = _shared_context(proc._ctx_cache, full_map, full, n + 2)
        y = ops.processor_fwd(x, ctx2, wq, wk, wv, wo, bo, attn.heads, mode=mode, fused=proc.is_fused, coef=coef,
                              begin=nctx, end=nctx + 1, ctx_map=ctx_map[:n], n_plain=proc.plain_tail,
                              kv_cached=_text_kv(attn, full, ctx2, idx2, wk, wv))
        return _epilogue(attn, y, residual, shape4)
    if ctx is not None and ctx_index is not None:
        ctx, ctx_map, idx = _shared_context(proc._ctx_cache, ctx_index, ctx, x.shape[0])
        if mode != "plain":
            begin, end = idx[0], idx[n_aid - 1]       # end-point rows of the key / value tensors
    elif ctx is not None:
        ctx = ctx.contiguous()
    fused = proc.is_fused if mode != "plain" else False
    y = ops.processor_fwd(x, ctx, wq, wk, wv, wo, bo, attn.heads, mode=mode, fused=fused, coef=coef,
                          begin=begin, end=end, ctx_map=ctx_map,
                          n_plain=proc.plain_tail if mode != "plain" else 0, ln=ln, residual=add_to, ln_folded=ln_folded,
                          seg_executed=ops.executed_segments(mode, fused, vals, x.shape[0], idx, begin, end),
                          kv_cached=_text_kv(attn, ehs, ctx, idx, wk, wv) if ctx is not None else None, attn_bias=bias)
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
        ctx, ctx_map, idx = encoder_hidden_states, None, None
        ctx_index = self.ctx_index if ctx_index is None else ctx_index
        if ctx is not None and ctx_index is not None:
            ctx, ctx_map, idx = _shared_context(self._ctx_cache, ctx_index, ctx, x.shape[0])
        elif ctx is not None:
            ctx = ctx.contiguous()
        return ops.processor_fwd(x, ctx, wq, wk, wv, wo, bo, attn.heads, mode="plain", ctx_map=ctx_map,
                                 ln=_ln_of(norm), residual=x, ln_folded=_ln_folded(attn, norm, ctx is not None),
                                 kv_cached=_text_kv(attn, encoder_hidden_states, ctx, idx, wk, wv) if ctx is not None else None)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 *args, ctx_index=None, **kwargs):
        residual, x, ctx, shape4, mask = _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        ehs = ctx
        wq, wk, wv, wo, bo = _weights(attn)
        bias = _score_bias(mask, x, x.shape[1] if ctx is None else ctx.shape[1], False, "HipAttnProcessor")
        ctx_map = idx = None
        ctx_index = self.ctx_index if ctx_index is None else ctx_index
        if ctx is not None and ctx_index is not None:
            ctx, ctx_map, idx = _shared_context(self._ctx_cache, ctx_index, ctx, x.shape[0])
        elif ctx is not None:
            ctx = ctx.contiguous()
        y = ops.processor_fwd(x, ctx, wq, wk, wv, wo, bo, attn.heads, mode="plain", ctx_map=ctx_map,
                              kv_cached=_text_kv(attn, ehs, ctx, idx, wk, wv) if ctx is not None else None, attn_bias=bias)
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
# IP-Adapter variants.  The reference hard-wires a batch of 3 (literal ``expand(3, ...)``, ``[::3]``, ``[6:9]``;
# SURVEY.md App. D5).  Here the batch is ``coef.numel()`` frames and the image-embedding tensor ``[R, 1, T, Cc]`` holds
# r = R / N copies per frame (r = 3, R = 9 in the reference's pipelines): ``[::3]`` becomes ``[::r]`` and ``[6:9]`` the
# rows of the last frame — identical for N = 3, and an N-frame sequence (one image-embedding row group per frame) runs
# too.  Every variant is ONE library call (aid_processor_fwd with the ip_* fields): the image keys / values are projected
# in the same grouped launch as q / k / V^T and the image attention accumulates into the text attention's output.
# ---------------------------------------------------------------------------------------------
def _split_ip(encoder_hidden_states, num_tokens):
    """interpolation.py:255-266: (text, [ip]) tuple or a concatenated tensor split at ``shape[1] - num_tokens[0]``."""
    if encoder_hidden_states is None:
        return None, None
    if isinstance(encoder_hidden_states, tuple):
        return encoder_hidden_states
    end_pos = encoder_hidden_states.shape[1] - num_tokens[0]
    return encoder_hidden_states[:, :end_pos, :], [encoder_hidden_states[:, end_pos:, :]]


def _ip_token_rows(ip_list) -> torch.Tensor:
    """ip_hidden_states[0] as [R, E*T, Cc]: a 4-D [R, E, T, Cc] input folds E into the token axis
    (head_to_batch_dim on 4-D tensors, interpolation.py:334-341)."""
    if len(ip_list) != 1:
        raise NotImplementedError("one IP-Adapter per layer (the reference reads to_k_ip[0] / scale[0] only)")
    rows = ip_list[0]
    if rows.ndim == 4:
        rows = rows.reshape(rows.shape[0], rows.shape[1] * rows.shape[2], rows.shape[3])
    if rows.ndim != 3:
        raise RuntimeError(f"image embeddings must be [R, T, Cc] or [R, E, T, Cc], got {tuple(rows.shape)}")
    return rows.contiguous()


class HipIPAdapterAttnProcessor(nn.Module):
    """What diffusers' ``IPAdapterAttnProcessor2_0`` computes (third-party, restated), on the HIP
    kernels: plain text attention + ``scale[0]`` x plain attention over the frame's image tokens.  The reference's IP
    processors call exactly this when de-activated (interpolation.py:248-251, 425-428) — every unconditional pass and
    every post-warm-up step of an IP run.  An image-embedding tensor with more rows than frames (the pipelines'
    ``[9, 1, T, Cc]`` for a batch of 3) is folded like diffusers' ``view(batch, -1, heads, head_dim)`` does: frame i
    attends over the tokens of rows ``[i r, (i + 1) r)``.
    Owns (or shares, see :meth:`wrap`) ``to_k_ip`` / ``to_v_ip`` / ``scale`` / ``num_tokens``."""

    def __init__(self, hidden_size: Optional[int] = None, cross_attention_dim: Optional[int] = None,
                 num_tokens=(4,), scale=1.0, dtype=None, device=None):
        super().__init__()
        self.num_tokens = tuple(num_tokens) if isinstance(num_tokens, (tuple, list)) else (num_tokens,)
        self.scale = list(scale) if isinstance(scale, (tuple, list)) else [scale] * len(self.num_tokens)
        if hidden_size is not None:
            kw = dict(dtype=dtype, device=device)
            self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False, **kw)
                                          for _ in self.num_tokens])
            self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False, **kw)
                                          for _ in self.num_tokens])

    @classmethod
    def wrap(cls, ip_attn) -> "HipIPAdapterAttnProcessor":
        """Share the weights / scale list / token counts of an installed IP-Adapter processor (diffusers' or a shim)."""
        self = cls(num_tokens=ip_attn.num_tokens, scale=ip_attn.scale)
        self.scale = ip_attn.scale                     # the SAME list: pipeline.set_ip_adapter_scale keeps working
        self.to_k_ip, self.to_v_ip = ip_attn.to_k_ip, ip_attn.to_v_ip
        return self

    def fused_sublayer(self, attn, norm, hidden_states, encoder_hidden_states=None, ctx_index=None):
        return hidden_states + self(attn, norm(hidden_states), encoder_hidden_states)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale: float = 1.0, ip_adapter_masks=None):
        if ip_adapter_masks is not None:
            raise NotImplementedError("ip_adapter_masks are not supported by the HIP path")
        text, ip = _split_ip(encoder_hidden_states, self.num_tokens)
        residual, x, text, shape4, mask = _prologue(attn, hidden_states, text, attention_mask, temb)
        wq, wk, wv, wo, bo = _weights(attn)
        # diffusers' IPAdapterAttnProcessor2_0 masks the TEXT attention only (the image branch runs with attn_mask=None)
        bias = _score_bias(mask, x, x.shape[1] if text is None else text.shape[1], False, "HipIPAdapterAttnProcessor")
        if bias is not None and ip is not None and float(self.scale[0]) != 0.0:
            raise NotImplementedError("attention_mask together with image embeddings: the one-call form has no mask for the text "
                                      "branch alone (AidProcessorArgs.attn_bias is refused with ip)")
        branch = None
        if ip is not None and float(self.scale[0]) != 0.0:
            rows = _ip_token_rows(ip)
            n = x.shape[0]
            if rows.shape[0] % n:
                raise RuntimeError(f"{rows.shape[0]} image-embedding rows do not fold into a batch of {n}")
            tokens = rows.reshape(n, -1, rows.shape[-1])          # diffusers: ip_key.view(batch, -1, heads, head_dim)
            branch = dict(tokens=tokens, wk=self.to_k_ip[0].weight, wv=self.to_v_ip[0].weight, mode="plain",
                          scale=float(self.scale[0]))
        y = ops.processor_fwd(x, None if text is None else text.contiguous(), wq, wk, wv, wo, bo, attn.heads,
                              mode="plain", ip=branch, attn_bias=bias)
        return _epilogue(attn, y, residual, shape4)


class _IPBase(InterpolatedAttnProcessor):
    def __init__(self, t=None, size=7, is_fused=False, alpha=1, beta=1, ip_attn=None):
        super().__init__(t=t, size=size, is_fused=is_fused, alpha=alpha, beta=beta)
        self.num_tokens = ip_attn.num_tokens if hasattr(ip_attn, "num_tokens") else (16,)
        self.scale = ip_attn.scale if hasattr(ip_attn, "scale") else None
        self.ip_attn = ip_attn

    def fused_sublayer(self, attn, norm, hidden_states, encoder_hidden_states=None, ctx_index=None):
        """``h + attn(norm(h), ctx)`` as three steps (the one-call form covers the text processors only)."""
        return hidden_states + self(attn, norm(hidden_states), encoder_hidden_states)

    def _fallback(self, attn, hidden_states, encoder_hidden_states, attention_mask, temb):
        """De-activated: the wrapped processor, like the reference (interpolation.py:248-251).  A wrapped object that
        cannot be called (a bare weight holder) runs the HIP IP-Adapter attention on its weights instead."""
        if callable(self.ip_attn) and not getattr(self.ip_attn, "weights_only", False):
            return self.ip_attn(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        if hasattr(self.ip_attn, "to_k_ip"):
            return HipIPAdapterAttnProcessor.wrap(self.ip_attn)(attn, hidden_states, encoder_hidden_states,
                                                                attention_mask, temb)
        return HipAttnProcessor()(attn, hidden_states, encoder_hidden_states, attention_mask, temb)

    def _frames(self, x) -> int:
        n = self.coef.numel()
        if x.shape[0] != n:
            raise RuntimeError(f"the IP processors run a batch of coef.numel() = {n} frames "
                               f"[start, ..., end] (the reference hard-wires 3, interpolation.py:300-303); "
                               f"got a batch of {x.shape[0]}")
        return n

    def _image_rows(self, ip, n: int, which: str):
        """Rows of the image-embedding tensor a variant attends with, as a (strided) [rows, T', Cc] view plus the
        frame -> row map (None = identity).  ``per_frame``: row group i of frame i — the reference's ``[::3]``
        (interpolation.py:330-331, 502-505); ``last``: the rows of the END frame — its ``[6:9]`` (:137-138, 187-188)."""
        rows = _ip_token_rows(ip)
        if rows.shape[0] % n:
            raise RuntimeError(f"{rows.shape[0]} image-embedding rows for {n} frames")
        r = rows.shape[0] // n
        if which == "per_frame":
            return rows[::r], None
        if r == n:                                       # N = 3, R = 9: literally rows [6:9], one per frame
            return rows[rows.shape[0] - r:], None
        last = rows[r * (n - 1): r * (n - 1) + 1]
        return last, _zeros_map(self._ctx_cache, n, rows.device)

    def _call(self, attn, hidden_states, encoder_hidden_states, attention_mask, temb, mode, branch_of):
        text, ip = _split_ip(encoder_hidden_states, self.num_tokens)
        residual, x, text, shape4, mask = _prologue(attn, hidden_states, text, attention_mask, temb)
        if mask is not None:
            # the reference hands the text-length mask to the IMAGE attention too (interpolation.py:141-143, 191-193, 352-359,
            # 525): T image tokens against an L-wide mask fails at the broadcast unless T == L, and fused text keys fail first
            raise RuntimeError(f"The expanded size of the tensor must match the existing size ({mask.shape[-1]}) at non-singleton "
                               "dimension 2: the IP processors apply the text attention_mask to the image-token scores as well "
                               "(interpolation.py:141-143, 352-359) — the reference fails at that broadcast; pass no mask")
        n = self._frames(x)
        wq, wk, wv, wo, bo = _weights(attn)
        coef, vals = self._coef_state(x.device, x.dtype, x.shape[0])
        fused = self.is_fused if mode != "plain" else False
        branch = branch_of(ip, n, coef) if ip is not None else None
        y = ops.processor_fwd(x, None if text is None else text.contiguous(), wq, wk, wv, wo, bo, attn.heads,
                              mode=mode, fused=fused, coef=coef, begin=0, end=n - 1, ip=branch,
                              seg_executed=ops.executed_segments(mode, fused, vals, n, None, 0, n - 1))
        return _epilogue(attn, y, residual, shape4)


def _zeros_map(cache, n: int, device) -> torch.Tensor:
    key = ("zeros", n, device)
    hit = cache.get(key)
    if hit is None:
        hit = cache[key] = torch.zeros(n, dtype=torch.int32, device=device)
    return hit


class OuterInterpolatedIPAttnProcessor(_IPBase):
    r"""Outer interpolation combined with the IP-Adapter image attention (interpolation.py:214-387):
    O = (1-c) [A_text_begin + s A_ip_begin] + c [A_text_end + s A_ip_end]."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        if not self.activated:
            return self._fallback(attn, hidden_states, encoder_hidden_states, attention_mask, temb)

        def branch(ip, n, coef):                                            # interpolation.py:329-367 (linear in O)
            tokens, _ = self._image_rows(ip, n, "per_frame")
            return dict(tokens=tokens, wk=self.ip_attn.to_k_ip[0].weight, wv=self.ip_attn.to_v_ip[0].weight,
                        mode="same", scale=float(self.scale[0]), begin=0, end=n - 1)
        return self._call(attn, hidden_states, encoder_hidden_states, attention_mask, temb, "outer", branch)


class InnerInterpolatedIPAttnProcessor(_IPBase):
    r"""Inner interpolation combined with the IP-Adapter image attention (interpolation.py:390-545).
    As in the reference the image branch attends with each frame's OWN image keys and is only
    shape-valid with ``is_fused=True`` (interpolation.py:512-527)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        if not self.activated:
            return self._fallback(attn, hidden_states, encoder_hidden_states, attention_mask, temb)

        def branch(ip, n, coef):                                            # interpolation.py:502-505, 525-530
            if not self.is_fused:
                raise RuntimeError("InnerInterpolatedIPAttnProcessor needs is_fused=True when image embeddings are "
                                   "passed: the reference's image branch multiplies un-split keys "
                                   "(batch1 dim mismatch in bmm, interpolation.py:525)")
            tokens, _ = self._image_rows(ip, n, "per_frame")
            return dict(tokens=tokens, wk=self.ip_attn.to_k_ip[0].weight, wv=self.ip_attn.to_v_ip[0].weight,
                        mode="plain", scale=float(self.scale[0]))
        return self._call(attn, hidden_states, encoder_hidden_states, attention_mask, temb, "inner", branch)


class ScaleControlIPAttnProcessor(_IPBase):
    r"""Image-prompt scale control (interpolation.py:51-211): text attention is outer-interpolated
    (activated) or plain (de-activated); the image attention of the END frame's rows ([6:9]) is added with the
    per-frame coefficient."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        def branch(ip, n, coef):                                            # interpolation.py:137-150 / 187-196
            tokens, row_map = self._image_rows(ip, n, "last")
            return dict(tokens=tokens, wk=self.ip_attn.to_k_ip[0].weight, wv=self.ip_attn.to_v_ip[0].weight,
                        mode="plain", scale=1.0, frame_scale=coef, map=row_map)
        return self._call(attn, hidden_states, encoder_hidden_states, attention_mask, temb,
                          "outer" if self.activated else "plain", branch)


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
    clear_weight_caches()                    # a (re-)installation is the boundary after which weights may have been edited
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


def load_aid_ip_adapter(unet, t: Optional[float] = 0.5, is_fused: bool = True, early: str = "fused_outer",
                        size: int = 7, alpha: float = 1, beta: float = 1, keep_original: bool = False) -> None:
    """Wrap every attention layer of a UNet whose IP-Adapter is ALREADY loaded (diffusers' ``load_ip_adapter`` —
    third-party, the reference calls it first, pipeline_interpolated_sd.py:986-992) with the IP variant ``early``
    selects (:993-1007): ``fused_outer`` / ``fused_inner`` / ``scale_control``.  The wrapped processor becomes
    ``ip_attn`` like in the reference; by default its HIP equivalent takes its place (sharing the adapter weights and
    the scale list) so the de-activated passes stay on the HIP kernels: IP-Adapter cross-attention layers get a
    :class:`HipIPAdapterAttnProcessor`, the other layers a :class:`HipAttnProcessor`.  ``keep_original=True`` keeps
    the installed processor itself."""
    classes = {"fused_outer": OuterInterpolatedIPAttnProcessor, "fused_inner": InnerInterpolatedIPAttnProcessor,
               "scale_control": ScaleControlIPAttnProcessor}
    if early not in classes:
        raise ValueError(f"early must be one of {sorted(classes)}, got {early!r}")
    clear_weight_caches()
    procs = {}
    current = unet.attn_processors
    for name, cur in current.items():
        if name.startswith("encoder"):
            procs[name] = cur
            continue
        if keep_original:
            ip_attn = cur
        elif hasattr(cur, "to_k_ip"):
            ip_attn = cur if isinstance(cur, HipIPAdapterAttnProcessor) else HipIPAdapterAttnProcessor.wrap(cur)
        else:
            ip_attn = HipAttnProcessor()
        procs[name] = classes[early](t=t, size=size, is_fused=is_fused, alpha=alpha, beta=beta, ip_attn=ip_attn)
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
