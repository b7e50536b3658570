"""A test double of the part of diffusers the drop-in boundary talks to — restated from the published
behaviour of diffusers 0.27 – 0.31, NOT copied: diffusers is absent from the build image and the GPU box, so the processors have
otherwise only ever met this repository's own AttnShim (VERDICT r4 missing #2).  What is restated is the PROTOCOL, which is what a
processor can get wrong:

  * ``Attention.forward`` dispatch: ``cross_attention_kwargs`` are filtered by ``inspect.signature(self.processor.__call__)`` — a
    keyword the processor does not name is dropped with a warning, never passed — and the call is
    ``self.processor(self, hidden_states, encoder_hidden_states=..., attention_mask=..., **filtered)``;
  * ``Attention.set_processor`` / ``get_processor``; a processor that is an ``nn.Module`` is registered as a sub-module (so the IP
    processors' ``to_k_ip`` weights follow ``unet.to(...)``, interpolation.py:10, 70-74);
  * ``UNet2DConditionModel.attn_processors``: a dict keyed ``"<module path>.processor"`` built by walking the module tree and
    collecting every module that has ``get_processor``; ``set_attn_processor(dict | processor)`` with diffusers' count check and the
    same walk (keys look like ``down_blocks.1.attentions.0.transformer_blocks.0.attn1.processor``);
  * ``BasicTransformerBlock.forward``: ``x = attn1(norm1(x), **kw) + x`` (self: no encoder states), ``x = attn2(norm2(x),
    encoder_hidden_states=ctx, **kw) + x``; ``Transformer2DModel`` hands the blocks 3-D ``[N, S, C]`` tensors;
  * with an IP-Adapter the UNet passes ``encoder_hidden_states = (text, [image_embeds])`` to every attention layer's processor
    (attn1 gets ``None``).

The arithmetic helpers of Attention that the HIP processors never call (head_to_batch_dim, get_attention_scores ...) are left out on
purpose: a processor that reached for them would fail here, as it should."""
from __future__ import annotations

import inspect
import warnings
from typing import Dict, Optional, Union

import torch
from torch import nn


class Attention(nn.Module):
    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64,
                 bias: bool = False, out_bias: bool = True, processor=None, dtype=None, device=None):
        super().__init__()
        inner = heads * dim_head
        kw = dict(dtype=dtype, device=device)
        self.heads, self.inner_dim, self.query_dim = heads, inner, query_dim
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=bias, **kw)
        self.to_k = nn.Linear(self.cross_attention_dim, inner, bias=bias, **kw)
        self.to_v = nn.Linear(self.cross_attention_dim, inner, bias=bias, **kw)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias, **kw), nn.Dropout(0.0)])
        self.spatial_norm = self.group_norm = self.norm_cross = None
        self.residual_connection, self.rescale_output_factor = False, 1.0
        self.upcast_attention = self.upcast_softmax = False
        self.set_processor(processor)

    def set_processor(self, processor) -> None:
        # diffusers: a module processor replaces the registered sub-module, a plain object is a plain attribute
        if hasattr(self, "processor") and isinstance(self.processor, nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def get_processor(self):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        params = set(inspect.signature(self.processor.__call__).parameters.keys())
        quiet = {"ip_adapter_masks", "ip_hidden_states"}
        unused = [k for k in cross_attention_kwargs if k not in params and k not in quiet]
        if unused:
            warnings.warn(f"cross_attention_kwargs {unused} are not expected by {self.processor.__class__.__name__} and will be ignored.")
        kwargs = {k: v for k, v in cross_attention_kwargs.items() if k in params}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask,
                              **kwargs)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, cross_attention_dim: int, dtype=None, device=None):
        super().__init__()
        kw = dict(dtype=dtype, device=device)
        self.norm1 = nn.LayerNorm(dim, **kw)
        self.attn1 = Attention(dim, None, heads, dim_head, **kw)
        self.norm2 = nn.LayerNorm(dim, **kw)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head, **kw)

    def forward(self, hidden_states, encoder_hidden_states=None, cross_attention_kwargs=None):
        kw = dict(cross_attention_kwargs or {})
        hidden_states = self.attn1(self.norm1(hidden_states), encoder_hidden_states=None, **kw) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states, **kw) + hidden_states
        return hidden_states


class _Transformer2D(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim, depth, **kw):
        super().__init__()
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, dim_head, cross_dim, **kw) for _ in range(depth)])

    def forward(self, x, ctx, cak):
        for blk in self.transformer_blocks:
            x = blk(x, ctx, cak)
        return x


class _Block(nn.Module):
    def __init__(self, n_attn, *a, **kw):
        super().__init__()
        self.attentions = nn.ModuleList([_Transformer2D(*a, **kw) for _ in range(n_attn)])


class UNetDouble(nn.Module):
    """The attention skeleton of a small SD-shaped UNet with diffusers' module naming: two down blocks, a mid block, two up blocks;
    each resolution level keeps its own token stream [N, S, C] (the convolutions between the levels are not part of the boundary)."""

    LEVELS = (("down_blocks.0", 2, 64, 80, 2, 40), ("down_blocks.1", 2, 16, 160, 2, 80), ("mid_block", 1, 16, 160, 2, 80),
              ("up_blocks.0", 3, 16, 160, 2, 80), ("up_blocks.1", 3, 64, 80, 2, 40))       # (name, attentions, S, C, heads, dim_head)

    def __init__(self, cross_dim: int = 96, depth: int = 1, dtype=None, device=None):
        super().__init__()
        kw = dict(dtype=dtype, device=device)
        self.cross_dim = cross_dim
        self.down_blocks = nn.ModuleList([_Block(n, c, h, dh, cross_dim, depth, **kw) for (nm, n, s, c, h, dh) in self.LEVELS if nm.startswith("down")])
        nm, n, s, c, h, dh = self.LEVELS[2]
        self.mid_block = _Block(n, c, h, dh, cross_dim, depth, **kw)
        self.up_blocks = nn.ModuleList([_Block(n, c, h, dh, cross_dim, depth, **kw) for (nm, n, s, c, h, dh) in self.LEVELS if nm.startswith("up")])
        self.encoder_hid_proj = None

    # ---- diffusers' processor surface -----------------------------------------------------------------------------------------
    @property
    def attn_processors(self) -> Dict[str, object]:
        processors: Dict[str, object] = {}

        def walk(name: str, module: nn.Module):
            if hasattr(module, "get_processor"):
                processors[f"{name}.processor"] = module.get_processor()
            for sub, child in module.named_children():
                walk(f"{name}.{sub}", child)
        for name, module in self.named_children():
            walk(name, module)
        return processors

    def set_attn_processor(self, processor: Union[object, Dict[str, object]]) -> None:
        count = len(self.attn_processors.keys())
        if isinstance(processor, dict) and len(processor) != count:
            raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not match the"
                             f" number of attention layers: {count}. Please make sure to pass {count} processor classes.")

        def walk(name: str, module: nn.Module):
            if hasattr(module, "set_processor"):
                module.set_processor(processor if not isinstance(processor, dict) else processor.pop(f"{name}.processor"))
            for sub, child in module.named_children():
                walk(f"{name}.{sub}", child)
        for name, module in self.named_children():
            walk(name, module)

    # ---- forward: every attention layer in UNet order --------------------------------------------------------------------------
    def forward(self, streams: Dict[str, torch.Tensor], encoder_hidden_states, cross_attention_kwargs=None):
        out = {}
        blocks = list(self.down_blocks) + [self.mid_block] + list(self.up_blocks)
        for (nm, n, s, c, h, dh), blk in zip(self.LEVELS, blocks):
            x = streams[nm]
            for tr in blk.attentions:
                x = tr(x, encoder_hidden_states, cross_attention_kwargs)
            out[nm] = x
        return out

    def streams(self, n: int, generator=None, dtype=None, device=None) -> Dict[str, torch.Tensor]:
        return {nm: torch.randn(n, s, c, generator=generator).to(dtype=dtype, device=device) for (nm, _, s, c, _, _) in self.LEVELS}


class IPAdapterAttnProcessor2_0(nn.Module):
    """State of diffusers' IPAdapterAttnProcessor2_0 (what ``load_ip_adapter`` installs on every attn2): to_k_ip / to_v_ip
    ModuleLists, the scale list, num_tokens.  Its own __call__ (plain text attention + scale x image attention via
    F.scaled_dot_product_attention) is restated in fp32 torch — the de-activated reference path when ``keep_original=True``."""

    def __init__(self, hidden_size: int, cross_attention_dim: int, num_tokens=(4,), scale=1.0, dtype=None, device=None):
        super().__init__()
        self.num_tokens = tuple(num_tokens)
        self.scale = [scale] * len(self.num_tokens)
        kw = dict(dtype=dtype, device=device)
        self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False, **kw) for _ in self.num_tokens])
        self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False, **kw) for _ in self.num_tokens])

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale: float = 1.0,
                 ip_adapter_masks=None):
        text, ip = encoder_hidden_states if isinstance(encoder_hidden_states, tuple) else (encoder_hidden_states, None)
        f = lambda t: t.float()           # noqa: E731
        b, s, _ = hidden_states.shape
        h = attn.heads

        def sdpa(q, k, v):
            d = q.shape[-1] // h
            sp = lambda t: t.view(b, -1, h, d).transpose(1, 2)      # noqa: E731
            o = torch.nn.functional.scaled_dot_product_attention(sp(q), sp(k), sp(v))
            return o.transpose(1, 2).reshape(b, -1, h * d)
        q = f(hidden_states) @ f(attn.to_q.weight).T
        e = f(hidden_states if text is None else text)
        o = sdpa(q, e @ f(attn.to_k.weight).T, e @ f(attn.to_v.weight).T)
        if ip is not None:
            for cur, sc, wk, wv in zip(ip, self.scale, self.to_k_ip, self.to_v_ip):
                cur = f(cur)
                o = o + sc * sdpa(q, (cur @ f(wk.weight).T).reshape(b, -1, q.shape[-1]), (cur @ f(wv.weight).T).reshape(b, -1, q.shape[-1]))
        o = o @ f(attn.to_out[0].weight).T + f(attn.to_out[0].bias)
        return o.to(hidden_states.dtype)
