"""Minimal stand-ins for the diffusers objects the AID processors talk to.

diffusers is not a dependency of this package (and is not installable in the
build image).  The processors are duck-typed against the
``attn`` argument, so they work with a real ``diffusers.models.attention_processor.Attention``
when diffusers is present and with :class:`AttnShim` otherwise.  The shim
restates the part of ``Attention`` the reference path touches (SURVEY.md
App. A; call sites interpolation.py:604-667) and is what the tests, the golden
generator and the attention-stack benchmark drive.

:class:`AttnStackUNet` is a UNet-shaped container exposing diffusers'
``attn_processors`` / ``set_attn_processor`` surface over the exact ordered
list of attention layers of one SD1.5 / SDXL UNet forward.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import nn


class AttnShim(nn.Module):
    """What one UNet ``Attention`` layer owns: to_q/to_k/to_v (no bias),
    to_out = [Linear(bias), Dropout(0)], heads, scale = dim_head**-0.5."""

    def __init__(self, query_dim: int, heads: int, cross_attention_dim: Optional[int] = None,
                 dtype: torch.dtype = torch.float32, device=None, processor=None):
        super().__init__()
        assert query_dim % heads == 0
        ctx = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.inner_dim = query_dim
        self.cross_attention_dim = ctx
        self.scale = (query_dim // heads) ** -0.5
        kw = dict(dtype=dtype, device=device)
        self.to_q = nn.Linear(query_dim, query_dim, bias=False, **kw)
        self.to_k = nn.Linear(ctx, query_dim, bias=False, **kw)
        self.to_v = nn.Linear(ctx, query_dim, bias=False, **kw)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim, bias=True, **kw), nn.Dropout(0.0)])
        # flags of SD / SDXL UNet attention
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.upcast_attention = False
        self.upcast_softmax = False
        self.processor = processor

    # -- the helpers the reference processors call ---------------------------
    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        """diffusers' ``Attention.prepare_attention_mask`` (third-party, restated from diffusers 0.27 - 0.31; not in SURVEY.md App. A,
        which only needed the ``None`` case): an additive mask ``[B, 1 | S, L]`` is padded when its length differs from the key
        sequence's — by ``target_length`` zeros, diffusers' own quirk — and repeated per head to ``[B * H, 1 | S, L]``."""
        if attention_mask is None:
            return None
        if attention_mask.shape[-1] != target_length:
            attention_mask = torch.nn.functional.pad(attention_mask, (0, target_length), value=0.0)
        if out_dim == 3:
            if attention_mask.shape[0] < batch_size * self.heads:
                attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
        elif out_dim == 4:
            attention_mask = attention_mask.unsqueeze(1).repeat_interleave(self.heads, dim=1)
        return attention_mask

    def head_to_batch_dim(self, tensor: torch.Tensor, out_dim: int = 3) -> torch.Tensor:
        h = self.heads
        if tensor.ndim == 3:
            b, l, dim = tensor.shape
            e = 1
        else:
            b, e, l, dim = tensor.shape
        tensor = tensor.reshape(b, l * e, h, dim // h).permute(0, 2, 1, 3)
        if out_dim == 3:
            tensor = tensor.reshape(b * h, l * e, dim // h)
        return tensor

    def batch_to_head_dim(self, tensor: torch.Tensor) -> torch.Tensor:
        h = self.heads
        bh, l, d = tensor.shape
        return tensor.reshape(bh // h, h, l, d).permute(0, 2, 1, 3).reshape(bh // h, l, d * h)

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if self.upcast_attention:
            query, key = query.float(), key.float()
        if attention_mask is None:
            base = torch.empty(query.shape[0], query.shape[1], key.shape[1],
                               dtype=query.dtype, device=query.device)
            beta = 0
        else:
            base, beta = attention_mask, 1
        scores = torch.baddbmm(base, query, key.transpose(-1, -2), beta=beta, alpha=self.scale)
        if self.upcast_softmax:
            scores = scores.float()
        return scores.softmax(dim=-1).to(dtype)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class IPAdapterShim(nn.Module):
    """The state the IP processors share with diffusers' IPAdapterAttnProcessor2_0
    (interpolation.py:70-74): to_k_ip / to_v_ip ModuleLists, scale list, num_tokens.  A weight holder only
    (``weights_only``): a de-activated IP processor that wraps it runs ``HipIPAdapterAttnProcessor`` on these weights."""
    weights_only = True

    def __init__(self, hidden_size: int, cross_attention_dim: int, num_tokens: int = 4,
                 scale: float = 1.0, dtype=torch.float32, device=None):
        super().__init__()
        kw = dict(dtype=dtype, device=device)
        self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False, **kw)])
        self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False, **kw)])
        self.scale = [scale]
        self.num_tokens = (num_tokens,)


# ---------------------------------------------------------------------------
# attention call list of one UNet forward
# ---------------------------------------------------------------------------
#   (location, transformer blocks, S, C, heads)
SD15_LAYERS: List[Tuple[str, int, int, int, int]] = [
    ("down_blocks.0", 2, 4096, 320, 8),
    ("down_blocks.1", 2, 1024, 640, 8),
    ("down_blocks.2", 2, 256, 1280, 8),
    ("mid_block", 1, 64, 1280, 8),
    ("up_blocks.1", 3, 256, 1280, 8),
    ("up_blocks.2", 3, 1024, 640, 8),
    ("up_blocks.3", 3, 4096, 320, 8),
]
SDXL_LAYERS: List[Tuple[str, int, int, int, int]] = [
    ("down_blocks.1", 4, 4096, 640, 10),
    ("down_blocks.2", 20, 1024, 1280, 20),
    ("mid_block", 10, 1024, 1280, 20),
    ("up_blocks.0", 30, 1024, 1280, 20),
    ("up_blocks.1", 6, 4096, 640, 10),
]
MODEL_SPECS = {
    "sd15": dict(layers=SD15_LAYERS, cross_dim=768, text_len=77, latent=(4, 64, 64)),
    "sdxl": dict(layers=SDXL_LAYERS, cross_dim=2048, text_len=77, latent=(4, 128, 128)),
}


class AttnStackUNet(nn.Module):
    """Ordered attention layers of one UNet forward with the diffusers processor
    surface (``attn_processors`` / ``set_attn_processor``).  ``forward`` replays
    the attention calls only: per transformer block ``x += attn1(x)`` (self) and
    ``x += attn2(x, ctx)`` (text cross-attention); everything between the
    attention calls of the real UNet is third-party and out of scope.
    """

    def __init__(self, model: str = "sd15", dtype=torch.float16, device=None, seed: int = 1002,
                 scale_down: int = 1, channel_div: int = 1, head_div: int = 1):
        """``scale_down`` divides every S, ``channel_div`` every width (structure-only uses in tests), ``head_div`` the number of
        heads AND the width together — the head dims of the real layers (40 / 80 / 160, 64) are kept, so the shipped attention
        kernels run; used by the 50-step end-to-end parity tests, whose fp64 oracle loop is bound by the weight bytes."""
        super().__init__()
        spec = MODEL_SPECS[model]
        self.model = model
        self.cross_dim = spec["cross_dim"] // channel_div
        self.text_len = spec["text_len"]
        names, mods, shapes = [], [], []
        for loc, nblk, s, c, h in spec["layers"]:
            s = max(s // scale_down, 1)
            c = c // channel_div
            if head_div > 1:
                assert h % head_div == 0, (model, h, head_div)
                c, h = c // head_div, h // head_div
            for b in range(nblk):
                for which, cd in (("attn1", None), ("attn2", self.cross_dim)):
                    names.append(f"{loc}.attentions.{b}.transformer_blocks.0.{which}.processor")
                    mods.append(AttnShim(c, h, cd, dtype=dtype, device=device))
                    shapes.append((s, c, h, cd is not None))
        self.names = names
        self.layers = nn.ModuleList(mods)
        self.shapes = shapes
        # the LayerNorm in front of every attention layer (BasicTransformerBlock.norm1 / norm2), used by the
        # ``sublayers`` modes of forward()
        self.norms = nn.ModuleList([nn.LayerNorm(c, eps=1e-5).to(device=device, dtype=dtype) for (_, c, _, _) in shapes])
        self.sublayers = "off"
        # weights ~ N(0, 1/fan_in), bias ~ N(0, .01); generated on the target device
        gdev = self.layers[0].to_q.weight.device
        g = torch.Generator(device=gdev).manual_seed(seed)
        with torch.no_grad():
            for m in self.layers:
                for lin in (m.to_q, m.to_k, m.to_v, m.to_out[0]):
                    w = torch.randn(lin.weight.shape, generator=g, device=gdev) / lin.weight.shape[1] ** 0.5
                    lin.weight.copy_(w.to(lin.weight.dtype))
                m.to_out[0].bias.copy_((0.01 * torch.randn(m.to_out[0].bias.shape, generator=g, device=gdev))
                                       .to(m.to_out[0].bias.dtype))
            for nrm in self.norms:
                nrm.weight.copy_((1.0 + 0.1 * torch.randn(nrm.weight.shape, generator=g, device=gdev)).to(nrm.weight.dtype))
                nrm.bias.copy_((0.05 * torch.randn(nrm.bias.shape, generator=g, device=gdev)).to(nrm.bias.dtype))

    def load_ip_adapter(self, num_tokens: int = 4, scale: float = 1.0, seed: int = 7):
        """Stand-in for diffusers' ``load_ip_adapter`` on this stack (third-party; the reference calls it first,
        pipeline_interpolated_sd.py:986-992): every text cross-attention layer (attn2) gets an IP-Adapter processor with
        synthetic ``to_k_ip`` / ``to_v_ip`` weights ~ N(0, 1/fan_in); self-attention layers keep theirs.  The cross layers
        are then fed ``encoder_hidden_states = (text, [image_embeds])``."""
        from .processors import HipAttnProcessor, HipIPAdapterAttnProcessor
        dev, dt = self.layers[0].to_q.weight.device, self.layers[0].to_q.weight.dtype
        g = torch.Generator(device=dev).manual_seed(seed)
        with torch.no_grad():
            for m, (s, c, h, is_cross) in zip(self.layers, self.shapes):
                if not is_cross:
                    if m.processor is None:
                        m.processor = HipAttnProcessor()
                    continue
                p = HipIPAdapterAttnProcessor(c, self.cross_dim, num_tokens=(num_tokens,), scale=scale, dtype=dt, device=dev)
                for lin in (p.to_k_ip[0], p.to_v_ip[0]):
                    lin.weight.copy_((torch.randn(lin.weight.shape, generator=g, device=dev) / lin.weight.shape[1] ** 0.5).to(dt))
                m.processor = p

    @property
    def attn_processors(self) -> Dict[str, object]:
        return {n: m.processor for n, m in zip(self.names, self.layers)}

    def set_attn_processor(self, processor: Union[object, Dict[str, object]]):
        if isinstance(processor, dict):
            if len(processor) != len(self.names):
                raise ValueError(f"A dict of processors was passed, but the number of processors "
                                 f"{len(processor)} does not match the number of attention layers: {len(self.names)}.")
            for n, m in zip(self.names, self.layers):
                m.processor = processor[n]
        else:
            for m in self.layers:
                m.processor = processor

    def forward(self, xs: Dict[Tuple[int, int], torch.Tensor], encoder_hidden_states: torch.Tensor):
        """``xs`` maps (S, C) -> the (post-LayerNorm-scale) hidden state [N, S, C] every
        attention call at that resolution level is fed with.  Returns the last
        attention output per level.  Calls are issued in UNet order; they are not
        chained through the residual stream because the norms / convs / MLPs that
        sit between them in the real UNet are not part of this path."""
        outs: Dict[Tuple[int, int], torch.Tensor] = {}
        if self.sublayers == "off":
            for m, (s, c, h, is_cross) in zip(self.layers, self.shapes):
                outs[(s, c)] = m(xs[(s, c)], encoder_hidden_states if is_cross else None)
            return outs
        # ``sublayers``: the residual stream of every resolution level runs through  h = h + attn(norm(h))  (what the
        # transformer block does around attn1 / attn2) — "steps": torch LayerNorm, processor call,
        # torch add; "fused": one library call per layer (processors' fused_sublayer)
        hs = dict(xs)
        for m, nrm, (s, c, h, is_cross) in zip(self.layers, self.norms, self.shapes):
            ctx = encoder_hidden_states if is_cross else None
            if self.sublayers == "fused":
                hs[(s, c)] = m.processor.fused_sublayer(m, nrm, hs[(s, c)], ctx)
            else:
                hs[(s, c)] = hs[(s, c)] + m(nrm(hs[(s, c)]), ctx)
        return hs

    def level_shapes(self) -> List[Tuple[int, int]]:
        seen: List[Tuple[int, int]] = []
        for s, c, _, _ in self.shapes:
            if (s, c) not in seen:
                seen.append((s, c))
        return seen
