"""GPU, only where diffusers is importable (it is not in the build image): drive a REAL
``diffusers.models.attention_processor.Attention`` module with the HIP processors installed through its own
``set_processor`` and compare with (a) the oracle and (b) diffusers' AttnProcessor2_0 for the de-activated path.
The goldens were generated with this repository's AttnShim standing in for that class."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

diffusers = pytest.importorskip("diffusers")

from oracle import aid_oracle as O  # noqa: E402
from util import TOL, rel_l2, to_np64  # noqa: E402
import aid_amd  # noqa: E402


@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("kind", ["outer", "inner"])
def test_real_diffusers_attention_module(kind, cross):
    from diffusers.models.attention_processor import Attention, AttnProcessor2_0
    dtype, n, s, c, heads, cc, l = torch.float16, 3, 48, 80, 2, 64, 77
    attn = Attention(query_dim=c, cross_attention_dim=cc if cross else None, heads=heads, dim_head=c // heads,
                     bias=False, out_bias=True).to("cuda:0", dtype)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, s, c, generator=g).to(dtype).cuda()
    ctx = torch.randn(n, l, cc, generator=g).to(dtype).cuda() if cross else None
    cls = aid_amd.OuterInterpolatedAttnProcessor if kind == "outer" else aid_amd.InnerInterpolatedAttnProcessor
    proc = cls(t=0.3, is_fused=True)
    attn.set_processor(proc)
    y = attn(x, encoder_hidden_states=ctx)                 # Attention.forward -> processor protocol
    w = O.AttnWeights(*(to_np64(t) for t in (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
                                              attn.to_out[0].weight, attn.to_out[0].bias)), heads)
    coef = torch.tensor([0.0, 0.3, 1.0]).to(dtype).float().numpy()
    fn = O.outer_attention if kind == "outer" else O.inner_attention
    assert rel_l2(to_np64(y), fn(to_np64(x), None if ctx is None else to_np64(ctx), w, coef, True)) < TOL[dtype]
    proc.deactivate()                                      # plain attention on the HIP kernel ...
    y_plain = attn(x, encoder_hidden_states=ctx)
    attn.set_processor(AttnProcessor2_0())                 # ... against diffusers' own processor
    y_ref = attn(x, encoder_hidden_states=ctx)
    assert rel_l2(to_np64(y_plain), to_np64(y_ref)) < 2 * TOL[dtype]
