"""GPU: re-activating ONE processor object with many different t (the reference's usage pattern:
pipeline_interpolated_sd.py:1845-1848 calls activate_aid(it) every step, prior.py:94 / gradio app.py:233-268 call
interpolate_single with a new t per run) must use [0, t, 1] every time (interpolation.py:37-42, 662-664) — eagerly
and from replayed hipGraphs."""
import numpy as np
import pytest
import torch

from oracle import aid_oracle as O
from util import TOL, rel_l2, to_np64

pytestmark = pytest.mark.gpu

import aid_amd  # noqa: E402
from aid_amd.loop import AidDenoiseLoop, install_sequence_processors  # noqa: E402

DEV = "cuda:0"


def _weights(attn, heads):
    return O.AttnWeights(*(to_np64(t) for t in (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
                                                 attn.to_out[0].weight, attn.to_out[0].bias)), heads)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["float16", "bfloat16"])
@pytest.mark.parametrize("kind", ["outer", "inner"])
def test_one_processor_reactivated_with_60_different_t(kind, dtype):
    heads, d, s = 2, 40, 48
    c = heads * d
    g = torch.Generator().manual_seed(11)
    attn = aid_amd.AttnShim(c, heads, dtype=dtype, device=DEV)
    x = torch.randn(3, s, c, generator=g).to(dtype)
    xd, xn, w = x.to(DEV), to_np64(x), _weights(attn, heads)
    cls = aid_amd.OuterInterpolatedAttnProcessor if kind == "outer" else aid_amd.InnerInterpolatedAttnProcessor
    fn = O.outer_attention if kind == "outer" else O.inner_attention
    proc = cls(t=0.5, is_fused=True)
    rs = np.random.RandomState(5)
    ts = [0.5] + [float(t) for t in rs.uniform(0.02, 0.98, 60)]
    worst = 0.0
    for i, t in enumerate(ts):
        if i:
            proc.activate(t)
        if i == 30:                              # once by plain assignment instead (prior.py-style callers)
            proc.coef = torch.tensor([0.0, t, 1.0])
        y = proc(attn, xd)
        coef = torch.tensor([0.0, t, 1.0]).to(dtype).float().numpy()
        err = rel_l2(to_np64(y), fn(xn, None, w, coef, True))
        worst = max(worst, err)
        assert err < TOL[dtype], (i, t, err)
    # a different t really gives a different middle frame (the test would pass trivially otherwise)
    proc.activate(0.1); a = proc(attn, xd)
    proc.activate(0.9); b = proc(attn, xd)
    assert rel_l2(to_np64(a[1]), to_np64(b[1])) > 20 * TOL[dtype]
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])           # end points do not depend on t


def test_activate_aid_over_the_whole_stack_for_consecutive_t():
    """activate_aid(unet, t) on the 32-layer SD1.5 stack (reduced S and width), two consecutive t, every level output
    against the oracle evaluated with that t."""
    dtype = torch.float16
    unet = aid_amd.AttnStackUNet("sd15", dtype=dtype, device=DEV, scale_down=64)
    aid_amd.load_aid(unet, t=0.5, is_fused=True, atype="fused_outer")
    g = torch.Generator().manual_seed(3)
    xs = {(s, c): torch.randn(3, s, c, generator=g).to(dtype) for (s, c) in unet.level_shapes()}
    ctx = torch.randn(3, unet.text_len, unet.cross_dim, generator=g).to(dtype)
    xs_d = {k: v.to(DEV) for k, v in xs.items()}
    for t in (0.3, 0.7, 0.31):
        aid_amd.activate_aid(unet, t)
        outs = unet(xs_d, ctx.to(DEV))
        coef = torch.tensor([0.0, t, 1.0]).to(dtype).float().numpy()
        last = {}
        for m, (s, c, h, is_cross) in zip(unet.layers, unet.shapes):
            last[(s, c)] = (m, h, is_cross)
        for key, (m, h, is_cross) in last.items():
            ref = O.outer_attention(to_np64(xs[key]), to_np64(ctx) if is_cross else None, _weights(m, h), coef, True)
            assert rel_l2(to_np64(outs[key]), ref) < TOL[dtype], (t, key)


def test_replayed_graph_sees_a_new_schedule():
    """AidDenoiseLoop captures the passes into hipGraphs; assigning a new schedule afterwards rewrites the device
    buffers the graphs read (in place), so the next replay uses it."""
    dtype = torch.float16
    n = 4
    unet = aid_amd.AttnStackUNet("sd15", dtype=dtype, device=DEV, scale_down=64)
    install_sequence_processors(unet, n, early="fused_inner", num_inference_steps=10)
    g = torch.Generator().manual_seed(4)
    xs = {(s, c): torch.randn(n, s, c, generator=g).to(dtype).to(DEV) for (s, c) in unet.level_shapes()}
    cond = torch.randn(n, unet.text_len, unet.cross_dim, generator=g).to(dtype).to(DEV)
    unc = torch.randn(n, unet.text_len, unet.cross_dim, generator=g).to(dtype).to(DEV)

    def run(use_graphs, coef):
        for p in unet.attn_processors.values():
            p.coef = coef.clone()
        loop = AidDenoiseLoop(unet, xs, cond, unc, num_inference_steps=10, use_graphs=use_graphs, batched_cfg=True)
        return loop

    c1 = torch.tensor([0.0, 0.2, 0.6, 1.0])
    c2 = torch.tensor([0.0, 0.45, 0.55, 1.0])
    loop = run(True, c1)
    out1 = {k: v.clone() for k, v in loop.step(0).items()}
    for p in unet.attn_processors.values():                       # new schedule, graphs already captured
        p.coef = c2.clone()
    out2 = {k: v.clone() for k, v in loop.step(0).items()}
    eager = run(False, c2)
    ref2 = eager.step(0)
    eager1 = run(False, c1)
    ref1 = eager1.step(0)
    for k in out1:
        assert torch.equal(out1[k], ref1[k]) and torch.equal(out2[k], ref2[k])
        assert not torch.equal(out1[k], out2[k])
