"""GPU: (1) the pipeline loops with their UNet passes captured into hipGraphs equal the eager loops bit for bit over a full
50-step run (reference loop pipeline_interpolated_sd.py:1834-1907); (2) the Beta-prior exploration with
the HIP renderer follows the same exploration with an oracle (fp64) renderer: coefficients, fitted (alpha, beta), picked
path (reference prior.py:119-199, 212-297)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import aid_amd  # noqa: E402
from aid_amd import prior as PR  # noqa: E402
from aid_amd.pipelines import (DDIMSchedulerLite, InterpolationStableDiffusionPipeline,  # noqa: E402
                               InterpolationStableDiffusionXLPipeline, StackDenoiser)
from test_hip_depth_and_pipelines import OracleDenoiser, _embs  # noqa: E402

DEV = "cuda:0"


def _setup(model, dtype, seed):
    hip = StackDenoiser(model, dtype=dtype, device=DEV, scale_down=16 if model == "sd15" else 64, latent_hw=(8, 8))
    g = torch.Generator().manual_seed(seed)
    l0, l1 = torch.randn(1, 4, 8, 8, generator=g).to(dtype), torch.randn(1, 4, 8, 8, generator=g).to(dtype)
    rd = lambda t: tuple(e.to(dtype).float() for e in t)     # noqa: E731
    xl = model == "sdxl"
    es, ee, eg = rd(_embs(g, hip.stack.cross_dim, xl)), rd(_embs(g, hip.stack.cross_dim, xl)), rd(_embs(g, hip.stack.cross_dim, xl))
    return hip, l0, l1, es, ee, eg


@pytest.mark.parametrize("model,dtype,atype,steps", [("sd15", torch.float16, "fused_inner", 50),
                                                      ("sdxl", torch.bfloat16, "fused_outer", 12)])
def test_interpolate_single_captured_equals_eager(model, dtype, atype, steps):
    hip, l0, l1, es, ee, eg = _setup(model, dtype, 21)
    cls = InterpolationStableDiffusionXLPipeline if model == "sdxl" else InterpolationStableDiffusionPipeline
    pipe = cls(hip, DDIMSchedulerLite())
    pipe.load_aid(t=0.5, is_fused=True, atype=atype)
    kw = dict(latent_start=l0, latent_end=l1, embeds_start=es, embeds_end=ee, embeds_guide=eg,
              num_inference_steps=steps, warmup_ratio=0.5, output_type="latent")
    seen = []
    cb = lambda p, i, t, d: seen.append(i) or {}             # noqa: E731   the step callback still fires every step
    captured = pipe.interpolate_single(0.35, use_graphs=True, callback_on_step_end=cb, **kw)["images"]
    eager = pipe.interpolate_single(0.35, use_graphs=False, **kw)["images"]
    assert seen == list(range(steps))
    assert torch.isfinite(captured).all() and torch.equal(captured, eager)
    # another coefficient on the same pipeline: activate_aid(it) rewrites the coefficient buffers in place, a NEW run captures
    # its own graphs — and differs from the first
    other = pipe.interpolate_single(0.8, use_graphs=True, **kw)["images"]
    assert torch.equal(other, pipe.interpolate_single(0.8, use_graphs=False, **kw)["images"])
    assert not torch.equal(other[1], captured[1])


@pytest.mark.parametrize("guided,batched", [(True, True), (False, True), (True, False)])
def test_n_frame_interpolate_captured_equals_eager_50_steps(guided, batched):
    hip, l0, l1, es, ee, eg = _setup("sd15", torch.float16, 22)
    pipe = InterpolationStableDiffusionPipeline(hip, DDIMSchedulerLite())
    pipe.load_aid(t=0.5, is_fused=True, atype="fused_inner")
    before = dict(hip.attn_processors)
    kw = dict(embeds_start=es, embeds_end=ee, embeds_guide=eg if guided else None, size=5, num_inference_steps=50,
              warmup_ratio=0.5, early="fused_outer", guidance_scale=4.0, output_type="latent", batched_cfg=batched)
    captured = pipe.interpolate(l0, l1, use_graphs=True, **kw)
    eager = pipe.interpolate(l0, l1, use_graphs=False, **kw)
    assert torch.isfinite(captured).all() and torch.equal(captured, eager)
    # the processors load_aid installed are back (interpolate() used to leave its own behind)
    after = dict(hip.attn_processors)
    assert after.keys() == before.keys() and all(after[k] is before[k] for k in before)
    pipe.interpolate_single(0.5, latent_start=l0, latent_end=l1, embeds_start=es, embeds_end=ee, num_inference_steps=2,
                            output_type="latent")            # ... and a batch-3 run works right after


class _Feat(torch.nn.Module):
    """Deterministic stand-in for the CLIP image tower: a fixed linear map of the frame (the CLIP weights are third-party
    and not available offline)."""

    def __init__(self, dtype, device):
        super().__init__()
        g = torch.Generator().manual_seed(4)
        self.w = torch.nn.Parameter(torch.randn(4 * 8 * 8, 24, generator=g, dtype=torch.float64).to(device=device, dtype=dtype),
                                    requires_grad=False)

    def get_image_features(self, pixel_values):
        return pixel_values.reshape(pixel_values.shape[0], -1).to(self.w.dtype) @ self.w


@pytest.mark.parametrize("batch", [1, 2])
def test_beta_prior_exploration_hip_renderer_vs_oracle_renderer(batch):
    dtype, steps = torch.float16, 6
    hip, l0, l1, es, ee, eg = _setup("sd15", dtype, 23)
    kw = dict(embeds_start=es, embeds_end=ee, output_type="latent", warmup_ratio=0.5)
    pipe = InterpolationStableDiffusionPipeline(hip, DDIMSchedulerLite())
    pipe.load_aid(t=0.5, is_fused=True, atype="fused_outer")
    bp = PR.BetaPriorPipeline(pipe, model=_Feat(torch.float32, DEV))
    got = bp.generate_interpolation(None, None, None, l0, l1, num_inference_steps=steps, exploration_size=7,
                                    interpolation_size=4, batch=batch, **kw)
    ora = InterpolationStableDiffusionPipeline(OracleDenoiser(hip), DDIMSchedulerLite())
    ora._aid_early = pipe._aid_early
    dbl = lambda t: tuple(e.double() for e in t)             # noqa: E731
    bo = PR.BetaPriorPipeline(ora, model=_Feat(torch.float64, "cpu"))
    ref = bo.generate_interpolation(None, None, None, l0.double(), l1.double(), num_inference_steps=steps, exploration_size=7,
                                    interpolation_size=4, batch=batch,
                                    **dict(kw, embeds_start=dbl(es), embeds_end=dbl(ee)))
    assert len(bp.xs) == len(bo.xs) == 7
    np.testing.assert_allclose(bp.xs, bo.xs, rtol=0, atol=1e-3)
    # 1 - cos of neighbouring frames: an absolute error of ~1e-4 is what fp16 rendering (2e-3 rel-L2 per run) leaves of it
    np.testing.assert_allclose([float(d) for d in bp.ds], [float(d) for d in bo.ds], rtol=2e-2, atol=5e-4)
    # the least-squares fit amplifies the distances' fp16 rendering noise (measured 1.6e-3 relative); coefficients above: 1e-3
    np.testing.assert_allclose([bp.alpha, bp.beta_param], [bo.alpha, bo.beta_param], rtol=5e-3)
    # identical picked path: the returned frames are the same explored frames
    picked = [next(i for i, f in enumerate(bp.images) if f is g_) for g_ in got]
    picked_ref = [next(i for i, f in enumerate(bo.images) if f is r_) for r_ in ref]
    assert picked == picked_ref and picked[0] == 0 and picked[-1] == 6
    for a, b in zip(got, ref):
        assert float((a.double().cpu() - b).norm() / b.norm()) < 1e-2
