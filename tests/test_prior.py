"""Beta-prior exploration against fixtures produced by the reference's own BetaPriorPipeline methods
(tests/golden/make_prior_goldens.py; renderer and CLIP features replaced by the deterministic stand-ins of cases.py)."""
import os

import numpy as np
import pytest
import torch

import cases as C
import aid_amd
from aid_amd import prior as P

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prior_goldens.npz"))


def _generator(calls):
    def generate(ts):
        calls.append(list(ts))
        pts = [0.0] + [float(t) for t in ts] + [1.0]
        return pts, [torch.from_numpy(C.prior_feature(t))[None] for t in pts]
    return generate


@pytest.mark.parametrize("r", range(len(C.PRIOR_RUNS)))
def test_exploration_follows_the_reference(r):
    calls = []
    ex = P.BetaPriorExplorer(_generator(calls))
    frames, features, ds, xs, alpha, beta = ex.explore(**C.PRIOR_RUNS[r])
    np.testing.assert_allclose(xs, G[f"run{r}_xs"], rtol=0, atol=1e-7)
    np.testing.assert_allclose([float(d) for d in ds], G[f"run{r}_ds"], rtol=0, atol=1e-6)
    np.testing.assert_allclose([alpha, beta], G[f"run{r}_ab"], rtol=1e-6)
    assert frames == list(xs)                                        # the stand-in "image" of a frame is its coefficient
    assert len(calls) == len(xs) - 2 and all(len(c) == 1 for c in calls)     # one batch-3 run per point, like the reference
    assert P.extract_uniform_points_plus(features, 5) == G[f"run{r}_path5"].tolist()
    assert P.extract_uniform_points(ds, 5) == G[f"run{r}_uniform5"].tolist()


def test_fit_and_uniform_pick_known_answers():
    for i, (xs, ds) in enumerate(C.PRIOR_FITS):
        np.testing.assert_allclose(P.update_alpha_beta(xs, ds), G[f"fit{i}"], rtol=1e-6)
    for i, (ds, n) in enumerate(C.PRIOR_UNIFORM):
        assert P.extract_uniform_points(ds, n) == G[f"uniform{i}"].tolist()


def test_batched_exploration_fills_several_gaps_per_run():
    calls = []
    ex = P.BetaPriorExplorer(_generator(calls))
    frames, features, ds, xs, alpha, beta = ex.explore(exploration_size=12, batch=3)
    assert len(xs) == 12 and xs == sorted(xs) and xs[0] == 0.0 and xs[-1] == 1.0 and len(set(xs)) == 12
    assert [len(c) for c in calls] == [1, 2, 3, 3, 1] and all(c == sorted(c) for c in calls)   # 9 single runs in the reference
    assert len(ds) == 11 and len(frames) == 12 and frames == xs
    np.testing.assert_allclose([float(d) for d in ds],
                               [float(P.clip_distance(features[i], features[i + 1])) for i in range(11)], atol=1e-7)
    out = ex.generate_interpolation(interpolation_size=5, exploration_size=10, batch=2)
    assert len(out) == 5 and out[0] == 0.0 and out[-1] == 1.0 and out == sorted(out)


def test_beta_prior_pipeline_facade_over_the_pipeline_classes():
    """BetaPriorPipeline (prior.py:12-340 surface) over InterpolationStableDiffusionPipeline with a recording stub UNet and a
    stand-in feature extractor: batch = 1 renders one interpolate_single per point like the reference, batch = 3 fills
    several gaps per N-frame run; both leave images / ds / xs / alpha / beta_param behind (:332-338)."""
    import torch
    import aid_amd
    from test_pipelines import RecordingUNet
    from aid_amd.pipelines import DDIMSchedulerLite, InterpolationStableDiffusionPipeline

    class Feat:
        def get_image_features(self, x):
            v = x.float().reshape(x.shape[0], -1)
            return torch.cat([v[:, :8], v[:, :8].sin()], dim=1)

    g = torch.Generator().manual_seed(3)
    l0, l1 = torch.randn(1, 4, 4, 4, generator=g), torch.randn(1, 4, 4, 4, generator=g)
    mk = lambda: (torch.randn(1, 7, 12, generator=g), torch.randn(1, 7, 12, generator=g))     # noqa: E731
    es, ee = mk(), mk()
    runs = {}
    for batch in (1, 3):
        unet = RecordingUNet()
        pipe = InterpolationStableDiffusionPipeline(unet, DDIMSchedulerLite())
        pipe.load_aid(t=0.5, is_fused=True, atype="fused_outer")
        bp = aid_amd.BetaPriorPipeline(pipe, model=Feat())
        frames = bp.generate_interpolation(None, None, None, l0, l1, num_inference_steps=4, exploration_size=8,
                                           interpolation_size=5, batch=batch, embeds_start=es, embeds_end=ee,
                                           output_type="latent", early="fused_outer")
        assert len(frames) == 5 and len(bp.xs) == 8 and bp.xs[0] == 0.0 and bp.xs[-1] == 1.0
        assert bp.xs == sorted(bp.xs) and len(bp.ds) == 7 and bp.alpha > 0 and bp.beta_param > 0
        runs[batch] = [c["n"] for c in unet.calls]
    assert set(runs[1]) == {3}                          # batch-3 interpolate_single runs only
    assert max(runs[3]) > 3                             # N-frame runs ([cond ; uncond] batched) fill several gaps
    assert len(runs[3]) < len(runs[1])


def test_get_feature_goes_through_the_image_processor_like_the_reference():
    """with transformers' CLIPImageProcessor as ``preprocess`` the frame must be passed with
    return_tensors="pt" and the ``pixel_values`` of the returned BatchFeature must reach the model — on the model's
    device and dtype (reference prior.py:24-33)."""
    seen = {}

    class BatchFeature(dict):                       # what CLIPImageProcessor returns: a dict with attribute access
        def __getattr__(self, k):
            return self[k]

    def processor(image, return_tensors=None, do_rescale=True):
        seen["kw"] = (return_tensors, do_rescale)
        x = torch.as_tensor(np.asarray(image)).float()
        if do_rescale:
            x = x / 255.0
        return BatchFeature(pixel_values=x.permute(2, 0, 1)[None])

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3, dtype=torch.float64))

        def get_image_features(self, pixel_values):
            seen["dtype"], seen["shape"] = pixel_values.dtype, tuple(pixel_values.shape)
            return pixel_values.mean(dim=(2, 3)) * self.w

    bp = P.BetaPriorPipeline(pipe=None, model=Model(), preprocess=processor)
    frame_u8 = (np.arange(4 * 5 * 3) % 255).astype(np.uint8).reshape(4, 5, 3)
    f = bp._get_feature(frame_u8)
    assert seen["kw"] == ("pt", True) and seen["dtype"] == torch.float64 and seen["shape"] == (1, 3, 4, 5)
    assert tuple(f.shape) == (1, 3) and float(f.max()) <= 1.0
    bp._get_feature(frame_u8.astype(np.float32) / 255.0)            # float frames in [0, 1] are not rescaled again
    assert seen["kw"] == ("pt", False)
    np.testing.assert_allclose(bp._get_feature(frame_u8.astype(np.float32) / 255.0).detach().numpy(), f.detach().numpy(), rtol=1e-6)


def test_smoothest_path_search_equals_the_reference_on_random_graphs():
    """232 randomised weight matrices (metric-like, i.i.d., tied / quantised, with missing edges; m = 3 .. 14, n = 2 .. m)
    through the reference's own find_minimal_spread_and_path (make_prior_goldens.py): same width bit for bit, same path, and
    ``(None, None)`` exactly where the reference returns it (its window scan skips every window that ends past the largest weight)."""
    cases = C.prior_path_cases()
    assert len(cases) == 232
    none = 0
    for i, (kind, m, n, seed) in enumerate(cases):
        d, path = P.find_minimal_spread_and_path(n, m, C.prior_path_weights(kind, m, seed))
        gd, gp = float(G[f"path{i}_d"]), G[f"path{i}_p"].tolist()
        if not gp:
            assert d is None and path is None and np.isnan(gd), (kind, m, n, seed)
            none += 1
        else:
            assert path == gp and d == gd, (kind, m, n, seed, path, gp)
            assert path[0] == 0 and path[-1] == m - 1 and len(path) == n and path == sorted(set(path))
    assert none == 60
