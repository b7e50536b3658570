#!/usr/bin/env python3
"""Golden fixtures for the Beta-prior exploration, produced by the REFERENCE's own ``BetaPriorPipeline`` methods
(/root/reference/prior.py, imported — never copied — with the stubs of make_goldens.py).  The renderer and the CLIP
feature extractor, which need diffusers / downloaded weights, are replaced by the deterministic stand-ins of cases.py
(`prior_feature`); everything the fixture pins — which coefficient is explored next, the distances, the fitted
(alpha, beta), the picked paths — is the reference's arithmetic.

Output: prior_goldens.npz          usage:  python tests/golden/make_prior_goldens.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases as C  # noqa: E402
from make_goldens import import_reference  # noqa: E402


def main():
    _, prior = import_reference()
    P = prior.BetaPriorPipeline
    out = {}

    class FakePipe:
        def interpolate_single(self, t, **kw):
            return types.SimpleNamespace(images=[0.0, float(t), 1.0])     # an "image" is just its coefficient

    for r, run in enumerate(C.PRIOR_RUNS):
        self = P.__new__(P)
        self.pipe = FakePipe()
        self._get_feature = lambda image: torch.from_numpy(C.prior_feature(float(image)))[None]
        images, features, ds, xs, alpha, beta = self.explore_with_beta(
            "a", "b", "", None, None, num_inference_steps=1, exploration_size=run["exploration_size"],
            init_alpha=run["init_alpha"], init_beta=run["init_beta"], uniform=run["uniform"])
        out[f"run{r}_xs"] = np.array(xs, dtype=np.float64)
        out[f"run{r}_ds"] = np.array([float(d) for d in ds], dtype=np.float64)
        out[f"run{r}_ab"] = np.array([alpha, beta], dtype=np.float64)
        out[f"run{r}_path5"] = np.array(self.extract_uniform_points_plus(features, 5))
        out[f"run{r}_uniform5"] = np.array(self.extract_uniform_points(ds, 5))
    self = P.__new__(P)
    for i, (xs, ds) in enumerate(C.PRIOR_FITS):
        out[f"fit{i}"] = np.array(self._update_alpha_beta(list(xs), list(ds)), dtype=np.float64)
    for i, (ds, n) in enumerate(C.PRIOR_UNIFORM):
        out[f"uniform{i}"] = np.array(self.extract_uniform_points(list(ds), n))
    # randomised smoothest-path searches by the reference's own find_minimal_spread_and_path (width NaN / empty path = None)
    import contextlib, io
    for i, (kind, m, n, seed) in enumerate(C.prior_path_cases()):
        w = C.prior_path_weights(kind, m, seed)
        with contextlib.redirect_stdout(io.StringIO()):
            d, path = self.find_minimal_spread_and_path(n, m, [list(map(float, row)) for row in w])
        out[f"path{i}_d"] = np.array(np.nan if d is None else d, dtype=np.float64)
        out[f"path{i}_p"] = np.array([] if path is None else path, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "prior_goldens.npz"), **out)
    for k, v in out.items():
        if not k.startswith("path"):
            print(k, v)
    print(sum(1 for k in out if k.endswith("_p")), "path cases,", sum(1 for k, v in out.items() if k.endswith("_p") and v.size == 0), "without a path")


if __name__ == "__main__":
    main()
