"""Beta-prior exploration of the interpolation path (reference ``prior.py:35-340``).

The reference's ``BetaPriorPipeline`` grows a set of interpolation coefficients ``xs`` one at a time: it takes the widest
perceptual gap (CLIP cosine distance between neighbouring frames), places the next coefficient at the Beta-CDF midpoint
of that gap, renders it with a batch-3 ``interpolate_single`` run, and refits Beta(alpha, beta) to the cumulative
distances.  Here the same search runs over a *generator callback* — ``generate(ts) -> (frames, features)`` renders the
frames at the coefficients ``ts`` (one N-frame AID run over ``[0, *ts, 1]`` on the HIP path: the processors take any
coefficient vector) — so it does not depend on diffusers / CLIP being importable, and with ``batch > 1`` several gaps are
filled per run instead of one batch-3 run per point.  With ``batch = 1`` the sequence of coefficients, distances and
fitted parameters is the reference's (pinned against the reference in ``tests/test_prior.py``).

Host-side logic only (numpy / scipy); nothing here touches the GPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
from scipy.optimize import curve_fit
from scipy.stats import beta as beta_distribution


def clip_distance(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """1 - cosine similarity of two [1, D] feature rows (prior.py:18-22)."""
    return 1 - torch.nn.functional.cosine_similarity(a, b)[0]


def update_alpha_beta(xs: Sequence[float], ds: Sequence[float]) -> Tuple[float, float]:
    """Fit Beta(alpha, beta) so that its CDF at ``xs`` follows the normalised cumulative distances (prior.py:35-56)."""
    total = sum(ds)
    uniform_points = np.cumsum([0] + [d / total for d in ds])
    xs = np.asarray(xs)
    uniform_points = np.asarray(uniform_points)

    def beta_cdf(x, alpha, beta_param):
        return beta_distribution.cdf(x, alpha, beta_param)

    params, _ = curve_fit(beta_cdf, xs, uniform_points, p0=[1.0, 1.0], bounds=([1e-6, 1e-6], [np.inf, np.inf]))
    return params[0], params[1]


def next_point(xs: Sequence[float], ds: Sequence[float], alpha: float, beta_param: float,
               uniform: bool = False, rank: int = 0) -> Tuple[int, float]:
    """(segment index, coefficient) of the next point (prior.py:74-90).  ``rank`` > 0 selects the rank-th widest gap
    (batched exploration); rank 0 is the reference's choice."""
    order = np.argsort(-np.asarray([float(d) for d in ds]), kind="stable")
    idx = int(np.argmax(ds)) if rank == 0 else int(order[rank])
    f_a = beta_distribution.cdf(xs[idx], alpha, beta_param)
    f_b = beta_distribution.cdf(xs[idx + 1], alpha, beta_param)
    t = beta_distribution.ppf((f_a + f_b) / 2, alpha, beta_param)
    if uniform:                                                   # the reference's uniform variant (:86-88), quirk kept
        if rank == 0:
            idx = int(np.argmax(np.array(xs) - np.array([0] + list(xs[:-1])))) - 1
        t = (xs[idx] + xs[idx + 1]) / 2
    return idx, float(t)


def extract_uniform_points(ds: Sequence[float], interpolation_size: int) -> List[int]:
    """Greedy pick of frames at equal cumulative distance (prior.py:201-210)."""
    expected = sum(ds) / (interpolation_size - 1)
    current, out = 0, [0]
    for idx, d in enumerate(ds):
        current += d
        if current >= expected:
            out.append(idx)
            current = 0
    return out


def is_path_possible(D, n, m, weights, W):
    """prior.py:256-297: is there a path 0 -> m-1 over n nodes whose edge weights lie in a window of width D?"""
    for w_min in W:
        w_max = w_min + D
        if w_max > W[-1]:
            break
        dp = [[None] * (n + 1) for _ in range(m)]
        dp[0][1] = (float("-inf"), float("inf"), [0])
        for l in range(1, n):
            for i in range(m):
                if dp[i][l] is None:
                    continue
                max_w, min_w, path = dp[i][l]
                for j in range(i + 1, m):
                    w = weights[i][j]
                    if w != -1 and w_min <= w <= w_max:
                        new_max, new_min = max(max_w, w), min(min_w, w)
                        if new_max - new_min <= D:
                            cur = dp[j][l + 1]
                            if cur is None or new_max - new_min < cur[0] - cur[1]:
                                dp[j][l + 1] = (new_max, new_min, path + [j])
        if dp[m - 1][n] is not None:
            return dp[m - 1][n][2]
    return None


def find_minimal_spread_and_path(n: int, m: int, weights) -> Tuple[Optional[float], Optional[List[int]]]:
    """Bisection on the spread (max - min edge weight) of an n-node path through the m explored frames (prior.py:223-254)."""
    W = sorted({weights[i][j] for i in range(m - 1) for j in range(i + 1, m) if weights[i][j] != -1})
    low, high = 0.0, W[-1] - W[0]
    best_d, best_path = None, None
    while high - low > 1e-6:
        D = (low + high) / 2
        result = is_path_possible(D, n, m, weights, W)
        if result is not None:
            high, best_d, best_path = D, D, result
        else:
            low = D
    return best_d, best_path


def extract_uniform_points_plus(features: Sequence[torch.Tensor], interpolation_size: int,
                                distance: Callable = clip_distance) -> Optional[List[int]]:
    """Smoothest path of ``interpolation_size`` frames through the explored ones (prior.py:212-221)."""
    m = len(features)
    weights = -1 * np.ones((m, m))
    for i in range(m):
        for j in range(i + 1, m):
            weights[i][j] = distance(features[i], features[j])
    return find_minimal_spread_and_path(interpolation_size, m, weights)[1]


class BetaPriorExplorer:
    """``explore_with_beta`` / ``generate_interpolation`` (prior.py:119-340) over a generator callback.

    ``generate(ts)`` renders the frames at coefficients ``ts`` (strictly inside (0, 1), ascending) between the two fixed
    end points and returns ``(frames, features)`` — one entry per element of ``[0, *ts, 1]``, features as [1, D]
    tensors.  The reference calls its pipeline once per new point (batch 3); ``batch`` > 1 fills that many gaps per call.
    """

    def __init__(self, generate: Callable[[List[float]], Tuple[list, List[torch.Tensor]]],
                 distance: Callable = clip_distance):
        self.generate = generate
        self.distance = distance

    def explore(self, exploration_size: int = 16, init_alpha: float = 3, init_beta: float = 3, uniform: bool = False,
                batch: int = 1):
        frames, features = self.generate([0.5])
        frames, features = list(frames), list(features)
        xs = [0.0, 0.5, 1.0]
        ds = [self.distance(features[0], features[1]), self.distance(features[1], features[2])]
        alpha, beta_param = init_alpha, init_beta
        while len(xs) < exploration_size:
            picks = []
            for rank in range(min(batch, len(ds), exploration_size - len(xs))):
                idx, t = next_point(xs, ds, alpha, beta_param, uniform=uniform, rank=rank)
                if t < 0 or t > 1:
                    break
                picks.append((idx, t))
            if not picks:
                break
            picks.sort(key=lambda p: p[1])
            new_frames, new_feats = self.generate([t for _, t in picks])
            for k, (idx, t) in sorted(enumerate(picks), key=lambda e: -e[1][0]):     # insert from the back: indices stay valid
                f = new_feats[1 + k]
                d1, d2 = self.distance(features[idx], f), self.distance(features[idx + 1], f)
                frames.insert(idx + 1, new_frames[1 + k])
                features.insert(idx + 1, f)
                xs.insert(idx + 1, t)
                del ds[idx]
                ds.insert(idx, d1)
                ds.insert(idx + 1, d2)
            alpha, beta_param = update_alpha_beta(xs, ds)
            if uniform:
                alpha, beta_param = 1, 1
        return frames, features, ds, xs, alpha, beta_param

    def generate_interpolation(self, interpolation_size: int = 7, **explore_kwargs):
        frames, features, ds, xs, alpha, beta_param = self.explore(**explore_kwargs)
        idxs = extract_uniform_points_plus(features, interpolation_size, self.distance)
        self.frames, self.ds, self.xs, self.alpha, self.beta_param = frames, ds, xs, alpha, beta_param   # prior.py:332-338
        return [frames[i] for i in idxs]


class BetaPriorPipeline:
    """The reference's ``BetaPriorPipeline`` (prior.py:12-340) over this package's pipeline classes: same constructor
    shape (``pipe``, ``model_ID``) and methods (``_compute_clip``, ``_get_feature``, ``explore_with_beta``,
    ``generate_interpolation``; attributes ``images / ds / xs / alpha / beta_param`` after a run, :332-338).

    Feature extractor: ``model`` = anything with ``get_image_features(pixel_values) -> [B, D]`` (transformers'
    ``CLIPModel`` when its weights are available — third-party; ``model_ID`` is tried through ``from_pretrained`` only if
    no model is given) plus an optional ``preprocess``; features and the cosine distances stay on the device the frames
    are on (the reference moves every frame through CPU PIL preprocessing).
    Rendering: ``pipe.interpolate_single(t, ...)`` per new point like the reference, or — ``batch`` > 1 — one N-frame
    ``pipe.interpolate`` run over ``[0, *ts, 1]`` filling several gaps at once (the processors take any coefficient vector).
    """

    def __init__(self, pipe, model_ID: str = "openai/clip-vit-base-patch32", model=None, preprocess=None):
        if model is None:
            from transformers import CLIPImageProcessor, CLIPModel      # needs the checkpoint on disk (no network here)
            model = CLIPModel.from_pretrained(model_ID)
            preprocess = preprocess or CLIPImageProcessor.from_pretrained(model_ID)
        self.model, self.preprocess, self.pipe = model, preprocess, pipe

    def _compute_clip(self, embedding_a, embedding_b):
        return clip_distance(embedding_a, embedding_b)

    @torch.no_grad()
    def _get_feature(self, image):
        if self.preprocess is not None:
            image = self.preprocess(image)
        if not torch.is_tensor(image):
            image = torch.as_tensor(np.asarray(image))
        if image.ndim == 3:
            image = image[None]
        return self.model.get_image_features(image)

    def _render(self, ts, prompt_start, prompt_end, negative_prompt, latent_start, latent_end, num_inference_steps, kw):
        """Frames at ``[0, *ts, 1]``: batch-3 ``interpolate_single`` for one point (the reference's call, prior.py:94-104),
        an N-frame ``interpolate`` with the coefficient vector for several."""
        if len(ts) == 1:
            out = self.pipe.interpolate_single(ts[0], prompt_start=prompt_start, prompt_end=prompt_end,
                                               negative_prompt=negative_prompt, latent_start=latent_start,
                                               latent_end=latent_end, num_inference_steps=num_inference_steps, **kw)
            frames = out["images"] if isinstance(out, dict) else out.images
        else:
            frames = self.pipe.interpolate(latent_start, latent_end, prompt_start, prompt_end,
                                           negative_prompt=negative_prompt, size=len(ts) + 2,
                                           num_inference_steps=num_inference_steps, coef=[0.0] + list(ts) + [1.0], **kw)
        frames = [frames[i] for i in range(len(ts) + 2)]
        return frames, [self._get_feature(f) for f in frames]

    def explore_with_beta(self, prompt_start, prompt_end, negative_prompt, latent_start, latent_end,
                          num_inference_steps=28, exploration_size=16, init_alpha=3, init_beta=3, uniform=False,
                          batch: int = 1, **kwargs):
        kwargs.pop("early", None)                      # accepted and ignored like the reference
        ex = BetaPriorExplorer(lambda ts: self._render(ts, prompt_start, prompt_end, negative_prompt, latent_start,
                                                       latent_end, num_inference_steps, kwargs), self._compute_clip)
        return ex.explore(exploration_size, init_alpha, init_beta, uniform=uniform, batch=batch)

    def generate_interpolation(self, prompt_start, prompt_end, negative_prompt, latent_start, latent_end,
                               num_inference_steps=28, exploration_size=16, init_alpha=3, init_beta=3,
                               interpolation_size=7, uniform=False, **kwargs):
        images, features, ds, xs, alpha, beta_param = self.explore_with_beta(
            prompt_start, prompt_end, negative_prompt, latent_start, latent_end, num_inference_steps, exploration_size,
            init_alpha, init_beta, uniform=uniform, **kwargs)
        idxs = extract_uniform_points_plus(features, interpolation_size, self._compute_clip)
        self.images, self.ds, self.xs, self.alpha, self.beta_param = images, ds, xs, alpha, beta_param
        return [images[i] for i in idxs]
