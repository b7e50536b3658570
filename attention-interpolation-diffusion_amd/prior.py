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
    """Beta(alpha, beta) whose CDF, sampled at the explored coefficients ``xs``, follows the share of the total perceptual
    distance covered up to each of them (prior.py:35-56): least squares from (1, 1), both parameters kept positive."""
    total = sum(ds)
    # accumulated in the precision of the distances themselves (fp32 CLIP distances in the reference's runs): the fitted
    # parameters move at the 1e-7 level with the rounding of these targets, and the goldens pin them
    share = np.asarray(np.cumsum([0] + [d / total for d in ds]))
    fit, _ = curve_fit(beta_distribution.cdf, np.asarray(xs), share, p0=[1.0, 1.0],
                       bounds=([1e-6, 1e-6], [np.inf, np.inf]))
    return fit[0], fit[1]


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
    """Frames at (roughly) equal arc length: walk the gaps and emit the gap index every time the distance accumulated since
    the last emitted frame reaches total / (size - 1) (prior.py:201-210; the reference emits the index of the GAP, quirk kept)."""
    quota = sum(ds) / (interpolation_size - 1)
    picks, walked = [0], 0
    for gap, length in enumerate(ds):
        walked += length
        if walked >= quota:
            picks.append(gap)
            walked = 0
    return picks


# ---- smoothest path through the explored frames (prior.py:212-297) ----------------------------------------------------------
# Task: among the increasing node sequences 0 = p_0 < ... < p_{n-1} = m - 1 pick one whose edge weights w[p_k][p_{k+1}] fit
# into the narrowest window.  The reference bisects the window width D to 1e-6 and, per trial D, slides a window
# [w_min, w_min + D] over the sorted distinct weights — ONLY windows that end inside the weight range (it stops at the first
# w_min + D > W[-1], prior.py:258-259) — running a greedy label DP per window.  That predicate is NOT "D >= the exact
# threshold": a window that needs the largest weight is feasible only when some distinct weight sits exactly D below W[-1] or
# lower, ties and small graphs make it non-monotone, and the reference returns None for about a quarter of small random graphs
# (tests/golden/prior_goldens.npz `path*` pins 232 randomised cases, 60 of them None).  Formulation here:
#   1. the reference's scalar bisection, trial by trial, on its real predicate — "some window [w, w + D], w a distinct weight,
#      w + D <= W[-1], carries an n-node path" — where a window's test is a boolean reachability product (inside a window of
#      width D every path's spread is <= D, so the reference's label DP succeeds exactly when a path exists);
#   2. one label pass in the first feasible window of the last feasible trial, level-synchronous and vectorised over the
#      predecessors, with parent pointers instead of stored paths.  Labels and tie-breaks are the reference's (keep, per
#      (node, length), the partial path of smallest spread; among equals the smallest predecessor index), so the picked path
#      is identical.
def _edges_in(weights: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """Upper-triangular mask of the existing edges whose weight lies in [lo, hi]."""
    m = weights.shape[0]
    return np.triu(np.ones((m, m), dtype=bool), 1) & (weights != -1) & (weights >= lo) & (weights <= hi)


def _reachable(mask: np.ndarray, n: int) -> bool:
    """Is node m - 1 reachable from node 0 in exactly n - 1 steps over the edges of ``mask``?"""
    front = np.zeros(mask.shape[0], dtype=bool)
    front[0] = True
    for _ in range(n - 1):
        front = (front.astype(np.int64) @ mask.astype(np.int64)) > 0
        if not front.any():
            return False
    return bool(front[-1])


def _first_window(D: float, n: int, weights: np.ndarray, W: np.ndarray) -> Optional[float]:
    """Lower end of the first window the reference's scan accepts for trial width D (prior.py:256-295), or None."""
    for w_lo in W:
        if w_lo + D > W[-1]:
            break                                    # the reference never tries a window that ends beyond the largest weight
        if _reachable(_edges_in(weights, w_lo, w_lo + D), n):
            return float(w_lo)
    return None


def _label_pass(n: int, weights: np.ndarray, mask: np.ndarray, width: float) -> Optional[List[int]]:
    """One sweep of (max, min) labels over path lengths 1 .. n inside one window; returns the path to node m - 1."""
    m = weights.shape[0]
    top = np.full((n + 1, m), np.nan)            # label of the kept partial path ending in node j with l nodes
    low = np.full((n + 1, m), np.nan)
    parent = np.full((n + 1, m), -1, dtype=np.int64)
    alive = np.zeros((n + 1, m), dtype=bool)
    top[1, 0], low[1, 0], alive[1, 0] = -np.inf, np.inf, True
    for l in range(1, n):
        src = np.flatnonzero(alive[l])
        if src.size == 0:
            return None
        for j in range(1, m):
            cand = src[(src < j) & mask[src, j]]
            if cand.size == 0:
                continue
            w = weights[cand, j]
            hi_, lo_ = np.maximum(top[l, cand], w), np.minimum(low[l, cand], w)
            spread = hi_ - lo_
            ok = spread <= width
            if not ok.any():
                continue
            k = int(np.argmin(np.where(ok, spread, np.inf)))          # first minimum = smallest predecessor index
            top[l + 1, j], low[l + 1, j], parent[l + 1, j], alive[l + 1, j] = hi_[k], lo_[k], cand[k], True
    if not alive[n, m - 1]:
        return None
    path, node = [m - 1], m - 1
    for l in range(n, 1, -1):
        node = int(parent[l, node])
        path.append(node)
    return path[::-1]


def find_minimal_spread_and_path(n: int, m: int, weights) -> Tuple[Optional[float], Optional[List[int]]]:
    """(window width, node path) of the smoothest n-node path 0 -> m - 1 through the explored frames — what the reference's
    bisection + window scan returns (prior.py:223-297), including ``(None, None)`` where the reference finds nothing."""
    weights = np.asarray(weights, dtype=np.float64)
    iu = np.triu_indices(m, 1)
    present = weights[iu]
    W = np.unique(present[present != -1])
    low, high = 0.0, float(W[-1] - W[0])
    width, start = None, None
    while high - low > 1e-6:
        D = (low + high) / 2
        w_lo = _first_window(D, n, weights, W)
        if w_lo is not None:
            high, width, start = D, D, w_lo
        else:
            low = D
    if width is None:
        return None, None
    return width, _label_pass(n, weights, _edges_in(weights, start, start + width), width)


def extract_uniform_points_plus(features: Sequence[torch.Tensor], interpolation_size: int,
                                distance: Callable = clip_distance) -> Optional[List[int]]:
    """Smoothest path of ``interpolation_size`` frames through the explored ones (prior.py:212-221)."""
    m = len(features)
    weights = np.full((m, m), -1.0)
    for a in range(m - 1):
        for b in range(a + 1, m):
            weights[a, b] = float(distance(features[a], features[b]))       # float() also brings a device scalar to the host
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
        self._distance = distance

    def distance(self, a, b):
        """Distance of two feature rows as a HOST scalar: features (and the cosine) stay on the frames' device, the search
        decisions (widest gap, curve fit) are host arithmetic on 0-dim fp32 values like the reference's."""
        d = self._distance(a, b)
        return d.detach().cpu() if torch.is_tensor(d) and d.device.type != "cpu" else d

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
        """CLIP image feature of one frame, [1, D] (prior.py:24-33).  With a ``preprocess`` component the frame goes through
        it the way the reference calls transformers' ``CLIPImageProcessor``: ``preprocess(image, return_tensors="pt"
        [, do_rescale=False for numpy frames]).pixel_values`` — float frames in [0, 1] must not be rescaled again; uint8
        frames (what this package's pipelines return for ``output_type="np"``) keep the processor's 1/255 rescale.  The
        pixel tensor is then moved to the feature model's device and dtype."""
        if self.preprocess is not None:
            kw = {"return_tensors": "pt"}
            if isinstance(image, np.ndarray) and image.dtype.kind == "f":
                kw["do_rescale"] = False
            out = self.preprocess(image, **kw)
            image = getattr(out, "pixel_values", None)
            if image is None:
                image = out["pixel_values"] if isinstance(out, dict) else out
        if not torch.is_tensor(image):
            image = torch.as_tensor(np.asarray(image))
        if image.ndim == 3:
            image = image[None]
        ref = next(iter(self.model.parameters()), None) if hasattr(self.model, "parameters") else None
        if ref is not None:
            image = image.to(ref.device, ref.dtype) if image.is_floating_point() else image.to(ref.device)
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
            # the N-frame run renders with the mode load_aid / load_aid_ip_adapter selected (the batch-3 call above does too)
            early = getattr(self.pipe, "_aid_early", None)
            if early not in ("pure_inner", "fused_inner", "pure_outer", "fused_outer"):
                early = "fused_outer"
            frames = self.pipe.interpolate(latent_start, latent_end, prompt_start, prompt_end,
                                           negative_prompt=negative_prompt, size=len(ts) + 2, early=early,
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
