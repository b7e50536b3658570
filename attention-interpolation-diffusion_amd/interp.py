"""Host-side preparation of an interpolation sequence: Beta-scheduled coefficients and the
latent / embedding initialisation.  Same names, arguments and behaviour as the reference
(prior.py:481-502, interpolation.py:807-918).  These run once per sequence on small tensors
and are not kernel targets."""
from __future__ import annotations

from typing import Optional

import torch
from torch import FloatTensor


def generate_beta_tensor(size: int, alpha: float = 3, beta: float = 3) -> torch.FloatTensor:
    """[x_0 .. x_{n-1}] with F(x_i) = i/(n-1) for the Beta(alpha, beta) CDF F (prior.py:481-502)."""
    from scipy.stats import beta as beta_distribution
    prob_values = [i / (size - 1) for i in range(size)]
    return torch.tensor(beta_distribution.ppf(prob_values, alpha, beta), dtype=torch.float32)


def linear_interpolation(l1: FloatTensor, l2: FloatTensor, ts: Optional[FloatTensor] = None,
                         size: int = 5) -> FloatTensor:
    """(size, *) linear interpolation between (1, *) tensors (interpolation.py:807-835)."""
    assert l1.shape == l2.shape, "shapes of l1 and l2 must match"
    if ts is None:
        ts = [i / (size - 1) for i in range(size)]
    return torch.cat([torch.lerp(l1, l2, t) for t in ts], dim=0)


def spherical_interpolation(l1: FloatTensor, l2: FloatTensor, size=5) -> FloatTensor:
    """(size, *) spherical interpolation between (1, *) tensors (interpolation.py:838-858)."""
    assert l1.shape == l2.shape, "shapes of l1 and l2 must match"
    return torch.cat([slerp(l1, l2, i / (size - 1)) for i in range(size)], dim=0)


def slerp(v0: FloatTensor, v1: FloatTensor, t, threshold=0.9995):
    """Row-wise (last dim) spherical linear interpolation with a lerp fallback for rows that are
    (anti-)colinear (|cos| > threshold) or contain a zero vector (NaN cosine)
    (interpolation.py:861-918)."""
    assert v0.shape == v1.shape, "shapes of v0 and v1 must match"
    n0 = torch.norm(v0, dim=-1, keepdim=True)
    n1 = torch.norm(v1, dim=-1, keepdim=True)
    dot = ((v0 / n0) * (v1 / n1)).sum(-1)
    mag = dot.abs()
    gotta_lerp = mag.isnan() | (mag > threshold)
    lerped = torch.lerp(v0, v1, t)
    theta_0 = dot.arccos().unsqueeze(-1)
    sin_theta_0 = theta_0.sin()
    theta_t = theta_0 * t
    s0 = (theta_0 - theta_t).sin() / sin_theta_0
    s1 = theta_t.sin() / sin_theta_0
    slerped = s0 * v0 + s1 * v1
    return torch.where(gotta_lerp.unsqueeze(-1), lerped, slerped)
