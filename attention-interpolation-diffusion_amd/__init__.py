"""attention-interpolation-diffusion_amd — MI355X-native (gfx950 / CDNA4) interpolated attention
for diffusion (AID / PAID): a drop-in for the attention-processor hot path of
QY-H00/attention-interpolation-diffusion.

The directory name contains hyphens; import the package through the ``aid_amd`` alias module at
the repository root (``import aid_amd``).

Layout:  csrc/ (HIP kernels + C ABI, built to libaid_hip.so)  ·  _lib.py (ctypes binding)  ·
ops.py (tensor-level entry points)  ·  processors.py (the reference's AttnProcessor classes)  ·
interp.py (coefficients, slerp / lerp initialisation)  ·  attn_shim.py (diffusers stand-ins)  ·
dist.py (frame sharding over RCCL)  ·  loop.py (denoising-loop harness)  ·  sequence.py (batch assembly of
interpolate_single / N-frame interpolate)  ·  pipelines.py (the reference's pipeline classes over components)  ·  prior.py (Beta-prior exploration of the coefficient path).
"""
from .interp import generate_beta_tensor, linear_interpolation, slerp, spherical_interpolation
from .processors import (HipAttnProcessor, HipIPAdapterAttnProcessor, InnerInterpolatedAttnProcessor,
                         InnerInterpolatedIPAttnProcessor, InterpolatedAttnProcessor, OuterInterpolatedAttnProcessor,
                         OuterInterpolatedIPAttnProcessor, ScaleControlIPAttnProcessor, activate_aid,
                         clear_text_kv_cache, clear_weight_caches, deactivate_aid, load_aid, load_aid_ip_adapter)
from .attn_shim import AttnShim, AttnStackUNet, IPAdapterShim
from . import ops, _lib, sequence, loop, dist, prior, pipelines
from .prior import BetaPriorExplorer, BetaPriorPipeline
from .pipelines import (DDIMSchedulerLite, InterpolationStableDiffusionPipeline,
                        InterpolationStableDiffusionXLPipeline, StackDenoiser)

__all__ = [
    "generate_beta_tensor", "linear_interpolation", "slerp", "spherical_interpolation",
    "InterpolatedAttnProcessor", "OuterInterpolatedAttnProcessor", "InnerInterpolatedAttnProcessor",
    "OuterInterpolatedIPAttnProcessor", "InnerInterpolatedIPAttnProcessor", "ScaleControlIPAttnProcessor",
    "HipAttnProcessor", "HipIPAdapterAttnProcessor", "load_aid", "load_aid_ip_adapter", "activate_aid", "deactivate_aid",
    "clear_text_kv_cache", "clear_weight_caches",
    "AttnShim", "AttnStackUNet", "IPAdapterShim", "ops",
    "BetaPriorPipeline", "BetaPriorExplorer", "InterpolationStableDiffusionPipeline", "InterpolationStableDiffusionXLPipeline", "DDIMSchedulerLite", "StackDenoiser",
]
__version__ = "0.1.0"
