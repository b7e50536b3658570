"""Pipeline surface of the reference over the HIP attention path.

``InterpolationStableDiffusionPipeline`` / ``InterpolationStableDiffusionXLPipeline`` keep the method names and keyword
arguments of the reference's pipelines —

    load_aid(t, is_fused, atype)                        pipeline_interpolated_sd.py:950-970
    load_aid_ip_adapter(..., t, is_fused, early)        :972-1007
    activate_aid(it) / deactivate_aid()                 :1008-1017
    interpolate_single(it, prompt_start, prompt_end, latent_start, latent_end, image_start, image_end,
                       guide_prompt, warmup_ratio, is_fused, atype, init, ...)           :1407-1963, sdxl :1693-2411
    interpolate(latent_start, latent_end, prompt_start, prompt_end, guide_prompt, negative_prompt, size,
                num_inference_steps, warmup_ratio, early, late, alpha, beta, guidance_scale)
                                                        gradio_src/pipeline_interpolated_stable_diffusion.py:163-304

— and run the same host logic: batch ``[start, interior ..., end]``, slerp'd latents, lerp / slerp / guide-prompt
embeddings, the conditional pass with AID on for ``i < int(T * warmup_ratio)`` and the unconditional pass plain,
``uncond + gs * (text - uncond)``, scheduler step.  Everything the reference inherits from diffusers (UNet, scheduler,
VAE, text / image encoders — "Copied from diffusers" in the reference, out of scope per SURVEY.md §2) is a COMPONENT
handed to the constructor, duck-typed on the diffusers interfaces:

    unet(sample, t, encoder_hidden_states=..., added_cond_kwargs=..., return_dict=False)[0]
        + ``attn_processors`` / ``set_attn_processor``          scheduler.set_timesteps / timesteps / scale_model_input /
    step(noise, t, latents, return_dict=False)[0] / init_noise_sigma          vae.decode(latents, return_dict=False)[0]
    encode_prompt(prompt, negative_prompt) -> (cond, uncond)   [SDXL: (cond, uncond, pooled, negative_pooled)]

so ``from_pipe(diffusers_pipeline)`` wraps a loaded diffusers pipeline where diffusers exists, and the tests / the build
image (no diffusers, no weights) use :class:`DDIMSchedulerLite` and :class:`StackDenoiser` — a stand-in UNet whose blocks
are the ordered attention layers of the real UNet (``AttnStackUNet``) on a chained residual stream.

MI355X-first: ``interpolate`` runs the two passes of a step as ONE UNet call ``[cond ; uncond]`` by default (the AID
processors treat the second half as plain riders), every attention call is one library call, and the interior frames of
a guide-prompt run share one projected text context (``ctx_index``).
"""
from __future__ import annotations

import warnings

from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import nn

from . import ops
from . import processors as P
from .attn_shim import AttnStackUNet
from .interp import generate_beta_tensor, linear_interpolation, slerp, spherical_interpolation
from .loop import EARLY_MODES, install_sequence_processors, set_aid_active, set_ctx_index
from .sequence import SequenceBatch, prepare_sequence, prepare_single


# ---------------------------------------------------------------------------------------------
# components used where diffusers is absent
# ---------------------------------------------------------------------------------------------
class DDIMSchedulerLite:
    """Deterministic DDIM (eta = 0) on the scaled-linear beta schedule of SD / SDXL — the published update
    x_{t-1} = sqrt(a_{t-1}) x0 + sqrt(1 - a_{t-1}) eps with x0 = (x_t - sqrt(1 - a_t) eps) / sqrt(a_t) — behind the
    diffusers scheduler interface the pipelines use (third-party there; north_star names 50-step DDIM)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 steps_offset: int = 1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self.num_inference_steps = num_train_timesteps

    def set_timesteps(self, num_inference_steps: int, device=None):
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (torch.arange(0, num_inference_steps) * ratio).flip(0) + self.steps_offset      # "leading" spacing
        self.timesteps = ts.to(device) if device is not None else ts
        self.num_inference_steps = num_inference_steps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict: bool = False, **kwargs):
        t = int(timestep)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else torch.tensor(1.0, dtype=torch.float64)
        a_t, a_p = float(a_t), float(a_p)
        eps = model_output.to(torch.float32)
        x = sample.to(torch.float32)
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        out = (a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps).to(sample.dtype)
        return (out,)


class StackDenoiser(nn.Module):
    """Stand-in UNet for the build image: latents ``[N, 4, h, w]`` are lifted to the token streams of every resolution
    level by fixed linear maps, run through the ordered attention layers of the real UNet (``AttnStackUNet``; per block
    ``h += attn1(norm(h))``, ``h += attn2(norm(h), ctx)``) and projected back to a noise prediction.  All the arithmetic
    between the attention calls of the real UNet (convs, MLPs, time embedding) is third-party and absent here; the
    attention layers are exactly the ones the reference wraps.  ``sublayers``: "fused" = one library call per layer,
    "steps" = torch LayerNorm / processor call / torch add."""

    def __init__(self, model: str = "sd15", dtype=torch.float16, device=None, seed: int = 1002, scale_down: int = 16,
                 latent_hw: Tuple[int, int] = (16, 16), sublayers: str = "fused", head_div: int = 1):
        super().__init__()
        self.stack = AttnStackUNet(model, dtype=dtype, device=device, seed=seed, scale_down=scale_down, head_div=head_div)
        self.stack.sublayers = sublayers
        self.latent_hw = latent_hw
        self.in_channels = 4
        dev = self.stack.layers[0].to_q.weight.device
        g = torch.Generator(device=dev).manual_seed(seed + 1)
        n_lat = 4 * latent_hw[0] * latent_hw[1]
        self.lift, self.drop = nn.ParameterDict(), nn.ParameterDict()
        for (s, c) in self.stack.level_shapes():
            key = f"{s}_{c}"
            self.lift[key] = nn.Parameter((torch.randn(n_lat, s * 8, generator=g, device=dev) / n_lat ** 0.5).to(dtype),
                                          requires_grad=False)
            self.drop[key] = nn.Parameter((torch.randn(s * 8, n_lat, generator=g, device=dev) / (s * 8) ** 0.5).to(dtype),
                                          requires_grad=False)
        self.time_scale = 1.0 / 1000.0

    # diffusers' processor surface
    @property
    def attn_processors(self):
        return self.stack.attn_processors

    def set_attn_processor(self, processor):
        self.stack.set_attn_processor(processor)

    @property
    def dtype(self):
        return self.stack.layers[0].to_q.weight.dtype

    @property
    def device(self):
        return self.stack.layers[0].to_q.weight.device

    def forward(self, sample, timestep=None, encoder_hidden_states=None, added_cond_kwargs=None, return_dict=False, **kw):
        n = sample.shape[0]
        flat = sample.reshape(n, -1).to(self.dtype)
        # the timestep is used as a device scalar (no host read): the call can be captured into a hipGraph and replayed with
        # another timestep in the same buffer, like a real UNet's timestep embedding
        tshift = None
        if timestep is not None:
            tshift = (timestep if torch.is_tensor(timestep) else torch.tensor(float(timestep))).to(flat.device, torch.float32) * self.time_scale
        xs = {}
        for (s, c) in self.stack.level_shapes():
            tok = (flat @ self.lift[f"{s}_{c}"]).view(n, s, 8)                       # [N, S, 8] tokens ...
            wide = tok.repeat(1, 1, c // 8)                                          # ... tiled to the level width
            xs[(s, c)] = (wide if tshift is None else wide + tshift).contiguous()
        ehs = encoder_hidden_states
        if added_cond_kwargs and added_cond_kwargs.get("image_embeds") is not None:  # IP-Adapter UNets hand (text, [ip])
            ehs = (encoder_hidden_states, list(added_cond_kwargs["image_embeds"]))
        hs = self.stack(xs, ehs)
        out = torch.zeros_like(flat)
        for (s, c) in self.stack.level_shapes():
            h = hs[(s, c)].view(n, s, c // 8, 8).mean(dim=2)                          # [N, S, 8]
            out = out + h.reshape(n, s * 8) @ self.drop[f"{s}_{c}"]
        out = (out / len(self.stack.level_shapes())).view_as(sample).to(sample.dtype)
        return (out,)


# ---------------------------------------------------------------------------------------------
class _PassGraphs:
    """hipGraph capture of the distinct UNet passes of ONE denoising run (the loop it serves is
    pipeline_interpolated_sd.py:1834-1907).  A run has at most three kinds of pass — conditional with AID on, conditional
    plain, unconditional (or the two batched variants) — each launching a few hundred kernels per call.  The first call of
    a kind runs once eagerly (lazy kernel attributes, coefficient buffers, workspaces, the text K / V cache), is captured, and
    every later step only copies the step's inputs (scaled latents, timestep) into the captured buffers and replays.
    ``activate_aid(it)`` rewrites the coefficient buffers in place, so a replay reads the live schedule.

    What a replay FREEZES: every Python-side decision of the first call of a pass kind — the frame -> context map
    (``ctx_index``), ``plain_tail``, the IP-Adapter scales, which processor object sits on which layer.  A
    ``callback_on_step_end`` that swaps processors or embeddings between steps has no effect on replayed passes: run such loops
    with ``use_graphs=False``.  Each pass kind also holds a private pool with one UNet pass of activations.

    A UNet whose forward cannot be captured (a host synchronisation or a host -> device copy inside it) does not abort the
    run: the failed capture is abandoned, graphs are switched off for the rest of the run (``self.enabled = False``,
    ``self.fallback_reason`` says why) and the pass runs eagerly."""

    def __init__(self, enabled: bool):
        self.enabled = bool(enabled)
        self.fallback_reason: Optional[str] = None
        self._ent: Dict[Any, Tuple] = {}
        self._cap = None
        self._ws = ops.WorkspaceOwner()          # the captures' workspaces go when this object (and its graphs) goes

    def _capture(self, fn: Callable, inputs: Dict[str, torch.Tensor]):
        static = {k: v.clone() for k, v in inputs.items()}
        cur = torch.cuda.current_stream()
        if self._cap is None:
            self._cap = torch.cuda.Stream()         # warm-up AND capture stream: the capture adopts the warm-up's workspace (ops.workspace)
        cap = self._cap
        cap.wait_stream(cur)
        with torch.cuda.stream(cap):
            fn(**static)
        cur.wait_stream(cap)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with self._ws, torch.cuda.graph(graph, stream=cap):
            out = fn(**static)
        return graph, static, out

    def run(self, key, fn: Callable, **inputs: torch.Tensor):
        if not self.enabled:
            return fn(**inputs)
        ent = self._ent.get(key)
        if ent is None:
            try:
                ent = self._capture(fn, inputs)
            except RuntimeError as e:           # what a foreign UNet does that a stream capture forbids surfaces as RuntimeError
                # (hipErrorStreamCaptureUnsupported & co. through torch); anything else is a bug and propagates.  The fallback is
                # loud: a run that silently lost its graphs is a large performance regression
                self.enabled = False
                self.fallback_reason = f"{type(e).__name__}: {e}"
                self._ent.clear()
                self._ws.release()              # the graphs are gone (and so is the failed capture): their workspaces too
                warnings.warn("hipGraph capture of a UNet pass failed, the run continues eagerly: " + self.fallback_reason,
                              RuntimeWarning, stacklevel=3)
                try:
                    torch.cuda.synchronize()    # an invalidated capture may leave a sticky error behind: report it, do not hide it
                except RuntimeError as e2:
                    raise RuntimeError(f"device unusable after a failed stream capture ({self.fallback_reason})") from e2
                return fn(**inputs)
            self._ent[key] = ent
        graph, static, out = ent
        for k, v in inputs.items():
            static[k].copy_(v)
        graph.replay()
        return out                      # captured output buffer: consume it before the same kind of pass runs again


def _use_graphs(flag: Optional[bool], device: torch.device) -> bool:
    return (device.type == "cuda" and torch.cuda.is_available()) if flag is None else bool(flag)


def _dev_scalar(t, device: torch.device) -> torch.Tensor:
    """The step's timestep as a device tensor (graph input); schedulers hand out device or host scalars."""
    if torch.is_tensor(t):
        return t if t.device == device else t.to(device)
    return torch.tensor(t, device=device)


# ---------------------------------------------------------------------------------------------
class InterpolationStableDiffusionPipeline:
    """See the module docstring.  ``encode_prompt`` (callable) turns a prompt string into ``(cond, uncond)`` embeddings
    ``[1, L, Cc]``; without it pass embeddings directly (``embeds_start`` / ``embeds_end`` / ``embeds_guide``)."""

    is_xl = False
    default_guidance_scale = 7.5

    def __init__(self, unet, scheduler, vae=None, encode_prompt: Optional[Callable] = None,
                 encode_image: Optional[Callable] = None, vae_scaling_factor: float = 0.18215):
        self.unet, self.scheduler, self.vae = unet, scheduler, vae
        self._encode_prompt, self._encode_image = encode_prompt, encode_image
        self.vae_scaling_factor = vae_scaling_factor
        self._guidance_scale = self.default_guidance_scale
        self._aid_early: Optional[str] = None

    # -- wrapping a loaded diffusers pipeline (only where diffusers is importable) -----------------
    @classmethod
    def from_pipe(cls, pipe):
        """Take ``unet / scheduler / vae / encode_prompt`` from a loaded diffusers (SD or SDXL) pipeline."""
        def enc(prompt, negative_prompt=None):
            out = pipe.encode_prompt(prompt, pipe._execution_device, 1, True, negative_prompt=negative_prompt) \
                if not cls.is_xl else pipe.encode_prompt(prompt=prompt, device=pipe._execution_device,
                                                         num_images_per_prompt=1, do_classifier_free_guidance=True,
                                                         negative_prompt=negative_prompt)
            return out
        sf = getattr(getattr(pipe.vae, "config", None), "scaling_factor", 0.18215)
        enc_img = None
        if hasattr(pipe, "prepare_ip_adapter_image_embeds"):
            def enc_img(image):
                """(negative, positive) image embeddings of one image, as the reference takes them from diffusers'
                prepare_ip_adapter_image_embeds with classifier-free guidance on (pipeline_interpolated_sd.py:1763-1802)."""
                embeds = pipe.prepare_ip_adapter_image_embeds(image, None, pipe._execution_device, 1, True)[0]
                neg, pos = embeds.chunk(2)
                return neg, pos
        obj = cls(pipe.unet, pipe.scheduler, pipe.vae, encode_prompt=enc, encode_image=enc_img, vae_scaling_factor=sf)
        obj._load_ip_adapter = getattr(pipe, "load_ip_adapter", None)      # diffusers' loader: load_aid_ip_adapter(path, ...) uses it
        return obj

    # -- the reference's AID methods ----------------------------------------------------------------
    def load_aid(self, t: Optional[float] = 0.5, is_fused: bool = True, atype: str = "fused_outer", **kw):
        P.load_aid(self.unet, t=t, is_fused=is_fused, atype=atype, **kw)
        self._aid_early = atype

    def load_aid_ip_adapter(self, pretrained_model_name_or_path_or_dict=None, subfolder=None, weight_name=None,
                            t: Optional[float] = 0.5, is_fused: bool = True,
                            image_encoder_folder: Optional[str] = "image_encoder", early: str = "fused_outer", **kwargs):
        """pipeline_interpolated_sd.py:972-1007.  Loading the adapter weights is diffusers' ``load_ip_adapter``
        (third-party): it is called on the wrapped pipeline / UNet when that offers it; a UNet whose IP-Adapter
        processors are already installed (``AttnStackUNet.load_ip_adapter`` in the build image) is wrapped as is."""
        loader = getattr(self.unet, "load_ip_adapter_weights", None) or getattr(self, "_load_ip_adapter", None)
        if pretrained_model_name_or_path_or_dict is not None:
            if loader is None:
                raise RuntimeError("loading IP-Adapter weights needs diffusers' load_ip_adapter (wrap a diffusers pipeline "
                                   "with from_pipe), or install the adapter on the UNet first")
            loader(pretrained_model_name_or_path_or_dict, subfolder=subfolder, weight_name=weight_name,
                   image_encoder_folder=image_encoder_folder, **kwargs)
        P.load_aid_ip_adapter(self.unet, t=t, is_fused=is_fused, early=early)
        self._aid_early = early

    def activate_aid(self, it: float):
        P.activate_aid(self.unet, it)

    def deactivate_aid(self):
        P.deactivate_aid(self.unet)

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    # -- helpers ---------------------------------------------------------------------------------------
    def _embed(self, prompt, negative_prompt, given):
        if given is not None:
            return tuple(given)
        if prompt is None or self._encode_prompt is None:
            raise ValueError("pass a prompt and an `encode_prompt` component, or the embeddings themselves "
                             "(embeds_start / embeds_end / embeds_guide)")
        return tuple(self._encode_prompt(prompt, negative_prompt))

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        """pipeline_interpolated_sd.py:881-920: N(0, 1) * scheduler.init_noise_sigma unless given."""
        if latents is None:
            shape = (batch_size, num_channels_latents, height, width)
            latents = torch.randn(shape, generator=generator, dtype=torch.float32).to(device=device, dtype=dtype)
        else:
            latents = latents.to(device=device, dtype=dtype)
        return latents * getattr(self.scheduler, "init_noise_sigma", 1.0)

    def _added_cond(self, pooled, n: int, image_embeds=None) -> Optional[Dict[str, Any]]:
        return None if image_embeds is None else {"image_embeds": image_embeds}

    def _unet(self, latents, t, ctx, added):
        return self.unet(latents, t, encoder_hidden_states=ctx, added_cond_kwargs=added, return_dict=False)[0]

    def _decode(self, latents, output_type: str):
        if output_type == "latent" or self.vae is None:
            return latents
        image = self.vae.decode(latents / self.vae_scaling_factor, return_dict=False)[0]
        if output_type == "pt":
            return image
        return ((image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1) * 255).to(torch.uint8).cpu().numpy()

    # -- interpolate_single (batch 3) ---------------------------------------------------------------------
    @torch.no_grad()
    def interpolate_single(self, it: float = 0.5, prompt_start: Optional[str] = None, prompt_end: Optional[str] = None,
                           latent_start: Optional[torch.Tensor] = None, latent_end: Optional[torch.Tensor] = None,
                           image_start=None, image_end=None, guide_prompt: Optional[str] = None,
                           warmup_ratio: float = 0.5, is_fused: bool = True, atype: str = "outer", init: str = "linear",
                           height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 50,
                           timesteps: Optional[List[int]] = None, sigmas: Optional[List[float]] = None,
                           guidance_scale: Optional[float] = None, negative_prompt=None,
                           num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                           latents: Optional[torch.Tensor] = None, prompt_embeds=None, negative_prompt_embeds=None,
                           ip_adapter_image=None, ip_adapter_image_embeds=None, output_type: Optional[str] = "pil",
                           return_dict: bool = True, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                           guidance_rescale: float = 0.0, clip_skip: Optional[int] = None,
                           callback_on_step_end: Optional[Callable] = None,
                           callback_on_step_end_tensor_inputs: Sequence[str] = ("latents",),
                           # build-specific: embeddings instead of prompt strings / images
                           embeds_start=None, embeds_end=None, embeds_guide=None,
                           image_embeds_start=None, image_embeds_end=None, use_graphs: Optional[bool] = None, **kwargs):
        """Batch ``[start, target(it), end]`` through the denoising loop (pipeline_interpolated_sd.py:1407-1963).
        ``is_fused`` / ``atype`` are accepted and ignored exactly like the reference (:1418-1419):
        behaviour is set by ``load_aid``.  ``image_embeds_start / _end``: (negative, positive) image-embedding pairs
        ``[1, 1, emb]`` in place of ``image_start`` / ``image_end`` PIL images (whose encoding is diffusers' job).
        ``use_graphs`` (build-specific; None = on a GPU): the three kinds of UNet pass of the loop are captured into
        hipGraphs at their first step and replayed afterwards (:class:`_PassGraphs`); results are bit-identical to eager."""
        if image_start is not None and image_end is None:
            raise ValueError("Please provide both `image_start` and `image_end` to interpolate, or only `image_end` to "
                             "control the scale.")                                     # pipeline_interpolated_sd.py:1608-1612
        if (image_start is not None or image_end is not None) and self._encode_image is None:
            raise ValueError("image inputs need an `encode_image` component; pass image_embeds_start / image_embeds_end")
        if timesteps is not None or sigmas is not None or eta != 0.0 or guidance_rescale != 0.0:
            raise NotImplementedError("custom timesteps / sigmas / eta / guidance_rescale are the scheduler's (third-party) "
                                      "business; configure the scheduler component instead")
        gs = self.default_guidance_scale if guidance_scale is None else guidance_scale
        self._guidance_scale = gs
        dev, dtype = self._unet_device_dtype()
        cond_s, unc_s = self._split(self._embed(prompt_start, negative_prompt, embeds_start))
        cond_e, unc_e = self._split(self._embed(prompt_end, negative_prompt, embeds_end))
        guide = None
        if guide_prompt is not None or embeds_guide is not None:
            guide = self._split(self._embed(guide_prompt, negative_prompt, embeds_guide))
        h = height or self._latent_hw()[0] * 8
        w = width or self._latent_hw()[1] * 8
        nc = getattr(self.unet, "in_channels", 4)
        latent_start = self.prepare_latents(1, nc, h // 8, w // 8, dtype, dev, generator, latent_start)
        latent_end = self.prepare_latents(1, nc, h // 8, w // 8, dtype, dev, generator, latent_end)
        batch = prepare_single(it, latent_start, latent_end, cond_s[0], cond_e[0], unc_s[0], unc_e[0],
                               None if guide is None else guide[0][0], None if guide is None else guide[1][0], init=init)
        pooled = self._pooled_single(cond_s, cond_e, unc_s, unc_e, guide, it, init)
        img = self._image_embeds_single(image_start, image_end, image_embeds_start, image_embeds_end, it, init, dev, dtype)
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        warmup_steps = int(num_inference_steps * warmup_ratio)                         # :1831
        lat = batch.latents
        cond, unc = batch.cond.to(dev, dtype), batch.uncond.to(dev, dtype)
        graphs = _PassGraphs(_use_graphs(use_graphs, dev))
        P.clear_weight_caches()              # caches derived from weights live for ONE run (in-place weight edits are invisible to their keys)

        # loop-invariant conditioning, moved to the device once (the reference rebuilds the dict every step, :1850-1867)
        added_c = self._added_cond(None if pooled is None else pooled[0], 3, None if img is None else img[0])
        added_u = self._added_cond(None if pooled is None else pooled[1], 3, None if img is None else img[1])

        def cond_pass(x, t):
            return self._unet(x, t, cond, added_c)

        def uncond_pass(x, t):
            return self._unet(x, t, unc, added_u)

        for i, t in enumerate(self.scheduler.timesteps):
            x = self.scheduler.scale_model_input(lat, t)
            td = _dev_scalar(t, dev) if graphs.enabled else t
            if i < warmup_steps:                                                       # :1845-1848
                self.activate_aid(it)
            else:
                self.deactivate_aid()
            noise_text = graphs.run(("cond", i < warmup_steps), cond_pass, x=x, t=td)
            self.deactivate_aid()                                                      # :1870
            noise_unc = graphs.run(("uncond",), uncond_pass, x=x, t=td)
            noise = noise_unc + gs * (noise_text - noise_unc)                          # :1892
            lat = self.scheduler.step(noise, t, lat, return_dict=False)[0]
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, t, {"latents": lat}) or {}
                lat = out.pop("latents", lat)
        image = self._decode(lat, "latent" if output_type == "latent" else ("np" if output_type in ("pil", "np") else output_type))
        return {"images": image} if return_dict else (image, None)

    # -- N-frame interpolate (gradio) -------------------------------------------------------------------
    @torch.no_grad()
    def interpolate(self, latent_start: torch.Tensor, latent_end: torch.Tensor, prompt_start: Optional[str] = None,
                    prompt_end: Optional[str] = None, guide_prompt: Optional[str] = None, negative_prompt: str = "",
                    size: int = 7, num_inference_steps: int = 25, warmup_ratio: float = 0.5,
                    early: str = "fused_outer", late: str = "self", alpha: Optional[float] = None,
                    beta: Optional[float] = None, guidance_scale: Optional[float] = None,
                    # build-specific
                    embeds_start=None, embeds_end=None, embeds_guide=None, batched_cfg: bool = True,
                    output_type: str = "np", coef: Optional[Sequence[float]] = None, use_graphs: Optional[bool] = None):
        """gradio_src/pipeline_interpolated_stable_diffusion.py:163-304 with the root pipelines' warm-up count
        (``i < int(T * warmup_ratio)`` with 0-based ``i``).  ``late`` must be "self" (plain
        attention) or one of the AID modes.  ``batched_cfg``: the two passes of a step run as one UNet call.
        ``coef`` (build-specific, ``[0, t_1, ..., 1]``): render the frames at THESE coefficients the way
        ``interpolate_single(t_k)`` would — latents slerp'd and embeddings lerp'd at ``t_k``, attention coefficient
        ``t_k`` — in one N-frame run (the Beta-prior exploration fills several gaps per run with it, prior.py).
        ``use_graphs`` as in ``interpolate_single``.  The processors ``load_aid`` / ``load_aid_ip_adapter`` installed are put
        back when the run ends (the run works with its own N-frame processors)."""
        if early not in EARLY_MODES or (late != "self" and late not in EARLY_MODES):
            raise ValueError(f"early / late must be in {EARLY_MODES} (late also 'self')")
        gs = self.default_guidance_scale if guidance_scale is None else guidance_scale
        dev, dtype = self._unet_device_dtype()
        cond_s, unc_s = self._split(self._embed(prompt_start, negative_prompt, embeds_start))
        cond_e, unc_e = self._split(self._embed(prompt_end, negative_prompt, embeds_end))
        guide = None
        if guide_prompt is not None or embeds_guide is not None:
            guide = self._split(self._embed(guide_prompt, negative_prompt, embeds_guide))
        batch = prepare_sequence(latent_start.to(dev, dtype), latent_end.to(dev, dtype), cond_s[0], cond_e[0], unc_s[0],
                                 unc_e[0], size=size, guide_emb=None if guide is None else guide[0][0],
                                 uncond_guide=None if guide is None else guide[1][0],
                                 num_inference_steps=num_inference_steps, alpha=alpha, beta=beta)
        pooled = self._pooled_sequence(cond_s, cond_e, unc_s, unc_e, guide, size)
        if coef is not None:
            ts = [float(t) for t in coef]
            if len(ts) != size or ts[0] != 0.0 or ts[-1] != 1.0:
                raise ValueError("coef must be [0, ..., 1] with `size` entries")
            l0, l1 = latent_start.to(dev, dtype), latent_end.to(dev, dtype)
            batch.latents = torch.cat([l0] + [slerp(l0, l1, t) for t in ts[1:-1]] + [l1], dim=0)
            batch.coef = torch.tensor(ts, dtype=torch.float32)
            if guide is None:
                batch.cond = linear_interpolation(cond_s[0], cond_e[0], ts=ts)
                batch.uncond = linear_interpolation(unc_s[0], unc_e[0], ts=ts)
                if pooled is not None:
                    pooled = (linear_interpolation(cond_s[1], cond_e[1], ts=ts), linear_interpolation(unc_s[1], unc_e[1], ts=ts))
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        warmup_steps = int(num_inference_steps * warmup_ratio)
        installed = dict(self.unet.attn_processors)          # what load_aid / load_aid_ip_adapter put there: restored at the end
        try:
            procs = {}
            for mode in {early} | ({late} - {"self"}):
                install_sequence_processors(self.unet, size, early=mode, alpha=alpha, beta=beta,
                                            num_inference_steps=num_inference_steps, coef=batch.coef)
                procs[mode] = dict(self.unet.attn_processors)
            cond, unc = batch.cond.to(dev, dtype), batch.uncond.to(dev, dtype)
            idx = [int(i) for i in batch.ctx_index.tolist()] if batch.n_distinct_ctx != size else None
            lat = batch.latents
            graphs = _PassGraphs(_use_graphs(use_graphs, dev))
            ctx_both = torch.cat([cond, unc]) if batched_cfg else None      # ONE tensor for the whole run: the text K / V cache keys on it
            if batched_cfg:
                added_both = self._added_cond(None if pooled is None else torch.cat(pooled), 2 * size)
            else:
                added_c = self._added_cond(None if pooled is None else pooled[0], size)
                added_u = self._added_cond(None if pooled is None else pooled[1], size)

            def both_pass(x, t):
                return self._unet(torch.cat([x, x]), t, ctx_both, added_both)

            def cond_pass(x, t):
                return self._unet(x, t, cond, added_c)

            def uncond_pass(x, t):
                return self._unet(x, t, unc, added_u)

            for i, t in enumerate(self.scheduler.timesteps):
                x = self.scheduler.scale_model_input(lat, t)
                td = _dev_scalar(t, dev) if graphs.enabled else t
                mode = early if i < warmup_steps else late
                self.unet.set_attn_processor(procs[mode] if mode != "self" else procs[early])
                if batched_cfg:
                    set_aid_active(self.unet, mode != "self", plain_tail=size if mode != "self" else 0)
                    set_ctx_index(self.unet, None if idx is None else idx + [j + batch.n_distinct_ctx for j in idx])
                    both = graphs.run(("both", mode), both_pass, x=x, t=td)
                    noise_text, noise_unc = both[:size], both[size:]
                else:
                    set_ctx_index(self.unet, idx)
                    set_aid_active(self.unet, mode != "self")
                    noise_text = graphs.run(("cond", mode), cond_pass, x=x, t=td)
                    set_aid_active(self.unet, False)
                    noise_unc = graphs.run(("uncond", mode), uncond_pass, x=x, t=td)
                noise = noise_unc + gs * (noise_text - noise_unc)
                lat = self.scheduler.step(noise, t, lat, return_dict=False)[0]
        finally:
            set_ctx_index(self.unet, None)
            self.unet.set_attn_processor(installed)
        return self._decode(lat, output_type)

    # -- small shape helpers (overridden by the XL class) ----------------------------------------------
    def _split(self, embs):
        """(cond, uncond) -> ((cond,), (uncond,)); the XL class carries pooled embeddings in slot 1."""
        return (embs[0],), (embs[1],)

    def _pooled_single(self, cond_s, cond_e, unc_s, unc_e, guide, it, init):
        return None

    def _pooled_sequence(self, cond_s, cond_e, unc_s, unc_e, guide, size):
        return None

    def _image_embeds_single(self, image_start, image_end, ies, iee, it, init, dev, dtype):
        """pipeline_interpolated_sd.py:1763-1802: [start x3, target x3, end x3] image embeddings ``[9, 1, emb]`` for the
        conditional pass and their negative counterparts for the unconditional one; only ``image_end`` given = scale
        control (the start rows are the NEGATIVE embeddings of the end image)."""
        if image_end is not None and iee is None:
            iee = self._encode_image(image_end)
        if image_start is not None and ies is None:
            ies = self._encode_image(image_start)
        if iee is None:
            return None
        neg_e, pos_e = (t_.to(dev, dtype).expand(3, *t_.shape[1:]) for t_ in iee)
        if ies is None:
            neg_s, pos_s = neg_e, neg_e
        else:
            neg_s, pos_s = (t_.to(dev, dtype).expand(3, *t_.shape[1:]) for t_ in ies)
        mix = (lambda a, b: torch.lerp(a, b, it)) if init == "linear" else (lambda a, b: slerp(a, b, it))
        pos = torch.cat([pos_s, mix(pos_s, pos_e), pos_e], dim=0)
        neg = torch.cat([neg_s, mix(neg_s, neg_e), neg_e], dim=0)
        return [pos], [neg]

    def _unet_device_dtype(self):
        p = next(self.unet.parameters())
        return p.device, p.dtype

    def _latent_hw(self):
        return getattr(self.unet, "latent_hw", None) or (getattr(getattr(self.unet, "config", None), "sample_size", 64),) * 2


class InterpolationStableDiffusionXLPipeline(InterpolationStableDiffusionPipeline):
    """SDXL variant (pipeline_interpolated_sdxl.py:1693-2411): ``encode_prompt`` returns
    ``(cond, uncond, pooled, negative_pooled)``; the pooled text embeddings and the micro-conditioning ``time_ids`` ride
    in ``added_cond_kwargs`` of every UNet call (:2230-2272).  The interior frame's pooled embedding is the guide
    prompt's or the lerp / slerp of the end points', like the token embeddings (:1998-2040)."""

    is_xl = True
    default_guidance_scale = 5.0

    def __init__(self, *a, original_size=(1024, 1024), crops_coords_top_left=(0, 0), target_size=(1024, 1024), **kw):
        super().__init__(*a, **kw)
        self.time_ids = torch.tensor([list(original_size) + list(crops_coords_top_left) + list(target_size)],
                                     dtype=torch.float32)

    def _split(self, embs):
        if len(embs) != 4:
            raise ValueError("SDXL needs (cond, uncond, pooled, negative_pooled) embeddings")
        return (embs[0], embs[2]), (embs[1], embs[3])

    def _pooled_single(self, cond_s, cond_e, unc_s, unc_e, guide, it, init):
        mix = (lambda a, b: torch.lerp(a, b, it)) if init == "linear" else (lambda a, b: slerp(a, b, it))
        pos = torch.cat([cond_s[1], guide[0][1] if guide else mix(cond_s[1], cond_e[1]), cond_e[1]], dim=0)
        neg = torch.cat([unc_s[1], guide[1][1] if guide else mix(unc_s[1], unc_e[1]), unc_e[1]], dim=0)
        return pos, neg

    def _pooled_sequence(self, cond_s, cond_e, unc_s, unc_e, guide, size):
        if guide:
            pos = torch.cat([cond_s[1]] + [guide[0][1]] * (size - 2) + [cond_e[1]], dim=0)
            neg = torch.cat([unc_s[1]] + [guide[1][1]] * (size - 2) + [unc_e[1]], dim=0)
        else:
            pos = linear_interpolation(cond_s[1], cond_e[1], size=size)
            neg = linear_interpolation(unc_s[1], unc_e[1], size=size)
        return pos, neg

    def _added_cond(self, pooled, n: int, image_embeds=None):
        dev, dtype = self._unet_device_dtype()
        added = {"text_embeds": pooled.to(dev, dtype), "time_ids": self.time_ids.to(dev, dtype).expand(n, -1)}
        if image_embeds is not None:
            added["image_embeds"] = image_embeds
        return added
