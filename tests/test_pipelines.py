"""CPU: host logic of the pipeline classes — API surface against the reference's signatures, batch
layout, warm-up toggling, classifier-free guidance, scheduler — over a recording stub UNet (no GPU, no HIP calls)."""
import inspect

import pytest
import torch

import aid_amd
from aid_amd.pipelines import (DDIMSchedulerLite, InterpolationStableDiffusionPipeline,
                               InterpolationStableDiffusionXLPipeline)

# leading parameters of the reference's methods, in order (pipeline_interpolated_sd.py:1407-1450, :950-952, :973-985;
# gradio_src/pipeline_interpolated_stable_diffusion.py:163-179)
REF_SINGLE = ["it", "prompt_start", "prompt_end", "latent_start", "latent_end", "image_start", "image_end",
              "guide_prompt", "warmup_ratio", "is_fused", "atype", "init", "height", "width", "num_inference_steps",
              "timesteps", "sigmas", "guidance_scale", "negative_prompt", "num_images_per_prompt", "eta", "generator",
              "latents", "prompt_embeds", "negative_prompt_embeds", "ip_adapter_image", "ip_adapter_image_embeds",
              "output_type", "return_dict", "cross_attention_kwargs", "guidance_rescale", "clip_skip",
              "callback_on_step_end", "callback_on_step_end_tensor_inputs"]
REF_INTERPOLATE = ["latent_start", "latent_end", "prompt_start", "prompt_end", "guide_prompt", "negative_prompt", "size",
                   "num_inference_steps", "warmup_ratio", "early", "late", "alpha", "beta", "guidance_scale"]
REF_LOAD_AID = ["t", "is_fused", "atype"]
REF_LOAD_IP = ["pretrained_model_name_or_path_or_dict", "subfolder", "weight_name", "t", "is_fused",
               "image_encoder_folder", "early"]


def _params(fn):
    return [p for p in inspect.signature(fn).parameters if p != "self"]


@pytest.mark.parametrize("cls", [InterpolationStableDiffusionPipeline, InterpolationStableDiffusionXLPipeline])
def test_method_names_and_keyword_order_match_the_reference(cls):
    assert _params(cls.interpolate_single)[:len(REF_SINGLE)] == REF_SINGLE
    assert _params(cls.interpolate)[:len(REF_INTERPOLATE)] == REF_INTERPOLATE
    assert _params(cls.load_aid)[:3] == REF_LOAD_AID
    assert _params(cls.load_aid_ip_adapter)[:len(REF_LOAD_IP)] == REF_LOAD_IP
    sig = inspect.signature(cls.interpolate_single).parameters
    assert sig["it"].default == 0.5 and sig["warmup_ratio"].default == 0.5 and sig["init"].default == "linear"
    assert sig["num_inference_steps"].default == 50 and sig["atype"].default == "outer" and sig["is_fused"].default is True
    isig = inspect.signature(cls.interpolate).parameters
    assert isig["size"].default == 7 and isig["num_inference_steps"].default == 25 and isig["early"].default == "fused_outer"
    assert isig["late"].default == "self"
    for name in ("activate_aid", "deactivate_aid"):
        assert callable(getattr(cls, name))


class RecordingUNet(torch.nn.Module):
    """Linear toy denoiser over the processor surface of AttnStackUNet; records (aid state, batch, ctx rows) per call."""

    def __init__(self):
        super().__init__()
        self.inner = aid_amd.AttnStackUNet("sd15", dtype=torch.float32, scale_down=64, channel_div=8)
        self.w = torch.nn.Parameter(torch.tensor(0.0), requires_grad=False)
        self.in_channels, self.latent_hw = 4, (4, 4)
        self.calls = []

    attn_processors = property(lambda self: self.inner.attn_processors)

    def set_attn_processor(self, p):
        self.inner.set_attn_processor(p)

    def forward(self, sample, t, encoder_hidden_states=None, added_cond_kwargs=None, return_dict=False):
        procs = [p for p in self.attn_processors.values() if isinstance(p, aid_amd.InterpolatedAttnProcessor)]
        active = [p.activated for p in procs]
        assert all(active) or not any(active)
        self.calls.append(dict(active=bool(active and active[0]), n=sample.shape[0], t=int(t),
                               coef=None if not procs else procs[0].coef.clone(), tail=procs[0].plain_tail if procs else 0,
                               ctx_index=None if not procs else procs[0].ctx_index,
                               added=None if added_cond_kwargs is None else sorted(added_cond_kwargs)))
        n_aid = sample.shape[0] - (procs[0].plain_tail if procs else 0)
        bonus = torch.zeros(sample.shape[0], 1, 1, 1)
        if active and active[0]:
            bonus[:n_aid] = 0.05
        ctx = encoder_hidden_states
        return (0.1 * sample + 0.01 * ctx.mean(dim=(1, 2)).view(-1, 1, 1, 1) + bonus,)


def _manual(latents, cond, unc, steps, ratio, gs, sched):
    sched.set_timesteps(steps)
    lat = latents.clone()
    for i, t in enumerate(sched.timesteps):
        text = 0.1 * lat + 0.01 * cond.mean(dim=(1, 2)).view(-1, 1, 1, 1) + (0.05 if i < int(steps * ratio) else 0.0)
        un = 0.1 * lat + 0.01 * unc.mean(dim=(1, 2)).view(-1, 1, 1, 1)
        lat = sched.step(un + gs * (text - un), t, lat)[0]
    return lat


def test_interpolate_single_follows_the_root_loop():
    g = torch.Generator().manual_seed(0)
    unet = RecordingUNet()
    pipe = InterpolationStableDiffusionPipeline(unet, DDIMSchedulerLite())
    pipe.load_aid(t=0.5, is_fused=True, atype="fused_inner")
    l0, l1 = torch.randn(1, 4, 4, 4, generator=g), torch.randn(1, 4, 4, 4, generator=g)
    es, ee = (torch.randn(1, 7, 12, generator=g), torch.randn(1, 7, 12, generator=g)), \
             (torch.randn(1, 7, 12, generator=g), torch.randn(1, 7, 12, generator=g))
    out = pipe.interpolate_single(0.3, latent_start=l0, latent_end=l1, embeds_start=es, embeds_end=ee,
                                  num_inference_steps=10, warmup_ratio=0.4, guidance_scale=3.0, output_type="latent")
    lat = out["images"]
    assert lat.shape == (3, 4, 4, 4)
    calls = unet.calls
    assert len(calls) == 20 and all(c["n"] == 3 for c in calls)
    assert [c["active"] for c in calls[0::2]] == [True] * 4 + [False] * 6        # cond pass: i < int(10 * 0.4)
    assert not any(c["active"] for c in calls[1::2])                            # uncond pass always plain
    assert torch.allclose(calls[0]["coef"], torch.tensor([0.0, 0.3, 1.0]))       # activate_aid(it)
    cond = torch.cat([es[0], torch.lerp(es[0], ee[0], 0.3), ee[0]])
    unc = torch.cat([es[1], torch.lerp(es[1], ee[1], 0.3), ee[1]])
    lat0 = torch.cat([l0, aid_amd.slerp(l0, l1, 0.3), l1])
    want = _manual(lat0, cond, unc, 10, 0.4, 3.0, DDIMSchedulerLite())
    assert torch.allclose(lat, want, atol=1e-5)
    with pytest.raises(ValueError, match="image_end"):
        pipe.interpolate_single(0.3, latent_start=l0, latent_end=l1, embeds_start=es, embeds_end=ee, image_start=object())
    with pytest.raises(AssertionError):
        pipe.interpolate_single(1.3, latent_start=l0, latent_end=l1, embeds_start=es, embeds_end=ee, num_inference_steps=2)


@pytest.mark.parametrize("batched", [True, False])
def test_n_frame_interpolate_layout_guide_prompt_and_toggling(batched):
    g = torch.Generator().manual_seed(1)
    unet = RecordingUNet()
    pipe = InterpolationStableDiffusionPipeline(unet, DDIMSchedulerLite())
    l0, l1 = torch.randn(1, 4, 4, 4, generator=g), torch.randn(1, 4, 4, 4, generator=g)
    mk = lambda: (torch.randn(1, 7, 12, generator=g), torch.randn(1, 7, 12, generator=g))     # noqa: E731
    es, ee, eg = mk(), mk(), mk()
    lat = pipe.interpolate(l0, l1, embeds_start=es, embeds_end=ee, embeds_guide=eg, size=5, num_inference_steps=8,
                           warmup_ratio=0.5, early="fused_outer", guidance_scale=2.0, batched_cfg=batched,
                           output_type="latent")
    assert lat.shape == (5, 4, 4, 4)
    calls = unet.calls
    if batched:
        assert len(calls) == 8 and all(c["n"] == 10 for c in calls)
        assert [c["active"] for c in calls] == [True] * 4 + [False] * 4
        assert [c["tail"] for c in calls] == [5] * 4 + [0] * 4
        assert calls[0]["ctx_index"] == [0, 1, 1, 1, 2, 3, 4, 4, 4, 5]         # guide prompt: 3 distinct contexts per pass
    else:
        assert len(calls) == 16 and all(c["n"] == 5 for c in calls)
        assert [c["active"] for c in calls[0::2]] == [True] * 4 + [False] * 4 and not any(c["active"] for c in calls[1::2])
    want_coef = aid_amd.generate_beta_tensor(5, 8, 8)
    want_coef[0], want_coef[-1] = 0, 1
    assert torch.allclose(calls[0]["coef"], want_coef)                          # alpha = beta = num_inference_steps
    cond = torch.cat([es[0]] + [eg[0]] * 3 + [ee[0]])
    unc = torch.cat([es[1]] + [eg[1]] * 3 + [ee[1]])
    want = _manual(aid_amd.spherical_interpolation(l0, l1, 5), cond, unc, 8, 0.5, 2.0, DDIMSchedulerLite())
    assert torch.allclose(lat, want, atol=1e-5)


def test_xl_pipeline_carries_pooled_embeddings_and_time_ids():
    g = torch.Generator().manual_seed(2)
    unet = RecordingUNet()
    pipe = InterpolationStableDiffusionXLPipeline(unet, DDIMSchedulerLite())
    pipe.load_aid(t=0.5, is_fused=True, atype="fused_outer")
    l0, l1 = torch.randn(1, 4, 4, 4, generator=g), torch.randn(1, 4, 4, 4, generator=g)
    mk = lambda: (torch.randn(1, 7, 12, generator=g), torch.randn(1, 7, 12, generator=g),            # noqa: E731
                  torch.randn(1, 6, generator=g), torch.randn(1, 6, generator=g))
    out = pipe.interpolate_single(0.5, latent_start=l0, latent_end=l1, embeds_start=mk(), embeds_end=mk(),
                                  num_inference_steps=4, output_type="latent", return_dict=False)
    assert out[0].shape == (3, 4, 4, 4) and pipe.guidance_scale == 5.0
    assert all(c["added"] == ["text_embeds", "time_ids"] for c in unet.calls)
    with pytest.raises(ValueError, match="pooled"):
        pipe.interpolate_single(0.5, latent_start=l0, latent_end=l1, embeds_start=mk()[:2], embeds_end=mk()[:2])


def test_ddim_scheduler_known_answers():
    s = DDIMSchedulerLite()
    s.set_timesteps(50)
    assert s.timesteps[0].item() == 981 and s.timesteps[-1].item() == 1 and len(s.timesteps) == 50
    x = torch.full((1, 4, 2, 2), 0.5)
    eps = torch.full((1, 4, 2, 2), -0.25)
    a_t, a_p = float(s.alphas_cumprod[981]), float(s.alphas_cumprod[961])
    x0 = (0.5 - (1 - a_t) ** 0.5 * -0.25) / a_t ** 0.5
    want = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * -0.25
    assert torch.allclose(s.step(eps, 981, x)[0], torch.full_like(x, want), atol=1e-6)
    last = s.step(eps, 1, x)[0]                                                   # final step lands on x0 (alpha_prev = 1)
    a1 = float(s.alphas_cumprod[1])
    assert torch.allclose(last, torch.full_like(x, (0.5 - (1 - a1) ** 0.5 * -0.25) / a1 ** 0.5), atol=1e-6)


def test_load_aid_ip_adapter_wraps_like_the_reference():
    unet = aid_amd.AttnStackUNet("sdxl", dtype=torch.float32, scale_down=64, channel_div=8)
    unet.load_ip_adapter(num_tokens=4, scale=0.5)
    pipe = InterpolationStableDiffusionXLPipeline(unet, DDIMSchedulerLite())
    for early, cls in (("fused_outer", aid_amd.OuterInterpolatedIPAttnProcessor),
                       ("fused_inner", aid_amd.InnerInterpolatedIPAttnProcessor),
                       ("scale_control", aid_amd.ScaleControlIPAttnProcessor)):
        unet.load_ip_adapter(num_tokens=4, scale=0.5)
        pipe.load_aid_ip_adapter(t=0.4, is_fused=True, early=early)
        procs = unet.attn_processors
        assert len(procs) == 140 and all(isinstance(p, cls) for p in procs.values())
        cross = [p for n, p in procs.items() if ".attn2." in n]
        selfs = [p for n, p in procs.items() if ".attn1." in n]
        assert all(isinstance(p.ip_attn, aid_amd.HipIPAdapterAttnProcessor) and p.scale == [0.5] and p.num_tokens == (4,)
                   for p in cross)
        assert all(isinstance(p.ip_attn, aid_amd.HipAttnProcessor) for p in selfs)
        assert torch.allclose(cross[0].coef, torch.tensor([0.0, 0.4, 1.0]))
    pipe.deactivate_aid()
    assert not any(p.activated for p in unet.attn_processors.values())
    pipe.activate_aid(0.25)
    assert all(p.activated and abs(float(p.coef[1]) - 0.25) < 1e-7 for p in unet.attn_processors.values())
    with pytest.raises(ValueError, match="early"):
        pipe.load_aid_ip_adapter(early="nope")


def test_from_pipe_wires_the_ip_adapter_loader_and_the_image_encoder():
    """a wrapped diffusers pipeline must hand over `load_ip_adapter` (so load_aid_ip_adapter(path, ...) can load
    weights) and an image encoder built on `prepare_ip_adapter_image_embeds` (so image_start / image_end work)."""
    calls = {}

    class FakeVae:
        config = type("C", (), {"scaling_factor": 0.13025})()

    class FakePipe:
        unet, scheduler, vae = object(), object(), FakeVae()
        _execution_device = "cpu"

        def encode_prompt(self, *a, **k):
            return ("cond", "uncond")

        def load_ip_adapter(self, path, subfolder=None, weight_name=None, image_encoder_folder=None, **kw):
            calls["load"] = (path, subfolder, weight_name, image_encoder_folder)

        def prepare_ip_adapter_image_embeds(self, image, embeds, device, n, cfg):
            calls["img"] = (image, device, n, cfg)
            return [torch.arange(2 * 1 * 4, dtype=torch.float32).view(2, 1, 4)]

    pipe = aid_amd.InterpolationStableDiffusionPipeline.from_pipe(FakePipe())
    assert pipe.vae_scaling_factor == 0.13025 and pipe._load_ip_adapter is not None
    neg, pos = pipe._encode_image("an image")
    assert calls["img"] == ("an image", "cpu", 1, True)
    assert neg.tolist() == [[[0.0, 1.0, 2.0, 3.0]]] and pos.tolist() == [[[4.0, 5.0, 6.0, 7.0]]]
    pipe._load_ip_adapter("h94/IP-Adapter", subfolder="models", weight_name="w.bin", image_encoder_folder="image_encoder")
    assert calls["load"] == ("h94/IP-Adapter", "models", "w.bin", "image_encoder")
