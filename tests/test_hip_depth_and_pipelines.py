"""GPU: depth parity and end-to-end pipeline parity (north_star "latents within 1e-3 rel-L2").

(1) Depth: the residual stream ``h += attn(norm(h), ctx)`` chained through ALL 32 (SD1.5, fp16) / 140 (SDXL, bf16)
    attention layers of one UNet pass at reduced S (full widths, so the shipped head dims and GEMM paths run), for an AID
    pass and a plain pass, HIP vs the fp64 oracle chain — the error after every layer is recorded, its growth bounded.
(2) Pipelines: ``interpolate_single`` (batch 3, activate_aid / deactivate_aid per step) and the N-frame ``interpolate``
    (batched CFG, guide-prompt contexts) over the stand-in denoiser for several DDIM steps, HIP vs the same loop with
    every attention layer evaluated by the fp64 oracle.
Measured values are written to gpurun_out/depth_parity.json when AID_WRITE_MEASUREMENTS=1."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import aid_oracle as O
from util import rel_l2, to_np64

pytestmark = pytest.mark.gpu

import aid_amd  # noqa: E402
from aid_amd.loop import install_sequence_processors, set_aid_active  # noqa: E402
from aid_amd.pipelines import (DDIMSchedulerLite, InterpolationStableDiffusionPipeline,  # noqa: E402
                               InterpolationStableDiffusionXLPipeline, StackDenoiser)

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(key, value):
    if os.environ.get("AID_WRITE_MEASUREMENTS") != "1":
        return
    path = os.path.join(ROOT, "gpurun_out", "depth_parity.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = value
    json.dump(data, open(path, "w"), indent=1)


_W64 = {}


def _w(m, heads):
    """fp64 copies of a layer's weights, converted once per module (the 50-step oracle loops call every layer 100 times)."""
    hit = _W64.get(id(m))
    if hit is None or hit[0] is not m:
        hit = (m, O.AttnWeights(*(to_np64(t) for t in (m.to_q.weight, m.to_k.weight, m.to_v.weight, m.to_out[0].weight,
                                                        m.to_out[0].bias)), heads))
        _W64[id(m)] = hit
    return hit[1]


def _oracle_layer(stack, i, h64, ctx64, mode, fused, coef):
    """One sublayer  h + attn(LayerNorm(h), ctx)  in fp64 with the layer's (fp16 / bf16-valued) weights."""
    m, nrm, (s, c, heads, is_cross) = stack.layers[i], stack.norms[i], stack.shapes[i]
    gb = _W64.get(id(nrm))
    if gb is None or gb[0] is not nrm:
        gb = (nrm, to_np64(nrm.weight), to_np64(nrm.bias))
        _W64[id(nrm)] = gb
    hn = O.layer_norm(h64, gb[1], gb[2], nrm.eps)
    ctx = ctx64 if is_cross else None
    w = _w(m, heads)
    if mode == "plain":
        a = O.plain_attention(hn, ctx, w)
    elif mode == "outer":
        a = O.outer_attention(hn, ctx, w, coef, fused)
    else:
        a = O.inner_attention(hn, ctx, w, coef, fused)
    return h64 + a


# BOUNDS: rel-L2 of the residual stream after the LAST layer of each resolution level (HIP storage dtype vs fp64).
# One sublayer measures ~3e-4 (fp16) / ~2.5e-3 (bf16) on the attention term; the stream itself is re-rounded to the
# storage dtype after every layer (eps/2 = 2.4e-4 fp16, 2e-3 bf16 relative per rounding), errors add in quadrature over
# the L layers of a level.  Bounds = 1.3 x the measured values (profiles/r05_depth_parity.json: 8.0e-4 / 1.46e-2 / 1.78e-3; the runs are
# deterministic, the margin is for other boxes' clocks changing nothing and for future kernel changes to have to argue).
# SDXL in fp16 storage (the kernels serve d = 64 in both dtypes): the 140-layer stream stays at the fp16 level — this is the
# configuration that meets north_star's 1e-3 per layer; bf16 pays 8x the rounding step.
DEPTH_BOUND = {("sd15", torch.float16): 1.05e-3, ("sdxl", torch.bfloat16): 1.9e-2, ("sdxl", torch.float16): 2.35e-3}


@pytest.mark.parametrize("model,dtype,early", [("sd15", torch.float16, "fused_inner"), ("sdxl", torch.bfloat16, "fused_outer"),
                                               ("sdxl", torch.float16, "fused_outer")])
def test_depth_parity_chained_through_every_layer(model, dtype, early):
    n = 7 if model == "sd15" else 5
    stack = aid_amd.AttnStackUNet(model, dtype=dtype, device=DEV, scale_down=16 if model == "sd15" else 32)
    install_sequence_processors(stack, n, early=early, num_inference_steps=50)
    g = torch.Generator().manual_seed(77)
    xs = {k: torch.randn(n, k[0], k[1], generator=g).to(dtype) for k in stack.level_shapes()}
    ctx = torch.randn(n, stack.text_len, stack.cross_dim, generator=g).to(dtype)
    coef = next(iter(stack.attn_processors.values())).coef.to(dtype).float().numpy()
    mode, fused = ("outer" if early.endswith("outer") else "inner"), early.startswith("fused")
    report = {}
    for step_mode in (mode, "plain"):
        set_aid_active(stack, step_mode != "plain")
        hs = {k: v.to(DEV) for k, v in xs.items()}
        hs64 = {k: to_np64(v) for k, v in xs.items()}
        ctx_d, ctx64 = ctx.to(DEV), to_np64(ctx)
        curve = []
        for i, (m, nrm, (s, c, h, is_cross)) in enumerate(zip(stack.layers, stack.norms, stack.shapes)):
            hs[(s, c)] = m.processor.fused_sublayer(m, nrm, hs[(s, c)], ctx_d if is_cross else None)
            hs64[(s, c)] = _oracle_layer(stack, i, hs64[(s, c)], ctx64, step_mode, fused, coef)
            curve.append(rel_l2(to_np64(hs[(s, c)]), hs64[(s, c)]))
        final = {f"{k[0]}x{k[1]}": rel_l2(to_np64(hs[k]), hs64[k]) for k in hs}
        report[step_mode] = dict(first_layer=curve[0], worst=max(curve), final=final, layers=len(curve),
                                 every_8th=curve[7::8])
        assert max(final.values()) < DEPTH_BOUND[(model, dtype)], (step_mode, final)
        assert max(curve) < DEPTH_BOUND[(model, dtype)]
        # growth over depth stays far below linear accumulation of the per-layer error
        assert max(curve) < 0.5 * len(curve) * max(curve[0], 1e-4)
    _record(f"depth_{model}" + ("_fp16" if (model == "sdxl" and dtype == torch.float16) else ""), report)


class OracleDenoiser(torch.nn.Module):
    """StackDenoiser.forward in fp64 with every attention layer evaluated by the oracle; reads the activation state /
    coefficients / riders of the processors installed on the HIP denoiser it mirrors."""

    def __init__(self, hip: StackDenoiser, storage: "torch.dtype | None" = None):
        """``storage``: round to that dtype at every point where ANY implementation that keeps its tensors in that dtype must round —
        the projected q / k / v, the attention output, the layer output and the residual stream — and nowhere else (fp64 arithmetic,
        un-rounded probabilities, exact LayerNorm): the FLOOR of the storage type for this loop, independent of any kernel."""
        super().__init__()
        self.hip = hip
        self.storage = storage
        self.latent_hw, self.in_channels = hip.latent_hw, 4
        self.dummy = torch.nn.Parameter(torch.zeros(1, dtype=torch.float64), requires_grad=False)

    def _layer(self, st, i, h, ctx, mode, fused, coef):
        if self.storage is None:
            return _oracle_layer(st, i, h, ctx, mode, fused, coef)
        rd = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.storage).double().numpy()      # noqa: E731
        m, nrm, (s, c, heads, is_cross) = st.layers[i], st.norms[i], st.shapes[i]
        hn = O.layer_norm(h, to_np64(nrm.weight), to_np64(nrm.bias), nrm.eps)
        w = _w(m, heads)
        q, k, v = O._project(hn, ctx if is_cross else None, w)
        o = O.attn_core(rd(q), rd(k), rd(v), heads, w.scale, mode, fused and mode != "plain", coef)
        return rd(h + rd(O._out(rd(o), w)))

    attn_processors = property(lambda self: self.hip.attn_processors)

    def set_attn_processor(self, p):
        self.hip.set_attn_processor(p)

    def forward(self, sample, timestep=None, encoder_hidden_states=None, added_cond_kwargs=None, return_dict=False, **kw):
        st = self.hip.stack
        n = sample.shape[0]
        # (storage mode also rounds where the scaffolding around the layers stores: the latents and embeddings handed in, the lifted
        # tokens, the stream the first layer reads, the predicted noise)
        rd = (lambda a: a) if self.storage is None else \
            (lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.storage).double().numpy())
        flat = rd(sample.reshape(n, -1).double().numpy())
        tshift = 0.0 if timestep is None else float(timestep) * self.hip.time_scale
        hs = {}
        for (s, c) in st.level_shapes():
            tok = rd(flat @ to_np64(self.hip.lift[f"{s}_{c}"])).reshape(n, s, 8)
            hs[(s, c)] = rd(np.tile(tok, (1, 1, c // 8)) + tshift)
        ctx_all = rd(encoder_hidden_states.double().numpy())
        for i, (m, (s, c, h, is_cross)) in enumerate(zip(st.layers, st.shapes)):
            proc = m.processor
            n_aid = n - proc.plain_tail if proc.activated else 0
            idx = proc.ctx_index
            ctx = ctx_all if idx is None or ctx_all.shape[0] == n else None
            if ctx is None:
                ctx = ctx_all[idx]
            elif idx is not None:                      # repeated contexts were passed: rows by first occurrence
                first = [idx.index(r) for r in range(max(idx) + 1)]
                ctx = ctx_all[[first[j] for j in idx]]
            mode = "plain" if not proc.activated else proc._mode
            coef = proc.coef.to(self.hip.dtype).float().numpy()
            hcur = hs[(s, c)]
            if mode == "plain":
                hs[(s, c)] = self._layer(st, i, hcur, ctx, "plain", False, None)
            else:
                a = self._layer(st, i, hcur[:n_aid], ctx[:n_aid], mode, proc.is_fused, coef)
                if n_aid < n:
                    a = np.concatenate([a, self._layer(st, i, hcur[n_aid:], ctx[n_aid:], "plain", False, None)])
                hs[(s, c)] = a
        out = np.zeros_like(flat)
        for (s, c) in st.level_shapes():
            hm = hs[(s, c)].reshape(n, s, c // 8, 8).mean(axis=2)
            out = out + hm.reshape(n, s * 8) @ to_np64(self.hip.drop[f"{s}_{c}"])
        out = rd(out / len(st.level_shapes()))
        return (torch.from_numpy(out).view_as(sample),)

    def parameters(self, recurse=True):
        return iter([self.dummy])


class StorageScheduler(DDIMSchedulerLite):
    """DDIM with the guided noise and the latents rounded to a storage dtype where a loop that keeps them in that dtype rounds them
    (the arithmetic of the step itself stays un-rounded) — the scheduler side of OracleDenoiser(storage=...)."""

    def __init__(self, storage, **kw):
        super().__init__(**kw)
        self.storage = storage

    def step(self, model_output, timestep, sample, **kw):
        rd = lambda a: a.to(self.storage).to(a.dtype)          # noqa: E731
        return (rd(super().step(rd(model_output), timestep, rd(sample), **kw)[0]),)


PIPE_BOUND = {torch.float16: 2.5e-3, torch.bfloat16: 2.1e-2}     # rel-L2 of the final latents, 1.3 x measured (1.92e-3 / 1.60e-2)


def _embs(g, cc, xl=False):
    base = (torch.randn(1, 77, cc, generator=g), torch.randn(1, 77, cc, generator=g))
    return base + ((torch.randn(1, 32, generator=g), torch.randn(1, 32, generator=g)) if xl else ())


@pytest.mark.parametrize("model,dtype,atype", [("sd15", torch.float16, "fused_inner"), ("sdxl", torch.bfloat16, "fused_outer")])
def test_interpolate_single_end_to_end_vs_oracle_loop(model, dtype, atype):
    steps = 6 if model == "sd15" else 3
    hip = StackDenoiser(model, dtype=dtype, device=DEV, scale_down=16 if model == "sd15" else 64, latent_hw=(8, 8))
    cls = InterpolationStableDiffusionXLPipeline if model == "sdxl" else InterpolationStableDiffusionPipeline
    g = torch.Generator().manual_seed(9)
    l0, l1 = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    es, ee = _embs(g, hip.stack.cross_dim, model == "sdxl"), _embs(g, hip.stack.cross_dim, model == "sdxl")
    rd = lambda t: tuple(e.to(dtype).float() for e in t)     # noqa: E731   both runs see the dtype-rounded embeddings
    es, ee = rd(es), rd(ee)
    pipe = cls(hip, DDIMSchedulerLite())
    pipe.load_aid(t=0.5, is_fused=True, atype=atype)
    out = pipe.interpolate_single(0.35, latent_start=l0, latent_end=l1, embeds_start=es, embeds_end=ee,
                                  num_inference_steps=steps, warmup_ratio=0.5, output_type="latent")["images"]
    ora = cls(OracleDenoiser(hip), DDIMSchedulerLite())
    ref = ora.interpolate_single(0.35, latent_start=l0.to(dtype).double(), latent_end=l1.to(dtype).double(),
                                 embeds_start=tuple(e.double() for e in es), embeds_end=tuple(e.double() for e in ee),
                                 num_inference_steps=steps, warmup_ratio=0.5, output_type="latent")["images"]
    err = rel_l2(to_np64(out), ref.numpy())
    _record(f"interpolate_single_{model}", dict(steps=steps, rel_l2=err))
    assert out.shape == (3, 4, 8, 8) and err < PIPE_BOUND[dtype], err


def _slerp64(a, b, t):
    """interpolation.py:861-918 restated on numpy fp64: row-wise over the last dim, lerp fallback at |cos| > 0.9995."""
    a2, b2 = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    out = np.empty_like(a2)
    for r in range(a2.shape[0]):
        u, v = a2[r], b2[r]
        cos = float(np.dot(u, v) / (np.linalg.norm(u) * np.linalg.norm(v)))
        if abs(cos) > 0.9995:
            out[r] = (1 - t) * u + t * v
        else:
            th = np.arccos(cos)
            out[r] = (np.sin((1 - t) * th) * u + np.sin(t * th) * v) / np.sin(th)
    return out.reshape(a.shape)


def test_interpolate_single_against_an_independently_written_loop():
    """The orchestration of interpolate_single — batch [start, target(it), end], latents slerp'd / embeddings lerp'd at `it`, AID active
    on the text pass of steps i < int(T * warmup_ratio) with coefficients [0, it, 1], de-activated for the unconditional pass and
    afterwards, classifier-free guidance, DDIM update — written out AGAIN here from reference pipeline_interpolated_sd.py:1690-1747 and
    :1831-1907 (numpy fp64, oracle attention, no pipelines.py / sequence.py / interp.py), against the HIP pipeline."""
    dtype, steps, it, gs, ratio = torch.float16, 6, 0.35, 5.0, 0.5
    hip = StackDenoiser("sd15", dtype=dtype, device=DEV, scale_down=16, latent_hw=(8, 8))
    g = torch.Generator().manual_seed(12)
    l0, l1 = torch.randn(1, 4, 8, 8, generator=g).to(dtype), torch.randn(1, 4, 8, 8, generator=g).to(dtype)
    rd = lambda t: tuple(e.to(dtype).float() for e in t)     # noqa: E731
    es, ee = rd(_embs(g, 768)), rd(_embs(g, 768))             # (positive, negative) embeddings of the two prompts
    pipe = InterpolationStableDiffusionPipeline(hip, DDIMSchedulerLite())
    pipe.load_aid(t=0.5, is_fused=True, atype="fused_inner")
    out = pipe.interpolate_single(it, latent_start=l0, latent_end=l1, embeds_start=es, embeds_end=ee, num_inference_steps=steps,
                                  warmup_ratio=ratio, guidance_scale=gs, output_type="latent")["images"]

    # ---- the same run, written out again ----
    den = OracleDenoiser(hip)
    procs = list(hip.attn_processors.values())
    a64 = lambda t: t.double().numpy()                          # noqa: E731
    lat = np.concatenate([a64(l0), _slerp64(a64(l0), a64(l1), it), a64(l1)])                 # :1690-1706
    cond = np.concatenate([a64(es[0]), (1 - it) * a64(es[0]) + it * a64(ee[0]), a64(ee[0])])   # :1716-1730 ("linear" init)
    unc = np.concatenate([a64(es[1]), (1 - it) * a64(es[1]) + it * a64(ee[1]), a64(ee[1])])
    # SD's scheduler configuration: scaled-linear betas 0.00085 .. 0.012 over 1000 steps, "leading" spacing, steps_offset 1
    ac = np.cumprod(1.0 - np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2)
    ts = [int(v) for v in (np.arange(steps) * (1000 // steps))[::-1] + 1]
    warm = int(steps * ratio)                                                               # :1831
    for i, t in enumerate(ts):
        for p_ in procs:                                                                    # :1845-1848
            if i < warm:
                p_.activate(it)
            else:
                p_.deactivate()
        eps_text = den(torch.from_numpy(lat), t, torch.from_numpy(cond))[0].numpy()
        for p_ in procs:                                                                    # :1870
            p_.deactivate()
        eps_unc = den(torch.from_numpy(lat), t, torch.from_numpy(unc))[0].numpy()
        eps = eps_unc + gs * (eps_text - eps_unc)                                           # :1892
        a_t = ac[t]                                                                         # DDIM, eta = 0 (Song et al. 2021, eq. 12)
        a_prev = ac[ts[i + 1]] if i + 1 < len(ts) else 1.0
        x0 = (lat - np.sqrt(1 - a_t) * eps) / np.sqrt(a_t)
        lat = np.sqrt(a_prev) * x0 + np.sqrt(1 - a_prev) * eps
    err = rel_l2(to_np64(out), lat)
    assert out.shape == (3, 4, 8, 8) and err < PIPE_BOUND[dtype], err


@pytest.mark.parametrize("guided", [False, True])
def test_n_frame_interpolate_end_to_end_vs_oracle_loop(guided):
    dtype, steps, size = torch.float16, 4, 5
    hip = StackDenoiser("sd15", dtype=dtype, device=DEV, scale_down=16, latent_hw=(8, 8))
    g = torch.Generator().manual_seed(10)
    l0, l1 = torch.randn(1, 4, 8, 8, generator=g).to(dtype), torch.randn(1, 4, 8, 8, generator=g).to(dtype)
    rd = lambda t: tuple(e.to(dtype).float() for e in t)     # noqa: E731
    es, ee, eg = rd(_embs(g, 768)), rd(_embs(g, 768)), rd(_embs(g, 768))
    kw = dict(size=size, num_inference_steps=steps, warmup_ratio=0.5, early="fused_outer", guidance_scale=4.0,
              output_type="latent")
    pipe = InterpolationStableDiffusionPipeline(hip, DDIMSchedulerLite())
    out = pipe.interpolate(l0, l1, embeds_start=es, embeds_end=ee, embeds_guide=eg if guided else None, **kw)
    two = pipe.interpolate(l0, l1, embeds_start=es, embeds_end=ee, embeds_guide=eg if guided else None,
                           batched_cfg=False, **kw)
    assert torch.equal(out, two)                               # one batched call per step == the reference's two calls
    ora = InterpolationStableDiffusionPipeline(OracleDenoiser(hip), DDIMSchedulerLite())
    dbl = lambda t: tuple(e.double() for e in t)               # noqa: E731
    ref = ora.interpolate(l0.double(), l1.double(), embeds_start=dbl(es), embeds_end=dbl(ee),
                          embeds_guide=dbl(eg) if guided else None, **kw)
    err = rel_l2(to_np64(out), ref.numpy())
    _record(f"interpolate_n{size}_{'guided' if guided else 'lerp'}", dict(steps=steps, rel_l2=err))
    assert out.shape == (size, 4, 8, 8) and err < PIPE_BOUND[dtype], err


# ---- 50 steps end to end ---------------------------------------------------------------------------------------------
# The reference loop is 50 DDIM steps (pipeline_interpolated_sd.py:1831-1870); the tests above run 3 - 6.  Here: interpolate_single and
# the 7-frame interpolate for the full 50 steps, warm-up ratio 0.5 (25 AID steps + 25 plain), HIP vs the same loop with every attention
# layer evaluated by the fp64 oracle.  All 32 / 140 layers at reduced S; the default also divides heads and widths together (head_div:
# head dims 40 / 80 / 160 / 64 kept, so the shipped kernels run) because the fp64 loop is bound by the weight bytes (SDXL: 8.4 GB of
# fp64 weights per pass at full width); AID_E2E_FULL_WIDTH=1 runs the full widths (tools/refresh_profiles.sh does, once per round).
# STATED TOLERANCE of the final latents after 50 steps (rel-L2 vs fp64) = 1.3 x the largest value measured for the configuration
# (profiles/r05_depth_parity.json; VERDICT r4 next #5b: a 2x bound lets a doubling of the accumulated error pass):
# measured (round 5): SD1.5 fp16 1.32e-3 / 1.36e-3 (single / 7 frames; 1.41e-3 at full width), SDXL fp16 1.80e-3 (1.54e-3 at full width),
# SDXL bf16 1.52e-2 / 1.67e-2
E2E50_BOUND = {("sd15", torch.float16): 1.85e-3, ("sdxl", torch.float16): 2.35e-3, ("sdxl", torch.bfloat16): 2.2e-2}
E2E50 = [("sd15", torch.float16, "fused_inner"), ("sdxl", torch.bfloat16, "fused_outer"), ("sdxl", torch.float16, "fused_outer")]
E2E50_N7 = E2E50 if os.environ.get("AID_E2E_ALL") == "1" else E2E50[:2]      # (the fp64 loop of a 14-frame SDXL run takes a minute)


def _e2e50_denoiser(model, dtype):
    full = os.environ.get("AID_E2E_FULL_WIDTH") == "1"
    hd = 1 if full else (4 if model == "sd15" else 5)
    return StackDenoiser(model, dtype=dtype, device=DEV, scale_down=16 if model == "sd15" else 64, latent_hw=(8, 8), head_div=hd), full


@pytest.mark.parametrize("model,dtype,atype", E2E50, ids=lambda v: str(v).split(".")[-1])
def test_interpolate_single_50_steps_vs_oracle_loop(model, dtype, atype):
    hip, full = _e2e50_denoiser(model, dtype)
    cls = InterpolationStableDiffusionXLPipeline if model == "sdxl" else InterpolationStableDiffusionPipeline
    g = torch.Generator().manual_seed(50)
    l0, l1 = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    rd = lambda t: tuple(e.to(dtype).float() for e in t)     # noqa: E731
    es, ee = rd(_embs(g, hip.stack.cross_dim, model == "sdxl")), rd(_embs(g, hip.stack.cross_dim, model == "sdxl"))
    kw = dict(num_inference_steps=50, warmup_ratio=0.5, guidance_scale=5.0, output_type="latent")
    pipe = cls(hip, DDIMSchedulerLite())
    pipe.load_aid(t=0.5, is_fused=True, atype=atype)
    out = pipe.interpolate_single(0.35, latent_start=l0, latent_end=l1, embeds_start=es, embeds_end=ee, **kw)["images"]
    ora = cls(OracleDenoiser(hip), DDIMSchedulerLite())
    ref = ora.interpolate_single(0.35, latent_start=l0.to(dtype).double(), latent_end=l1.to(dtype).double(),
                                 embeds_start=tuple(e.double() for e in es), embeds_end=tuple(e.double() for e in ee), **kw)["images"]
    err = rel_l2(to_np64(out), ref.numpy())
    _record(f"e2e50_single_{model}_{str(dtype).split('.')[-1]}" + ("_fullwidth" if full else ""), dict(steps=50, rel_l2=err))
    assert out.shape == (3, 4, 8, 8) and torch.isfinite(out).all() and err < E2E50_BOUND[(model, dtype)], err


@pytest.mark.parametrize("model,dtype,atype", E2E50_N7, ids=lambda v: str(v).split(".")[-1])
def test_seven_frame_interpolate_50_steps_vs_oracle_loop(model, dtype, atype):
    hip, full = _e2e50_denoiser(model, dtype)
    cls = InterpolationStableDiffusionXLPipeline if model == "sdxl" else InterpolationStableDiffusionPipeline
    g = torch.Generator().manual_seed(51)
    l0, l1 = torch.randn(1, 4, 8, 8, generator=g).to(dtype), torch.randn(1, 4, 8, 8, generator=g).to(dtype)
    rd = lambda t: tuple(e.to(dtype).float() for e in t)     # noqa: E731
    xl = model == "sdxl"
    es, ee, eg = rd(_embs(g, hip.stack.cross_dim, xl)), rd(_embs(g, hip.stack.cross_dim, xl)), rd(_embs(g, hip.stack.cross_dim, xl))
    kw = dict(size=7, num_inference_steps=50, warmup_ratio=0.5, early=atype, guidance_scale=5.0, output_type="latent")
    pipe = cls(hip, DDIMSchedulerLite())
    out = pipe.interpolate(l0, l1, embeds_start=es, embeds_end=ee, embeds_guide=eg, **kw)        # PAID: guide prompt, batched CFG
    ora = cls(OracleDenoiser(hip), DDIMSchedulerLite())
    dbl = lambda t: tuple(e.double() for e in t)               # noqa: E731
    ref = ora.interpolate(l0.double(), l1.double(), embeds_start=dbl(es), embeds_end=dbl(ee), embeds_guide=dbl(eg), **kw)
    err = rel_l2(to_np64(out), ref.numpy())
    per_frame = [rel_l2(to_np64(out[f]), ref[f].numpy()) for f in range(7)]
    _record(f"e2e50_n7_{model}_{str(dtype).split('.')[-1]}" + ("_fullwidth" if full else ""),
            dict(steps=50, rel_l2=err, per_frame=per_frame))
    assert out.shape == (7, 4, 8, 8) and torch.isfinite(out).all() and err < E2E50_BOUND[(model, dtype)], (err, per_frame)


def test_interpolate_single_50_steps_sdxl_fp16_at_full_width():
    """One FULL-WIDTH 50-step case in the suite the driver runs (VERDICT r4 next #5a; the cases above divide heads and widths together
    unless AID_E2E_FULL_WIDTH=1): SDXL, fp16 storage, all 140 attention layers at their real widths (C = 640 / 1280, 10 / 20 heads),
    interpolate_single, HIP against the fp64 oracle loop."""
    dtype, model = torch.float16, "sdxl"
    hip = StackDenoiser(model, dtype=dtype, device=DEV, scale_down=64, latent_hw=(8, 8), head_div=1)
    g = torch.Generator().manual_seed(50)
    l0, l1 = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    rd = lambda t: tuple(e.to(dtype).float() for e in t)     # noqa: E731
    es, ee = rd(_embs(g, hip.stack.cross_dim, True)), rd(_embs(g, hip.stack.cross_dim, True))
    kw = dict(num_inference_steps=50, warmup_ratio=0.5, guidance_scale=5.0, output_type="latent")
    pipe = InterpolationStableDiffusionXLPipeline(hip, DDIMSchedulerLite())
    pipe.load_aid(t=0.5, is_fused=True, atype="fused_outer")
    out = pipe.interpolate_single(0.35, latent_start=l0, latent_end=l1, embeds_start=es, embeds_end=ee, **kw)["images"]
    ora = InterpolationStableDiffusionXLPipeline(OracleDenoiser(hip), DDIMSchedulerLite())
    ref = ora.interpolate_single(0.35, latent_start=l0.to(dtype).double(), latent_end=l1.to(dtype).double(),
                                 embeds_start=tuple(e.double() for e in es), embeds_end=tuple(e.double() for e in ee), **kw)["images"]
    err = rel_l2(to_np64(out), ref.numpy())
    _record("e2e50_single_sdxl_float16_fullwidth_suite", dict(steps=50, rel_l2=err))
    assert out.shape == (3, 4, 8, 8) and torch.isfinite(out).all() and err < E2E50_BOUND[(model, dtype)], err


@pytest.mark.parametrize("model,dtype,atype", [("sd15", torch.float16, "fused_inner"), ("sdxl", torch.float16, "fused_outer")],
                         ids=lambda v: str(v).split(".")[-1])
def test_fp16_storage_floor_of_the_50_step_loop(model, dtype, atype):
    """north_star's "latents within 1e-3 rel-L2" against what fp16 STORAGE alone costs (VERDICT r4 next #5d).  The same 50-step loop
    in fp64 ARITHMETIC, rounded to fp16 only where a loop that keeps its tensors in fp16 stores them — q / k / v, the attention
    output, the layer output and the residual stream; the latents and embeddings handed to the denoiser, the predicted noise, the
    guided noise and the latents the scheduler returns (StorageScheduler) — with un-rounded probabilities and exact LayerNorm, i.e.
    better than any kernel can be, against the un-rounded fp64 loop.  That is the floor of the storage type: measured 1.25e-3 (SD1.5,
    32 layers) and 1.73e-3 (SDXL, 140 layers), ABOVE 1e-3 — which is why the fp16 rows of E2E50_BOUND are not 1e-3.  (Rounding only
    inside the attention layers gives 2.3e-4 / 1.06e-3: for SD1.5 most of the floor is the fp16 latents / noise of the loop around the
    layers, which the reference's fp16 pipeline stores the same way.)  The HIP path measures 1.32e-3 / 1.80e-3: within 6 % of the floor,
    asserted at 1.15x — the kernels' own roundings (fp16 probabilities, the pre-scaled q, folded LayerNorm weights) add almost nothing,
    so the lever VERDICT r4 names (row sums from the un-rounded probabilities) has nothing left to recover."""
    hip, full = _e2e50_denoiser(model, dtype)
    cls = InterpolationStableDiffusionXLPipeline if model == "sdxl" else InterpolationStableDiffusionPipeline
    g = torch.Generator().manual_seed(50)
    l0, l1 = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    rd = lambda t: tuple(e.to(dtype).float() for e in t)     # noqa: E731
    es, ee = rd(_embs(g, hip.stack.cross_dim, model == "sdxl")), rd(_embs(g, hip.stack.cross_dim, model == "sdxl"))
    kw = dict(num_inference_steps=50, warmup_ratio=0.5, guidance_scale=5.0, output_type="latent")
    pipe = cls(hip, DDIMSchedulerLite())
    pipe.load_aid(t=0.5, is_fused=True, atype=atype)
    out = pipe.interpolate_single(0.35, latent_start=l0, latent_end=l1, embeds_start=es, embeds_end=ee, **kw)["images"]
    args = dict(latent_start=l0.to(dtype).double(), latent_end=l1.to(dtype).double(), embeds_start=tuple(e.double() for e in es),
                embeds_end=tuple(e.double() for e in ee), **kw)
    ref = cls(OracleDenoiser(hip), DDIMSchedulerLite()).interpolate_single(0.35, **args)["images"].numpy()
    floor = cls(OracleDenoiser(hip, storage=dtype), StorageScheduler(dtype)).interpolate_single(0.35, **args)["images"].numpy()
    e_hip, e_floor = rel_l2(to_np64(out), ref), rel_l2(floor, ref)
    _record(f"storage_floor_{model}_float16", dict(steps=50, hip_rel_l2=e_hip, storage_only_rel_l2=e_floor))
    assert e_floor > 1e-3 and e_hip < 1.15 * e_floor, (e_hip, e_floor)
