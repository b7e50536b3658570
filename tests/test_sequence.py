"""Host harness rows (SURVEY.md §2 ★ host harness): batch assembly of interpolate_single / interpolate,
the warm-up toggling rule and the CFG combine of loop.py — CPU only, the UNet is a recording stub."""
import numpy as np
import pytest
import torch

import cases as C
import aid_amd
from aid_amd import sequence as S
from aid_amd.loop import AidDenoiseLoop, install_sequence_processors, set_aid_active

MISC = C.load_fixture("misc_goldens.npz")


def _embs(seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(1, 7, 12, generator=g) for _ in range(6)]


def test_prepare_sequence_matches_reference_helpers():
    v0, v1 = torch.from_numpy(MISC["slerp_v0"]), torch.from_numpy(MISC["slerp_v1"])
    e0, e1 = torch.from_numpy(MISC["emb0"]), torch.from_numpy(MISC["emb1"])
    b = S.prepare_sequence(v0, v1, e0, e1, e0 * 0, e1 * 0, size=4, num_inference_steps=25)
    np.testing.assert_allclose(b.latents.numpy(), MISC["spherical_size4"], atol=2e-6)       # uniform-t slerp
    b5 = S.prepare_sequence(v0, v1, e0, e1, e0, e1, size=5, num_inference_steps=25)
    np.testing.assert_allclose(b5.cond.numpy(), MISC["linear_size5"], atol=1e-6)            # uniform-t lerp of embeds
    ref = MISC["beta_5_25_25"].copy(); ref[0], ref[-1] = 0, 1
    np.testing.assert_array_equal(b5.coef.numpy(), ref)                                     # Beta-PPF coefficients
    assert b5.ctx_index.tolist() == [0, 1, 2, 3, 4] and b5.n_distinct_ctx == 5


def test_prepare_sequence_with_guide_prompt_shares_interior_context():
    es, ee, us, ue, g, ug = _embs()
    lat0, lat1 = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    b = S.prepare_sequence(lat0, lat1, es, ee, us, ue, size=7, guide_emb=g, uncond_guide=ug, num_inference_steps=50)
    assert b.cond.shape == (7, 7, 12)
    assert torch.equal(b.cond[0], es[0]) and torch.equal(b.cond[-1], ee[0])
    assert all(torch.equal(b.cond[i], g[0]) for i in range(1, 6))
    assert all(torch.equal(b.uncond[i], ug[0]) for i in range(1, 6))
    assert b.ctx_index.tolist() == [0, 1, 1, 1, 1, 1, 2] and b.n_distinct_ctx == 3
    assert b.coef[0] == 0 and b.coef[-1] == 1 and abs(float(b.coef[3]) - 0.5) < 1e-6
    cond3, unc3, idx = S.distinct_contexts(b)
    assert idx == [0, 1, 1, 1, 1, 1, 2] and cond3.shape == (3, 7, 12)
    assert torch.equal(cond3[idx], b.cond) and torch.equal(unc3[idx], b.uncond)


def test_shared_context_map_validation():
    from aid_amd.processors import _shared_context
    ctx3 = torch.randn(3, 5, 8)
    full = ctx3[[0, 1, 1, 2]]
    c, m, idx = _shared_context({}, [0, 1, 1, 2], ctx3, 4)
    assert torch.equal(c, ctx3) and m.tolist() == [0, 1, 1, 2] and m.dtype == torch.int32 and idx == [0, 1, 1, 2]
    c2, _, _ = _shared_context({}, torch.tensor([0, 1, 1, 2]), full, 4)          # repeated rows in: de-duplicated
    assert torch.equal(c2, ctx3)
    for bad, ctx, n in (([0, 1, 1], ctx3, 4), ([0, 2, 2, 3], ctx3, 4), ([0, 1, 1, 2], ctx3[:2], 4), ([-1, 0, 0, 1], ctx3, 4)):
        with pytest.raises(RuntimeError):
            _shared_context({}, bad, ctx, n)


def test_prepare_single_batch3_layout():
    es, ee, us, ue, g, ug = _embs(1)
    lat0, lat1 = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    b = S.prepare_single(0.3, lat0, lat1, es, ee, us, ue, init="linear")
    assert b.latents.shape[0] == 3 and torch.equal(b.latents[0:1], lat0) and torch.equal(b.latents[2:3], lat1)
    torch.testing.assert_close(b.latents[1:2], aid_amd.slerp(lat0, lat1, 0.3))
    torch.testing.assert_close(b.cond[1:2], torch.lerp(es, ee, 0.3))
    torch.testing.assert_close(b.uncond[1:2], torch.lerp(us, ue, 0.3))
    assert b.coef.tolist() == [0.0, 0.30000001192092896, 1.0]
    bs = S.prepare_single(0.3, lat0, lat1, es, ee, us, ue, init="slerp")
    torch.testing.assert_close(bs.cond[1:2], aid_amd.slerp(es, ee, 0.3))
    bg = S.prepare_single(0.3, lat0, lat1, es, ee, us, ue, guide_emb=g, uncond_guide=ug)
    assert torch.equal(bg.cond[1:2], g) and bg.ctx_index.tolist() == [0, 1, 2]


class _RecordingUNet(torch.nn.Module):
    """UNet-shaped stub: records (AID active?, plain_tail, batch) of every call; output = sample * marker."""
    def __init__(self):
        super().__init__()
        self.inner = aid_amd.AttnStackUNet("sd15", dtype=torch.float32, scale_down=64, channel_div=8)
        self.calls = []
        self.ctx_indices = []

    @property
    def attn_processors(self):
        return self.inner.attn_processors

    def set_attn_processor(self, p):
        self.inner.set_attn_processor(p)

    def forward(self, sample, ctx):
        p = next(iter(self.attn_processors.values()))
        self.calls.append((p.activated, p.plain_tail, sample.shape[0], ctx.shape[0]))
        self.ctx_indices.append(p.ctx_index)
        return sample * (2.0 if p.activated else 1.0) + ctx.mean()


def test_loop_toggling_follows_root_pipeline_rule():
    """AID on for the conditional pass of steps i < int(T * warmup_ratio) (0-based), off otherwise and always
    off for the unconditional pass (pipeline_interpolated_sd.py:1831, 1845-1848, 1870)."""
    unet = _RecordingUNet()
    install_sequence_processors(unet, 5, early="fused_outer", num_inference_steps=7)
    x, cond, uncond = torch.ones(5, 2), torch.zeros(5, 3), torch.ones(5, 3)
    loop = AidDenoiseLoop(unet, x, cond, uncond, num_inference_steps=7, warmup_ratio=0.5, guidance_scale=3.0,
                          use_graphs=False)
    assert loop.warmup_steps == 3
    for i in range(7):
        out = loop.step(i)
        text = x * (2.0 if i < 3 else 1.0) + 0.0
        unc = x * 1.0 + 1.0
        torch.testing.assert_close(out, unc + 3.0 * (text - unc))                  # CFG combine (:1892)
    assert unet.calls == [(i < 3, 0, 5, 5) if k == 0 else (False, 0, 5, 5) for i in range(7) for k in (0, 1)]
    # batched CFG: ONE call per step over [cond ; uncond]; the uncond half rides as plain frames
    unet.calls.clear()
    loop2 = AidDenoiseLoop(unet, x, cond, uncond, num_inference_steps=7, use_graphs=False, batched_cfg=True)
    for i in range(7):
        loop2.step(i)
    assert unet.calls == [(i < 3, 5 if i < 3 else 0, 10, 10) for i in range(7)]
    set_aid_active(unet, True)
    assert all(p.activated and p.plain_tail == 0 for p in unet.attn_processors.values())
    # shared contexts: the loop hands the DISTINCT contexts to the UNet and the frame -> row map to every processor
    unet.calls.clear(); unet.ctx_indices.clear()
    idx = [0, 1, 1, 1, 2]
    loop3 = AidDenoiseLoop(unet, x, cond[:3], uncond[:3], num_inference_steps=7, use_graphs=False, batched_cfg=True,
                           ctx_index=idx)
    loop3.step(0)
    assert unet.calls == [(True, 5, 10, 6)] and unet.ctx_indices == [idx + [3, 4, 4, 4, 5]]
    loop4 = AidDenoiseLoop(unet, x, cond[:3], uncond[:3], num_inference_steps=7, use_graphs=False, ctx_index=idx)
    loop4.step(6)
    assert unet.calls[1:] == [(False, 0, 5, 3)] * 2 and unet.ctx_indices[1:] == [idx, idx]
    assert all(p.original_attn.ctx_index == idx for p in unet.attn_processors.values())
    with pytest.raises(ValueError):
        AidDenoiseLoop(unet, x, cond[:3], uncond[:3], use_graphs=False)            # 3 contexts for 5 frames, no map
    with pytest.raises(ValueError):
        AidDenoiseLoop(unet, x, cond[:3], uncond[:3], use_graphs=False, ctx_index=[0, 1, 2])
