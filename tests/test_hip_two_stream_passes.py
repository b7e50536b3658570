"""GPU: the two passes of a step on two STREAMS (AidDenoiseLoop(concurrent_cfg=True)) — what the IP-Adapter workload does, whose cond and
uncond UNet calls carry different image embeddings (pipeline_interpolated_sd.py:1763-1802) and cannot share one batched call — give the
SAME tensors, bit for bit, as the two calls back to back on one stream, eagerly and replayed from one hipGraph with the fork / join inside.
(tools/dev/two_stream_loop.py repeats the comparison at the full SDXL size: 0 mismatches in 720 outputs.  The pipelines do NOT fork
their passes: the third-party ops of a UNet — the stand-in denoiser's rocBLAS matmuls already — are not stream-safe on this stack,
profiles/r04_bench_runs.txt.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

import aid_amd  # noqa: E402
from aid_amd.loop import AidDenoiseLoop, install_sequence_processors  # noqa: E402

DEV = torch.device("cuda:0")


def _inputs(unet, n, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    xs = {lv: torch.randn(n, lv[0], lv[1], generator=g).to(dtype).to(DEV) for lv in unet.level_shapes()}
    cond = torch.randn(n, unet.text_len, unet.cross_dim, generator=g).to(dtype).to(DEV)
    unc = torch.randn(n, unet.text_len, unet.cross_dim, generator=g).to(dtype).to(DEV)
    return xs, cond, unc, g


@pytest.mark.parametrize("graphs", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("ip", [False, True], ids=["text", "ip"])
def test_two_stream_passes_equal_serial_passes(ip, graphs):
    dtype, n, steps = torch.bfloat16, 5, 4
    unet = aid_amd.AttnStackUNet("sdxl", dtype=dtype, device=DEV, scale_down=8)
    xs, cond, unc, g = _inputs(unet, n, dtype, 31 + ip)
    coef = aid_amd.generate_beta_tensor(n, steps, steps)
    coef[0], coef[-1] = 0, 1
    if ip:
        unet.load_ip_adapter(num_tokens=4, scale=0.6)
        aid_amd.load_aid_ip_adapter(unet, t=None, size=n, is_fused=True, early="fused_outer", alpha=steps, beta=steps)
        for p in unet.attn_processors.values():
            p.coef = coef.detach().to(torch.float32).cpu().clone()
        rep3 = lambda t: t.repeat_interleave(3, dim=0).contiguous()      # noqa: E731
        pos = torch.randn(n, 1, 4, unet.cross_dim, generator=g).to(dtype).to(DEV)
        neg = torch.randn(n, 1, 4, unet.cross_dim, generator=g).to(dtype).to(DEV)
        cond, unc = (cond, [rep3(pos)]), (unc, [rep3(neg)])
    else:
        install_sequence_processors(unet, n, early="fused_outer", num_inference_steps=steps, coef=coef)
    outs = []
    for conc in (False, True):
        loop = AidDenoiseLoop(unet, xs, cond, unc, num_inference_steps=steps, use_graphs=graphs, concurrent_cfg=conc)
        assert loop.concurrent_cfg == conc and not loop.batched_cfg
        res = []
        for i in (0, 1, steps - 1, 0):                                  # AID step, AID step, plain step, AID again (graph replays)
            out = loop.step(i)
            torch.cuda.synchronize()
            res.append({k: v.clone() for k, v in out.items()})
        outs.append(res)
    for a, b in zip(*outs):
        for k in a:
            assert torch.isfinite(a[k].float()).all() and torch.equal(a[k], b[k]), k
    assert not torch.equal(outs[0][0][unet.level_shapes()[0]], outs[0][2][unet.level_shapes()[0]])   # AID and plain steps differ


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_cu_share_hint_changes_the_engine_not_the_result(dtype, tuning):
    """CU_SHARE = n (n launch streams share the device) only moves the GEMM engine choice: the same bits come out of the projection
    shapes of a two-stream SDXL / SD1.5 step under every value, through whichever engine the hint selects."""
    from aid_amd import ops
    g = torch.Generator().manual_seed(17)
    picked = set()
    for m, n, k in ((7168, 1280, 1280), (28672, 640, 640), (28672, 320, 320), (3584, 1280, 1280), (8192, 1280, 2048)):
        x = (torch.randn(m, k, generator=g) * 0.5).to(dtype).to(DEV)
        w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype).to(DEV)
        b = torch.randn(n, generator=g).to(dtype).to(DEV)
        outs = []
        for share in (-1, 2, 4, 8):
            tuning("CU_SHARE", share)
            outs.append(ops.linear(x, w, b))
            picked.add((m, share, ops.last_gemm_variant()))
        torch.cuda.synchronize()
        ref = (x.float() @ w.float().T + b.float())
        assert float((outs[0].float() - ref).norm() / ref.norm()) < (1e-3 if dtype == torch.float16 else 8e-3)
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), (m, n, k)
    assert len({v for (m, s, v) in picked if m == 7168}) > 1, picked          # the hint did move the choice for the half-round launch
