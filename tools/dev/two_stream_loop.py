"""Development: the two passes of a step on two streams (AidDenoiseLoop(concurrent_cfg=True)) against one stream at the FULL SDXL size,
30 repetitions per configuration, eager and graph: counts outputs that differ.  usage: python tools/dev/two_stream_loop.py off|fused [ip]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import aid_amd
from aid_amd.loop import AidDenoiseLoop, install_sequence_processors
DEV = torch.device("cuda:0")
dtype, n, steps = torch.bfloat16, 7, 8
sub = sys.argv[1] if len(sys.argv) > 1 else "off"
ip = len(sys.argv) > 2 and sys.argv[2] == "ip"
unet = aid_amd.AttnStackUNet("sdxl", dtype=dtype, device=DEV, scale_down=1)
unet.sublayers = sub
g = torch.Generator().manual_seed(3)
xs = {lv: torch.randn(n, lv[0], lv[1], generator=g).to(dtype).to(DEV) for lv in unet.level_shapes()}
cond = torch.randn(n, unet.text_len, unet.cross_dim, generator=g).to(dtype).to(DEV)
unc = torch.randn(n, unet.text_len, unet.cross_dim, generator=g).to(dtype).to(DEV)
coef = aid_amd.generate_beta_tensor(n, steps, steps); coef[0], coef[-1] = 0, 1
if ip:
    unet.load_ip_adapter(num_tokens=4, scale=0.6)
    aid_amd.load_aid_ip_adapter(unet, t=None, size=n, is_fused=True, early="fused_outer", alpha=steps, beta=steps)
    for p in unet.attn_processors.values():
        p.coef = coef.detach().to(torch.float32).cpu().clone()
    rep3 = lambda t: t.repeat_interleave(3, dim=0).contiguous()
    pos = torch.randn(n, 1, 4, unet.cross_dim, generator=g).to(dtype).to(DEV)
    neg = torch.randn(n, 1, 4, unet.cross_dim, generator=g).to(dtype).to(DEV)
    cond, unc = (cond, [rep3(pos)]), (unc, [rep3(neg)])
else:
    install_sequence_processors(unet, n, early="fused_outer", num_inference_steps=steps, coef=coef)
for graphs in (False, True):
    ref_loop = AidDenoiseLoop(unet, xs, cond, unc, num_inference_steps=steps, use_graphs=graphs, concurrent_cfg=False)
    ref = {}
    for i in (0, steps - 1):
        o = ref_loop.step(i); torch.cuda.synchronize(); ref[i] = {k: v.clone() for k, v in o.items()}
    loop = AidDenoiseLoop(unet, xs, cond, unc, num_inference_steps=steps, use_graphs=graphs, concurrent_cfg=True)
    bad = 0
    for rep in range(30):
        for i in (0, steps - 1):
            o = loop.step(i); torch.cuda.synchronize()
            for k in o:
                if not torch.equal(o[k], ref[i][k]):
                    bad += 1
                    d = (o[k].float() - ref[i][k].float()).abs()
                    if bad <= 4: print("  mismatch", "graphs" if graphs else "eager", "rep", rep, "step", i, "level", k, "elements", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()))
    print(sub, "ip" if ip else "text", "graphs" if graphs else "eager", "mismatching (step, level) outputs:", bad, "of", 30 * 2 * len(ref[0]))
