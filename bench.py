#!/usr/bin/env python3
"""Headline benchmark: interpolation-frames/sec (50-step) of the AID attention stack on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model sd15|sdxl] [--early fused_inner|...]

One "step" = one denoising step of the interpolation run = the ordered attention calls of one UNet
forward (SD1.5 32 calls, SDXL 140 calls) executed twice: the conditional pass (AID
processors active for steps i < int(K * warmup_ratio), plain attention afterwards) and the
unconditional pass (plain), exactly as the reference loop toggles them
(pipeline_interpolated_sd.py:1831-1870).  Weights / hidden states / text context are synthetic
(seed 1002) and resident in HBM before the timed region.

Default workload = BASELINE.json configs[1]: SD1.5 512x512, 7-frame fused-inner AID, 50 steps, fp16
on 1 GPU.  ``--model sdxl`` = configs[2] (SDXL 1024x1024, 7-frame fused-outer, bf16).
For N > 1 (launched under torch.distributed.run) ONE sequence of 7*N frames is sharded by frame
with replicated end points (dist.py): zero per-layer communication, one broadcast of the conditioning
before and one all_gather of the owned outputs after the K steps (inside the timed region).

Rank 0 prints ONE JSON line (see the task contract); N=1 additionally measures
  roofline     — HIP-event timing of every kernel launch of one AID step + one plain step
                 (aid_profile_begin/end in the C ABI) -> achieved algorithmic TFLOP/s of the dominant kernel
  cpu_baseline — the numpy oracle (oracle/aid_oracle.py, "port") timed on the host cores on a bounded
                 sample and extrapolated to the same 50-step unit.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16 / bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

DEFAULT_EARLY = {"sd15": "fused_inner", "sdxl": "fused_outer"}
DTYPES = {"sd15": torch.float16, "sdxl": torch.bfloat16}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", default="sd15", choices=["sd15", "sdxl"])
    ap.add_argument("--early", default=None, help="pure_inner|fused_inner|pure_outer|fused_outer")
    ap.add_argument("--frames-per-gpu", type=int, default=7)
    ap.add_argument("--warmup-ratio", type=float, default=0.5)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--separate-passes", action="store_true",
                    help="run the cond and the uncond pass as two UNet calls like the reference loop "
                         "(default: one call over [cond ; uncond], same work, same results)")
    ap.add_argument("--guide-prompt", default="auto", choices=["auto", "on", "off"],
                    help="PAID: interior frames share the guide prompt's text context (3 distinct contexts); "
                         "auto = on for sdxl (BASELINE configs[2]), off for sd15 (configs[1]: per-frame embeddings)")
    ap.add_argument("--sublayers", default="off", choices=["off", "steps", "fused"],
                    help="widened workload (SURVEY.md 8f.2): every attention call with the LayerNorm in front of it and the "
                         "residual add behind it, as three steps (torch LayerNorm / call / torch add) or as ONE library call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def make_inputs(unet, n_frames, dtype, device, seed=1002, n_ctx=None):
    """Hidden state per resolution level ~ N(0,1) (post-LayerNorm scale) and ``n_ctx`` text contexts ~ N(0,1)
    (one per frame, or the 3 distinct ones [start, guide, end] of a PAID run)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    n_ctx = n_frames if n_ctx is None else n_ctx
    xs = {}
    for (s, c) in unet.level_shapes():
        xs[(s, c)] = torch.randn(n_frames, s, c, generator=g).to(dtype).to(device)
    cond = torch.randn(n_ctx, unet.text_len, unet.cross_dim, generator=g).to(dtype).to(device)
    uncond = torch.randn(n_ctx, unet.text_len, unet.cross_dim, generator=g).to(dtype).to(device)
    return xs, cond, uncond


def recorded_traffic(model, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC collection (profiles/r01_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same command, gfx950 x2 correction
    on FETCH_SIZE applied).  PMC counters cannot be collected from inside this process; null if absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["models"][model][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def roofline_pass(loop, aid_amd):
    """HIP-event timing (on the launch stream) of every kernel of one AID step + one plain step."""
    lib = aid_amd._lib.load()
    was = loop.use_graphs
    loop.use_graphs = False
    loop.step(0); loop.step(loop.num_inference_steps - 1)        # eager warm-up
    torch.cuda.synchronize()
    lib.aid_profile_begin()
    loop.step(0)                                                 # AID step
    loop.step(loop.num_inference_steps - 1)                      # plain step
    buf = (aid_amd._lib.AidProfileEntry * 4096)()
    n = lib.aid_profile_end(buf, 4096)
    loop.use_graphs = was
    if n < 0:
        raise RuntimeError(lib.aid_strerror(n).decode())
    agg = {}
    for e in buf[:n]:
        a = agg.setdefault(e.kernel.decode(), dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        a["ms"] += e.ms; a["flops"] += e.flops; a["bytes"] += e.bytes; a["launches"] += 1
    return agg


def cpu_baseline(model, n_frames, early, steps, warmup_ratio):
    """Oracle ("port") on the host cores: one transformer block (self + cross call) per resolution level,
    AID mode and plain mode, on the 3-frame sub-batch [first, middle, last] of the same synthetic inputs,
    scaled by n_frames/3 and by the block counts to one UNet pass, then to the 50-step unit."""
    import numpy as np
    from oracle import aid_oracle as O
    from aid_amd.attn_shim import MODEL_SPECS
    spec = MODEL_SPECS[model]
    rs = np.random.RandomState(1002)
    mode = "outer" if early.endswith("outer") else "inner"
    fused = early.startswith("fused")
    coef = O.beta_coefs(n_frames, steps, steps)
    sub = np.asarray([0.0, float(coef[n_frames // 2]), 1.0], dtype=np.float32)
    levels = {}
    for loc, nblk, s, c, h in spec["layers"]:
        levels.setdefault((s, c, h), 0)
        levels[(s, c, h)] += nblk
    t_aid = t_plain = 0.0
    t0_all = time.time()
    for (s, c, h), nblk in levels.items():
        cc = spec["cross_dim"]
        x = rs.standard_normal((3, s, c)).astype(np.float32)
        ctx = rs.standard_normal((3, spec["text_len"], cc)).astype(np.float32)
        ws = O.AttnWeights(*(rs.standard_normal(sh).astype(np.float32) / np.sqrt(sh[-1]) for sh in
                             ((c, c), (c, c), (c, c), (c, c))), rs.standard_normal(c).astype(np.float32) * .01, h)
        wx = O.AttnWeights(ws.wq, rs.standard_normal((c, cc)).astype(np.float32) / np.sqrt(cc),
                           rs.standard_normal((c, cc)).astype(np.float32) / np.sqrt(cc), ws.wo, ws.bo, h)
        fn = O.outer_attention if mode == "outer" else O.inner_attention
        t0 = time.time(); fn(x, None, ws, sub, fused); fn(x, ctx, wx, sub, fused); ta = time.time() - t0
        t0 = time.time(); O.plain_attention(x, None, ws); O.plain_attention(x, ctx, wx); tp = time.time() - t0
        t_aid += nblk * ta * n_frames / 3.0
        t_plain += nblk * tp * n_frames / 3.0
    n_aid = int(steps * warmup_ratio)
    total = n_aid * (t_aid + t_plain) + (steps - n_aid) * 2 * t_plain
    scale50 = 50.0 / steps
    cores = os.cpu_count() or 1
    return dict(value=n_frames / (total * scale50), unit="interpolation-frames/sec (50-step)", cores=cores,
                kind="port",
                sample=(f"numpy fp32 oracle, 1 transformer block (self+cross call) per resolution level in {early} and "
                        f"plain mode on the 3-frame sub-batch [first, middle, last]; scaled x{n_frames}/3 frames, x blocks per "
                        f"level, x({n_aid} AID + {2 * steps - n_aid} plain passes); measured {time.time() - t0_all:.1f} s of CPU work"))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # development aid: AID_BENCH_ONE_DEVICE=1 runs all ranks on cuda:0 with gloo collectives, so the N>1 code
    # path (sharding, broadcast, all_gather, max-over-ranks) can be exercised on a 1-GPU box
    one_dev = os.environ.get("AID_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_dev else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    import aid_amd
    from aid_amd import dist as adist
    from aid_amd.loop import AidDenoiseLoop, install_sequence_processors
    aid_amd._lib.load()                     # fail loudly before anything else if the HIP library is missing

    model = args.model
    early = args.early or DEFAULT_EARLY[model]
    dtype = DTYPES[model]
    n_total = args.frames_per_gpu * world
    shard = adist.frame_shard(n_total, world, rank)
    steps = args.steps

    unet = aid_amd.AttnStackUNet(model, dtype=dtype, device=device)
    unet.sublayers = args.sublayers
    coef = aid_amd.generate_beta_tensor(n_total, steps, steps)
    coef[0], coef[-1] = 0, 1
    guided = args.guide_prompt == "on" or (args.guide_prompt == "auto" and model == "sdxl")
    # PAID (gradio ...stable_diffusion.py:221-229): contexts [start, guide x (N-2), end] -> 3 distinct rows
    global_ctx = ([0] + [1] * (n_total - 2) + [2]) if guided else list(range(n_total))
    xs, cond, uncond = make_inputs(unet, n_total, dtype, device, n_ctx=3 if guided else None)
    if world > 1:                            # conditioning comes from rank 0 (north_star: RCCL broadcast)
        named = {f"x{s}_{c}": t for (s, c), t in xs.items()}
        named.update(cond=cond, uncond=uncond)
        adist.broadcast_conditioning(named, src=0)
    xs = {k: adist.shard_rows(v, shard) for k, v in xs.items()}
    rows = [global_ctx[f] for f in shard.index]              # context row of every local frame ...
    used = sorted(set(rows), key=rows.index)                 # ... renumbered in order of first local use
    ctx_index = [used.index(r) for r in rows] if guided else None
    sel = torch.tensor(used, device=device)
    cond, uncond = cond.index_select(0, sel).contiguous(), uncond.index_select(0, sel).contiguous()
    local_coef = coef[list(shard.index)]
    install_sequence_processors(unet, shard.n_local, early=early, num_inference_steps=steps, coef=local_coef)

    loop = AidDenoiseLoop(unet, xs, cond, uncond, num_inference_steps=steps, warmup_ratio=args.warmup_ratio,
                          use_graphs=not args.no_graph, batched_cfg=not args.separate_passes, ctx_index=ctx_index)
    gather_key = unet.level_shapes()[-1]

    def run_steps(idx):
        out = None
        for i in idx:
            out = loop.step(i)
        return out

    # untimed warm-up: alternate AID / plain steps so every graph is captured and warm
    warm_idx = [0 if (j % 2 == 0) else steps - 1 for j in range(max(args.warmup, 2))]
    run_steps(warm_idx)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_steps(range(steps))
    final = adist.gather_owned(out[gather_key], shard)          # all_gather of the owned frames' outputs
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if one_dev else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert final.shape[0] == n_total and torch.isfinite(final.float()).all()

    ms_per_step = elapsed * 1000.0 / steps
    value = n_total / (ms_per_step * 50.0 / 1000.0)

    result = {
        "metric": "interpolation-frames/sec (50-step)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": len(warm_idx),
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if dtype == torch.float16 else "bf16", "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[1]: SD1.5 512x512 attention stack (32 attention calls / UNet pass)" if model == "sd15"
                         else "BASELINE configs[2]: SDXL-base 1024x1024 attention stack (140 attention calls / UNet pass)"),
            "frames": n_total, "frames_per_gpu": args.frames_per_gpu, "local_batch": shard.n_local,
            "early": early, "late": "plain", "warmup_ratio": args.warmup_ratio,
            "aid_steps": loop.warmup_steps,
            "passes_per_step": ("cond + uncond (CFG), two UNet calls" if args.separate_passes
                                else "cond + uncond (CFG) batched in one UNet call [cond ; uncond]"),
            "contexts": ("PAID guide prompt: interior frames share one text context (3 distinct per pass), keys/values "
                         "projected once per distinct context" if guided else "one text context per frame"),
            "sublayers": {"off": "attention calls only (the BASELINE metric)",
                          "steps": "LayerNorm + call + residual add per layer, three steps (torch LayerNorm / add)",
                          "fused": "LayerNorm + call + residual add per layer in one library call"}[args.sublayers],
            "coef": f"BetaPPF(alpha=beta={steps})", "hipgraph": not args.no_graph,
            "parallelism": f"frame-shard x{world} (replicated end points, no per-layer collective)",
        },
    }

    if rank == 0 and world == 1 and not args.no_roofline:
        agg = roofline_pass(loop, aid_amd)
        attn = {k: v for k, v in agg.items()}
        dom = max(attn, key=lambda k: attn[k]["ms"])
        d = attn[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        result["roofline"] = {
            "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
            "traffic": recorded_traffic(model, dom), "kernel": dom, "launches": d["launches"],
            "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
            "avg_launch_us": d["ms"] * 1e3 / d["launches"], "avg_launch_gflop": d["flops"] / d["launches"] / 1e9,
            "note": "algorithmic flops (SURVEY.md §8d) / HIP-event time on the launch stream, 1 AID step + 1 plain step",
            "kernels": {k: {"ms": round(v["ms"], 4), "launches": v["launches"],
                            "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                            "min_bytes_gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in sorted(agg.items())},
        }
        tot_ms = sum(v["ms"] for v in agg.values())
        tot_fl = sum(v["flops"] for v in agg.values())
        result["roofline"]["stack_tflops"] = tot_fl / (tot_ms * 1e-3) / 1e12
        result["roofline"]["kernel_ms_per_2steps"] = tot_ms
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(model, n_total, early, steps, args.warmup_ratio)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
