#!/usr/bin/env python3
"""Headline benchmark: interpolation-frames/sec (50-step) of the AID attention stack on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sdxl|sd15|seq16|ip] [--early ...]

One "step" = one denoising step of the interpolation run = the ordered attention calls of one UNet
forward (SD1.5 32 calls, SDXL 140 calls) executed twice: the conditional pass (AID
processors active for steps i < int(K * warmup_ratio), plain attention afterwards) and the
unconditional pass (plain), exactly as the reference loop toggles them
(pipeline_interpolated_sd.py:1831-1870).  Weights / hidden states / text context are synthetic
(seed 1002) and resident in HBM before the timed region.

Workloads (BASELINE.json configs):
  sdxl   configs[2]  SDXL 1024x1024, 7-frame PAID (guide prompt) fused-outer, bf16          <- default at --gpus 1
  sd15   configs[1]  SD1.5 512x512, 7-frame fused-inner AID, fp16   (also reported under "also" by the default run)
  seq16  configs[3]  SDXL 1024x1024, ONE 16-frame sequence sharded by frame over the ranks  <- default at --gpus N > 1
  ip     configs[4]  SDXL + IP-Adapter image-conditioned morphing, 8-frame outer-IP, image-embed cross-attention
`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with N ranks
(one per GPU, backend nccl = RCCL); under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  The sequence is sharded by
frame with replicated end points (dist.py): zero per-layer communication, one broadcast of the conditioning before and
one all_gather of the owned outputs after the steps (inside the timed region).  `--weak` keeps 7 owned frames per GPU.

The timed region is EXACTLY K steps, repeated `repeats` times back to back when K steps take less than a second
(ms_per_step is the mean over K * repeats steps).  Rank 0 prints ONE JSON line; N = 1 additionally measures
  roofline     — HIP-event timing of every kernel launch of one AID step + one plain step (aid_profile_begin/end in the
                 C ABI): algorithmic AND executed TFLOP/s per kernel symbol, the dominant symbol reported on top
  cpu_baseline — the threaded CPU port of the path (oracle/aid_cpu_port.py, "port") timed on the host cores on a bounded
                 sample and extrapolated to the same 50-step unit.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16 / bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    #          stack   dtype   early          frames  guide  what
    "sdxl":  ("sdxl", "bf16", "fused_outer", 7, True,
              "BASELINE configs[2]: SDXL-base 1024x1024 attention stack (140 attention calls / UNet pass), 7-frame PAID"),
    "sd15":  ("sd15", "f16", "fused_inner", 7, False,
              "BASELINE configs[1]: SD1.5 512x512 attention stack (32 attention calls / UNet pass), 7-frame AID"),
    "seq16": ("sdxl", "bf16", "fused_outer", 16, True,
              "BASELINE configs[3]: SDXL 1024x1024, ONE 16-frame PAID sequence sharded by frame over the GPUs"),
    "ip":    ("sdxl", "bf16", "fused_outer", 8, False,
              "BASELINE configs[4]: SDXL + IP-Adapter image-conditioned morphing, 8-frame outer-IP, image-embed cross-attention"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="auto", choices=["auto"] + sorted(WORKLOADS))
    ap.add_argument("--model", default=None, choices=["sd15", "sdxl"], help="alias of --workload sd15 / sdxl")
    ap.add_argument("--early", default=None, help="pure_inner|fused_inner|pure_outer|fused_outer (ip: fused_outer|fused_inner|scale_control)")
    ap.add_argument("--frames", type=int, default=None, help="total frames of the sequence (default: the workload's)")
    ap.add_argument("--weak", action="store_true", help="N > 1: 7 owned frames per GPU (7 N frames) instead of configs[3]")
    ap.add_argument("--frames-per-gpu", type=int, default=7, help="--weak: owned frames per GPU")
    ap.add_argument("--warmup-ratio", type=float, default=0.5)
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the K steps until the timed region is this long")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--passes", default="auto", choices=["auto", "batched", "streams", "serial"],
                    help="the cond and the uncond pass of a step: ONE UNet call over [cond ; uncond] (batched; how stock diffusers "
                         "pipelines batch CFG) | two UNet calls on two streams, forked and joined inside one graph (streams; the library "
                         "is told per call that two launch streams share the device: cu_share = 2) | two UNet calls back to back like the reference "
                         "loop (serial).  All three give the same results.  auto: batched for the SDXL text workloads (streams is faster "
                         "there too, `also.sdxl_two_streams`, but halves every launch), streams for sd15 (5.71 -> 5.37 ms/step) and ip "
                         "(cannot batch: its passes carry different image embeddings)")
    ap.add_argument("--separate-passes", action="store_true", help="alias of --passes serial")
    ap.add_argument("--guide-prompt", default="auto", choices=["auto", "on", "off"],
                    help="PAID: interior frames share the guide prompt's text context (3 distinct contexts)")
    ap.add_argument("--ip-tokens", type=int, default=4, help="ip: image tokens per frame (4: ImageProjection, 16: plus)")
    ap.add_argument("--sublayers", default="off", choices=["off", "steps", "fused"],
                    help="widened workload (SURVEY.md 8f.2): every attention call with the LayerNorm in front of it and the "
                         "residual add behind it, as three steps (torch LayerNorm / call / torch add) or as ONE library call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements of the default run")
    ap.add_argument("--no-text-kv-cache", action="store_true",
                    help="project the text keys / values in every cross-attention call like the reference does (default: "
                         "once per layer and context; the contexts do not change over the steps)")
    ap.add_argument("--dtype", default=None, choices=["f16", "bf16", "f32"],
                    help="override the workload's storage dtype (f32: the reference's SD1.x default, on the correctness-first fp32 kernels)")
    ap.add_argument("--endpoints", default="replicate", choices=["replicate", "exchange"],
                    help="N > 1 layouts (SURVEY §8e / §8f.4): every rank recomputes the two end-point frames (no per-layer collective) | "
                         "ranks run their owned frames only and the owners broadcast the end points' keys / values per self-attention "
                         "layer on a side stream (eager launches, no fused sublayers)")
    return ap.parse_args()


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run."""
    import torch
    one_dev = os.environ.get("AID_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not one_dev:
        print(f"[bench] --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to report a {args.gpus}-GPU number "
              "(AID_BENCH_ONE_DEVICE=1 runs all ranks on cuda:0 over gloo for development)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; with the legacy mode RCCL's
    # peer-buffer exchange between the rank processes fails in hipIpcGetMemHandle ("invalid argument").  The image exports
    # it already; it is set here too so that a bare `python bench.py --gpus N` works from any shell.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


# ----------------------------------------------------------------------------------------------------
def make_inputs(unet, n_frames, dtype, device, seed=1002, n_ctx=None):
    """Hidden state per resolution level ~ N(0,1) (post-LayerNorm scale) and ``n_ctx`` text contexts ~ N(0,1)
    (one per frame, or the 3 distinct ones [start, guide, end] of a PAID run)."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    n_ctx = n_frames if n_ctx is None else n_ctx
    xs = {}
    for (s, c) in unet.level_shapes():
        xs[(s, c)] = torch.randn(n_frames, s, c, generator=g).to(dtype).to(device)
    cond = torch.randn(n_ctx, unet.text_len, unet.cross_dim, generator=g).to(dtype).to(device)
    uncond = torch.randn(n_ctx, unet.text_len, unet.cross_dim, generator=g).to(dtype).to(device)
    return xs, cond, uncond, g


def recorded_traffic(stack, kernel):
    """(HBM bytes per launch of `kernel`, where the number comes from).  PMC counters cannot be collected from inside this process:
    the figure is READ from the newest committed PMC collection that has the kernel (profiles/rNN_pmc.json, written by
    tools/pmc_collect.py on the builder's GPU box: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same command,
    gfx950 x2 correction on FETCH_SIZE applied) — it is NOT a measurement of the run that prints it; (None, None) if absent."""
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return (json.load(f)["models"][stack][kernel]["hbm_bytes_per_launch"],
                        f"profiles/{name} (builder's box, rocprofv3 --pmc over this command; not measured by this run)")
        except Exception:
            continue
    return None, None


def parity_statement(dt):
    """Stated tolerance of a storage dtype (rel-L2 against the fp64 oracle: one processor call; final latents of a 50-step run over the
    stand-in denoiser, tests/test_hip_depth_and_pipelines.py) next to the values MEASURED on the GPU for every storage dtype
    (profiles/r05_depth_parity.json, written by those tests under AID_WRITE_MEASUREMENTS=1) — north_star asks for 1e-3 on the latents:
    met in fp32 storage by two orders of magnitude, in fp16 only per call (use fp16 or fp32 where parity with an fp32
    run matters; bf16, the dtype BASELINE names for SDXL, pays 8x the rounding step)."""
    stated = {"f16": (1e-3, 2.35e-3), "bf16": (8e-3, 2.2e-2), "f32": (1e-5, 1e-5)}[dt]     # (= 1.3 x the largest measured value)
    out = {"per_call_rel_l2": stated[0], "latents_50_steps_rel_l2": stated[1]}
    for name in ("r06_depth_parity.json", "r05_depth_parity.json", "r04_depth_parity.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            out["measured_50_steps_rel_l2"] = {k.replace("e2e50_", "").replace("e2e20_", "20 steps: "): round(v["rel_l2"], 6)
                                               for k, v in sorted(d.items()) if k.startswith(("e2e50_", "e2e20_"))}
            floor = {k.replace("storage_floor_", ""): {"hip": round(v["hip_rel_l2"], 6), "fp16_storage_alone": round(v["storage_only_rel_l2"], 6)}
                     for k, v in sorted(d.items()) if k.startswith("storage_floor_")}
            if floor:      # fp64 arithmetic rounded only where an fp16 loop stores its tensors, vs the same fp64 oracle
                out["fp16_storage_floor_50_steps"] = floor
            out["measured_from"] = "profiles/" + name
            break
        except Exception:
            continue
    return out


def roofline_pass(loop, aid_amd, torch):
    """HIP-event timing (on the launch stream) of every kernel of one AID step + one plain step."""
    lib = aid_amd._lib.load()
    was, was_conc = loop.use_graphs, getattr(loop, "concurrent_cfg", False)
    loop.use_graphs = False
    loop.concurrent_cfg = False                                  # every kernel alone on the device: the two passes back to back ...
    import contextlib
    # ... launched as in the two-stream run (per-call hint cu_share = 2: engine choice for half the CUs)
    with (aid_amd.ops.cu_share(2) if was_conc else contextlib.nullcontext()):
        loop.step(0); loop.step(loop.num_inference_steps - 1)    # eager warm-up
        torch.cuda.synchronize()
        lib.aid_profile_begin()
        loop.step(0)                                             # AID step
        loop.step(loop.num_inference_steps - 1)                  # plain step
        buf = (aid_amd._lib.AidProfileEntry * 8192)()
        n = lib.aid_profile_end(buf, 8192)
    loop.use_graphs, loop.concurrent_cfg = was, was_conc
    if n < 0:
        raise RuntimeError(lib.aid_strerror(n).decode())
    agg = {}
    for e in buf[:n]:
        a = agg.setdefault(e.kernel.decode(), dict(ms=0.0, flops=0.0, flops_executed=0.0, bytes=0.0, launches=0))
        a["ms"] += e.ms; a["flops"] += e.flops; a["flops_executed"] += e.flops_executed
        a["bytes"] += e.bytes; a["launches"] += 1
    return agg


_CEILING = {}


def power_limited_peak():
    """The dense-MFMA rate THIS box sustains on random operands (tools/ubench/mfma_ceiling: four waves per CU issuing v_mfma_f32_32x32x16
    back to back from registers, nothing else — 32.0 cycles per MFMA and SIMD whatever the data, but the clock the power budget allows
    drops from ~2.3 GHz on zeros to ~1.6 GHz on random sign / mantissa / exponent bits).  Measured once per bench run, outside every
    timed region; None when the helper binary is absent (python -c 'import __graft_entry__ as g; g.build()' builds it)."""
    if "v" not in _CEILING:
        _CEILING["v"] = None
        exe = os.path.join(ROOT, "tools", "ubench", "mfma_ceiling")
        try:
            if os.path.exists(exe):
                out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1]
                _CEILING["v"] = json.loads(out)
        except Exception:
            _CEILING["v"] = None
    return _CEILING["v"]


def roofline_object(agg, stack, dtype="bf16"):
    dom = max(agg, key=lambda k: agg[k]["ms"])          # the kernel SYMBOL with the largest total time
    d = agg[dom]
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    ach_x = d["flops_executed"] / (d["ms"] * 1e-3) / 1e12
    tot_ms = sum(v["ms"] for v in agg.values())
    traffic, traffic_source = recorded_traffic(stack, dom)
    ceil = power_limited_peak()
    pl = None
    if ceil and dtype in ("bf16", "f16"):
        rate = ceil["bf16_random_tflops" if dtype == "bf16" else "f16_random_tflops"]
        pl = {"peak_on_random_operands": rate, "clock_ghz": ceil["bf16_random_ghz" if dtype == "bf16" else "f16_random_ghz"],
              "peak_on_zeros": ceil["bf16_zeros_tflops"], "clock_on_zeros_ghz": ceil["bf16_zeros_ghz"],
              "frac": ach / rate, "frac_executed": ach_x / rate,
              "stack_frac": sum(v["flops"] for v in agg.values()) / (tot_ms * 1e-3) / 1e12 / rate,
              "source": "tools/ubench/mfma_ceiling run by this bench on this box, outside the timed region: dense v_mfma_f32_32x32x16 issue "
                        "from registers, nothing else — the ceiling the POWER budget leaves on random operands; `frac` above stays against "
                        "the nominal 2.5 PFLOP/s"}
    return {
        "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
        "achieved_executed": ach_x, "frac_executed": ach_x / MFMA_PEAK_TFLOPS, "power_limited": pl,
        "traffic": traffic, "traffic_source": traffic_source, "kernel": dom, "launches": d["launches"],
        "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
        "avg_launch_us": d["ms"] * 1e3 / d["launches"], "avg_launch_gflop": d["flops"] / d["launches"] / 1e9,
        "avg_launch_gflop_executed": d["flops_executed"] / d["launches"] / 1e9,
        "share_of_kernel_time": d["ms"] / tot_ms,
        "note": ("per kernel symbol; `achieved` = algorithmic flops (SURVEY.md §8d: a fused-outer frame counts 3 key "
                 "segments, fused-inner 2) / HIP-event time on the launch stream over 1 AID step + 1 plain step; "
                 "`*_executed` counts the segment passes the kernel really runs (a fused END-POINT frame runs 1)"),
        "kernels": {k: {"ms": round(v["ms"], 4), "launches": v["launches"],
                        "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                        "tflops_executed": round(v["flops_executed"] / (v["ms"] * 1e-3) / 1e12, 1),
                        "min_bytes_gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in sorted(agg.items())},
        "stack_tflops": sum(v["flops"] for v in agg.values()) / (tot_ms * 1e-3) / 1e12,
        "stack_tflops_executed": sum(v["flops_executed"] for v in agg.values()) / (tot_ms * 1e-3) / 1e12,
        "kernel_ms_per_2steps": tot_ms,
    }


def cpu_baseline(stack, n_frames, early, steps, warmup_ratio, budget_s=25.0):
    """The CPU port ("port", oracle/aid_cpu_port.py: the oracle's arithmetic on torch CPU ops) on the host cores with
    torch.set_num_threads(the cpus this process may use).  Recipe of SURVEY.md §8d on a bounded sample (~25 s):
    ONE transformer block (self + cross call) per resolution level, in the AID mode and in plain mode, on the 3-frame
    sub-batch [first, middle, last] of the same synthetic inputs.  Every call runs IN FULL (all heads, all query rows, real
    S): levels with S <= 1024 one warm-up + the median of three runs, the S = 4096 level single shots (its calls take
    seconds each); when the budget runs out before the S = 4096 plain calls, those are scaled from the measured AID call by
    the flop ratio (stated in `sample`).  Scaled x n_frames / 3, x blocks per level, x pass counts, to the 50-step unit."""
    import statistics
    import torch
    from oracle import aid_cpu_port as P
    from oracle import aid_oracle as O
    from aid_amd.attn_shim import MODEL_SPECS
    spec = MODEL_SPECS[stack]
    cores = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
    try:                                                   # a container may see every host cpu but own a smaller quota
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            usable = max(1, min(usable, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    torch.set_num_threads(usable)
    g = torch.Generator().manual_seed(1002)
    mode = "outer" if early.endswith("outer") else "inner"
    fused = early.startswith("fused")
    segs = (2 if mode == "outer" else 1) + (1 if fused else 0)
    full = O.beta_coefs(n_frames, steps, steps)
    coef = torch.tensor([0.0, float(full[n_frames // 2]), 1.0])
    levels = {}
    for loc, nblk, s, c, h in spec["layers"]:
        levels[(s, c, h)] = levels.get((s, c, h), 0) + nblk
    t_aid = t_plain = 0.0
    lo_aid = lo_plain = hi_aid = hi_plain = 0.0           # the same sums from the fastest / the slowest run of every call
    t0_all = time.time()
    detail, scaled = [], []

    def timed(x, ctx, w, h, md, fu, cf, single):
        def once():
            t0 = time.time()
            P.processor_call(x, ctx, *w, h, md, fu, cf)
            return time.time() - t0
        if single:
            first = once()                               # seconds per call: single shot ...
            if first < 3.0 and time.time() - t0_all < 0.5 * budget_s:
                return sorted([first, once(), once()])   # ... unless the budget clearly allows three
            return [first]
        once()                                           # warm-up: allocator, thread pool
        return sorted(once() for _ in range(3))

    for (s, c, h), nblk in sorted(levels.items()):          # small levels first, the S = 4096 calls last
        cc, l = spec["cross_dim"], spec["text_len"]
        rn = lambda *sh, sc=1.0: torch.randn(*sh, generator=g) * sc      # noqa: E731
        x, ctx = rn(3, s, c), rn(3, l, cc)
        ws = (rn(c, c, sc=c ** -0.5), rn(c, c, sc=c ** -0.5), rn(c, c, sc=c ** -0.5), rn(c, c, sc=c ** -0.5), rn(c, sc=.01))
        wx = (ws[0], rn(c, cc, sc=cc ** -0.5), rn(c, cc, sc=cc ** -0.5), ws[3], ws[4])
        single = s > 1024
        med, lo, hi = {}, {}, {}
        for tag, xx, cx, w, md, fu, cf in (("aid_self", x, None, ws, mode, fused, coef), ("aid_cross", x, ctx, wx, mode, fused, coef),
                                           ("plain_self", x, None, ws, "plain", False, None),
                                           ("plain_cross", x, ctx, wx, "plain", False, None)):
            if tag == "plain_self" and single and time.time() - t0_all > budget_s:
                # out of budget before the longest plain call: scale the measured AID self call by the flop ratio
                fa = 8.0 * 3 * s * c * c + 4.0 * 3 * s * s * c * segs
                fp = 8.0 * 3 * s * c * c + 4.0 * 3 * s * s * c
                med[tag] = med["aid_self"] * fp / fa
                lo[tag], hi[tag] = lo["aid_self"] * fp / fa, hi["aid_self"] * fp / fa
                scaled.append(f"S={s} plain self call = AID self call x {fp / fa:.2f} (flop ratio)")
                continue
            ts = timed(xx, cx, w, h, md, fu, cf, single)
            med[tag] = statistics.median(ts)
            lo[tag], hi[tag] = ts[0], ts[-1]
            detail.append(f"S={s} C={c} {tag}: " + "/".join(f"{t:.2f}" for t in (ts[0], med[tag], ts[-1])) + " s")
        t_aid += nblk * (med["aid_self"] + med["aid_cross"]) * n_frames / 3.0
        t_plain += nblk * (med["plain_self"] + med["plain_cross"]) * n_frames / 3.0
        lo_aid += nblk * (lo["aid_self"] + lo["aid_cross"]) * n_frames / 3.0
        lo_plain += nblk * (lo["plain_self"] + lo["plain_cross"]) * n_frames / 3.0
        hi_aid += nblk * (hi["aid_self"] + hi["aid_cross"]) * n_frames / 3.0
        hi_plain += nblk * (hi["plain_self"] + hi["plain_cross"]) * n_frames / 3.0
    n_aid = int(steps * warmup_ratio)
    tot = lambda a, p_: n_aid * (a + p_) + (steps - n_aid) * 2 * p_      # noqa: E731
    total = tot(t_aid, t_plain)
    val = lambda t_: n_frames / (t_ * 50.0 / steps)                       # noqa: E731
    return dict(value=val(total), unit="interpolation-frames/sec (50-step)",
                cores=torch.get_num_threads(), kind="port",
                # how far the extrapolation moves with the run-to-run noise of the sampled calls (rounds 2 - 5 printed 0.00088 - 0.0018
                # for this workload): the value from every call's slowest / fastest run; the S = 4096 calls are single shots (lo = hi)
                spread={"value_low": val(tot(hi_aid, hi_plain)), "value_high": val(tot(lo_aid, lo_plain)),
                        "note": "same extrapolation from the slowest / the fastest run of every sampled call; a baseline, not a target"},
                sample=(f"oracle/aid_cpu_port.py (torch CPU ops, fp32, torch.get_num_threads() = {torch.get_num_threads()} = the cpus this "
                        f"process may use, of {cores} host cpus): 1 transformer block (self + cross call) per resolution level in {early} and in "
                        f"plain mode on the 3-frame sub-batch [first, middle, last]; every call in full (all heads, all rows, real S): S <= 1024 one "
                        f"warm-up + median of 3, S = 4096 single shots" + ("; " + "; ".join(scaled) if scaled else "") +
                        f"; scaled x{n_frames}/3 frames, x blocks per level, x({n_aid} AID + {2 * steps - n_aid} plain passes), x50/{steps}; "
                        f"measured {time.time() - t0_all:.1f} s of CPU work"),
                calls_min_median_max_s=detail)


# ----------------------------------------------------------------------------------------------------
def build_workload(name, args, world, rank, device, torch, aid_amd):
    """Stack + inputs + loop of one workload on this rank.  Returns a dict."""
    from aid_amd import dist as adist
    from aid_amd.loop import AidDenoiseLoop, install_sequence_processors
    stack, dt, early_default, frames_default, guide_default, what = WORKLOADS[name]
    dt = getattr(args, "dtype", None) or dt
    dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dt]
    early = args.early or early_default
    steps = args.steps
    if args.frames is not None:
        n_total = args.frames
    elif args.weak and world > 1:
        n_total = args.frames_per_gpu * world
    else:
        n_total = frames_default
    exch = args.endpoints == "exchange"
    shard = adist.owned_shard(n_total, world, rank) if exch else adist.frame_shard(n_total, world, rank)
    guided = guide_default if args.guide_prompt == "auto" else args.guide_prompt == "on"
    if name == "ip":
        guided = False

    unet = aid_amd.AttnStackUNet(stack, dtype=dtype, device=device)
    unet.sublayers = args.sublayers
    coef = aid_amd.generate_beta_tensor(n_total, steps, steps)
    coef[0], coef[-1] = 0, 1
    # PAID (gradio ...stable_diffusion.py:221-229): contexts [start, guide x (N-2), end] -> 3 distinct rows
    global_ctx = ([0] + [1] * (n_total - 2) + [2]) if guided else list(range(n_total))
    xs, cond, uncond, g = make_inputs(unet, n_total, dtype, device, n_ctx=3 if guided else None)
    ip_pos = ip_neg = None
    if name == "ip":
        # image embeddings after encoder_hid_proj, the reference's layout: 3 copies per frame, [3 N, 1, T, Cc]
        # (pipeline_interpolated_sd.py:1763-1802); negative image embeddings for the unconditional pass
        ip_pos = torch.randn(n_total, 1, args.ip_tokens, unet.cross_dim, generator=g).to(dtype).to(device)
        ip_neg = torch.randn(n_total, 1, args.ip_tokens, unet.cross_dim, generator=g).to(dtype).to(device)
    if world > 1:                            # conditioning comes from rank 0 (north_star: RCCL broadcast)
        named = {f"x{s}_{c}": t for (s, c), t in xs.items()}
        named.update(cond=cond, uncond=uncond)
        if ip_pos is not None:
            named.update(ip_pos=ip_pos, ip_neg=ip_neg)
        adist.broadcast_conditioning(named, src=0)
    xs = {k: adist.shard_rows(v, shard) for k, v in xs.items()}
    end_ctx = torch.stack([cond[global_ctx[0]], cond[global_ctx[-1]]]) if exch else None    # text contexts of frames 0 / N-1
    rows = [global_ctx[f] for f in shard.index]              # context row of every local frame ...
    used = sorted(set(rows), key=rows.index)                 # ... renumbered in order of first local use
    ctx_index = [used.index(r) for r in rows] if guided else None
    sel = torch.tensor(used, device=device)
    cond, uncond = cond.index_select(0, sel).contiguous(), uncond.index_select(0, sel).contiguous()
    local_coef = coef[list(shard.index)]
    passes = "serial" if args.separate_passes else args.passes
    if passes == "auto":
        # sd15: small launches, two streams fill the device better; ip: cannot batch.  The SDXL text workloads are faster on two
        # streams as well (36.8 -> 35.7 ms/step, `also.sdxl_two_streams`) but stay batched by default: every launch then owns the
        # device, which keeps the per-kernel roofline of the top-level line unambiguous and comparable across rounds
        passes = "streams" if (name in ("sd15", "ip") and not exch) else "batched"
    if name == "ip" and passes == "batched":
        raise SystemExit("--workload ip: the two passes carry different image embeddings (--passes streams | serial)")
    batched = passes == "batched"
    if name == "ip":
        unet.load_ip_adapter(num_tokens=args.ip_tokens, scale=0.6)
        aid_amd.load_aid_ip_adapter(unet, t=None, size=shard.n_local, is_fused=True, early=early, alpha=steps, beta=steps)
        for p in unet.attn_processors.values():
            p.coef = local_coef.detach().to(torch.float32).cpu().clone()
        rep3 = lambda t: adist.shard_rows(t, shard).repeat_interleave(3, dim=0).contiguous()      # noqa: E731
        cond, uncond = (cond, [rep3(ip_pos)]), (uncond, [rep3(ip_neg)])
    else:
        install_sequence_processors(unet, shard.n_local, early=early, num_inference_steps=steps, coef=local_coef)
    if exch:
        if name == "ip" or args.sublayers != "off":
            raise SystemExit("--endpoints exchange: text processors, --sublayers off")
        ex = adist.EndpointExchange(n_total, world, rank)
        for p in unet.attn_processors.values():
            p.endpoint_exchange, p.endpoint_ctx = ex, end_ctx
        if os.environ.get("AID_BENCH_ONE_DEVICE") == "1":
            args.no_graph = True             # development mode: gloo collectives run on the host and cannot be captured
        # a collective per self-attention layer: RCCL records broadcast / all_gather and the exchange's side-stream fork / join into a
        # hipGraph on this ROCm (tools/dev/rccl_graph_capture.py, profiles/r06_rccl_graph_capture.txt), so the passes are captured like
        # the other layouts'; a stack that cannot falls back to eager launches inside AidDenoiseLoop and says so (config.graph_fallback)
    loop = AidDenoiseLoop(unet, xs, cond, uncond, num_inference_steps=steps, warmup_ratio=args.warmup_ratio,
                          use_graphs=not args.no_graph, batched_cfg=batched, ctx_index=ctx_index,
                          concurrent_cfg=(passes == "streams"))
    return dict(name=name, stack=stack, dtype=dt, early=early, what=what, n_total=n_total, shard=shard, guided=guided,
                unet=unet, loop=loop, batched=batched, passes=passes, gather_key=unet.level_shapes()[-1])


def time_workload(wl, args, world, device, torch, dist):
    """W untimed warm-up steps, then EXACTLY K steps x `repeats`, bracketed by barrier + synchronize; max over ranks."""
    from aid_amd import dist as adist
    loop, steps = wl["loop"], args.steps
    one_dev = os.environ.get("AID_BENCH_ONE_DEVICE") == "1"

    def run_steps(idx):
        out = None
        for i in idx:
            out = loop.step(i)
        return out

    # untimed warm-up: alternate AID / plain steps so every graph is captured and warm
    warm_idx = [0 if (j % 2 == 0) else steps - 1 for j in range(max(args.warmup, 2))]
    run_steps(warm_idx[:2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(warm_idx[2:] or warm_idx[:2])
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / max(len(warm_idx[2:] or warm_idx[:2]), 1)
    reps = max(1, int(math.ceil(args.min_seconds / max(est * steps, 1e-6))))
    if world > 1:
        r = torch.tensor([reps], device="cpu" if one_dev else device, dtype=torch.int64)
        dist.all_reduce(r, op=dist.ReduceOp.MAX)
        reps = int(r.item())
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(reps):
        out = run_steps(range(steps))
    final = adist.gather_owned(out[wl["gather_key"]], wl["shard"])      # all_gather of the owned frames' outputs
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if one_dev else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert final.shape[0] == wl["n_total"] and torch.isfinite(final.float()).all()
    ms_per_step = elapsed * 1000.0 / (steps * reps)
    return dict(ms_per_step=ms_per_step, value=wl["n_total"] / (ms_per_step * 50.0 / 1000.0), repeats=reps,
                timed_seconds=elapsed, warmup=len(warm_idx))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # development aid: AID_BENCH_ONE_DEVICE=1 runs all ranks on cuda:0 with gloo collectives, so the N>1 code
    # path (sharding, broadcast, all_gather, max-over-ranks) can be exercised on a 1-GPU box
    one_dev = os.environ.get("AID_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_dev else local_rank
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if one_dev else "nccl"
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    import aid_amd
    from aid_amd import dist as adist
    from aid_amd import processors as aproc
    aid_amd._lib.load()                     # fail loudly before anything else if the HIP library is missing
    aproc.TEXT_KV_CACHE = not args.no_text_kv_cache

    name = args.workload
    if name == "auto":
        name = args.model or ("sdxl" if world == 1 else "seq16")
    wl = build_workload(name, args, world, rank, device, torch, aid_amd)
    tm = time_workload(wl, args, world, device, torch, dist)
    shard, loop, steps = wl["shard"], wl["loop"], args.steps

    result = {
        "metric": "interpolation-frames/sec (50-step)",
        "value": tm["value"], "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": tm["warmup"],
        "ms_per_step": tm["ms_per_step"], "higher_is_better": True,
        "scaling": "weak" if (args.weak or world == 1) else "strong", "vs_baseline": None,
        "dtype": wl["dtype"], "data": "synthetic",
        "repeats": tm["repeats"], "timed_seconds": tm["timed_seconds"],
        "config": {
            "workload": wl["what"],
            "frames": wl["n_total"], "local_batch": shard.n_local, "owned_frames": shard.n_owned,
            "early": wl["early"], "late": "plain", "warmup_ratio": args.warmup_ratio,
            "aid_steps": loop.warmup_steps,
            "passes_per_step": {"batched": "cond + uncond (CFG) batched in one UNet call [cond ; uncond]",
                                "streams": "cond + uncond (CFG), two UNet calls on two streams, forked and joined inside one graph",
                                "serial": "cond + uncond (CFG), two UNet calls back to back"}[wl["passes"]],
            "contexts": ("PAID guide prompt: interior frames share one text context (3 distinct per pass), keys/values "
                         "projected once per distinct context" if wl["guided"] else "one text context per frame"),
            "sublayers": {"off": "attention calls only (the BASELINE metric)",
                          "steps": "LayerNorm + call + residual add per layer, three steps (torch LayerNorm / add)",
                          "fused": "LayerNorm + call + residual add per layer in one library call"}[args.sublayers],
            "text_kv": ("keys / values of the text contexts projected ONCE per (layer, context) and reused by every step: the "
                        "contexts are loop-invariant (reference pipeline_interpolated_sd.py:1859-1867 passes the same prompt_embeds "
                        "each step and re-projects them 50 times); `also.sdxl_text_kv_per_call` times the per-call projection"
                        if not args.no_text_kv_cache else "projected in every cross-attention call (like the reference)"),
            "coef": f"BetaPPF(alpha=beta={steps})", "hipgraph": bool(loop.use_graphs),
            "graph_fallback": getattr(loop, "fallback_reason", None),
            # stated tolerance of this storage dtype: rel-L2 of the final latents of a 50-step run vs the fp64 oracle loop
            # (tests/test_hip_depth_and_pipelines.py E2E50_BOUND; measured values in profiles/r04_depth_parity.json) and of one call
            "parity_tolerance": parity_statement(wl["dtype"]),
            "parallelism": (f"frame-shard x{world} (replicated end points, no per-layer collective)" if args.endpoints == "replicate"
                            else f"frame-shard x{world} (owned frames only; end-point keys / values broadcast per self-attention "
                                 "layer on a side stream)"),
            "ranks": world, "backend": backend,
            "expected_speedup_vs_1gpu": (adist.expected_speedup(wl["n_total"], world) if args.endpoints == "replicate"
                                         else wl["n_total"] / max(adist.owned_shard(wl["n_total"], world, r).n_local
                                                                  for r in range(world))),
        },
    }
    if name == "ip":
        result["config"]["ip_adapter"] = (f"{args.ip_tokens} image tokens per frame, image embeddings [3 N, 1, T, Cc]; AID pass = "
                                          f"{wl['early']} IP processors, other passes = IP-Adapter attention (text + scale x image)")
    if world > 1:
        # every rank contributes (rank, device index) through the job's own collective backend: the driver can check that N distinct
        # ranks on N distinct devices took part
        mine = torch.tensor([rank, dev_index], device="cpu" if one_dev else device, dtype=torch.int64)
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        result["rccl_ranks_seen"] = sorted(int(t[0]) for t in seen)
        result["devices_seen"] = sorted(int(t[1]) for t in seen)
        mk = adist.owned_shard if args.endpoints == "exchange" else adist.frame_shard
        mlb = max(mk(wl["n_total"], world, r).n_local for r in range(world))
        result["config"]["max_local_batch"] = mlb
        # what the busiest rank delivers, in frames of ITS batch per second: compare with the 1-GPU line at that batch size
        # (replicated end points make the ideal speed-up n_total / max_local_batch, not the rank count)
        result["config"]["busiest_rank_local_frames_per_s"] = mlb / (tm["ms_per_step"] * 50.0 / 1000.0)

    if not args.no_roofline and (rank == 0 or args.endpoints == "exchange"):
        # N > 1: rank 0's LOCAL batch.  Replicated end points: no collective inside a step, so rank 0 times its kernels alone;
        # the exchange layout broadcasts per layer, so every rank walks the profiled steps and rank 0 reports
        prof = roofline_pass(loop, aid_amd, torch)
        if rank == 0:
            result["roofline"] = roofline_object(prof, wl["stack"], wl["dtype"])
            if world > 1:
                result["roofline"]["scope"] = f"rank 0 of {world}: local batch of {shard.n_local} frames"
    if rank == 0 and world == 1 and not args.no_cpu_baseline and name != "ip":
        result["cpu_baseline"] = cpu_baseline(wl["stack"], wl["n_total"], wl["early"], steps, args.warmup_ratio)
    if rank == 0 and world == 1 and name == "sdxl" and args.workload == "auto" and not args.model and not args.no_also:
        # BASELINE's metric names SD1.5 512^2 AND SDXL 1024^2: the SD1.5 half rides along (same steps / repeats rule)
        del wl, loop
        torch.cuda.empty_cache()
        w2 = build_workload("sd15", args, world, rank, device, torch, aid_amd)
        t2 = time_workload(w2, args, world, device, torch, dist)
        also = {"workload": w2["what"], "value": t2["value"], "unit": "frames/s", "ms_per_step": t2["ms_per_step"],
                "repeats": t2["repeats"], "dtype": w2["dtype"], "early": w2["early"], "frames": w2["n_total"]}
        if not args.no_roofline:
            r2 = roofline_object(roofline_pass(w2["loop"], aid_amd, torch), "sd15", w2["dtype"])
            also["roofline"] = {k: r2[k] for k in ("kernel", "achieved", "frac", "achieved_executed", "frac_executed", "power_limited",
                                                   "traffic", "avg_launch_us", "share_of_kernel_time", "stack_tflops", "kernels")}
        result["also"] = {"sd15": also}
        # the headline workload in fp16 storage (north_star's 1e-3 rel-L2 holds in fp16) ...
        del w2
        torch.cuda.empty_cache()
        args.dtype = "f16"
        w3 = build_workload("sdxl", args, world, rank, device, torch, aid_amd)
        t3 = time_workload(w3, args, world, device, torch, dist)
        result["also"]["sdxl_fp16"] = {"workload": w3["what"], "value": t3["value"], "unit": "frames/s",
                                       "ms_per_step": t3["ms_per_step"], "repeats": t3["repeats"], "dtype": "f16"}
        del w3
        torch.cuda.empty_cache()
        # ... on two streams (the reference's two UNet calls, concurrently) instead of one batched call
        args.dtype = None
        if args.passes == "auto" and not args.separate_passes:
            args.passes = "streams"
            w5 = build_workload("sdxl", args, world, rank, device, torch, aid_amd)
            t5 = time_workload(w5, args, world, device, torch, dist)
            result["also"]["sdxl_two_streams"] = {"value": t5["value"], "unit": "frames/s", "ms_per_step": t5["ms_per_step"],
                                                  "repeats": t5["repeats"], "dtype": w5["dtype"],
                                                  "passes_per_step": "two UNet calls on two streams inside one graph, cu_share = 2 per call"}
            args.passes = "auto"
            del w5
            torch.cuda.empty_cache()
            # ... the reference's order: the two UNet calls back to back on one stream
            args.passes = "serial"
            w6 = build_workload("sdxl", args, world, rank, device, torch, aid_amd)
            t6 = time_workload(w6, args, world, device, torch, dist)
            result["also"]["sdxl_serial"] = {"value": t6["value"], "unit": "frames/s", "ms_per_step": t6["ms_per_step"],
                                             "repeats": t6["repeats"], "dtype": w6["dtype"],
                                             "passes_per_step": "two UNet calls back to back (the reference loop's order)"}
            args.passes = "auto"
            del w6
            torch.cuda.empty_cache()
        # ... the CHAINED workload: every layer with its LayerNorm in front and the residual add behind it in one library call, the
        # residual stream of a level running through its layers (activations no longer the same cache-resident tensor for every layer)
        if args.sublayers == "off":
            args.sublayers = "fused"
            w7 = build_workload("sdxl", args, world, rank, device, torch, aid_amd)
            t7 = time_workload(w7, args, world, device, torch, dist)
            result["also"]["sdxl_sublayers_fused"] = {"value": t7["value"], "unit": "frames/s", "ms_per_step": t7["ms_per_step"],
                                                      "repeats": t7["repeats"], "dtype": w7["dtype"],
                                                      "sublayers": "h += attn(LayerNorm(h), ctx) per layer in one library call, residual stream chained"}
            # the same workload as a second TOP-LEVEL object with its own per-kernel roofline: the number a real
            # UNet is closer to — every layer reads what the previous one wrote, LayerNorm statistics and the residual add included
            chained = dict(result["also"]["sdxl_sublayers_fused"], metric=result["metric"],
                           workload=w7["what"] + "; every attention call as h += attn(LayerNorm(h), ctx), the residual stream of a "
                                                  "resolution level chained through its layers")
            if not args.no_roofline:
                r7 = roofline_object(roofline_pass(w7["loop"], aid_amd, torch), "sdxl", w7["dtype"])
                chained["roofline"] = {k: r7[k] for k in ("bound", "achieved", "peak", "unit", "frac", "achieved_executed", "frac_executed", "power_limited",
                                                          "traffic", "traffic_source", "kernel", "avg_launch_us", "share_of_kernel_time",
                                                          "stack_tflops", "stack_tflops_executed", "kernel_ms_per_2steps", "kernels")}
            result["chained"] = chained
            args.sublayers = "off"
            del w7
            torch.cuda.empty_cache()
        # ... BASELINE configs[0] on the GPU: SD1.5, batch 3, 20 steps, float32 storage (the reference's SD1.x default) on the
        # correctness-first fp32 kernels (v_mfma_f32_32x32x2_f32: 157 TFLOP/s peak)
        if not args.frames and not args.early:
            keep = (args.dtype, args.frames, args.steps, args.min_seconds)
            args.dtype, args.frames, args.steps, args.min_seconds = "f32", 3, 20, 0.0
            w8 = build_workload("sd15", args, world, rank, device, torch, aid_amd)
            t8 = time_workload(w8, args, world, device, torch, dist)
            f32 = {"workload": "BASELINE configs[0] on the GPU: SD1.5 512x512 attention stack, 3 frames [start, target, end], 20 steps, fp32",
                   "value": w8["n_total"] / (t8["ms_per_step"] * 20.0 / 1000.0), "unit": "interpolation-frames/sec (20-step)",
                   "ms_per_step": t8["ms_per_step"], "repeats": t8["repeats"], "dtype": "f32", "frames": 3, "steps": 20}
            if not args.no_roofline:
                r8 = roofline_object(roofline_pass(w8["loop"], aid_amd, torch), "sd15", "f32")
                f32["stack_tflops"] = r8["stack_tflops"]
                f32["peak_tflops"] = 157.3
                f32["kernels"] = r8["kernels"]
            result["also"]["sd15_fp32_batch3"] = f32
            args.dtype, args.frames, args.steps, args.min_seconds = keep
            del w8
            torch.cuda.empty_cache()
        # ... BASELINE configs[3] on ONE device (the 16-frame sequence a rank of an 8-GPU run would see as 4 frames) and configs[4]
        # (SDXL + IP-Adapter, 8 frames, two passes on two streams): every BASELINE config has a driver-timed line
        if not args.frames and not args.early and args.passes == "auto" and not args.separate_passes:
            for key, wname in (("seq16_1gpu", "seq16"), ("ip", "ip")):
                w9 = build_workload(wname, args, world, rank, device, torch, aid_amd)
                t9 = time_workload(w9, args, world, device, torch, dist)
                result["also"][key] = {"workload": w9["what"], "value": t9["value"], "unit": "frames/s", "ms_per_step": t9["ms_per_step"],
                                       "repeats": t9["repeats"], "dtype": w9["dtype"], "frames": w9["n_total"], "early": w9["early"],
                                       "passes": w9["passes"]}
                del w9
                torch.cuda.empty_cache()
        # ... and with the text keys / values projected in every call, like the reference
        if not args.no_text_kv_cache:
            aproc.TEXT_KV_CACHE = False
            w4 = build_workload("sdxl", args, world, rank, device, torch, aid_amd)
            t4 = time_workload(w4, args, world, device, torch, dist)
            result["also"]["sdxl_text_kv_per_call"] = {"value": t4["value"], "unit": "frames/s", "ms_per_step": t4["ms_per_step"],
                                                       "repeats": t4["repeats"], "dtype": w4["dtype"]}
            aproc.TEXT_KV_CACHE = True
            del w4

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()                      # the other ranks wait for rank 0's roofline pass
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
