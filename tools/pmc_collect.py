#!/usr/bin/env python3
"""Collect rocprofv3 PMC counters per launch of every aid_* kernel of the bench workloads -> profiles/r06_pmc.json.

Procedure (MI355X_MICROARCH.md, HBM / PMC-slot sections): every counter group is its OWN rocprofv3 pass with
--kernel-trace only (never combined with sys / hip / memory-copy tracing), over
    bench.py --workload W --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-also
  pass 1  FETCH_SIZE                      pass 2  WRITE_SIZE        (they do not fit one TCC pass)
  pass 3  SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY
          SQ_ACTIVE_INST_ANY
  pass 4  SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES
          SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU
  pass 5  GRBM_GUI_ACTIVE                 (kernel duration in shader-clock cycles -> effective clock, MFMA-busy fraction)
Per kernel (named as bench.py's roofline object names them) the averages per launch are stored, plus
  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950: FETCH_SIZE reports half the bytes of a 16 B/lane
                                                                 streaming read; WRITE_SIZE is uncalibrated)
  cycles               = GRBM_GUI_ACTIVE / 8            (the counter is summed over the 8 XCDs' GRBMs)
  mfma_busy_frac       = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * cycles)   (the counter = 32 x MFMAs issued: matrix-pipe
                         occupancy incl. padded / row-sum MFMAs, at the clock the kernel really ran at)
  valu_per_mfma        = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA   (SQ_INSTS_VALU counts the MFMAs too)
  coexec_frac_of_mfma_busy = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES  (matrix-pipe time with a VALU op in flight)
  clock_ghz            = cycles / rocprofv3 kernel duration
usage (on the GPU box, from the repo root):  python tools/pmc_collect.py [out.json] [--workloads sdxl,sd15]"""
import collections, csv, glob, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = {"0": "plain", "1": "inner", "2": "outer"}
PASSES = [
    ["FETCH_SIZE"], ["WRITE_SIZE"],
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
     "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU",
     "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_SALU"],
    ["GRBM_GUI_ACTIVE"],
]


WORKLOAD_DTYPE = {"sdxl": "bf16", "seq16": "bf16", "ip": "bf16", "sd15": "f16"}
CURRENT = [None]                       # workload being collected (bench_name needs it for one broken demangling)


def bench_name(sym):
    """kernel symbol -> the name bench.py's roofline object uses"""
    dt = "bf16" if "IDF16b" in sym else "f16"
    m = re.search(r"aid_attn_kernelIDF16b?_?Li(\d+)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)", sym)
    if m:
        if m.group(6) == "1":
            return f"aid_attn<{dt},d{m.group(1)},{MODES[m.group(2)]},res>"
        sfx = ",qb2" if m.group(4) == "2" else ",pipe" if m.group(5) == "1" else ""
        return f"aid_attn<{dt},d{m.group(1)},{MODES[m.group(2)]},nw{m.group(3)}{sfx}>"
    if "aid_attn_pp_kernel" in sym:                     # one device symbol behind aid_attn_pp<dt,d64> and aid_attn_pp<dt,d64,outer>
        return f"aid_attn_pp<{dt},d64>"
    if "aid_attn_tx_kernel" in sym:          # (rocprofv3 prints the one-region instantiation through a broken demangling: neither the
        m = re.search(r"aid_attn_tx_kernelIDF16b?_?Li(\d)", sym)      # dtype nor "Li1" survive; the workloads run it in one dtype each)
        if not m:
            dt = WORKLOAD_DTYPE.get(CURRENT[0], dt)
        return f"aid_attn_tx<{dt},d64,{'outer' if m and m.group(1) == '3' else 'plain'}>"
    for k in ("aid_gemm_rs_kernel", "aid_gemm_nt_ppx_kernel", "aid_gemm_nt_pp_kernel", "aid_gemm_nt_pipe_kernel", "aid_gemm_nt_kernel", "aid_lerp_kv_kernel",
              "aid_layernorm_kernel", "aid_ln_stats_kernel"):
        if k in sym:
            short = {"aid_lerp_kv_kernel": "aid_lerp_kv", "aid_layernorm_kernel": "aid_layernorm", "aid_ln_stats_kernel": "aid_ln_stats"}
            return f"{short.get(k, k)}<{dt}>"
    return None


def collect(workload, counters):
    CURRENT[0] = workload
    out = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "2", "--no-graph",
           "--no-cpu-baseline", "--no-roofline", "--no-also", "--min-seconds", "0"] + (["--passes", "serial"] if workload in ("sd15", "ip") else [])
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        nm = bench_name(r["Kernel_Name"])
        if nm is None:
            continue
        per[nm][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[nm][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    res = {}
    for nm, disp in per.items():
        n = len(disp)
        avg = collections.defaultdict(float)
        for d in disp.values():
            for c, v in d.items():
                avg[c] += v / n
        res[nm] = dict(avg, launches=n)
        if dur[nm]:
            res[nm]["_ns"] = sum(dur[nm].values()) / len(dur[nm])
    return res


def derive(e, v):
    cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        e["cycles_per_launch"] = round(cyc, 1)
        e["mfma_busy_frac"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
    if v.get("SQ_INSTS_MFMA"):
        e["valu_per_mfma"] = round((v.get("SQ_INSTS_VALU", 0.0) - v["SQ_INSTS_MFMA"]) / v["SQ_INSTS_MFMA"], 2)
    # GRBM_GUI_ACTIVE also counts the dispatch ramp-up / drain around a launch: for launches under ~20 us the quotient is not a
    # clock (a 5 us lerp kernel came out at "5.9 GHz"), so it is reported for longer kernels only
    if cyc and v.get("profiled_ns_per_launch", 0) >= 20000:
        e["clock_ghz"] = round(cyc / v["profiled_ns_per_launch"], 3)
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and "SQ_VALU_MFMA_COEXEC_CYCLES" in v:
        e["coexec_frac_of_mfma_busy"] = round(v["SQ_VALU_MFMA_COEXEC_CYCLES"] / v["SQ_VALU_MFMA_BUSY_CYCLES"], 4)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--rederive":        # recompute the derived fields of an existing file
        res = json.load(open(sys.argv[2]))
        res["_comment"] = __doc__.split("usage")[0].strip()
        for t in res["models"].values():
            for e in t.values():
                for k in ("mfma_busy_frac", "valu_per_mfma", "clock_ghz", "coexec_frac_of_mfma_busy", "cycles_per_launch"):
                    e.pop(k, None)
                derive(e, e)
        json.dump(res, open(sys.argv[3] if len(sys.argv) > 3 else sys.argv[2], "w"), indent=1)
        return
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    wls = sys.argv[sys.argv.index("--workloads") + 1].split(",") if "--workloads" in sys.argv else ["sdxl", "sd15"]
    out = argv[0] if argv else os.path.join(ROOT, "profiles", "r06_pmc.json")
    res = {"_comment": __doc__.split("usage")[0].strip(), "models": {}}
    for wl in wls:
        merged = collections.defaultdict(dict)
        for counters in PASSES:
            try:
                got = collect(wl, counters)
            except Exception as e:                                   # a counter this rocprofv3 does not know: keep going
                print(f"[pmc] pass {counters} failed for {wl}: {e}", file=sys.stderr)
                continue
            for nm, vals in got.items():
                ns = vals.pop("_ns", None)
                merged[nm].update(vals)
                if ns is not None and counters == ["GRBM_GUI_ACTIVE"]:
                    merged[nm]["profiled_ns_per_launch"] = ns
        table = {}
        for nm, v in sorted(merged.items()):
            e = {k: round(x, 1) for k, x in v.items() if k != "launches"}
            e["launches"] = v.get("launches")
            if "FETCH_SIZE" in v:
                e["hbm_bytes_per_launch"] = int((2 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024)
            derive(e, v)
            table[nm] = e
        stack = {"sdxl": "sdxl", "sd15": "sd15"}.get(wl, wl)
        res["models"][stack] = table
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["models"], indent=1))


if __name__ == "__main__":
    main()
