#!/usr/bin/env python3
"""Collect HBM traffic per launch of every aid_* kernel of the bench workload and write profiles/r01_pmc_traffic.json.

Procedure (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do
not fit one pass) with --kernel-trace only, over `bench.py --model M --steps 2 --warmup 2 --no-graph`; per kernel
symbol average KB per launch; HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — on gfx950 FETCH_SIZE reports half the
bytes of a 16 B/lane streaming read; WRITE_SIZE is uncalibrated.
usage (on the GPU box, from the repo root):  python tools/pmc_traffic.py [out.json]"""
import collections, csv, glob, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = {"0": "plain", "1": "inner", "2": "outer"}


def bench_name(sym):
    """kernel symbol -> the name bench.py's roofline object uses"""
    dt = "bf16" if "IDF16b" in sym else "f16"
    m = re.search(r"aid_attn_kernelIDF16b?_?Li(\d+)ELi(\d)ELi(\d)", sym)
    if m:
        return f"aid_attn<{dt},d{m.group(1)},{MODES[m.group(2)]},nw{m.group(3)}>"
    if "aid_gemm_nt" in sym:
        return f"aid_gemm_nt<{dt}>"
    if "aid_lerp_kv" in sym:
        return f"aid_lerp_kv<{dt}>"
    return None


def collect(model, counter):
    out = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--model", model, "--steps", "2", "--warmup", "2", "--no-graph",
           "--no-cpu-baseline", "--no-roofline"]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: [0.0, 0])
    for sym, disp in per.items():
        nm = bench_name(sym)
        if nm:
            agg[nm][0] += sum(disp.values())
            agg[nm][1] += len(disp)
    return {k: (v[0] / v[1], v[1]) for k, v in agg.items()}


def main():
    res = {"_comment": __doc__.split("usage")[0].strip(), "models": {}}
    for model in ("sd15", "sdxl"):
        fe, wr = collect(model, "FETCH_SIZE"), collect(model, "WRITE_SIZE")
        res["models"][model] = {
            k: {"fetch_size_kb_per_launch": round(fe[k][0], 1), "launches": fe[k][1],
                "write_size_kb_per_launch": round(wr.get(k, (0, 0))[0], 1),
                "hbm_bytes_per_launch": int((2 * fe[k][0] + wr.get(k, (0, 0))[0]) * 1024)} for k in sorted(fe)}
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["models"], indent=1))


if __name__ == "__main__":
    main()
