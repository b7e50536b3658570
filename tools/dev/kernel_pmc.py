#!/usr/bin/env python3
"""rocprofv3 PMC counters of ONE kernel symbol over a small driver command (development; run on the GPU box from the repo root).
usage: python tools/dev/kernel_pmc.py <substring of the kernel name> -- <driver command ...>
Each counter group is its own rocprofv3 pass with --kernel-trace only (MI355X_MICROARCH.md: never combine --pmc with sys / hip tracing).
Prints per-launch averages and a few derived figures: matrix-pipe busy fraction, LDS busy / conflict share, clock."""
import collections, csv, glob, os, subprocess, sys, tempfile

PASSES = [
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
     "SQ_ACTIVE_INST_ANY"],
    ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_SALU",
     "SQ_LDS_ADDR_CONFLICT"],
    ["GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"], ["WRITE_SIZE"],
]
key = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]
tot = collections.defaultdict(float)
n_launch, dur = 0, 0.0
for counters in PASSES:
    out = tempfile.mkdtemp(prefix="kpmc_", dir="/tmp")
    r = subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "--"] + cmd,
                       cwd=os.getcwd(), env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    fs = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        print("pass failed:", counters, r.stderr[-400:])
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    d = {}
    for row in csv.DictReader(open(fs[0])):
        if key not in row["Kernel_Name"]:
            continue
        per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
        if row.get("Start_Timestamp") and row.get("End_Timestamp"):
            d[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    if not per:
        print("no launch of", key, "in pass", counters)
        continue
    n_launch = len(per)
    for disp in per.values():
        for c, v in disp.items():
            tot[c] += v / n_launch
    if d:
        dur = sum(d.values()) / len(d)
for c in sorted(tot):
    print(f"{c:32s} {tot[c]:16.1f}")
cyc = tot.get("GRBM_GUI_ACTIVE", 0) / 8
if cyc:
    print(f"launches {n_launch}, duration {dur / 1e3:.1f} us (profiled), cycles {cyc:.0f}, clock {cyc / dur:.2f} GHz")
    print(f"mfma_busy_frac        {tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}")
    print(f"lds_active_frac       {tot['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):.3f}   (LDS-array cycles / (CUs x cycles))")
    print(f"lds_conflict_share    {tot['SQ_LDS_BANK_CONFLICT'] / max(tot['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
    print(f"wave-cycles: wait_any {tot['SQ_WAIT_ANY'] / max(tot['SQ_WAVE_CYCLES'], 1):.3f}  wait_inst_any {tot['SQ_WAIT_INST_ANY'] / max(tot['SQ_WAVE_CYCLES'], 1):.3f}  "
          f"active_inst_any {tot['SQ_ACTIVE_INST_ANY'] / max(tot['SQ_WAVE_CYCLES'], 1):.3f}  wait_inst_lds {tot['SQ_WAIT_INST_LDS'] / max(tot['SQ_WAVE_CYCLES'], 1):.3f}")
    print(f"hbm bytes / launch    {(2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024 / 1e6:.1f} MB (2 x FETCH_SIZE + WRITE_SIZE)")
