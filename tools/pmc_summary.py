#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, counters of the LAST dispatch.
usage: pmc_summary.py counter_collection.csv [kernel-substring]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else "aid_"
agg = collections.OrderedDict()
for r in rows:
    if sub in r["Kernel_Name"]:
        agg.setdefault((r["Kernel_Name"][:60], r["Dispatch_Id"]), collections.OrderedDict())
        d = agg[(r["Kernel_Name"][:60], r["Dispatch_Id"])]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["_grid"] = r["Grid_Size"]; d["_wg"] = r["Workgroup_Size"]; d["_vgpr"] = r["VGPR_Count"]; d["_lds"] = r["LDS_Block_Size"]
        d["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
last = {}
for (name, disp), d in agg.items():
    last[name] = d
for name, d in last.items():
    print(name, {k: v for k, v in d.items() if k.startswith("_")})
    for k, v in d.items():
        if not k.startswith("_"):
            print(f"   {k:28s} {v:.4e}")
