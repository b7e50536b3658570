#!/usr/bin/env python3
"""Turn a rocprofv3 results DB (rocpd sqlite, `rocprofv3 --kernel-trace --stats`) into the plain-text
per-kernel summary that is committed under profiles/.   usage: rocprof_summary.py results.db [out.txt]"""
import sqlite3
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = [f"# rocprofv3 --kernel-trace --stats   source: {sys.argv[1]}",
             f"{'calls':>8} {'total_ms':>14} {'avg_us':>10} {'pct':>7}  kernel"]
    for name, calls, tot, avg, pct in rows:
        nm = demangle(name).replace("aid::", "")
        if len(nm) > 110:
            nm = nm[:107] + "..."
        lines.append(f"{calls:8d} {tot/1e3:14.2f} {avg:10.2f} {pct:7.2f}  {nm}")
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
