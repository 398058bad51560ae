#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table
(the same content as `rocprofv3 --stats` kernel_stats.csv): calls, total, avg, min, max, %."""
import re
import sqlite3
import subprocess
import sys


def demangle(n):
    try:
        out = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        out = n
    out = re.sub(r"\(anonymous namespace\)::", "", out)
    return re.sub(r"\(.*", "", out).replace("void ", "")


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':72s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for n, c, s, a, mn, mx in rows:
        lines.append(f"{demangle(n)[:72]:72s} {c:8d} {s / 1e6:10.3f} {a / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / tot:6.2f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
