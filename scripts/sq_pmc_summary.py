#!/usr/bin/env python
"""Per-kernel averages of the SQ counters of a `rocprofv3 --kernel-trace --pmc ...` run (rocpd sqlite database).
usage: sq_pmc_summary.py <db> [<out.txt>]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "").strip()


def main(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    res = {}
    for name, ctr, n, avg in rows:
        d = res.setdefault(short(name), {"launches": n})
        d[ctr] = avg
    lines = []
    for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1]["launches"])[:14]:
        lines.append(f"{k[:90]}  launches={d['launches']}")
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1
        for c in sorted(d):
            if c == "launches":
                continue
            lines.append(f"    {c:32s} {d[c]:16.1f}   {100 * d[c] / wc:7.2f} % of SQ_WAVE_CYCLES")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
