#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc run (rocpd sqlite) into per-kernel HBM traffic per launch.

hbm_bytes = (k * FETCH_SIZE + WRITE_SIZE) * 1024 with k = 2: on gfx950 this rocprofv3 reports exactly half the
bytes of a wide coalesced streaming read in FETCH_SIZE (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported.
FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 ("exceeds the capabilities of the hardware"), so they come
from two runs of the same command; pass both databases.
The output carries `_meta.decode_csrc_sha16` = the hash bench.py computes over the decode step's kernel sources
(bench.decode_csrc_sha16): bench.py refuses the file as stale when the sources have changed since the passes.
usage: pmc_summary.py <out.json> <db> [<db> ...]"""
import json
import re
import sqlite3
import subprocess
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "").strip()


def main(out, *dbs):
    res = {}
    for db in dbs:
        con = sqlite3.connect(db)
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                           "group by kernel_name, counter_name").fetchall()
        for name, ctr, n, avg, tot in rows:
            d = res.setdefault(short(name), {"launches": n})
            d[ctr] = avg
    for k, d in res.items():
        f, w = d.get("FETCH_SIZE"), d.get("WRITE_SIZE")
        if f is not None and w is not None:
            d["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
            d["fetch_kb_raw"], d["write_kb_raw"] = f, w
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import decode_csrc_sha16
    res["_meta"] = {"decode_csrc_sha16": decode_csrc_sha16(), "kernels": sorted(k for k in res if not k.startswith(("at::", "__amd")))}
    txt = json.dumps(res, indent=1, sort_keys=True)
    if out:
        open(out, "w").write(txt + "\n")
    for k, d in sorted(((k, d) for k, d in res.items() if k != "_meta"), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0) * kv[1]["launches"])[:16]:
        print(f"{k[:70]:70s} n={d['launches']:6d} hbm/launch {d.get('hbm_bytes_per_launch', 0) / 1e6:10.3f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:])
