#!/usr/bin/env python
"""Per-launch durations of the decode step from a rocprofv3 kernel trace (rocpd sqlite), by LAYER index and by STEP: where does the
avg - min gap of the small launches (qkv, attention, o_proj) come from - an address (layer) dependence or a context dependence?
usage: r05_decode_gaps.py <db> [out.txt]"""
import re, sqlite3, subprocess, sys
import numpy as np

def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", o)).replace("void ", "") for o in out]

def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    uniq = sorted({r[0] for r in rows})
    dm = dict(zip(uniq, demangle(uniq)))
    ev = [(dm[n], s, e) for n, s, e in rows]
    HEAD = "gemv_rowwave_kernel<4, 3, 1, 1, 0>"
    LAYER = ["gemv_rowwave_kernel<2, 3, 1, 1, 1024>", "attn_decode_pagesplit_kernel", "gemv_rowwave_kernel<2, 3, 1, 3, 8>",
             "gemv_rowwave_kernel<4, 3, 1, 1, 16>", "gemv_splitk_kernel"]
    SHORT = ["qkv", "attn", "o_proj", "gate_up", "down"]
    # steps: the 140 layer launches that precede every lm_head launch
    steps = []
    for i, (n, s, e) in enumerate(ev):
        if n.startswith(HEAD) and i >= 140:
            blk = ev[i - 140:i]
            ok = all(blk[5 * l + k][0].startswith(LAYER[k]) for l in range(28) for k in range(5))
            if ok:
                steps.append((blk, ev[i], ev[i - 141] if i >= 141 else None))
    lines = [f"{len(steps)} decode steps found in {db}"]
    if not steps:
        names = {}
        for n, s, e in ev: names[n] = names.get(n, 0) + 1
        lines += [f"{c:7d} {n}" for n, c in sorted(names.items(), key=lambda x: -x[1])[:30]]
        txt = "\n".join(lines); print(txt)
        if out: open(out, "w").write(txt + "\n")
        return
    S = len(steps)
    dur = np.zeros((S, 28, 5)); gap = np.zeros((S, 28, 5))
    for si, (blk, head, prev) in enumerate(steps):
        for l in range(28):
            for k in range(5):
                n, s, e = blk[5 * l + k]
                dur[si, l, k] = (e - s) / 1e3
                pe = blk[5 * l + k - 1][2] if (l or k) else (prev[2] if prev else s)
                gap[si, l, k] = (s - pe) / 1e3
    lines.append("us per launch (rocprofv3 begin -> end), all steps x layers:   " + "  ".join(f"{n:>9s}" for n in SHORT))
    for label, fn in (("avg", np.mean), ("min", np.min), ("p10", lambda a, axis=None: np.percentile(a, 10, axis=axis)), ("median", np.median),
                      ("p90", lambda a, axis=None: np.percentile(a, 90, axis=axis)), ("max", np.max)):
        lines.append(f"  {label:>8s} duration                                              " + "  ".join(f"{fn(dur[:, :, k]):9.2f}" for k in range(5)))
    lines.append(f"  {'avg':>8s} gap to the previous launch's end                       " + "  ".join(f"{gap[:, :, k].mean():9.2f}" for k in range(5)))
    lines.append(f"  sum of durations per layer {dur.sum(axis=2).mean():.2f} us, sum of gaps per layer {gap.sum(axis=2).mean():.2f} us; "
                 f"step wall (first qkv start -> lm_head end) {np.mean([(h[2] - b[0][1]) / 1e3 for b, h, _ in steps]):.1f} us, lm_head {np.mean([(h[2] - h[1]) / 1e3 for _, h, _ in steps]):.1f} us")
    lines.append("")
    lines.append("by LAYER (average over the steps): duration | gap")
    lines.append("layer  " + "  ".join(f"{n:>8s}" for n in SHORT) + "   |  " + "  ".join(f"{n:>8s}" for n in SHORT))
    for l in range(28):
        lines.append(f"{l:5d}  " + "  ".join(f"{dur[:, l, k].mean():8.2f}" for k in range(5)) + "   |  " + "  ".join(f"{gap[:, l, k].mean():8.2f}" for k in range(5)))
    lines.append("")
    lines.append("by STEP (average over the layers), every 16th step: duration | gap   (context grows by one token per step)")
    for si in list(range(0, S, 16)) + [S - 1]:
        lines.append(f"{si:5d}  " + "  ".join(f"{dur[si, :, k].mean():8.2f}" for k in range(5)) + "   |  " + "  ".join(f"{gap[si, :, k].mean():8.2f}" for k in range(5)))
    lines.append("")
    lines.append("spread inside ONE (layer, kernel) cell over the steps vs between cells: std of the per-layer means / mean of the per-layer stds")
    for k in range(5):
        lines.append(f"  {SHORT[k]:>8s}: between layers {dur[:, :, k].mean(axis=0).std():.3f} us, within a layer {dur[:, :, k].std(axis=0).mean():.3f} us, "
                     f"between steps {dur[:, :, k].mean(axis=1).std():.3f} us")
    # duration vs gap correlation: does a launch that starts right behind its predecessor (small gap) run longer (overlapped ramp)?
    for k in range(5):
        c = np.corrcoef(dur[:, :, k].ravel(), gap[:, :, k].ravel())[0, 1]
        lines.append(f"  {SHORT[k]:>8s}: corr(duration, gap before it) = {c:+.2f}; duration + gap = {np.mean(dur[:, :, k] + gap[:, :, k]):.2f} us (std {np.std(dur[:, :, k] + gap[:, :, k]):.2f})")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
