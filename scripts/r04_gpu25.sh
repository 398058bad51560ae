#!/bin/bash
# Session 25: kernel-level durations of sample_filter_kernel per filter configuration (rocprofv3 kernel trace)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu25
mkdir -p $O
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $O/prof -o s -- python $R/scripts/sampler_prof.py > $O/prof.log 2>&1; echo "rc=$?"
cd $R
python - <<'P'
import glob, sqlite3
db = glob.glob('gpurun_out/r04_gpu25/prof/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nc = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {nc}, start, end from kernels order by start").fetchall()
names = "top_p,top_k,min_p,classic chain,top_n_sigma,p_less,typical_p,xtc,min_keep".split(",")
d = [(e - s) / 1e3 for n, s, e in rows if "sample_filter_kernel" in n]
g2 = sorted((e - s) / 1e3 for n, s, e in rows if "gumbel_partial" in n)
l = sorted((e - s) / 1e3 for n, s, e in rows if 'lse_partial' in n)
a = sorted((e - s) / 1e3 for n, s, e in rows if 'logprob_argmax_kernel' in n)
out = open('gpurun_out/r04_gpu25/sampler_kernel_us.txt', 'w')
for i, n in enumerate(names):
    g = sorted(d[20 * i:20 * i + 20])
    line = f"{n:16s} sample_filter_kernel median {g[len(g)//2]:8.1f} us  min {g[0]:8.1f}  (V = 151,936, one row, 20 launches)"
    print(line); out.write(line + "\n")
line = f"lse_partial_kernel median {l[len(l)//2]:.1f} us, logprob_argmax_kernel median {a[len(a)//2]:.1f} us, gumbel_partial_kernel median {g2[len(g2)//2]:.1f} us (min {g2[0]:.1f}, max {g2[-1]:.1f}: all survivors of the plain case)"
print(line); out.write(line + "\n")
P
rm -rf $O/prof
