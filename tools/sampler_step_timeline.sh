#!/bin/bash
# Per-kernel totals of ONE denoising step of the fp16 sampler (B = 64, 3 CFG entries) from a rocprofv3 kernel trace of tools/bench_sampler.py
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tls
T=${T:-40} DTYPE=${DTYPE:-fp16} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tls -o p -- python3 $ROOT/tools/bench_sampler.py 64 > /tmp/tls.log 2>&1 || echo "profiler run failed"
tail -3 /tmp/tls.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/tls/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "step_select" in r[2]]
if len(starts) < 3:
    print(len(rows), "kernel rows;", collections.Counter(r[2][:60] for r in rows).most_common(12))
    raise SystemExit(1)
a, b = starts[-3], starts[-2]
step = rows[a:b]
wall = (rows[b][0] - step[0][0]) / 1e3
busy = sum(r[1] - r[0] for r in step) / 1e3
print(f"{len(step)} kernels in one denoising step: wall {wall:.1f} us, kernel time {busy:.1f} us")
by = collections.defaultdict(lambda: [0.0, 0])
for s, e, n in step:
    k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:95]
    by[k][0] += (e - s) / 1e3; by[k][1] += 1
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"  {t:8.1f} us  x{c:3d}  avg {t / c:7.2f}  {k}")
PY
