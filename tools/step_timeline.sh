#!/bin/bash
# Timeline of ONE hipGraph replay of the forward bench step from a rocprofv3 kernel trace: kernel time, idle gaps between
# consecutive kernels, per-kernel totals.   bash tools/step_timeline.sh [bench.py flags, e.g. --dtype f16x2]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
# PROG=<script under the repo root> [args]: trace that program instead of bench.py (e.g. PROG=tools/bench_sampler.py DELIM=step_select ... 64)
if [ -n "$PROG" ]; then
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python3 $ROOT/$PROG "$@" > /tmp/tl.log 2>&1 || echo "profiler run failed"
else
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-two-streams-leg --no-roofline --legs none "$@" > /tmp/tl.log 2>&1 || echo "profiler run failed"
fi
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
# steps start at the conv0 moments kernel; take the LAST complete step but one (inside the timed replays)
import os
delim = os.environ.get("DELIM", "conv0_moments_partial")      # a kernel that runs once per step (training: DELIM=adam_kernel)
starts = [i for i, r in enumerate(rows) if delim in r[2]]
starts = starts[0::2] if len(starts) > 1 and starts[1] - starts[0] < 3 else starts
a, b = starts[-3], starts[-2]
step = rows[a:b]
wall = (rows[b][0] - step[0][0]) / 1e3
busy = sum(r[1] - r[0] for r in step) / 1e3
gaps = [(step[i + 1][0] - step[i][1]) / 1e3 for i in range(len(step) - 1)]
print(f"{len(step)} kernels in one replay: wall {wall:.1f} us (start to next step's start), kernel time {busy:.1f} us, idle between kernels {sum(g for g in gaps if g > 0):.1f} us "
      f"(median gap {sorted(gaps)[len(gaps) // 2]:.2f} us, overlapped {sum(-g for g in gaps if g < 0):.1f} us)")
by = collections.defaultdict(lambda: [0.0, 0])
for s, e, n, *_ in step:
    k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
    by[k][0] += (e - s) / 1e3; by[k][1] += 1
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get('TOP', '22'))]:
    print(f"  {t:8.1f} us  x{c:3d}  avg {t / c:7.2f}  {k}")
print("longest single launches:")
for s_, e_, n_, *_ in sorted(step, key=lambda r: r[0] - r[1])[:14]:
    print(f"  {(e_ - s_) / 1e3:8.1f} us  at +{(s_ - step[0][0]) / 1e3:7.1f}  {n_[:70]}")
big = sorted(((g, step[i][2][:50], step[i + 1][2][:50]) for i, g in enumerate(gaps)), reverse=True)[:3]
if os.environ.get("SEQ"):
    print("sequence (index, us, grid / workgroup, kernel):")
    for i, (s_, e_, n_, gx, wx) in enumerate(step):
        short = n_.replace("_Z12gemm2_kernelIDF16b", "gemm2<").replace("_Z11attn_kernelIDF16b", "attn<")[:46]
        print(f"  {i:3d} {(e_ - s_) / 1e3:7.1f}  {gx:>8s}/{wx:<4s} {short}")
print("largest gaps (us, after kernel -> before kernel):")
for g, x, y in big:
    print(f"  {g:6.2f}  {x} -> {y}")
PY
