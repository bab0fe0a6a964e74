#!/bin/bash
# Kernel timeline of steady-state training steps (hipGraph replays): span vs busy time per step and the per-step top
# kernels, from rocprofv3 --kernel-trace of `bench.py --mode train`.  Usage: bash tools/prof_train_timeline.sh [tag]
TAG=${1:-train}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptl_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl_$TAG -o p -- python3 $ROOT/bench.py --mode train --steps 12 --warmup 2 > /tmp/ptl_$TAG.log 2>&1
tail -1 /tmp/ptl_$TAG.log | cut -c1-300
f=$(find /tmp/ptl_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$ROOT/gpurun_out/${TAG}_step_kernels.csv" <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
segs = [(adam[k], adam[k + 1]) for k in range(len(adam) - 1)][-8:]
tot = collections.defaultdict(lambda: [0, 0])
spans, busys, counts, gaps = [], [], [], []
for a, b in segs:
    seg = rows[a:b]
    spans.append(seg[-1][1] - seg[0][0]); busys.append(sum(e - s for s, e, _ in seg)); counts.append(len(seg))
    # idle = time when no kernel runs (union of intervals)
    cur_e, idle = seg[0][1], 0
    for s, e, _ in seg[1:]:
        if s > cur_e: idle += s - cur_e
        cur_e = max(cur_e, e)
    gaps.append(idle)
    for s, e, n in seg:
        tot[n][0] += e - s; tot[n][1] += 1
n = len(segs)
print(f"steady-state steps analysed: {n}; span {sum(spans)/n/1e6:.2f} ms, sum of kernel durations {sum(busys)/n/1e6:.2f} ms, idle (no kernel running) {sum(gaps)/n/1e6:.2f} ms, kernels per step {sum(counts)/n:.0f}")
with open(sys.argv[2], "w") as f:
    f.write("kernel,calls_per_step,us_per_step,avg_us\n")
    for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        f.write(f'"{k[:120]}",{c/n:.1f},{t/n/1e3:.1f},{t/c/1e3:.2f}\n')
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:32]:
    print(f"{t/n/1e3:9.1f} us/step  x{c/n:6.1f}  avg {t/c/1e3:8.2f} us  {k[:100]}")
PY
