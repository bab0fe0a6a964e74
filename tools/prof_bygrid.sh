#!/bin/bash
# rocprofv3 --kernel-trace of one bench.py configuration, aggregated per (kernel, grid size): the per-shape launch time
# of the GEMMs.  Usage: bash tools/prof_bygrid.sh <tag> <bench args...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o p -- python3 $ROOT/bench.py "$@" > /tmp/prof_$TAG.log 2>&1
f=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$ROOT/gpurun_out/${TAG}_bygrid.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    g = (r["Kernel_Name"][:90], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[g].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
with open(sys.argv[2], "w") as f:
    f.write("kernel,wgs_x,grid_y,grid_z,calls,avg_us,min_us,total_ms\n")
    for (k, x, y, z), v in rows:
        f.write(f'"{k}",{x},{y},{z},{len(v)},{sum(v)/len(v)/1e3:.2f},{min(v)/1e3:.2f},{sum(v)/1e6:.3f}\n')
for (k, x, y, z), v in rows[:30]:
    print(f"{sum(v)/1e6:8.3f} ms calls {len(v):5d} avg {sum(v)/len(v)/1e3:8.2f} min {min(v)/1e3:8.2f} us  wgs {x}x{y}x{z}  {k[:70]}")
PY
