#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python tool: bash tools/prof_cmd.sh <tag> tools/x.py [args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
SCRIPT=$ROOT/$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python3 $SCRIPT "$@" > /tmp/prof_$TAG.log 2>&1
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
mkdir -p $ROOT/gpurun_out
[ -n "$f" ] && cp "$f" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
python3 - "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:100]}')
PY
