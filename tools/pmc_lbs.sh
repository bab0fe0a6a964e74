#!/bin/bash
# Write-path PMC counters of the skinning kernel (one --pmc pass per group, --kernel-trace only, 3 launches each).
# Usage: bash tools/pmc_lbs.sh [frames]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-25600}
cd /tmp && export TMPDIR=/tmp
for grp in "$@"; do :; done
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_WRITE_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_WRREQ_STALL_sum TA_FLAT_WRITE_WAVEFRONTS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pl_$tag
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pl_$tag -o p -- python3 $ROOT/tools/lbs_once.py $N 3 > /tmp/pl.log 2>&1 || { echo "pass [$grp] failed/timeout"; tail -2 /tmp/pl.log; continue; }
  f=$(find /tmp/pl_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "lbs_" not in k: continue
    a = agg[(k[:40], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in agg.items(): print(f"{k[0]:42s} {k[1]:32s} avg/dispatch {v[0]/v[1]:16.1f}  ({v[1]} dispatches)")
PY
done
