#!/bin/bash
# HBM traffic per kernel of the forward bench step: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (rocprofv3
# guide), each with --kernel-trace only.  Usage: bash tools/pmc_hbm.sh <tag> [extra env for bench, e.g. MSMD_TUNE=7=1]
cd /tmp && export TMPDIR=/tmp
TAG=${1:-cur}
OUT=/root/repo/gpurun_out/pmc_hbm_$TAG; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_hbm_${TAG}_$c -o p -- python3 /root/repo/bench.py --eager --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc_hbm.log 2>&1
  f=$(find /tmp/pmc_hbm_${TAG}_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$OUT/$c.json" <<'PY'
import csv, sys, json, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    a = agg[r["Kernel_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
json.dump({k: {"kb_avg": v[0] / v[1], "launches": v[1]} for k, v in agg.items()}, open(sys.argv[2], "w"), indent=0)
PY
done
ls $OUT
