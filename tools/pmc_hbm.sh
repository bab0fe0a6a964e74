#!/bin/bash
# HBM traffic per kernel of the forward bench step: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (rocprofv3
# guide), each with --kernel-trace only; per-kernel averages merged into gpurun_out/<tag>_pmc_hbm_fetch_write_per_kernel.json.
# Usage: [MSMD_TUNE=7=1] bash tools/pmc_hbm.sh <tag> [bench dtype ...]      (tuning keys, if any, come from the environment)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-cur}; shift
DTYPES=${@:-bf16}
mkdir -p $ROOT/gpurun_out
for dt in $DTYPES; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${TAG}_${dt}_$c
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${TAG}_${dt}_$c -o p -- python3 $ROOT/bench.py --eager --steps 3 --warmup 1 --dtype $dt --no-cpu-baseline --no-roofline --no-parity --legs none > /tmp/pmc_hbm.log 2>&1 || echo "pass $dt $c failed or timed out"
  done
done
python3 - "$TAG" "$ROOT/gpurun_out/${TAG}_pmc_hbm_fetch_write_per_kernel.json" $DTYPES <<'PY'
import csv, sys, json, collections, glob
tag, out, dts = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for dt in dts:
    for c, key in (("FETCH_SIZE", "fetch_kb_avg"), ("WRITE_SIZE", "write_kb_avg")):
        fs = glob.glob(f"/tmp/pmc_{tag}_{dt}_{c}/**/*counter_collection.csv", recursive=True)
        if not fs: continue
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(fs[0])):
            a = agg[r["Kernel_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, v in agg.items():
            e = res.setdefault(k, {})
            e[key] = v[0] / v[1]; e["launches"] = max(e.get("launches", 0), v[1])
json.dump(res, open(out, "w"), indent=0)
for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("fetch_kb_avg", 0) * 2 + kv[1].get("write_kb_avg", 0)) * kv[1]["launches"])[:8]:
    print(f'{k[:90]:90s} fetch {v.get("fetch_kb_avg", 0):10.1f} KB  write {v.get("write_kb_avg", 0):10.1f} KB  x{v["launches"]}')
PY
