#!/bin/bash
# Round-6 measurement pass: rocprofv3 summaries behind every number bench.py prints.  Everything lands in gpurun_out/r06/
# (copy what is to be judged into profiles/).  One profiler run per leg; counters in their own runs (--kernel-trace only).
#   bash tools/prof_r06.sh [stats|pmc|timelines|all]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
WHAT=${1:-all}
O=$ROOT/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="--steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-two-streams-leg"
stats() {   # stats <tag> <program args...>: kernel-trace + stats of one command -> $O/r06_<tag>_kernel_stats.csv
  local tag=$1; shift
  rm -rf /tmp/pr_$tag
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$tag -o p -- python3 "$@" > /tmp/pr_$tag.log 2>&1 || echo "$tag: profiler run failed"
  f=$(find /tmp/pr_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/r06_${tag}_kernel_stats.csv && python3 $ROOT/tools/show_stats.py $f 6 | head -8
  grep '"metric"' /tmp/pr_$tag.log | tail -1 > $O/r06_${tag}_line.json
}
pmc_prog() {   # pmc_prog <tag> <program args...>: FETCH_SIZE / WRITE_SIZE per kernel (separate passes) -> $O/r06_pmc_<tag>_fetch_write_per_kernel.json
  local tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcp_${tag}_$c
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcp_${tag}_$c -o p -- python3 "$@" > /tmp/pmcp.log 2>&1 || echo "pmc pass $tag $c failed"
  done
  python3 - "$tag" "$O/r06_pmc_${tag}_fetch_write_per_kernel.json" <<'PY'
import csv, sys, json, collections, glob
tag, out = sys.argv[1], sys.argv[2]
res = {}
for c, key in (("FETCH_SIZE", "fetch_kb_avg"), ("WRITE_SIZE", "write_kb_avg")):
    fs = glob.glob(f"/tmp/pmcp_{tag}_{c}/**/*counter_collection.csv", recursive=True)
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
}
if [ "$WHAT" = "stats" ] || [ "$WHAT" = "all" ]; then
  stats bench_b32_bf16 $ROOT/bench.py $C --legs none
  stats bench_b32_f16x2 $ROOT/bench.py $C --legs none --dtype f16x2
  stats bench_b32_fp32 $ROOT/bench.py $C --legs none --dtype fp32 --steps 5
  MSMD_BENCH_LBS=6400 stats lbs_6400 $ROOT/bench.py $C --legs lbs --steps 2 --warmup 1 --no-roofline
  MSMD_BENCH_LBS=25600 stats lbs_25600 $ROOT/bench.py $C --legs lbs --steps 2 --warmup 1 --no-roofline
  MSMD_BENCH_LBS=25600_fp16 stats lbs_25600_fp16_vertices $ROOT/bench.py $C --legs lbs --steps 2 --warmup 1 --no-roofline
  MSMD_BENCH_LBS=shape stats lbs_25600_per_frame_shape $ROOT/bench.py $C --legs lbs --steps 2 --warmup 1 --no-roofline
  stats sampler_b64_t500 $ROOT/bench.py $C --legs sampler --steps 2 --warmup 1 --no-roofline
  MSMD_SAMPLER_LANES=1 stats sampler_b64_t500_one_lane $ROOT/bench.py $C --legs sampler --steps 2 --warmup 1 --no-roofline
  stats train_step_b32 $ROOT/bench.py $C --legs train --steps 2 --warmup 1 --no-roofline
  stats hubert_large_10s_b32 $ROOT/bench.py $C --legs hubert --steps 2 --warmup 1 --no-roofline
  stats rotations_landmarks_attention $ROOT/bench.py $C --legs rot,att --steps 2 --warmup 1 --no-roofline
  timeout 900 python3 $ROOT/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > /tmp/train_line.log 2>&1; grep '"metric"' /tmp/train_line.log | tail -1 > $O/r06_bench_train_full.json
fi
if [ "$WHAT" = "timelines" ] || [ "$WHAT" = "all" ]; then
  TOP=30 bash $ROOT/tools/step_timeline.sh > $O/r06_forward_step_timeline.txt 2>&1
  MSMD_SAMPLER_LANES=1 DTYPE=fp16 T=50 PROG=tools/bench_sampler.py DELIM=step_select TOP=30 bash $ROOT/tools/step_timeline.sh 64 > $O/r06_sampler_step_timeline.txt 2>&1
  DELIM=adam_kernel TOP=30 bash $ROOT/tools/step_timeline.sh --mode train --no-exchange-rehearsal > $O/r06_train_step_timeline.txt 2>&1
fi
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  bash $ROOT/tools/pmc_hbm.sh r06 bf16 f16x2 | tail -9
  cp $ROOT/gpurun_out/r06_pmc_hbm_fetch_write_per_kernel.json $O/ 2>/dev/null
  pmc_prog sampler $ROOT/tools/sampler_once.py 32 3
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rm -rf /tmp/pm_$tag
    timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm_$tag -o p -- python3 $ROOT/bench.py --eager --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --legs none > /tmp/pm.log 2>&1 || echo "pass [$grp] failed"
  done
  python3 - $O/r06_pmc_insitu_mfma_lds_per_kernel.json <<'PY'
import csv, glob, json, sys, collections
res = collections.defaultdict(dict)
for f in glob.glob("/tmp/pm_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        a = agg[(r["Kernel_Name"], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), v in agg.items():
        res[k][c + "_avg_per_dispatch"] = v[0] / v[1]; res[k]["dispatches"] = v[1]
json.dump(res, open(sys.argv[1], "w"), indent=0)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES_avg_per_dispatch", 0) * kv[1]["dispatches"])[:6]:
    print(k[:80], {c: round(x) for c, x in v.items()})
PY
  bash $ROOT/tools/pmc_lbs.sh 25600 > $O/r06_pmc_lbs_25600.txt 2>&1; tail -12 $O/r06_pmc_lbs_25600.txt
fi
ls $O
