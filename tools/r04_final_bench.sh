#!/bin/bash
# round 4: the two complete bench lines (default forward line with every leg; --mode train with the RCCL rehearsal leg)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench_full.json 2> gpurun_out/r04_bench_full.err; echo "forward rc=$? stdout lines=$(wc -l < gpurun_out/r04_bench_full.json)"
python bench.py --mode train > gpurun_out/r04_bench_train_full.json 2> gpurun_out/r04_bench_train.err; echo "train rc=$? stdout lines=$(wc -l < gpurun_out/r04_bench_train_full.json)"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_full.json"))
print(d["ms_per_step"], d["value"], d["max_abs_err_vs_oracle"], d["within_error_bound"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print([(p["dtype"], p["ms_per_step"]) for p in d["parity_mode"]], d["forward_two_streams"]["ms_per_step"])
for k, v in d["legs"].items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "ms", "hbm_frac", "mfma_frac", "frames_per_s")} or str(v)[:300])
t = json.load(open("gpurun_out/r04_bench_train_full.json"))
print(t["ms_per_step"], t["value"], t.get("exchange_rehearsal_world1"))
PY
