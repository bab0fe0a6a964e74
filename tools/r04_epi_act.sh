#!/bin/bash
# round 4: the activation as a constant of the GEMM kernel: parity tests, bench lines
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|FAILED" | tail -5
python bench.py --no-cpu-baseline > gpurun_out/epi_act_bench.json 2> gpurun_out/epi_act_bench.err; echo "rc=$?"
python bench.py --mode train --no-cpu-baseline > gpurun_out/epi_act_train.json 2> gpurun_out/epi_act_train.err; echo "rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/epi_act_bench.json"))
print(d["ms_per_step"], d["value"], d.get("max_abs_err_vs_oracle"), d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["tile_192x128"]["frac"])
print([(p["dtype"], p["ms_per_step"]) for p in d.get("parity_mode", [])], d.get("forward_two_streams", {}).get("ms_per_step"))
for k, v in d["legs"].items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "ms", "hbm_frac", "mfma_frac", "frames_per_s")})
t = json.load(open("gpurun_out/epi_act_train.json"))
print(t["ms_per_step"], t["value"], t.get("exchange_rehearsal_world1", {}).get("ms_per_step_with_exchange"))
PY
