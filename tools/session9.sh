#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s9
O=gpurun_out/s9
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_gpu.log
timeout 900 python bench.py --steps 30 --warmup 5 --legs none --no-cpu-baseline --no-two-streams-leg 2>/dev/null | tail -1 > $O/bench.json
python - <<'PY' >> $O/pytest_gpu.log
import json
d = json.load(open("gpurun_out/s9/bench.json"))
print("bf16", d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], [(p["dtype"], p["ms_per_step"], p["roofline"]["achieved"]) for p in d["parity_mode"]])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s9 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --legs none --no-cpu-baseline --no-two-streams-leg --no-parity --no-roofline > /tmp/prof_s9.log 2>&1
cp $(find /tmp/prof_s9 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/bf16_kernel_stats.csv
cd $GRAFT_REPO_ROOT; python tools/show_stats.py $O/bf16_kernel_stats.csv 14 23 >> $O/pytest_gpu.log
cat $O/pytest_gpu.log
