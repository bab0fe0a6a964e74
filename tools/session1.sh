#!/bin/bash
# GPU session 1 (round 3): does write-through (sc1) output storing cure the two-queue GEMM hazard?  + DP rehearsal + cost
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s1
O=gpurun_out/s1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_split_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
for cfg in "0 0" "0 1" "0 3" "13 0"; do
  set -- $cfg
  VARIANT=$1 FLAGS=$2 REPS=80 timeout 600 python tools/concurrent_pattern.py feat 2>&1 | grep -v "^priority" | tail -4 >> $O/pattern.log
done
QUIET_TWICE=1 timeout 900 python tools/dp_sidestream_check.py graph 40 2>&1 | tail -6 > $O/dp_quiet.log
timeout 1200 python tools/dp_sidestream_check.py graph 150 2>&1 | tail -12 > $O/dp_graph_default.log
FLAGS=1 timeout 1200 python tools/dp_sidestream_check.py graph 150 2>&1 | tail -12 > $O/dp_graph_wt.log
SIDE_GEMM=1 timeout 1200 python tools/dp_sidestream_check.py graph 150 2>&1 | tail -12 > $O/dp_graph_default_sidegemm.log
for f in 0 1 2 3; do
  MSMD_GEMM_FLAGS=$f timeout 600 python bench.py --steps 30 --warmup 5 --legs none --no-parity --no-roofline --no-cpu-baseline --no-two-streams-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags $f', d['ms_per_step'], d['value'], d.get('max_abs_err_vs_oracle'))" >> $O/bench_flags.log
done
cat $O/pytest.log $O/pattern.log $O/dp_*.log $O/bench_flags.log
