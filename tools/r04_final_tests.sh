#!/bin/bash
# round 4 close-out: the full GPU suite, then the HuBERT-large leg's kernel stats again (the 192-row tile back at two workgroups per CU)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 > gpurun_out/r04_pytest_gpu_final.txt; cat gpurun_out/r04_pytest_gpu_final.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_h
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_h -o p -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-two-streams-leg --legs hubert --steps 2 --warmup 1 --no-roofline > /tmp/pr_h.log 2>&1 || echo "profiler run failed"
f=$(find /tmp/pr_h -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $ROOT/gpurun_out/r04/r04_hubert_large_10s_b32_kernel_stats.csv && python3 $ROOT/tools/show_stats.py $f 8 | head -10
grep '"metric"' /tmp/pr_h.log | tail -1 > $ROOT/gpurun_out/r04/r04_hubert_large_10s_b32_line.json
python3 -c "
import json; d=json.load(open('$ROOT/gpurun_out/r04/r04_hubert_large_10s_b32_line.json')); print(d['legs']['hubert_large_10s_b32'])"
