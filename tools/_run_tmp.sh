cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 2400 python -m pytest tests/test_train_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | tail -8
for bw in 0 1 0 1; do
  MSMD_TRAIN_BATCH_WINDOWS=$bw python bench.py --mode train --no-cpu-baseline --no-exchange-rehearsal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch_windows=$bw', d['ms_per_step'], d['value'], d.get('config',{}).get('kernels_per_step'))"
done
