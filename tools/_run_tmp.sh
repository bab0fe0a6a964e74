cd ${GRAFT_REPO_ROOT:-/root/repo}
SHAPES=train python tools/bench_gemm_variants.py 0,17,15 20 2>&1 | grep -v amdgpu | head -4
timeout 2400 python -m pytest tests/test_train_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | tail -5
for i in 1 2; do
python bench.py --mode train --no-cpu-baseline --no-exchange-rehearsal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', d['ms_per_step'], d['value'])"
done
