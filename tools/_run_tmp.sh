cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "persistent or gemm_ln or bit_identical" 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
AB_ARGS="--legs hubert" bash tools/ab_env.sh MSMD_GEMM_ONE_TILE=1 bf16
