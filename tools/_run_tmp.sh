cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "persistent or bit_identical or gemm_ln" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
bash tools/ab_env.sh MSMD_GEMM_ONE_TILE=1 bf16
