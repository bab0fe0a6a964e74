#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s7
O=gpurun_out/s7
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
