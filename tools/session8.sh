#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s8
O=gpurun_out/s8
timeout 1200 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "side_stream" 2>&1 | grep -v Warning | tail -40 > $O/pytest_dp.log
cat $O/pytest_dp.log
