#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s11
O=gpurun_out/s11
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "^E|assert|passed|failed|FAILED" | head -20 > $O/pytest_gpu.log
timeout 1200 python tools/dp_sidestream_check.py graph 200 2>&1 | tail -4 > $O/dp_graph.log
timeout 1200 python tools/dp_sidestream_check.py eager 30 2>&1 | tail -4 > $O/dp_eager.log
cat $O/pytest_gpu.log $O/dp_*.log
