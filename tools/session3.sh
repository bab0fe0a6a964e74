#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s3
O=gpurun_out/s3
KEEP=1 timeout 900 python tools/concurrent_trace.py 100 2>&1 | grep -v amdgpu.ids > $O/trace_keep.log
KEEP=0 timeout 900 python tools/concurrent_trace.py 100 2>&1 | grep -v amdgpu.ids > $O/trace_nokeep.log
cat $O/trace_keep.log $O/trace_nokeep.log
