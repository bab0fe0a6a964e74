#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s5
O=gpurun_out/s5
KEEP=1 timeout 900 python tools/concurrent_trace.py 150 2>&1 | grep -v amdgpu.ids > $O/trace_product.log
MSMD_LIB=$PWD/ubisoft-laforge-msmd_amd/csrc/libmsmd_hip_noslp.so KEEP=1 timeout 900 python tools/concurrent_trace.py 150 2>&1 | grep -v amdgpu.ids > $O/trace_noslp.log
MSMD_LIB=$PWD/ubisoft-laforge-msmd_amd/csrc/libmsmd_hip_noslp.so KEEP=0 STAGE=feat timeout 900 python tools/concurrent_trace.py 150 2>&1 | grep -v amdgpu.ids > $O/trace_noslp_feat.log
cat $O/*.log
