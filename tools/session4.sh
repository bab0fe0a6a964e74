#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s4
O=gpurun_out/s4
timeout 600 python tools/conv0_corun.py 300 2>&1 | grep -v amdgpu.ids > $O/corun_product.log
MSMD_LIB=$PWD/ubisoft-laforge-msmd_amd/csrc/libmsmd_hip_noslp.so timeout 600 python tools/conv0_corun.py 300 2>&1 | grep -v amdgpu.ids > $O/corun_noslp.log
ODT=fp32 timeout 600 python tools/conv0_corun.py 200 2>&1 | grep -v amdgpu.ids > $O/corun_product_fp32.log
MSMD_LIB=$PWD/ubisoft-laforge-msmd_amd/csrc/libmsmd_hip_noslp.so REPS=80 timeout 600 python tools/concurrent_pattern.py enc 2>&1 | grep -v "^priority\|amdgpu.ids" | tail -3 > $O/pattern_noslp.log
cat $O/*.log
