#!/bin/bash
# round 4: v4 GEMM kernels against the routed ones (variant per call), isolated and hot
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/bench_gemm_variants.py ${1:-17,15,60,61,62,63,64} 20 > gpurun_out/gemm_ab.txt 2>&1
tail -12 gpurun_out/gemm_ab.txt
