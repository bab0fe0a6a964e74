cd ${GRAFT_REPO_ROOT:-/root/repo}
SHAPES=encoder MSMD_LIB=ubisoft-laforge-msmd_amd/csrc/libmsmd_hip_exp.so python tools/bench_gemm_variants.py 17,56,11,50,7,47,49 20 2>&1 | grep -v amdgpu | cut -c1-330
