cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s16
L=$PWD/ubisoft-laforge-msmd_amd/csrc/libmsmd_hip_exp.so
MSMD_LIB=$L timeout 900 python tools/ab_forward.py base=pass 'n96=ops.GEMM_ROUTER=lambda M,N,K,b: 37 if N==768 and M==6400 else None' 'n96s3=ops.GEMM_ROUTER=lambda M,N,K,b: 38 if N==768 and M==6400 else None' 'v13=ops.GEMM_ROUTER=lambda M,N,K,b: 13 if N>64 and M>=6400 else None' base2=pass 2>&1 | grep -v amdgpu.ids > gpurun_out/s16/ab.log
cat gpurun_out/s16/ab.log
