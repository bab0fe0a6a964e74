cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s17
timeout 600 python tools/bench_gemm_variants.py 17,41 20 2>&1 | grep -v amdgpu.ids > gpurun_out/s17/gemm.log
timeout 900 python tools/ab_forward.py base=pass 'stag=ops.GEMM_ROUTER=lambda M,N,K,b: 41 if (N>64 and ((M+127)//128)*((N+127)//128)*b>=192 and K%64==0) else None' base2=pass 2>&1 | grep -v amdgpu.ids >> gpurun_out/s17/gemm.log
cat gpurun_out/s17/gemm.log
