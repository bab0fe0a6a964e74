#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s6
O=gpurun_out/s6
L=$PWD/ubisoft-laforge-msmd_amd/csrc
PARTNERS=convgemm,attn,ln,fe,enc timeout 900 python tools/conv0_corun.py 200 2>&1 | grep -v amdgpu.ids > $O/corun_product.log
MSMD_LIB=$L/libmsmd_hip_nopk.so PARTNERS=fe,enc timeout 900 python tools/conv0_corun.py 200 2>&1 | grep -v amdgpu.ids > $O/corun_nopk.log
# whole-library "no packed fp32" build: parity + speed
MSMD_LIB=$L/libmsmd_hip_nopk.so timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_split_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -3 > $O/pytest_nopk.log
for lib in libmsmd_hip.so libmsmd_hip_nopk.so libmsmd_hip.so libmsmd_hip_nopk.so; do
  MSMD_LIB=$L/$lib timeout 600 python bench.py --steps 30 --warmup 5 --legs none --no-roofline --no-cpu-baseline --no-two-streams-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['dtype'], d['ms_per_step'], [(p['dtype'], p['ms_per_step']) for p in d.get('parity_mode', [])])" >> $O/bench_ab.log
done
MSMD_LIB=$L/libmsmd_hip_nopk.so REPS=100 timeout 600 python tools/concurrent_pattern.py feat 2>&1 | grep -v "^priority\|amdgpu.ids" | tail -3 > $O/pattern_nopk.log
cat $O/*.log
