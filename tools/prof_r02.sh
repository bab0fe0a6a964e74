#!/bin/bash
# Round-2 rocprofv3 --kernel-trace --stats summaries of every bench.py leg (one profiler run per leg); results under
# gpurun_out/<tag>_kernel_stats.csv, to be copied into profiles/.   Usage: bash tools/prof_r02.sh <tag-prefix>
P=${1:-r02}
D=$(dirname "$0")
C="--steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-two-streams-leg"
bash $D/prof_bench.sh ${P}_bench_b32_bf16 $C --legs none | tail -3
bash $D/prof_bench.sh ${P}_bench_b32_f16x2 $C --legs none --dtype f16x2 | tail -3
bash $D/prof_bench.sh ${P}_bench_b32_fp32 $C --legs none --dtype fp32 --steps 5 | tail -3
bash $D/prof_bench.sh ${P}_lbs_6400_25600 $C --legs lbs --steps 2 --warmup 1 --no-roofline | head -6
bash $D/prof_bench.sh ${P}_sampler_b64_t500 $C --legs sampler --steps 2 --warmup 1 --no-roofline | head -4
bash $D/prof_bench.sh ${P}_train_step_b32 $C --legs train --steps 2 --warmup 1 --no-roofline | head -4
bash $D/prof_bench.sh ${P}_hubert_large_10s_b32 $C --legs hubert --steps 2 --warmup 1 --no-roofline | head -4
