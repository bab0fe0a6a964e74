#!/bin/bash
# same-box A/B of two builds of the library on the bench step:  bash tools/ab_lib.sh <libA.so> <libB.so> [dtypes...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; shift 2
for dt in ${@:-bf16 f16x2}; do
  for rep in 1 2 3; do
    for lib in $A $B; do
      MSMD_LIB=$lib python bench.py --dtype $dt --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-two-streams-leg --legs none 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$dt', '$(basename $lib)', d['ms_per_step'], d['roofline']['frac'])"
    done
  done
done
