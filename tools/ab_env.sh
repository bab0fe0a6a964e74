#!/bin/bash
# same-box A/B of an environment switch on the bench step:  bash tools/ab_env.sh VAR=1 [dtypes...]   (A = with VAR, B = without)
cd ${GRAFT_REPO_ROOT:-/root/repo}
SW=$1; shift
ARGS=${AB_ARGS:---legs none}
for dt in ${@:-bf16}; do
  for rep in 1 2 3; do
    for on in 1 0; do
      if [ $on = 1 ]; then export $SW; else unset ${SW%%=*}; fi
      python bench.py --dtype $dt --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-two-streams-leg $ARGS 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$dt', '$SW' if $on else 'default', d['ms_per_step'], d['roofline']['frac'], {k: v.get('ms_per_step') for k, v in d.get('legs', {}).items()})"
    done
  done
done
