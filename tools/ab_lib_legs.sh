#!/bin/bash
# same-box A/B of two builds of the library on the FLAME / rotation legs:  bash tools/ab_lib_legs.sh <libA.so> <libB.so>
cd ${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2
for rep in 1 2; do
  for lib in $A $B; do
    MSMD_LIB=$lib MSMD_LIB_ALLOW_MISSING=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-two-streams-leg --no-roofline --legs lbs,rot 2>/dev/null |
      python -c "
import json,sys
d=json.loads(sys.stdin.read())['legs']
print('$(basename $lib)', {k: (v.get('ms'), v.get('gb_per_s')) for k, v in d.items() if k.startswith('lbs')})
r = d.get('rotations_landmarks', d)
print('   ', {k: (v.get('gb_per_s') if isinstance(v, dict) else v) for k, v in (r.items() if isinstance(r, dict) else [])})"
  done
done
