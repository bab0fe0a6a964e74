#!/bin/bash
# same-box A/B of two builds of the library on the sampler loop:  bash tools/ab_lib_sampler.sh <libA.so> <libB.so>   (env LANES, T, B)
cd ${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2
for rep in 1 2; do
  for lib in $A $B; do
    echo "== $(basename $lib)"
    MSMD_LIB=$lib MSMD_LIB_ALLOW_MISSING=1 python tools/ab_sampler_lanes.py 2>&1 | grep "ms/step"
  done
done
