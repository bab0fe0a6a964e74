#!/bin/bash
# Rehearsal of bench.py's multi-rank control flow on ONE GPU with N gloo ranks on cuda:0 (default 8; the real launch is RCCL, one
# device per rank): barriers, max-over-ranks timing, rank-0-only printing, the bucket reducer's exchange inside segmented
# hipGraphs, bucket order with LayerDrop, the all-rank non-finite stop and the rank-0 broadcast -- with N real processes
# touching the GPU path.  Numbers are meaningless (the processes share the device); it must not hang and must print one line.
#   bash tools/bench_ranks_one_gpu.sh [N=8] [modes="train"]      (BATCH=8 clips per rank: at the real 32 eight ranks need 8 x 36 GB)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
N=${1:-8}; MODES=${2:-train}
export MSMD_DIST_BACKEND=gloo MSMD_ONE_DEVICE=1
for mode in $MODES; do
  SECONDS=0
  timeout 3000 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus $N --steps 2 --warmup 1 --mode $mode --batch ${BATCH:-8} --no-cpu-baseline > /tmp/ranks_$mode.log 2>&1
  echo "mode $mode: exit $? after $SECONDS s"
  grep '"metric"' /tmp/ranks_$mode.log | cut -c1-600
  grep -iE "error|traceback" /tmp/ranks_$mode.log | head -5
done
