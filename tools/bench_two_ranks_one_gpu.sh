#!/bin/bash
# Rehearsal of bench.py's multi-rank control flow on ONE GPU: two ranks (gloo rendezvous, both on cuda:0) run the forward and the
# training line exactly as the driver launches them with --gpus 2 -- barriers, max-over-ranks timing, rank-0-only printing, the
# gradient exchange through the bucket reducer.  Numbers are meaningless (two processes share the device); it must not hang.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export MSMD_DIST_BACKEND=gloo MSMD_ONE_DEVICE=1
for mode in forward train; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 3 --warmup 1 --mode $mode --no-cpu-baseline > /tmp/two_$mode.log 2>&1
  echo "mode $mode: exit $?"
  grep '"metric"' /tmp/two_$mode.log | cut -c1-420
  grep -iE "error|traceback" /tmp/two_$mode.log | head -5
done
