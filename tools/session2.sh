#!/bin/bash
# GPU session 2 (round 3): same-stream ordering probe + stage bisection of the two-stream mismatch
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2
O=gpurun_out/s2
timeout 600 tools/_bin/stream_order_probe 3000 2 > $O/probe.log 2>&1
timeout 600 tools/_bin/stream_order_probe 2000 3 >> $O/probe.log 2>&1
for w in stats conv0 fe_fp enc feat; do
  REPS=60 timeout 600 python tools/concurrent_pattern.py $w 2>&1 | grep -v "^priority" | tail -3 >> $O/stages.log
done
for dt in f16x2 fp32; do
  DT=$dt REPS=40 timeout 600 python tools/concurrent_pattern.py feat 2>&1 | grep -v "^priority" | tail -3 >> $O/stages.log
done
cat $O/probe.log $O/stages.log
