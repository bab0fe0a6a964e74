cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests/test_autograd_gpu.py tests/test_train_gpu.py tests/test_kernels_gpu.py tests/test_losses_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | tail -5
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o p -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-exchange-rehearsal > /tmp/pl.log 2>&1
f=$(find /tmp/pl -name "*kernel_stats.csv" | head -1); grep -i "seq_loss\|colsum" $f | cut -c1-160
cd $GRAFT_REPO_ROOT; python bench.py --mode train --no-cpu-baseline --no-exchange-rehearsal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', d['ms_per_step'])"
