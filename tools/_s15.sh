cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s15
for cfg in "32 1" "16 2" "8 4" "16 1" "32 2"; do
  set -- $cfg
  timeout 600 python bench.py --batch $1 --streams $2 --steps 40 --warmup 5 --legs none --no-parity --no-roofline --no-cpu-baseline --no-two-streams-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $1 streams $2:', d['ms_per_step'], 'ms per step ->', round(d['ms_per_step']*32/$1, 3), 'ms per 32 clips;', d['value'], 'frames/s')" >> gpurun_out/s15/streams.log
done
cat gpurun_out/s15/streams.log
