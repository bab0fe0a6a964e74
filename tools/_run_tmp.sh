cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "sampl or infer" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for k in 1 10 50 1 10 50; do
  MSMD_SAMPLER_STEPS_PER_GRAPH=$k python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-two-streams-leg --no-roofline --legs sampler 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); v=d['legs']['sampler_b64_t500']; print('k=$k', v['ms_per_step'], v.get('f16x2',{}).get('ms_per_step'))"
done
