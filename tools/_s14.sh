cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s14
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -2 > gpurun_out/s14/pytest.log
( time python bench.py > gpurun_out/s14/bench_full.json 2> gpurun_out/s14/bench_full.err ) 2>> gpurun_out/s14/pytest.log
python - <<'PY' >> gpurun_out/s14/pytest.log
import json
d = json.load(open("gpurun_out/s14/bench_full.json"))
print({k: d[k] for k in ("value", "ms_per_step", "max_abs_err_vs_oracle")})
print("replay", d.get("replay_checked_vs_oracle"))
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "avg_launch_us", "launches_per_step")})
for p in d["parity_mode"]:
    print(p["dtype"], p["ms_per_step"], p["max_abs_err_vs_oracle"], p.get("replay_max_abs_err_vs_oracle"), p["roofline"]["achieved"], p["roofline"]["frac"])
print("two streams", d.get("forward_two_streams", {}).get("ms_per_step"), d.get("forward_two_streams", {}).get("mismatching_streams"), d.get("forward_two_streams_f16x2", {}).get("ms_per_step"))
for k, v in d["legs"].items():
    print(k, json.dumps(v)[:600])
print("cpu", json.dumps(d["cpu_baseline"])[:500])
PY
cat gpurun_out/s14/pytest.log
