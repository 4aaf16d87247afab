cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py > gpurun_out/bench_r05_a.json 2> gpurun_out/bench_r05_a.err; tail -c 600 gpurun_out/bench_r05_a.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r05_a.json"))
print("value", d["value"], "ms", d["ms_per_step"], "adv", d.get("advanced",{}).get("value"))
print(json.dumps(d.get("device_clock"), indent=1)[:1500])
print("nominal", d.get("value_at_nominal_clock"))
PY
