cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_poison.py tests/test_gpu_fir_modes.py tests/test_gpu_broker.py tests/test_gpu_settings.py -x -q 2>&1 | tail -3
python tools/scratch/determinism.py 512 2>&1 | tail -2
PEAQ_AMD_LIB=$GRAFT_REPO_ROOT/gstpeaq_amd/libpeaq_amd_fbprof.so python tools/fb_profile.py 1024 f64 > gpurun_out/r05_fb_phases_d.json 2>/dev/null
python - <<'PY'
import json
b=json.load(open('gpurun_out/r05_fb_phases_d.json'))
for w in ('wave0','wave3'):
    print(w, b[w]['cycles_per_tile'], {k[:10]: v for k, v in b[w]['phases'].items()})
PY
AB_ARGS="--no-scaling-reference --advanced" AB_STEPS=3 bash tools/ab_basic.sh base main 2>&1 | tail -4
