cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_poison.py tests/test_gpu_fir_modes.py -x -q 2>&1 | tail -4
AB_ARGS="--no-scaling-reference --advanced" AB_STEPS=3 bash tools/ab_basic.sh base main 2>&1 | tail -4
