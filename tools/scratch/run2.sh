cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backend_stage.py -x -q 2>&1 | tail -5
AB_ARGS="--no-scaling-reference" bash tools/ab_basic.sh base main 2>&1 | tail -8
