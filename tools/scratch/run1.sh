cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backend_stage.py -x -q 2>&1 | tail -5
AB_ARGS="--no-scaling-reference" bash tools/ab_basic.sh base main 2>&1 | tail -8
PEAQ_AMD_LIB=$GRAFT_REPO_ROOT/gstpeaq_amd/libpeaq_amd_prof.so timeout 300 python tools/fe_profile.py 1024 > gpurun_out/r05_frontend_phases_a.json 2>gpurun_out/fe_prof.err; tail -3 gpurun_out/fe_prof.err
timeout 600 python -m pytest tests/test_gpu_two_ranks.py -x -q 2>&1 | tail -5
