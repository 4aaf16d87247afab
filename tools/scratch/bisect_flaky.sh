cd $GRAFT_REPO_ROOT
run() { n=0; for i in 1 2 3 4 5 6; do python -m pytest tests/test_gpu_parity.py tests/test_gpu_broker.py -x -q -k "$1" 2>&1 | tail -1 | grep -q failed && n=$((n+1)); done; echo "fails $n/6 :: $1"; }
run "(one_filterbank_block and default) or (test_gpu_broker and default)"
run "one_filterbank_block or test_gpu_broker"
