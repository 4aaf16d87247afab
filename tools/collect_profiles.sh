# Runs on the GPU box (via gpurun): rocprofv3 passes of the bench, summarised on the spot
# (the rocpd databases are too large to travel back) into gpurun_out/*.json.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-advanced --no-scaling-reference > $O/prof_basic.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/p_stats/r_results.db > $O/stats_basic.json
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-advanced --no-scaling-reference > $O/prof_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-advanced --no-scaling-reference > $O/prof_write.log 2>&1
python $R/tools/rocprof_summary.py pmc /tmp/p_fetch/r_results.db /tmp/p_write/r_results.db > $O/pmc_hbm_basic.json
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_adv -o r -- python $R/bench.py --advanced --reduced-precision --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_adv.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/p_adv/r_results.db > $O/stats_adv.json
cd $R
python bench.py > $O/bench_basic.json 2> $O/bench_basic.err
python bench.py --advanced --steps 2 > $O/bench_adv.json 2> /dev/null
python bench.py --advanced --reduced-precision --steps 2 > $O/bench_adv_default.json 2> /dev/null
ls -la $O
