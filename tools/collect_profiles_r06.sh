# Runs on the GPU box (via gpurun): the round's committed profiles, summarised on the spot into gpurun_out/r06_*.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
# basic: kernel statistics, HBM traffic counters (separate passes), instruction mix
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-advanced --no-scaling-reference > $O/prof_basic.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/p_stats/r_results.db > $O/stats_basic.json
python $R/tools/rocprof_summary.py timeline /tmp/p_stats/r_results.db 40 > $O/r06_timeline_basic.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-advanced --no-scaling-reference > $O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-advanced --no-scaling-reference > $O/prof_write.log 2>&1
python $R/tools/rocprof_summary.py pmc /tmp/p_fetch/r_results.db /tmp/p_write/r_results.db > $O/pmc_hbm_basic.json
bash $R/tools/pmc_mix.sh
# advanced, the default FP64 engine (configs[2] headline): statistics, timeline of one pass, counters of the bank kernel
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_adv -o r -- python $R/bench.py --advanced --steps 1 --warmup 1 --no-cpu-baseline --no-scaling-reference > $O/prof_adv.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/p_adv/r_results.db > $O/stats_adv.json
python $R/tools/rocprof_summary.py timeline /tmp/p_adv/r_results.db 140 | grep "peaq::" | grep -v synth | tail -28 > $O/r06_timeline_adv.txt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_advd -o r -- python $R/bench.py --advanced --reduced-precision --steps 1 --warmup 1 --no-cpu-baseline --no-scaling-reference > $O/prof_advd.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/p_advd/r_results.db > $O/stats_adv_f16x3.json
bash $R/tools/pmc_fb.sh
bash $R/tools/pmc_busy.sh main > $O/r06_pmc_busy_print.txt 2>&1
cd $R
if [ -f gstpeaq_amd/libpeaq_amd_fbprof.so ]; then
  PEAQ_AMD_LIB=$R/gstpeaq_amd/libpeaq_amd_fbprof.so python tools/fb_profile.py 1024 f64 > $O/r06_fb_phases_f64.json 2>/dev/null
fi
if [ -f gstpeaq_amd/libpeaq_amd_serial.so ]; then
  cd /tmp
  PEAQ_AMD_LIB=$R/gstpeaq_amd/libpeaq_amd_serial.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_ser -o r -- python $R/bench.py --advanced --steps 1 --warmup 1 --no-cpu-baseline --no-scaling-reference > $O/prof_ser.log 2>&1
  python $R/tools/rocprof_summary.py stats /tmp/p_ser/r_results.db > $O/r06_stats_adv_serial.json
  cd $R
fi
if [ -f gstpeaq_amd/libpeaq_amd_prof.so ]; then
  PEAQ_AMD_LIB=$R/gstpeaq_amd/libpeaq_amd_prof.so python tools/fe_profile.py 1024 > $O/r06_frontend_phases.json 2>/dev/null
fi
python tools/make_pmc_frontend.py > /dev/null 2> $O/make_pmc.err; cp profiles/pmc_frontend.json $O/pmc_frontend.json
# instruction-cache counters of the basic kernels (round 6: the front end's code is 37 KB, the back end's 18 KB, a CU pair's cache 64 KB)
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES SQ_WAVE_CYCLES -d /tmp/p_ic -o r -- python $R/bench.py --pairs 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-advanced --no-scaling-reference > $O/prof_ic.log 2>&1
python $R/tools/rocprof_summary.py pmc /tmp/p_ic/r_results.db > $O/r06_pmc_icache.json
cd $R
python bench.py > $O/bench_basic.json 2> $O/bench_basic.err
python bench.py --advanced --steps 3 > $O/bench_adv.json 2> /dev/null
python bench.py --advanced --reduced-precision --steps 3 > $O/bench_adv_f16x3.json 2> /dev/null
ls -la $O | tail -30
