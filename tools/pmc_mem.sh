# vector-memory counters of the front end, summarised on the GPU box (gpurun_out/pmc_mem.json)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0; dbs=""
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcmem$i -o r -- python $R/bench.py --pairs 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-advanced > $O/pmcmem$i.log 2>&1
  [ -f /tmp/pmcmem$i/r_results.db ] && dbs="$dbs /tmp/pmcmem$i/r_results.db"
done
python $R/tools/rocprof_summary.py pmc $dbs > $O/pmc_mem.json
python - <<PY
import json
d=json.load(open("$O/pmc_mem.json"))
for k,v in d.items():
    if "frontend_kernel<109>" in k:
        for c,x in sorted(v.items()):
            print(c, "%.4g" % x["avg"], "dur_us %.1f" % (x["avg_duration_ns"]/1e3))
PY
