# Runs on the GPU box (via gpurun): what is busy while the basic front end runs -- vector ALU, LDS pipe, wave
# lifetimes -- per kernel, summarised into gpurun_out/pmc_busy[_<lib>].json.  usage: bash tools/pmc_busy.sh [lib ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in ${@:-main}; do
  if [ "$v" = main ]; then unset PEAQ_AMD_LIB; else export PEAQ_AMD_LIB=$R/gstpeaq_amd/libpeaq_amd_$v.so; fi
  i=0; dbs=""
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES" \
             "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64" \
             "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcbusy_$v$i -o r -- python $R/bench.py --pairs 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-advanced --no-scaling-reference $EXTRA > $O/pmcbusy_$v$i.log 2>&1
    [ -f /tmp/pmcbusy_$v$i/r_results.db ] && dbs="$dbs /tmp/pmcbusy_$v$i/r_results.db"
  done
  python $R/tools/rocprof_summary.py pmc $dbs > $O/pmc_busy_$v.json
  python - <<PY
import json
d = json.load(open("$O/pmc_busy_$v.json"))
for k in d:
    if "frontend_kernel" in k or "backend_kernel" in k:
        w = d[k].get("SQ_WAVES", {}).get("avg", 0) or 1
        print("$v", k[:50])
        for c, x in sorted(d[k].items()):
            print("    %-26s %16.1f  per wave %12.2f" % (c, x["avg"], x["avg"] / w))
PY
done
