# instruction mix / stall counters of both kernels, summarised on the GPU box (gpurun_out/pmc_mix.json)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0; dbs=""
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcmix$i -o r -- python $R/bench.py --pairs 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-advanced $EXTRA > $O/pmcmix$i.log 2>&1
  dbs="$dbs /tmp/pmcmix$i/r_results.db"
done
python $R/tools/rocprof_summary.py pmc $dbs > $O/pmc_mix.json
