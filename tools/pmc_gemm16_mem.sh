# memory-side counters of the 16-bit GEMM (one small counter group per pass, no trace flags; every pass under its own
# timeout: a rejected counter set makes rocprofv3 abort and then hang in finalisation)
export TMPDIR=/tmp
R=$PWD
cd /tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1)); mkdir -p $R/gpurun_out/pm$i; rm -rf $R/gpurun_out/pm$i/*
  ONLY=${ONLY:-qkv} REPS=3 VARIANTS=3 timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pm$i -- python3 $R/tools/gemm16_pl_ab.py > $R/gpurun_out/pm$i.log 2>&1
  echo "group $i rc=$?"
done
find $R/gpurun_out -name "*.db" -delete
