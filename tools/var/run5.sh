cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  for v in new head; do
    if [ $v = head ]; then export SPNET_HIP_LIB=$R/tools/libhead.so; else unset SPNET_HIP_LIB; fi
    rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/pmc5_${v}_$i -o run --output-format csv -- python3 $R/tools/gemm_sweep.py one fwd 6144 728 2912 6 > $R/gpurun_out/pmc5_${v}_$i.log 2>&1
    echo "== pass $i $v"; python3 $R/tools/pmc_fold.py gemm_f32 $(find $R/gpurun_out/pmc5_${v}_$i -name "*counter_collection.csv")
  done
done
