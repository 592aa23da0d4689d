cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc18
mkdir -p $OUT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Counter_Name|Name)|SQ_|TCP_|TCC_|TA_|GRBM" | head -400 > $OUT/counters.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 $NOX > /dev/null 2> $OUT/p$i.err
  DB=$(find $OUT/p$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python $GRAFT_REPO_ROOT/tools/gemm_pmc.py $DB > $OUT/set$i.txt; else tail -5 $OUT/p$i.err > $OUT/set$i.txt; fi
  rm -rf $OUT/p$i
done
ls $OUT
