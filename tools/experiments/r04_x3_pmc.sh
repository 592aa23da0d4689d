# SQ counters of the bf16-pipe GEMM (gemm_x3.h) at the C4 / C2 shapes: where its waves spend their cycles
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04z; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/x3run.py <<PY
import sys; sys.path.insert(0, '$R')
from sert_amd import _capi as C
C.bench_gemm(M=65536, N=300, K=300, tb=1, iters=6)
C.bench_gemm(M=65536, N=300, K=300, epi=2, iters=6)
C.bench_gemm(M=65536, N=128, K=128, tb=1, iters=6)
C.bench_gemm(M=300, N=300, K=65536, ta=1, splits=113, iters=6)
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/x3p$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/x3p$i -o p -- python /tmp/x3run.py > /dev/null 2> $OUT/pmc$i.err
  DB=$(find /tmp/x3p$i -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/gemm_pmc.py $DB | grep -A6 "gemm_x3" >> $OUT/x3_pmc.txt
done
cat $OUT/x3_pmc.txt
