#!/bin/bash
# Round 3: SQ stall / activity counters of gemm.h and of the vendor GEMM on one long-K shape (65536 x 128 x 1000)
# and on 4096^3.  Separate --pmc passes (no trace domain besides the kernel trace).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03q; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
cat > /tmp/mine.py <<PY
import sys; sys.path.insert(0, "$ROOT")
from sert_amd import _capi as C
C.bench_gemm(M=65536, N=128, K=1000, iters=30)
C.bench_gemm(M=4096, N=4096, K=4096, iters=10)
PY
cat > /tmp/vendor.py <<PY
import torch
torch.backends.cuda.matmul.allow_tf32 = False
a = torch.randn(65536, 1000, device="cuda"); b = torch.randn(1000, 128, device="cuda")
for _ in range(30): a @ b
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
for _ in range(10): a @ b
torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA"; do
  i=$((i+1))
  for who in mine vendor; do
    rocprofv3 --kernel-trace --pmc $set -d $OUT/p_${who}_$i -o p -- python /tmp/$who.py > /dev/null 2> $OUT/${who}_$i.err
    DB=$(find $OUT/p_${who}_$i -name '*.db' | head -1)
    [ -n "$DB" ] && python $ROOT/tools/gemm_pmc.py $DB > $OUT/${who}_$i.txt 2>&1
    rm -rf $OUT/p_${who}_$i
  done
done
ls $OUT
