# Round 6, item 2: gemm_x3 with TWO k steps of operand loads in flight (PF = 2) against one (SERT_X3_PF=2, variants library; the default is one):
# the three C2 GEMM shapes alone (sert_bench_gemm, HIP events), then the step (A/B x 3 on one box)
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06d
gemms() { python - <<PY
import sys; sys.path.insert(0, '$R')
from sert_amd import _capi as C
for name, kw in (('fwd h.W tanh', dict(M=65536, N=128, K=128, epi=2)), ('dh da.W^T', dict(M=65536, N=128, K=128, tb=1)),
                 ('dW h^T.da splitK 512', dict(M=128, N=128, K=65536, ta=1, splits=512)),
                 ('fwd 8192', dict(M=8192, N=128, K=128, epi=2)), ('fwd 32768', dict(M=32768, N=128, K=128, epi=2)),
                 ('ps fwd 4096x128x300', dict(M=4096, N=128, K=300, epi=2)),):
    print('  %-24s %7.2f us' % (name, C.bench_gemm(iters=200, **kw)))
PY
}
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{}); ki=r.get('kernel_us_instep',{})
print('%-8s %-6s ms/step %.4f  alone us: fwd %.1f dX %.1f dW %.1f   in-step: fwd %.1f dW %.1f' % ('$name', '$TAGV', r['ms_per_step'], k.get('gemm_fwd', 0), k.get('gemm_dX', 0), k.get('gemm_dW', 0), ki.get('gemm_fwd', 0), ki.get('gemm_dW', 0)))"
}
{
for v in pf2 pf1; do
  unset SERT_X3_PF; [ $v = pf2 ] && export SERT_X3_PF=2
  echo "== $v"; gemms
done
for rep in 1 2 3; do for v in pf2 pf1; do
  TAGV=$v; unset SERT_X3_PF; [ $v = pf2 ] && export SERT_X3_PF=2
  run c2 --batch 65536
  run c2_8192 --batch 8192
  run c2_16384 --batch 16384
done; done
} 2>&1 | tee $R/gpurun_out/r06d/x3_pf2.txt
