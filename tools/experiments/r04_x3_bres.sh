R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
for v in "SERT_X3_BRES=8" "SERT_X3_BRES=4" "A=1"; do
echo "== $v"
env $v python - <<'PY'
import sys; sys.path.insert(0, ".")
from sert_amd import _capi as C
for name, kw in [("c2 proj NN tanh", dict(M=65536, N=128, K=128, epi=2)), ("c2 dh NT", dict(M=65536, N=128, K=128, tb=1)), ("NT 32768", dict(M=32768, N=128, K=128, tb=1)), ("NT 131072", dict(M=131072, N=128, K=128, tb=1))]:
    print(name, round(C.bench_gemm(**kw),1))
PY
done
