R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python - <<'PY'
import sys; sys.path.insert(0, '.')
from sert_amd import _capi as C
for rep in range(2):
  for name, kw in [('c2 proj NN', dict(M=65536, N=128, K=128, epi=2)), ('c2 dh NT', dict(M=65536, N=128, K=128, tb=1)),
                 ('c4 proj NN', dict(M=65536, N=300, K=300, epi=2)), ('c4 dh NT', dict(M=65536, N=300, K=300, tb=1)),
                 ('fs dp NN 65536x128x1000', dict(M=65536, N=128, K=1000)), ('fs logits NT 65536x1000x128', dict(M=65536, N=1000, K=128, tb=1)),
                 ('c2 dW TN /512', dict(M=128, N=128, K=65536, ta=1, splits=512)), ('c4 dW TN /113', dict(M=300, N=300, K=65536, ta=1, splits=113))]:
    us = C.bench_gemm(**kw)
    print('%-34s %8.1f us' % (name, us))
PY
