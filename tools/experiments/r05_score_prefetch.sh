# filter kernel: the next k chunk's pieces fetched under the current chunk's MFMAs (product) against the strictly serial chunks
# (libsert_noprefetch.so = the same source without the change), A/B/A/B on one box
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in $R/sert_amd/libsert_hip.so $R/sert_amd/variants/libsert_noprefetch.so; do
SERT_LIB=$lib python - <<P
import sys, time, numpy as np
sys.path.insert(0, '$R')
from sert_amd import _capi
rng = np.random.RandomState(7)
Q, V, d, k = 10000, 100000, 128, 100
E = rng.randn(V, d).astype(np.float32); P = np.tanh(rng.randn(Q, d)).astype(np.float32)
sc = _capi.Scorer(E); sc.topk(P[:256], k)
Pq = sc.query_buffer(Q); np.copyto(Pq, P)
best = 1e9
for _ in range(6):
    t0 = time.perf_counter(); idx, val = sc.topk(Pq, k); best = min(best, time.perf_counter() - t0)
print('%-24s %.3f ms  %.2f M queries/s  checksum %d %.6f' % ('$lib'.split('/')[-1], 1e3 * best, Q / best / 1e6, int(idx.astype(np.int64).sum()), float(val.sum())))
P
done; done
