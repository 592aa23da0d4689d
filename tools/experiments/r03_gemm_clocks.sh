#!/bin/bash
# Round 3: does the chip hold its clock under the fp32 MFMA GEMM?  Sample rocm-smi while sert_bench_gemm (4096^3, many
# iterations) and, for comparison, the vendor GEMM through torch run.
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY' &
import sys; sys.path.insert(0, '.')
from sert_amd import _capi as C
us = C.bench_gemm(M=4096, N=4096, K=4096, iters=3000)
print('mine 4096^3: %.1f us %.1f TF' % (us, 2 * 4096**3 / us / 1e6), flush=True)
PY
PID=$!
sleep 2.0
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -6; sleep 0.4; done
wait $PID
python - <<'PY' &
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
a = torch.randn(4096, 4096, device='cuda'); b = torch.randn(4096, 4096, device='cuda')
for _ in range(5): a @ b
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3000): a @ b
torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 3000 * 1e6
print('vendor 4096^3: %.1f us %.1f TF' % (us, 2 * 4096**3 / us / 1e6), flush=True)
PY
PID=$!
sleep 4.0
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -6; sleep 0.4; done
wait $PID
