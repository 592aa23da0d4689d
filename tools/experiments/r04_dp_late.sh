# data parallel, world of one (SERT_FORCE_COMM=1): the late join (default), dW behind the chain (2), the old schedule (0) -- same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
NOX="--no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc"
for rep in 1 2; do
for b in 65536 32768 16384 8192; do
  for mode in 1 0 2; do
    SERT_DP_LATE=$mode SERT_FORCE_COMM=1 python bench.py --batch $b --steps 100 --warmup 10 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('batch $b dp_late $mode: %.4f ms' % r['ms_per_step'])"
  done
done
done
