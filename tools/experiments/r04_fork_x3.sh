R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
NOX="--no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc"
mkdir -p gpurun_out/r04g
run() { env "$@" python bench.py --steps 100 --warmup 10 $NOX > gpurun_out/r04g/b.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r04g/b.json')); print('C2 $*: %.4f ms' % r['ms_per_step'])"; }
run A=1
run SERT_FORK_AT=nce
run SERT_DW_FIRST=2
run SERT_FORK_LATE=0
run A=2
