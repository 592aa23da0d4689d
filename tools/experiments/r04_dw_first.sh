# dW / db first on the side stream (default) against dW on the main stream (variants build, SERT_DW_FIRST=0)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p gpurun_out/r04g
NOX="--no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc"
run() { env "$@" python bench.py --steps 100 --warmup 10 $NOX > gpurun_out/r04g/b.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r04g/b.json')); print('C2 $*: %.4f ms' % r['ms_per_step'])"; }
runb() { b=$1; shift; env "$@" python bench.py --batch $b --steps 100 --warmup 10 $NOX > gpurun_out/r04g/b.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r04g/b.json')); print('batch $b $*: %.4f ms' % r['ms_per_step'])"; }
run A=1
run SERT_DW_FIRST=0
run A=2
run SERT_DW_FIRST=0
runb 16384 A=1
runb 16384 SERT_DW_FIRST=0
runb 4096 A=1
runb 4096 SERT_DW_FIRST=0
