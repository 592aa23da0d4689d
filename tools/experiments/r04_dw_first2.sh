R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
NOX="--no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc"
runb() { b=$1; shift; env "$@" python bench.py --batch $b --steps 100 --warmup 10 $NOX > gpurun_out/r04g/b.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r04g/b.json')); print('batch $b $*: %.4f ms' % r['ms_per_step'])"; }
mkdir -p gpurun_out/r04g
for b in 65536 32768 8192; do runb $b A=1; runb $b SERT_DW_FIRST=0; runb $b A=2; runb $b SERT_DW_FIRST=0; done
