# split count of the projection's dW GEMM with the bf16-pipe kernel (variants build: SERT_DW_SPLITS)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p gpurun_out/r04g
run() { env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc > gpurun_out/r04g/b.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r04g/b.json')); print('C2 $*: %.4f ms' % r['ms_per_step'])"; }
run4() { env "$@" python bench.py --steps 30 --warmup 5 --vocab 500000 --entities 100000 --dim 300 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc > gpurun_out/r04g/b.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r04g/b.json')); print('C4 $*: %.4f ms' % r['ms_per_step'])"; }
run A=1
run SERT_DW_SPLITS=256
run SERT_DW_SPLITS=128
run SERT_DW_SPLITS=768
run A=2
run4 A=1
run4 SERT_DW_SPLITS=128
run4 SERT_DW_SPLITS=64
run4 SERT_DW_SPLITS=256
run4 A=2
