# (for the record: the code path this measured -- SERT_SORT_MAIN -- was reverted after the measurement, profiles/r04_experiments.txt item 16)
# C4 (side-heavy schedule): the counting sort of the entity chain on the main stream in front of the fork (default) against
# on the side stream (variants build, SERT_SORT_MAIN=0); also the product-search settings (sorted chain, not side-heavy)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
NOX="--no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc"
mkdir -p gpurun_out/r04g
run4() { env "$@" python bench.py --steps 30 --warmup 5 --vocab 500000 --entities 100000 --dim 300 $NOX > gpurun_out/r04g/b.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r04g/b.json')); print('C4 $*: %.4f ms' % r['ms_per_step'])"; }
run4 A=1
run4 SERT_SORT_MAIN=0
run4 A=2
run4 SERT_SORT_MAIN=0
