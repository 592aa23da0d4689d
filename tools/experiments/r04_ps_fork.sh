# the reference's product-search settings (sorted entity chain, V_e = 32768): where the side stream forks
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc --no-seed-extra"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']/1e6,2), 'M pairs/s', round(1000*d['ms_per_step'],1), 'us/step')"; }
for rep in 1 2; do
for e in "A=1" "SERT_FORK_AT=nce" "SERT_SIDE_HEAVY=2" "SERT_FORK_LATE=0"; do
env $e python bench.py --batch 4096 --entities 32768 --dim 300 --entity-dim 128 --steps 300 --warmup 30 $NOX 2>/dev/null | p "product_search $e"
done; done
