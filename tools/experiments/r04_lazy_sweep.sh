# lazy dense update of the word table: where does it pay?  (variants build: SERT_LAZY_MAX = 0 off, 1 always)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
NOX="--no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc"
for b in 4096 16384 32768 65536; do for lz in 0 1; do
  SERT_LAZY_MAX=$lz python bench.py --batch $b --steps 100 --warmup 10 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernels']; print('C2 dims batch $b lazy $lz: %.4f ms  word-table update %.1f us' % (r['ms_per_step'], k['optimizer_word_table']['us']))"
done; done
for lz in 0 1; do
  SERT_LAZY_MAX=$lz python bench.py --batch 4096 --entities 32768 --dim 300 --entity-dim 128 --steps 300 --warmup 30 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernels']; print('product-search lazy $lz: %.4f ms  word-table update %.1f us' % (r['ms_per_step'], k['optimizer_word_table']['us']))"
  SERT_LAZY_MAX=$lz python bench.py --model loglinear --batch 1024 --dim 300 --entities 715 --window 8 --steps 300 --warmup 30 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernels']; print('w3c loglinear lazy $lz: %.4f ms  word-table update %.1f us' % (r['ms_per_step'], k['optimizer_word_table']['us']))"
  SERT_LAZY_MAX=$lz python tools/bench_c4.py --kinds vectorspace --steps 10 2>/dev/null | python -c "
import json,sys; r=json.load(sys.stdin)['vectorspace']; print('C4 lazy $lz: %.3f ms' % r['ms_per_step'], r['kernels_us']['optimizer_word_table'], r['kernels_us']['optimizer_other'])"
  SERT_LAZY_MAX=$lz python bench.py --model loglinear --steps 20 --warmup 3 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernels']; print('LL C2 dims B=65536 lazy $lz: %.4f ms  word-table update %.1f us' % (r['ms_per_step'], k['optimizer_word_table']['us']))"
done
