cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in OB2048 OB1024 OB4096 OB8192 OB2048 OB4096; do
  SERT_LIB=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('variant=$v ms/step %.4f adam %.2f' % (d['ms_per_step'], k['optimizer_word_table']['us']))"
done
for v in OB2048 OB4096 OB8192; do
SERT_LIB=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so python tools/bench_c4.py --kinds vectorspace --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vectorspace']
print('C4 $v ms/step %.4f adam %.1f other %.1f' % (d['ms_per_step'], d['kernels_us']['optimizer_word_table'], d['kernels_us']['optimizer_other']))"
done
