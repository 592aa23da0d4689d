cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in U1 U2 U3 U4 U1 U2; do
  SERT_LIB=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('variant=$v ms/step %.4f adam %.2f loss %.6f' % (d['ms_per_step'], k['optimizer_word_table']['us'], d['last_loss']))"
done
for v in U1 U2 U4; do
SERT_LIB=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so python tools/bench_c4.py --kinds vectorspace --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vectorspace']
print('C4 $v ms/step %.4f adam %.1f other %.1f' % (d['ms_per_step'], d['kernels_us']['optimizer_word_table'], d['kernels_us']['optimizer_other']))"
done
