cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in T256 T64 T128 T512 T256 T128; do
  SERT_LIB=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('variant=$v ms/step %.4f segsum %.2f' % (d['ms_per_step'], k['word_grad_segsum']['us']))"
done
