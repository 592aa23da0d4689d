cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in base KO_COEF KO_MATH KO_ROWS KO_IDS; do
  L=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so
  SERT_LIB=$L python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('variant=[$v] ms/step %.4f gather %.1f loss %.1f segsum %.1f' % (d['ms_per_step'], k['gather']['us'], k['loss']['us'], k['word_grad_segsum']['us']))" || SERT_LIB=$L python bench.py --steps 50 --warmup 5 $NOX 2>&1 | tail -3
done
