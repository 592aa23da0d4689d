cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in "" adam_nt "" adam_nt; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so
  SERT_LIB=$L python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('variant=[$v] ms/step %.4f adam %.1f segsum %.1f gather %.1f' % (d['ms_per_step'], k['optimizer_word_table']['us'], k['word_grad_segsum']['us'], k['gather']['us']))"
done
for v in "" adam_nt; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so
  SERT_LIB=$L python tools/bench_c4.py --kinds vectorspace --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vectorspace']
print('C4 variant=[$v] ms/step %.4f adam %.1f' % (d['ms_per_step'], d['kernels_us']['optimizer_word_table']))"
done
