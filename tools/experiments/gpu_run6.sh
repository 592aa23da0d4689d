cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in "" b8 b32 b8db; do
  for rep in 1 2; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so
  SERT_LIB=$L python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('variant=[$v] ms/step %.4f Mpairs/s %.1f | bucket %.1f egrad %.1f sum %.1f' % (d['ms_per_step'], d['value']/1e6, k.get('entity_sort',{}).get('us',0), k['entity_grad_reduce']['us'], k['entity_grad_fixup']['us']))"
  done
done
