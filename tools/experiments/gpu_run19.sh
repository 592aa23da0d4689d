cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in "" base ""; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so
  SERT_LIB=$L python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('variant=[$v] ms/step %.4f gather %.1f loss %.1f segsum %.1f adam %.1f' % (d['ms_per_step'], k['gather']['us'], k['loss']['us'], k['word_grad_segsum']['us'], k['optimizer_word_table']['us']), {a:b['us'] for a,b in k.items() if a.startswith('entity')})"
done
python bench.py --model loglinear --steps 50 --warmup 5 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('loglinear ms/step %.4f' % d['ms_per_step'], {k:v['us'] for k,v in d['kernels'].items()})"
SERT_LIB=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_base.so python bench.py --model loglinear --steps 50 --warmup 5 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('loglinear[base] ms/step %.4f' % d['ms_per_step'], {k:v['us'] for k,v in d['kernels'].items()})"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E  " | tail -8
