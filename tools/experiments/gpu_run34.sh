cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
python bench.py --model loglinear --batch 8192 --steps 100 --warmup 10 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('LL B=8192 ms/step %.4f  %.1f M pairs/s' % (d['ms_per_step'], d['value']/1e6), {a:b['us'] for a,b in d['kernels'].items()})"
python bench.py --model loglinear --batch 1024 --dim 300 --entities 715 --window 8 --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('LL W3C ms/step %.4f  %.2f M pairs/s' % (d['ms_per_step'], d['value']/1e6), {a:b['us'] for a,b in d['kernels'].items()})"
python bench.py --batch 4096 --entities 32768 --dim 300 --entity-dim 128 --steps 300 --warmup 30 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('VS product-search ms/step %.4f  %.2f M pairs/s' % (d['ms_per_step'], d['value']/1e6), {a:b['us'] for a,b in d['kernels'].items()})"
python tools/bench_c1.py 2>/dev/null | tail -2
python tools/epoch_bench.py 2>/dev/null | tail -4
