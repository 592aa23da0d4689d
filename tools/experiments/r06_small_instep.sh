R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NOX="--num-batches 8 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc"
python bench.py --model loglinear --batch 1024 --dim 300 --entities 715 --window 8 --steps 300 --warmup 20 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('w3c %.4f ms' % r['ms_per_step']); print('  alone ', r.get('kernel_us')); print('  instep', r.get('kernel_us_instep'))"
python bench.py --batch 4096 --entities 32768 --dim 300 --entity-dim 128 --steps 300 --warmup 20 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('ps %.4f ms' % r['ms_per_step']); print('  alone ', r.get('kernel_us')); print('  instep', r.get('kernel_us_instep'))"
python bench.py --batch 8192 --steps 300 --warmup 20 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c2_8192 %.4f ms' % r['ms_per_step']); print('  alone ', r.get('kernel_us')); print('  instep', r.get('kernel_us_instep'))"
