R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NOX="--num-batches 8 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc"
for b in 65536 8192; do for fc in 0 1; do
SERT_FORCE_COMM=$fc python bench.py --batch $b --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('batch $b comm $fc %.4f ms' % r['ms_per_step']); print('  alone ', r.get('kernel_us')); print('  instep', r.get('kernel_us_instep'))"
done; done
