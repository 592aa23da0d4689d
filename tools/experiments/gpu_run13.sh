cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for rep in 1 2 3; do
python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('ms/step %.4f Mpairs/s %.1f deferred %.4f' % (d['ms_per_step'], d['value']/1e6, d['deferred_loss_readback']['ms_per_step']))"
done
