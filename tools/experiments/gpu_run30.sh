cd $GRAFT_REPO_ROOT
for v in 2 1 2 1; do
SERT_STREAMS=$v python tools/bench_c4.py --kinds vectorspace --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vectorspace']
print('C4 streams=$v ms/step %.4f' % d['ms_per_step'])"
done
