cd $GRAFT_REPO_ROOT
for v in "" 1 "" 1; do
if [ -n "$v" ]; then export SERT_NCE_PER_CANDIDATE=1; else unset SERT_NCE_PER_CANDIDATE; fi
python tools/bench_c4.py --kinds vectorspace --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vectorspace']
print('C4 per_candidate=[$v] ms/step %.4f loss kernel %.1f' % (d['ms_per_step'], d['kernels_us']['loss']))"
done
unset SERT_NCE_PER_CANDIDATE
timeout 900 python -m pytest tests/test_gpu_c4.py -x -q -m gpu 2>&1 | tail -2
