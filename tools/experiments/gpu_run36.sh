cd $GRAFT_REPO_ROOT
for i in 1 2; do
python tools/bench_c4.py --kinds vectorspace --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vectorspace']
print('C4 ms/step %.4f adam %.1f other %.1f' % (d['ms_per_step'], d['kernels_us']['optimizer_word_table'], d['kernels_us']['optimizer_other']))"
done
python tools/bench_c4.py --kinds loglinear --steps 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['loglinear']
print('C4 LL ms/step %.4f' % d['ms_per_step'], d['kernels_us'])"
timeout 900 python -m pytest tests/test_gpu_c4.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -2
