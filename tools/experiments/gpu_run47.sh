cd $GRAFT_REPO_ROOT
for v in "" 1 "" 1; do
if [ -n "$v" ]; then export SERT_SEG_LPI32_Y=1; else unset SERT_SEG_LPI32_Y; fi
python tools/bench_c4.py --kinds vectorspace --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vectorspace']
print('C4 lpi32y=[$v] ms/step %.4f segsum %.1f loss %.6f' % (d['ms_per_step'], d['kernels_us']['word_grad_segsum'], d.get('last_loss', 0)))"
done
