# loglinear at C2 dims: dW + combine on the side stream (default) against one stream
for e in "SERT_LL_DW_SIDE=1" "SERT_LL_DW_SIDE=0"; do echo $e; env $e python bench.py --model loglinear --steps 40 --warmup 5 --no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc --no-seed-extra | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value']/1e6)"; done
