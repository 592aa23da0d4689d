# re-measure the opt-in schedule variants on the round-4 code (C2, 100 steps)
mkdir -p gpurun_out/r04g
run() { env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc > gpurun_out/r04g/b.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r04g/b.json')); print('$*: %.4f ms' % r['ms_per_step'])"; }
run A=1
run SERT_ADAM_SPLIT=1
run SERT_DENSE_HEAVY=1
run SERT_BWD_FUSED=1
run SERT_ADAM_SPLIT=1 SERT_DENSE_HEAVY=1
run A=2
