# schedule knobs re-measured with the bf16-pipe GEMMs (C2, 100 steps; C4 through --vocab etc.)
mkdir -p gpurun_out/r04g
run() { env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc > gpurun_out/r04g/b.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r04g/b.json')); print('$*: %.4f ms' % r['ms_per_step'])"; }
run A=1
run SERT_STREAMS=3
run SERT_SIDE_HEAVY=1
run SERT_GEMM_FP32=1
run SERT_STREAMS=1
run A=2
