mkdir -p gpurun_out/r04e
SERT_SEG_PIPE=2 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "row_grouped or vectorspace_steps" 2>&1 | tail -3
for g in 1 8 16; do for pb in 0 1024 2048 4096; do
SERT_SEG_GROUPS=$g SERT_SEG_PIPE=$pb python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc > gpurun_out/r04e/bench_g${g}_p${pb}.json 2>/dev/null
python -c "
import json; r=json.load(open('gpurun_out/r04e/bench_g${g}_p${pb}.json')); k=r['kernels']; print('groups $g pipe $pb: %.4f ms  segsum %.1f adam %.1f egrad %.1f' % (r['ms_per_step'], k['word_grad_segsum']['us'], k['optimizer_word_table']['us'], k['entity_grad_reduce']['us']))"
done; done
