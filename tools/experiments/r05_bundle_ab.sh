# bundled level 0 of the word-gradient tree, A/B (same box, same process order): C2, C2 dims at 8192, product-search settings, C4
mkdir -p gpurun_out/r05c
R=$GRAFT_REPO_ROOT
for cfg in "c2 --batch 65536" "c2_8192 --batch 8192" "ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128" "c4 --vocab 500000 --entities 100000 --dim 300"; do
  set -- $cfg; name=$1; shift
  for b in 1 0 1 0; do
    SERT_SEG_BUNDLE=$b python $R/bench.py --num-batches 8 "$@" --steps 100 --warmup 10 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('$name bundle=$b ms/step %.4f  segsum %.1f us  adam %.1f' % (r['ms_per_step'], k.get('word_grad_segsum',0), k.get('optimizer_word_table',0)))"
  done
done
