# dense_update_skip: workgroups in its grid (SERT_SKIP_BLOCKS, variants library; - = the dense launches' grid: 2048, 4096 from 2^24 elements)
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 24 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s blocks=%-5s ms/step %.4f  word-table update %.1f us' % ('$name', '${SERT_SKIP_BLOCKS:--}', r['ms_per_step'], k.get('optimizer_word_table', 0)))"
}
for rep in 1 2; do for v in - 3072 4096 6144 8192; do
  if [ $v = - ]; then unset SERT_SKIP_BLOCKS; else export SERT_SKIP_BLOCKS=$v; fi
  run c2 --batch 65536
  run c2_8192 --batch 8192
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  run w3c --model loglinear --batch 1024 --window 8 --entities 715 --dim 300
  STEPS=60 run c4 --vocab 500000 --entities 100000 --dim 300
done; done
