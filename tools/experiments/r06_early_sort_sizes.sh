R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-150} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-14s %-8s ms/step %.4f in-step us: sort %.1f reduce %.1f tree %.1f update %.1f tail %.1f' % ('$name', '$TAGV', r['ms_per_step'], k.get('entity_sort', 0), k.get('entity_grad_reduce', 0), k.get('word_grad_segsum', 0), k.get('optimizer_word_table', 0), k.get('finalize', 0)))"
}
for rep in 1 2; do for v in early behind; do
  TAGV=$v; unset SERT_NO_EARLY_SORT
  [ $v = behind ] && export SERT_NO_EARLY_SORT=1
  run b32k_e32k_d128 --batch 32768 --entities 32768 --dim 128
  run b64k_e32k_d128 --batch 65536 --entities 32768 --dim 128
  run b64k_e100k_d128 --batch 65536 --entities 100000 --dim 128
  run b16k_e100k_d300 --batch 16384 --vocab 500000 --entities 100000 --dim 300
  run b32k_e100k_d300 --batch 32768 --vocab 500000 --entities 100000 --dim 300
done; done
