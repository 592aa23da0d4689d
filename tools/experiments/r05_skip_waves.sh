# dense_update_skip, one float4 column per lane: within the registers of eight waves per SIMD (product) against the 72 registers /
# seven waves of the plain build (variants library -D'SERT_SKIP_WAVES(CPL)=1'); A/B x 3 on one box
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 24 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s %-6s ms/step %.4f  word-table update %.1f us' % ('$name', '$TAGV', r['ms_per_step'], k.get('optimizer_word_table', 0)))"
}
for rep in 1 2 3; do for v in waves8 waves7; do
  TAGV=$v
  if [ $v = waves7 ]; then export SERT_LIB=$R/sert_amd/variants/libsert_skw1.so; else unset SERT_LIB; fi
  run c2 --batch 65536
  run c2_32k --batch 32768
  run c2_8192 --batch 8192
done; done
