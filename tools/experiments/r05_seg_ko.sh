# where do the 41 us of the word-gradient tree's level 0 go?  timing knock-outs (variant libraries built with -DSERT_KO_SEG=k; wrong results)
R=$GRAFT_REPO_ROOT
for lib in ${LIBS:-product ko_seg1 ko_seg2 ko_seg3 ko_seg4 ko_seg5}; do
  if [ $lib = product ]; then L=$R/sert_amd/libsert_hip.so; else L=$R/sert_amd/variants/libsert_$lib.so; fi
  SERT_SEG_BUNDLE=0 SERT_LIB=$L python $R/bench.py --num-batches 8 --steps 50 --warmup 10 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('$lib ms/step %.4f  segsum group %.1f us' % (r['ms_per_step'], k.get('word_grad_segsum',0)))"
done
