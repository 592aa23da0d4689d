# what bounds level 0 + the heavy words' stream (segsum_rows_plus)?  knock-outs, WRONG results: the tree alone / the stream alone
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
cd /tmp; export TMPDIR=/tmp
for ko in 0 1 2; do
  export SERT_KO_PLUS=$ko
  rm -rf /tmp/tl_ko
  rocprofv3 --kernel-trace -d /tmp/tl_ko -o t -- python $R/bench.py --profile-inner --num-batches 8 --batch 65536 --steps 40 --warmup 10 > /dev/null 2>&1
  echo "SERT_KO_PLUS=$ko"
  python $R/tools/rocpd_timeline.py $(find /tmp/tl_ko -name '*.db' | head -1) vs_gather_mean 24 | grep -E "segsum_rows_plus|egrad_acc|step span"
  python $R/bench.py --num-batches 8 --batch 65536 --steps 100 --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('   ms/step %.4f  word_grad_segsum alone %.1f us' % (r['ms_per_step'], k.get('word_grad_segsum', 0)))"
done
