# vectorspace word gradient: the dense heavy words INSIDE the tree's launches (segsum_rows_plus) against the two launches in
# front of the tree (SERT_HEAVY_NO_FUSE=1) and against the plain tree; variants library, A/B/C x 2 on one box + a timeline
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r05i
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s %-12s ms/step %.4f  word_grad_segsum %.1f us' % ('$name', '$TAGV', r['ms_per_step'], k.get('word_grad_segsum', 0)))"
}
for rep in 1 2; do for v in tree fused two_launches; do
  TAGV=$v; unset SERT_DENSE_HEAVY SERT_HEAVY_NO_FUSE
  [ $v = fused ] && export SERT_DENSE_HEAVY=1
  [ $v = two_launches ] && export SERT_DENSE_HEAVY=1 SERT_HEAVY_NO_FUSE=1
  run c2 --batch 65536
  run c2_8192 --batch 8192
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  run c4 --vocab 500000 --entities 100000 --dim 300
done; done
cd /tmp; export TMPDIR=/tmp
unset SERT_HEAVY_NO_FUSE; export SERT_DENSE_HEAVY=1
rm -rf /tmp/tl_f
rocprofv3 --kernel-trace -d /tmp/tl_f -o t -- python $R/bench.py --profile-inner --num-batches 8 --batch 65536 --steps 40 --warmup 10 > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/tl_f -name '*.db' | head -1) vs_gather_mean 24 > $R/gpurun_out/r05i/timeline_c2_heavy_fused.txt
tail -20 $R/gpurun_out/r05i/timeline_c2_heavy_fused.txt
