# C2 step timeline with the heavy words summed densely (SERT_DENSE_HEAVY=1) + ms/step A/B
mkdir -p gpurun_out/r05i; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SERT_DENSE_HEAVY=1
rm -rf /tmp/tl_dh
rocprofv3 --kernel-trace -d /tmp/tl_dh -o t -- python $R/bench.py --profile-inner --num-batches 8 --batch 65536 --steps 40 --warmup 10 > /dev/null 2>&1
DB=$(find /tmp/tl_dh -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB vs_gather_mean 24 > $R/gpurun_out/r05i/timeline_c2_dense_heavy.txt
unset SERT_DENSE_HEAVY
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps 200 --warmup 24 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s dense_heavy=%s ms/step %.4f  word_grad_segsum %.1f us' % ('$name', '${SERT_DENSE_HEAVY:-0}', r['ms_per_step'], k.get('word_grad_segsum', 0)))"
}
for rep in 1 2; do for v in 0 1; do
  export SERT_DENSE_HEAVY=$v
  run c2 --batch 65536
done; done
