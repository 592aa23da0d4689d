# the data-parallel step STRUCTURE on one GPU: RCCL attached with a world of one (SERT_FORCE_COMM=1), C2 and the per-GPU batches of strong scaling
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05f
NOX="--num-batches 8 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc"
for b in ${BATCHES:-65536 8192}; do
  for fc in 0 1; do
    SERT_FORCE_COMM=$fc python bench.py --batch $b --steps 100 --warmup 10 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('batch $b force_comm $fc: %.4f ms  %.1f M pairs/s' % (r['ms_per_step'], r['value']/1e6), r.get('kernel_us'))"
  done
done
cd /tmp; export TMPDIR=/tmp
for b in ${BATCHES:-65536 8192}; do
rm -rf /tmp/tl2
SERT_FORCE_COMM=1 rocprofv3 --kernel-trace -d /tmp/tl2 -o t -- python $R/bench.py --profile-inner --num-batches 8 --batch $b --steps 40 --warmup 10 > /dev/null 2>&1
DB=$(find /tmp/tl2 -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB vs_gather_mean 5 | tee $R/gpurun_out/r05f/timeline_dp_world1_$b.txt
done
