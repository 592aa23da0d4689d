# one step as the GPU sees it at the small-batch configurations (round-5 verdict item 3): product-search settings, C2 dims at 8192 rows
mkdir -p gpurun_out/r05b; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128" "c2_8192 --batch 8192"; do
  set -- $cfg; name=$1; shift
  rm -rf /tmp/tl_$name
  rocprofv3 --kernel-trace -d /tmp/tl_$name -o t -- python $R/bench.py --profile-inner --num-batches 8 "$@" --steps 40 --warmup 10 > /dev/null 2>&1
  DB=$(find /tmp/tl_$name -name '*.db' | head -1)
  python $R/tools/rocpd_timeline.py $DB vs_gather_mean 20 > $R/gpurun_out/r05b/timeline_$name.txt
  python $R/bench.py --num-batches 8 "$@" --steps 200 --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc > $R/gpurun_out/r05b/bench_$name.json 2> /dev/null
done
cat $R/gpurun_out/r05b/timeline_ps.txt $R/gpurun_out/r05b/timeline_c2_8192.txt
python - <<'P'
import json,os
for n in ('ps','c2_8192'):
    r=json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r05b/bench_%s.json'%n)))
    print(n, r['ms_per_step'], r.get('deferred_loss_readback'), r.get('kernel_us'))
P
