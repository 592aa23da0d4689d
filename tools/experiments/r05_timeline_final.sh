# one step as the GPU sees it, final tree of round 5: C2, C2 dims at 8192 rows, product-search settings
mkdir -p gpurun_out/r05i; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "c2 --batch 65536" "c2_8192 --batch 8192" "ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128"; do
  set -- $cfg; name=$1; shift
  rm -rf /tmp/tl_$name
  rocprofv3 --kernel-trace -d /tmp/tl_$name -o t -- python $R/bench.py --profile-inner --num-batches 8 "$@" --steps 40 --warmup 10 > /dev/null 2>&1
  DB=$(find /tmp/tl_$name -name '*.db' | head -1)
  python $R/tools/rocpd_timeline.py $DB vs_gather_mean 24 > $R/gpurun_out/r05i/timeline_$name.txt
done
