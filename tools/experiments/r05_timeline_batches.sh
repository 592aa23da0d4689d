# the same step without the host round trip (sert_train_batches): do the gaps in front of the word-table update and the tail stay?
mkdir -p gpurun_out/r05d; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "c2_8192 8192" "c2 65536"; do
  set -- $cfg; name=$1; shift
  rm -rf /tmp/tlb_$name
  rocprofv3 --kernel-trace -d /tmp/tlb_$name -o t -- python $R/tools/experiments/r05_inner_batches.py "$@" > /dev/null 2>&1
  DB=$(find /tmp/tlb_$name -name '*.db' | head -1)
  python $R/tools/rocpd_timeline.py $DB vs_gather_mean 20 > $R/gpurun_out/r05d/timeline_batches_$name.txt
  cat $R/gpurun_out/r05d/timeline_batches_$name.txt
done
