mkdir -p gpurun_out/r05h; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in 0 1; do
rm -rf /tmp/tlp$r
SERT_EGRAD_RANGES=$r rocprofv3 --kernel-trace -d /tmp/tlp$r -o t -- python $R/bench.py --profile-inner --num-batches 8 --batch 4096 --entities 32768 --dim 300 --entity-dim 128 --steps 40 --warmup 10 > /dev/null 2>&1
DB=$(find /tmp/tlp$r -name '*.db' | head -1)
echo "== SERT_EGRAD_RANGES=$r"; python $R/tools/rocpd_timeline.py $DB vs_gather_mean 20 | tee $R/gpurun_out/r05h/timeline_ps_ranges$r.txt
done
