# one C2 step as the GPU sees it (kernel trace of the bench's inner loop)
mkdir -p gpurun_out/r04i; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl
rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $R/bench.py --profile-inner --steps 40 --warmup 10 > /dev/null 2>&1
DB=$(find /tmp/tl -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB vs_gather_mean 5 > $R/gpurun_out/r04i/timeline_c2.txt
python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/r04i/kernels_c2.txt
cat $R/gpurun_out/r04i/timeline_c2.txt
