# one C4 step as the GPU sees it
mkdir -p gpurun_out/r04i; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl4
rocprofv3 --kernel-trace -d /tmp/tl4 -o t -- python $R/bench.py --profile-inner --vocab 500000 --entities 100000 --dim 300 --steps 12 --warmup 4 > /dev/null 2>&1
DB=$(find /tmp/tl4 -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB vs_gather_mean 8 > $R/gpurun_out/r04i/timeline_c4.txt
cat $R/gpurun_out/r04i/timeline_c4.txt
