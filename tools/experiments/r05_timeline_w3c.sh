# one step of the reference's W3C loglinear settings (batch 1024, window 8, 715 experts, d 300) as the GPU sees it
mkdir -p gpurun_out/r05g; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tlw
rocprofv3 --kernel-trace -d /tmp/tlw -o t -- python $R/bench.py --profile-inner --model loglinear --num-batches 8 --batch 1024 --window 8 --entities 715 --dim 300 --steps 40 --warmup 10 > /dev/null 2>&1
DB=$(find /tmp/tlw -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB ll_gather_rows 20 | tee $R/gpurun_out/r05g/timeline_w3c.txt
