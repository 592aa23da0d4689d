# one step at the reference's W3C expert-finding settings (loglinear, batch 1024, d 300, V_e 715, window 8) as the GPU sees it
mkdir -p gpurun_out/r04i; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tlw
rocprofv3 --kernel-trace -d /tmp/tlw -o t -- python $R/bench.py --profile-inner --model loglinear --batch 1024 --dim 300 --entities 715 --window 8 --steps 40 --warmup 10 > /dev/null 2>&1
DB=$(find /tmp/tlw -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB ll_gather_rows 20 > $R/gpurun_out/r04i/timeline_w3c.txt
cat $R/gpurun_out/r04i/timeline_w3c.txt
