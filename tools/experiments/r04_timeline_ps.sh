# one step at the reference's product-search settings (batch 4096, d_w 300, d_e 128, V_e 32768) as the GPU sees it
mkdir -p gpurun_out/r04i; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tlp
rocprofv3 --kernel-trace -d /tmp/tlp -o t -- python $R/bench.py --profile-inner --batch 4096 --entities 32768 --dim 300 --entity-dim 128 --steps 40 --warmup 10 > /dev/null 2>&1
DB=$(find /tmp/tlp -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB vs_gather_mean 20 > $R/gpurun_out/r04i/timeline_ps.txt
cat $R/gpurun_out/r04i/timeline_ps.txt
