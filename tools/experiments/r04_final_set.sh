# the round-4 profile set: tools/experiments/r04_final_set.sh <tag>   (GPU box; results under gpurun_out/<tag>/)
TAG=${1:-r04x}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG
bash $R/tools/profile_round.sh $TAG c2 c4 ll fs mfma > $R/gpurun_out/$TAG/profile_round.log 2>&1
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/${TAG}_bench_steps20.json 2> gpurun_out/$TAG/bench20.err
python bench.py --steps 100 --warmup 10 > gpurun_out/$TAG/${TAG}_bench_steps100.json 2> gpurun_out/$TAG/bench100.err
SERT_COMM=host SERT_DEVICE=0 python bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/$TAG/${TAG}_bench_dp8_host_transport.json 2> gpurun_out/$TAG/dp8.err
SERT_COMM=host SERT_DEVICE=0 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/$TAG/${TAG}_bench_dp2_host_transport.json 2> gpurun_out/$TAG/dp2.err
bash tools/experiments/r04_timeline.sh > /dev/null 2>&1; cp gpurun_out/r04i/timeline_c2.txt gpurun_out/$TAG/${TAG}_timeline_c2.txt
rm -f gpurun_out/r04z/x3_pmc.txt; bash tools/experiments/r04_x3_pmc.sh > /dev/null 2>&1; cp gpurun_out/r04z/x3_pmc.txt gpurun_out/$TAG/${TAG}_gemm_x3_pmc.txt
bash tools/experiments/r04_gemm_x3.sh > gpurun_out/$TAG/${TAG}_gemm_x3_vs_fp32.txt 2>&1
bash tools/experiments/r04_dp_world1.sh > gpurun_out/$TAG/dp_world1.log 2>&1; cp gpurun_out/r04j/timeline_c2_dp_world1.txt gpurun_out/$TAG/${TAG}_timeline_c2_dp_world1.txt
ls -la gpurun_out/$TAG | head -40
