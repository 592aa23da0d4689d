cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "vectorspace" 2>&1 | tail -3
timeout 900 bash tools/profile_round.sh r02e c2 > /dev/null 2>&1
head -22 gpurun_out/r02e/r02e_vs_c2_kernels.txt | cut -c1-150
head -24 gpurun_out/r02e/r02e_vs_c2_pmc.txt
