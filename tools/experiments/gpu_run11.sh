cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/r02g/bench.json 2> gpurun_out/r02g/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r02g/bench.err
timeout 600 python tools/bench_c4.py --kinds vectorspace_softmax --batch 65536 --steps 2 > gpurun_out/r02g/c4_fs_b65536.json 2> gpurun_out/r02g/c4_fs.err; echo "c4 fs rc=$?"; tail -2 gpurun_out/r02g/c4_fs.err; cat gpurun_out/r02g/c4_fs_b65536.json | head -30
timeout 600 python tools/bench_c4.py --kinds loglinear --steps 5 > gpurun_out/r02g/c4_ll.json 2>/dev/null; head -8 gpurun_out/r02g/c4_ll.json
