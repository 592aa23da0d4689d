set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r02a/pytest.log
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r02a/bench.err
head -c 1500 gpurun_out/r02a/bench.json
timeout 900 bash tools/profile_round.sh r02a c4 c2
