cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tests
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/tests/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/tests/pytest.log | tail -5
