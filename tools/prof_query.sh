#!/bin/bash
# kernel trace of the query path (tools/query_bench.py) on the GPU box -> stdout
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/qkt -o kt -- python $ROOT/tools/query_bench.py > $ROOT/gpurun_out/q.log 2>&1
cd $ROOT
tail -1 gpurun_out/q.log | cut -c1-250
python tools/rocpd_summary.py $(find gpurun_out/qkt -name "*.db" | head -1) | cut -c1-200
rm -rf gpurun_out/qkt
