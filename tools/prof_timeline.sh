#!/bin/bash
# GPU timeline of one bench step (tools/rocpd_timeline.py) on the GPU box -> stdout
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/tl -o tl -- python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-query-extra --no-loglinear-extra "$@" > /dev/null 2>&1
cd $ROOT
python tools/rocpd_timeline.py $(find gpurun_out/tl -name "*.db" | head -1) ${ANCHOR:-vs_gather_mean} 60
rm -rf gpurun_out/tl
