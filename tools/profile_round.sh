#!/bin/bash
# Profile one round on the GPU box:  tools/profile_round.sh <tag> [what ...]
#   what = c2 c4 ll fs mfma query   (default: all but query)
# For every workload: rocprofv3 --kernel-trace --stats        -> <tag>_<w>_kernels.txt
#                     rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE  -> <tag>_<w>_pmc.{txt,json}
#   (separate passes, no trace domain besides the kernel trace -- MI355X_MICROARCH.md)
# mfma: SQ counters of the GEMM kernels (MFMA busy cycles vs wave cycles) -> <tag>_mfma.txt
# Everything lands under gpurun_out/<tag>/ (merged back by gpurun); copy into profiles/.
set -u
TAG=${1:-r02_x}; shift || true
WHAT=${*:-c2 c4 ll fs mfma}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc --no-seed-extra"

profile() {   # profile <name> <command...>
    local name=$1; shift
    cd /tmp
    rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt -- "$@" > $OUT/${name}_run.json 2> $OUT/${name}_kt.err
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fe_$name -o fe -- "$@" > /dev/null 2> $OUT/${name}_fe.err
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/wr_$name -o wr -- "$@" > /dev/null 2> $OUT/${name}_wr.err
    cd $ROOT
    local KT=$(find $OUT/kt_$name -name '*.db' | head -1) FE=$(find $OUT/fe_$name -name '*.db' | head -1) WR=$(find $OUT/wr_$name -name '*.db' | head -1)
    [ -n "$KT" ] && python tools/rocpd_summary.py $KT > $OUT/${TAG}_${name}_kernels.txt
    [ -n "$FE" ] && [ -n "$WR" ] && python tools/rocpd_pmc.py $FE $WR --json $OUT/${TAG}_${name}_pmc.json > $OUT/${TAG}_${name}_pmc.txt
    rm -rf $OUT/kt_$name $OUT/fe_$name $OUT/wr_$name
}

for w in $WHAT; do
  case $w in
    c2) profile vs_c2 python $ROOT/bench.py --steps 50 --warmup 5 $NOX ;;
    c4) profile c4 python $ROOT/tools/bench_c4.py --kinds vectorspace --steps 10 ;;
    ll) profile ll_c2 python $ROOT/bench.py --model loglinear --steps 20 --warmup 3 $NOX ;;
    fs) profile fs_c2 python $ROOT/tools/bench_c4.py --kinds vectorspace_softmax --vocab 100000 --entities 1000 --dim 128 --batch 65536 --steps 10 ;;
    query)   # the C5 scoring call (bin/query.py:239-370 batched): kernel stats, HBM bytes, matrix-pipe counters of the filter GEMM
      profile query python $ROOT/bench.py --profile-query-inner
      cd /tmp
      rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/mq -o mq -- python $ROOT/bench.py --profile-query-inner > /dev/null 2> $OUT/mfma_query.err
      cd $ROOT
      MQ=$(find $OUT/mq -name '*.db' | head -1)
      [ -n "$MQ" ] && python tools/gemm_pmc.py $MQ > $OUT/${TAG}_query_mfma.txt
      rm -rf $OUT/mq ;;
    mfma)
      cd /tmp
      rocprofv3 -L 2>/dev/null | grep -i -E "mfma|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_INSTS_VALU " | head -40 > $OUT/${TAG}_counters_available.txt
      rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/mf -o mf -- python $ROOT/bench.py --steps 20 --warmup 3 $NOX > /dev/null 2> $OUT/mfma.err
      cd $ROOT
      MF=$(find $OUT/mf -name '*.db' | head -1)
      [ -n "$MF" ] && python tools/gemm_pmc.py $MF > $OUT/${TAG}_mfma.txt
      rm -rf $OUT/mf ;;
  esac
done
ls -la $OUT
