# fork behind the loss kernel with dW / db FIRST on the side stream (beside the dh GEMM), the entity chain behind them
# (SERT_FORK_AT=nce_dw, variants library) against the default (fork behind the dh GEMM); A/B/A/B on one box + a timeline
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r05i
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-10s fork=%-6s ms/step %.4f' % ('$name', '${SERT_FORK_AT:-dh}', r['ms_per_step']))"
}
for rep in 1 2; do for f in dh nce_dw nce; do
  export SERT_FORK_AT=$f
  run c2 --batch 65536
  run c2_32k --batch 32768
  run c2_8192 --batch 8192
done; done
cd /tmp; export TMPDIR=/tmp
export SERT_FORK_AT=nce_dw
rm -rf /tmp/tl_f
rocprofv3 --kernel-trace -d /tmp/tl_f -o t -- python $R/bench.py --profile-inner --num-batches 8 --batch 65536 --steps 40 --warmup 10 > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/tl_f -name '*.db' | head -1) vs_gather_mean 24 > $R/gpurun_out/r05i/timeline_c2_nce_dw.txt
tail -22 $R/gpurun_out/r05i/timeline_c2_nce_dw.txt
