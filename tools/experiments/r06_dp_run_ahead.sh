# Round 6, item 4: the data-parallel step with the run-ahead of the announced batch (forward + backward + its collectives) against without
# (SERT_DP_RUN_AHEAD=0, variants library); RCCL attached with a world of one (SERT_FORCE_COMM=1); and the single-GPU step beside it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06g
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
NOX="--num-batches 8 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc"
for rep in 1 2; do for b in 65536 32768 16384 8192; do
  for v in single dp_ahead dp_r5; do
    unset SERT_FORCE_COMM SERT_DP_RUN_AHEAD
    [ $v = dp_ahead ] && export SERT_FORCE_COMM=1
    [ $v = dp_r5 ] && export SERT_FORCE_COMM=1 SERT_DP_RUN_AHEAD=0
    python bench.py --batch $b --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('batch %6d %-9s %.4f ms  %.1f M pairs/s  loss %.6f' % ($b, '$v', r['ms_per_step'], r['value']/1e6, r['last_loss']))"
  done
done; done 2>&1 | tee gpurun_out/r06g/dp_run_ahead.txt
