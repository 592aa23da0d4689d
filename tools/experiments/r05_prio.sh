# stream priorities: the latency-bound side chain above / below the bandwidth-bound main chain (SERT_PRIO=side|main), A/B on one box
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps 200 --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('%-10s prio=%-5s ms/step %.4f' % ('$name', '${SERT_PRIO:-none}', r['ms_per_step']))"
}
for rep in 1 2; do for p in none side main; do
if [ $p = none ]; then unset SERT_PRIO; else export SERT_PRIO=$p; fi
run c2 --batch 65536
run c2_8192 --batch 8192
run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
run c4 --vocab 500000 --entities 100000 --dim 300
done; done
