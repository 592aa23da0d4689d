set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  for rep in 1 2; do
  SERT_ADAM_SPLIT=$1 SERT_DW_SIDE=$2 python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('split=$1 dwside=$2 ms/step %.4f  deferred %.4f  Mpairs/s %.1f' % (d['ms_per_step'], d['deferred_loss_readback']['ms_per_step'], d['value']/1e6))"
  done
done 2>&1 | tee gpurun_out/r02b/variants.txt
# C4 with and without the split
for s in 0 1; do
  SERT_ADAM_SPLIT=$s python tools/bench_c4.py --kinds vectorspace --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vectorspace']
print('C4 split=$s ms/step %.4f' % d['ms_per_step'])"
done 2>&1 | tee -a gpurun_out/r02b/variants.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02b/pytest.log
