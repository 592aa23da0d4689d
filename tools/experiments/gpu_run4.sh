cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
run() { env "$@" SERT_EGRAD_SORT=1 python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$*  ms/step %.4f' % d['ms_per_step'])"; }
run A=1
run SERT_KO_EGRAD=1
run SERT_STREAMS=1
run SERT_STREAMS=1 SERT_KO_EGRAD=1
