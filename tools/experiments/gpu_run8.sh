cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "vectorspace" 2>&1 | tail -4
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
run() { env "$@" python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('$*  ms/step %.4f Mpairs/s %.1f | fwd %.1f dW %.1f comb %.1f dX %.1f' % (d['ms_per_step'], d['value']/1e6, k['gemm_fwd']['us'], k['gemm_dW']['us'], k['splitk_combine']['us'], k['gemm_dX']['us']))"; }
run SERT_NO_STRIP_GEMM=1
run A=1
run SERT_STRIP_WGS=1
run SERT_STRIP_WGS=2 SERT_X=1
run SERT_STRIP_DW_WGS=256
run SERT_STRIP_DW_WGS=1024
