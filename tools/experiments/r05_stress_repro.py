"""One shape of tools/stress_parity.py again, with the traceback (argv: lib path or '-')."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] != '-':
    os.environ['SERT_LIB'] = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as T
dims = {'B': 129, 'n': 2, 'z': 17, 'Vw': 70000, 'Ve': 2, 'dw': 128, 'de': 128}
try:
    T.test_vectorspace_steps(None, dims, 'default', None)
    print('ok', os.environ.get('SERT_LIB', 'product'))
except Exception:
    traceback.print_exc()
    print('FAIL', os.environ.get('SERT_LIB', 'product'))
