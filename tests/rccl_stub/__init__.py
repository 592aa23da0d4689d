"""Build recipe of the TEST-ONLY librccl.so.1 stand-in (rccl_stub.cpp; see its header).  Nothing under
sert_amd/ refers to this directory; tests put `_build/` ahead on LD_LIBRARY_PATH of the ranks they start."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'rccl_stub.cpp')
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'librccl.so.1')


def build(force=False):
    """g++ (host code only; the HIP runtime API through libamdhip64) -> tests/rccl_stub/_build/librccl.so.1"""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = LIB + '.tmp.%d' % os.getpid()
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-D__HIP_PLATFORM_AMD__',
                           '-I' + os.path.join(rocm, 'include'), SRC, '-o', tmp,
                           '-L' + os.path.join(rocm, 'lib'), '-Wl,-rpath,' + os.path.join(rocm, 'lib'),
                           '-lamdhip64', '-lrt', '-lpthread', '-Wl,-soname,librccl.so.1'])
    os.replace(tmp, LIB)
    return LIB


def env_with_stub(env=None):
    """A copy of `env` whose LD_LIBRARY_PATH resolves the bare name librccl.so.1 to the stand-in."""
    env = dict(os.environ if env is None else env)
    build()
    env['LD_LIBRARY_PATH'] = OUT_DIR + (':' + env['LD_LIBRARY_PATH'] if env.get('LD_LIBRARY_PATH') else '')
    return env
