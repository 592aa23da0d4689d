"""Build recipe for libsert_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsert_hip.so')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')

SOURCES = ['sert_hip.hip']

def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]   # (every header of csrc/)
    vdir = os.path.join(CSRC, 'variants')                                             # (... and of csrc/variants/)
    if os.path.isdir(vdir):
        deps += [os.path.join(vdir, f) for f in os.listdir(vdir) if f.endswith('.h')]
    hdir = os.path.join(CSRC, 'host')                                                 # (... and the host-side parts of sert_hip.hip)
    if os.path.isdir(hdir):
        deps += [os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith('.inc')]
    deps.append(os.path.join(INCLUDE, 'sert_hip.h'))
    deps.append(os.path.join(INCLUDE, 'sert_hip_debug.h'))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into sert_amd/libsert_hip.so."""
    if not force and not _stale():
        return LIB
    # several ranks of one node may arrive here together (python bench.py --gpus N on a box whose library is stale): one
    # builds, the others wait for the lock and find the library fresh
    import fcntl
    try:
        lock = open(LIB + '.lock', 'w')
    except OSError:
        # read-only install: no lock file can be made -- build if there is a toolchain (it fails on the write as it
        # should), else return the library that is there
        return _build_locked(verbose)
    try:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            return LIB
        return _build_locked(verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(verbose):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB  # GPU box without a toolchain: use the prebuilt library
        raise RuntimeError('hipcc not found and no prebuilt libsert_hip.so')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-I' + INCLUDE]      # (no -munsafe-fp-atomics: the design has no floating-point atomics)
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    tmp = LIB + '.tmp.%d' % os.getpid()
    cmd += ['-o', tmp, '-ldl']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)       # (atomic: a process that loads the library meanwhile sees the old or the new file, never half of one)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
