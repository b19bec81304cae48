"""
Builds the HIP extension in-tree:  em_pose_amd/csrc/*.hip -> em_pose_amd/csrc/libempose_hip.so  (gfx950 only).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libempose_hip.so')
ARCH = 'gfx950'


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        glob.glob(os.path.join(CSRC, '..', '..', 'include', '*.h'))
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force=False, verbose=True):
    """Compile every .hip source to an object (parallel), then link the shared library."""
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    flags = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics',
             '-Wno-unused-result']
    objs, procs = [], []
    for src in sources():
        obj = src[:-4] + '.o'
        objs.append(obj)
        procs.append((src, subprocess.Popen([hipcc] + flags + ['-c', src, '-o', obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on {}:\n{}'.format(src, out.decode()))
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode != 0:
        raise RuntimeError('link failed:\n' + out.stdout.decode())
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
