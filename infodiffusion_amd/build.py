"""Build libinfodiff_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libinfodiff_hip.so')


# q_sample / sampler updates must round every mul/add separately to match the CPU path bitwise
EXTRA = {'idf_elementwise.hip': ['-ffp-contract=off']}


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.h'))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, 'build', os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17'] + EXTRA.get(os.path.basename(s), []) + ['-c', s, '-o', o]
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    subprocess.check_call(cmd)
    if verbose:
        print('built', LIB, file=sys.stderr)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
