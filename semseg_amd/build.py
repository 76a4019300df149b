"""Builds semseg_amd/libsemseg_hip.so (gfx950 only) from semseg_amd/csrc/*.hip with hipcc, in-tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', 'build')
LIB = os.path.join(HERE, 'libsemseg_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast', '-Wno-unused-result']


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + '.o')
        if force or _newer(src, obj, hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        r = subprocess.run([hipcc] + FLAGS + ['-c', src, '-o', obj], capture_output=True, text=True)
        return src, r.returncode, r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for src, rc, err in ex.map(cc, jobs):
            if verbose:
                print(f'[build] hipcc {os.path.basename(src)} -> rc {rc}')
            if rc != 0:
                raise RuntimeError(f'hipcc failed for {src}:\n{err}')
    objs = [os.path.join(OBJ, s[:-4] + '.o') for s in srcs]
    if force or jobs or not os.path.exists(LIB):
        r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr)
        if verbose:
            print('[build] linked', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
