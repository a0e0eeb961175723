"""Build libdetzero_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

    python -m detzero_amd.build [--force]
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJDIR = os.path.join(HERE, 'csrc', 'build')
LIB = os.path.join(HERE, 'libdetzero_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']
if os.environ.get('DZ_BUILD_EXPERIMENTAL', '0') not in ('', '0'):
    # the engines that were built, parity-tested and measured SLOWER than the shipped ones: the tile-resident sparse
    # convolution (sparse_conv_t.hip) and the direct-to-LDS gather variant (sparse_conv_d.h).  Not part of the default library; their
    # tests carry the `experimental` marker (tests/conftest.py)
    FLAGS.append('-DDZ_BUILD_EXPERIMENTAL')
FLAGS += os.environ.get('DZ_HIPCC_FLAGS', '').split()      # development only (e.g. -DDZ_SPCONV_DIAG: tools/gpu_diag.sh)


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(HERE, '..', 'include', 'detzero_hip.h'))
    return hs


def _digest(paths):
    h = hashlib.sha1()
    for p in sorted(paths):
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_digest = _digest(_headers())
    jobs, objs = [], []
    for src in _sources():
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, src[:-4] + '.o')
        stamp = obj + '.stamp'
        want = _digest([sp]) + hdr_digest
        have = open(stamp).read() if os.path.exists(stamp) else ''
        objs.append(obj)
        if force or not os.path.exists(obj) or have != want:
            jobs.append((sp, obj, stamp, want))

    def compile_one(job):
        sp, obj, stamp, want = job
        cmd = [HIPCC] + FLAGS + ['-c', sp, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp, 'w') as f:
            f.write(want)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
