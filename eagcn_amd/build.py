"""Build libeagcn_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is plain C ABI."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libeagcn_hip.so')
HIP_SOURCES = ['index.hip', 'gemm.hip', 'gemm3.hip', 'gemm_bx3.hip', 'gemm_bx3w.hip', 'agg.hip', 'lagg.hip', 'gat.hip', 'pool.hip', 'layer.hip', 'readout.hip', 'head.hip', 'head2.hip', 'loss.hip']
CPP_SOURCES = ['api.cpp']
HEADERS = ['common.h', 'kernels.h', 'gemm_x6.h', 'bx3.h', os.path.join('..', '..', 'include', 'eagcn_hip.h')]
ARCH = 'gfx950'


def _hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the EAGCN HIP library cannot be built')


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def needs_build():
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES + CPP_SOURCES + HEADERS if os.path.exists(os.path.join(CSRC, s))]
    return (not os.path.exists(LIB)) or os.path.getmtime(LIB) < _newest(srcs)


def build_library(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the shared library."""
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    flags = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-munsafe-fp-atomics']

    hdr_time = _newest([os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))])

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + '.o')
        # an object newer than its source and every header is kept (the library is relinked from all objects)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_time, os.path.getmtime(os.path.join(CSRC, src))):
            return obj
        cmd = [hipcc] + flags + ['-c', os.path.join(CSRC, src), '-o', obj]
        if src.endswith('.cpp'):
            cmd = [hipcc, '-O2', '-std=c++17', '-fPIC', '-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    sources = [s for s in HIP_SOURCES + CPP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(8, len(sources))) as ex:
        objs = list(ex.map(compile_one, sources))
    tmp = LIB + '.tmp'
    r = subprocess.run([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', tmp] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    os.replace(tmp, LIB)
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
