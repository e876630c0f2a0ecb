"""Builds libxrt_hip.so (gfx950) in-tree with hipcc. No cmake, no JIT cache:
the .so lands next to the package so that it travels with the source tree.

    python -m xrt_amd.csrc.build [--force] [--verbose]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, 'libxrt_hip.so')
SOURCES = ['kirchhoff.hip', 'reflect.hip', 'screen.hip', 'hist.hip', 'undulator.hip',
           'capi.hip']
HEADERS = ['fp64_math.h', 'kirchhoff.h', 'reflect.h', 'screen.h', 'hist.h', 'undulator.h',
           os.path.join('..', '..', 'include', 'xrt_hip.h')]
# -ffp-contract=off: fused multiply-add only where the source says fma();
# the reference (numpy) never fuses and ray states / the Kirchhoff phase
# depend on bit-identical intermediate roundings.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
         '-fPIC', '-fvisibility=hidden', '-Wall', '-Wno-unused-function']


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found (need ROCm to build libxrt_hip.so)')
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, 'build', s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src, __file__] + hdrs):
            cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd, cwd=HERE)
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd, cwd=HERE)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv))
