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
# the reflect kernels are instantiated in units of their own (reflect_tu.h): the slowest first
SOURCES = ['reflect_multi.hip', 'reflect_figured_x1.hip', 'reflect_figured_x0.hip', 'reflect_figured_f.hip',
           'reflect_exact1.hip', 'reflect_exact3.hip', 'reflect_exact0.hip',
           'reflect_exact2.hip', 'reflect_layered_x.hip', 'reflect_layered_f.hip',
           'reflect_generic.hip', 'reflect_xtal.hip', 'reflect.hip', 'reflect_hot.hip',
           'reflect_hot_scr.hip', 'reflect_hot_gen.hip', 'reflect_hot_plot.hip', 'reflect_hot_plate2.hip', 'reflect_hot_dcm_scr.hip', 'reflect_hot_xtal_scr.hip',
           'kirchhoff.hip', 'undulator.hip', 'capi.hip', 'screen.hip', 'hist.hip', 'source.hip']
HEADERS = ['fp64_math.h', 'kernarg.h', 'plot_tail.h', 'screen_impl.h', 'source_impl.h', 'kirchhoff.h', 'reflect.h', 'reflect_impl.h', 'reflect_tu.h',
           'reflect_multi_impl.h',
           'screen.h', 'hist.h', 'undulator.h', 'source.h',
           os.path.join('..', '..', 'include', 'xrt_hip.h')]
# headers only these sources depend on (everything else rebuilds on any header change)
ONLY_FOR = {'reflect_impl.h': 'reflect', 'reflect_tu.h': 'reflect',
            'reflect_multi_impl.h': 'reflect_multi', 'kirchhoff.h': ('kirchhoff', 'capi'),
            'hist.h': ('hist', 'capi'), 'screen.h': ('screen', 'capi'), 'source.h': ('source', 'capi'),
            'undulator.h': ('undulator', 'capi')}
# -ffp-contract=off: fused multiply-add only where the source says fma();
# the reference (numpy) never fuses and ray states / the Kirchhoff phase
# depend on bit-identical intermediate roundings.
# -instcombine-max-copied-from-constant-users: the by-value records (xrt_hip_pass, 1 KB) reach a
# kernel as a private copy of the kernarg segment, which the optimiser drops again only if it can
# see that the copy is never written -- and it stops looking after 300 uses. The generic kernels
# read the pass record more often than that; past the limit the whole record lives in scratch
# (1 KB per lane, 12 us of launch overhead on every pass).
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
         '-fPIC', '-fvisibility=hidden', '-Wall', '-Wno-unused-function',
         '-mllvm', '-instcombine-max-copied-from-constant-users=100000']


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


def _headers_of(source):
    out = []
    for h in HEADERS:
        users = ONLY_FOR.get(h)
        if users is None or source.startswith(users):
            out.append(os.path.join(HERE, h))
    return out


def build(force=False, verbose=False, jobs=None):
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    objs = []
    todo = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, 'build', s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src, __file__] + _headers_of(s)):
            todo.append([hipcc] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HERE)

    if todo:
        from concurrent.futures import ThreadPoolExecutor
        jobs = jobs or int(os.environ.get('XRT_HIP_BUILD_JOBS', 0)) or os.cpu_count() or 1
        with ThreadPoolExecutor(max_workers=max(1, min(jobs, len(todo)))) as pool:
            list(pool.map(run, todo))
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd, cwd=HERE)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv))
