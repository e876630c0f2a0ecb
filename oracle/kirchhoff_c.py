"""TEST INFRASTRUCTURE ONLY — ctypes wrapper and build recipe of the C/OpenMP
restatement oracle/kirchhoff_c.c (all-core CPU baseline for bench.py)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'kirchhoff_c.c')
LIB = os.path.join(HERE, '_build', 'libxrt_oracle_kirchhoff.so')
_lib = None


def build(force=False):
    """gcc -O2 -fopenmp; the .so is git-ignored but travels with the snapshot."""
    if not force and os.path.exists(LIB) and \
            os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = ['gcc', '-O2', '-fopenmp', '-shared', '-fPIC', SRC, '-o', LIB, '-lm']
    subprocess.check_call(cmd)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.xrt_oracle_max_threads.restype = ctypes.c_int
        _lib.xrt_oracle_kirchhoff.restype = None
    return _lib


def max_threads():
    return int(load().xrt_oracle_max_threads())


def kirchhoff(px, py, pz, sx, sy, sz, n, nl, k, Es, Ep):
    """Same arguments as kirchhoff_np.kirchhoff_conv but with the wavenumber k
    [1/mm] instead of E. Returns the five raw integrals (numpy convention)."""
    lib = load()
    f = lambda a, m: np.ascontiguousarray(  # noqa: E731
        np.broadcast_to(np.asarray(a, dtype=np.float64), (m,)))
    c = lambda a, m: np.ascontiguousarray(  # noqa: E731
        np.broadcast_to(np.asarray(a, dtype=np.complex128), (m,)))
    npix, ns = len(px), len(sx)
    ins = [f(px, npix), f(py, npix), f(pz, npix), f(sx, ns), f(sy, ns), f(sz, ns),
           f(n[0], ns), f(n[1], ns), f(n[2], ns), f(nl, ns), f(k, ns), c(Es, ns),
           c(Ep, ns)]
    outs = [np.zeros(npix, dtype=np.complex128) for _ in range(5)]
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    lib.xrt_oracle_kirchhoff(
        ctypes.c_int64(npix), ptr(ins[0]), ptr(ins[1]), ptr(ins[2]), ctypes.c_int64(ns),
        *[ptr(a) for a in ins[3:]], *[ptr(a) for a in outs])
    return tuple(outs)
