"""CPU: libxrt_hip.so builds, loads, and exports every symbol that
include/xrt_hip.h declares; the ctypes table covers them all. No compute."""
import os
import re

import pytest

from xrt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    inc = os.path.join(ROOT, 'include')
    for fn in sorted(os.listdir(inc)):
        if fn.endswith('.h'):
            text = open(os.path.join(inc, fn)).read()
            names += re.findall(r'XRT_HIP_API\s+[\w\s\*]+?\b(xrt_hip_\w+)\s*\(', text)
    return names


def test_header_declares_symbols():
    names = declared_symbols()
    assert 'xrt_hip_kirchhoff_f64' in names and 'xrt_hip_kirchhoff_f64_dev' in names
    assert len(names) == len(set(names))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name + ' not exported'


def test_ctypes_table_matches_header():
    assert set(_lib.SIGNATURES) == set(declared_symbols())


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.xrt_hip_version() >= 100
    assert isinstance(lib.xrt_hip_last_error(), bytes)


def test_argument_validation_without_gpu():
    lib = _lib.load()
    rc = lib.xrt_hip_kirchhoff_plan(-1, 10, 0, 0, None, None, None)
    assert rc == -1 and b'negative' in lib.xrt_hip_last_error()
    with pytest.raises(_lib.XrtHipError):
        _lib.check(rc, 'plan')


def test_crystal_on_a_conic_or_vfm_surface_is_refused():
    """A Bragg crystal on a surface kind only the family-1 / family-2 kernels evaluate
    (parametric conics, lenses, cone, blazed, VFM, DualVFM) is refused by the C ABI before any
    GPU work: the crystal kernels are compiled for family 0 and would trace it as flat."""
    import ctypes
    from xrt_amd import _lib, _structs
    lib = _lib.load(build_if_missing=False)
    lib.xrt_hip_last_error.restype = ctypes.c_char_p
    for kind in (3, 4, 5, 6, 9, 10):
        p, m = _structs.Pass(), _structs.Material()
        p.surf_kind = kind
        m.kind = 4          # XRT_HIP_MAT_CRYSTAL
        rc = lib.xrt_hip_reflect_pass_f64_dev(ctypes.byref(p), ctypes.byref(m), None, None, None,
                                              None, None, None, ctypes.c_size_t(0), None, None,
                                              None)
        assert rc != 0, kind
        assert b'crystals on blazed / parametric surfaces' in lib.xrt_hip_last_error()
