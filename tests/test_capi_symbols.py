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


def test_crystal_surface_combinations_the_c_abi_takes():
    """Round 4: Bragg crystals on conics, lens paraboloids, cones, VFM and DualVFM go through the
    generic exact sequence of the surface's family (tests/test_gpu_reflect.py has the goldens);
    a blazed profile has no Bragg planes and is refused by the C ABI before any GPU work, with a
    reason; a user-defined surface takes crystals (its one normal serves the atomic planes too,
    golden g3_user_crystal) on its general unit and Multilayer / Coated on the layered flavour
    of it (goldens g2_user_multilayer, g2_user_coated); the wrong flavour is refused."""
    import ctypes
    from xrt_amd import _lib, _structs
    lib = _lib.load(build_if_missing=False)
    lib.xrt_hip_last_error.restype = ctypes.c_char_p

    def call(kind, unit=None, mat=4):     # 4: XRT_HIP_MAT_CRYSTAL
        p, m = _structs.Pass(), _structs.Material()
        p.surf_kind = kind
        p.invert_normal = 1
        if unit:
            p.user_unit = unit
        m.kind = mat
        rc = lib.xrt_hip_reflect_pass_f64_dev(ctypes.byref(p), ctypes.byref(m), None, None, None,
                                              None, None, None, ctypes.c_size_t(0), None, None,
                                              None)
        return rc, lib.xrt_hip_last_error()
    rc, why = call(3)
    assert rc != 0 and b'crystals on blazed gratings' in why
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import user_surface_case as case
    from xrt_amd import usersurf
    general = usersurf.load_unit(usersurf.build_unit(case.HIP_LOCAL_Z, case.HIP_LOCAL_N))
    layered = usersurf.load_unit(usersurf.build_unit(case.HIP_LOCAL_Z, case.HIP_LOCAL_N,
                                                     layered=True))
    assert general != layered
    rc, why = call(12, unit=general, mat=_structs.MAT_MULTILAYER)
    assert rc != 0 and b'need the layered flavour' in why and b'this unit: general' in why
    rc, why = call(12, unit=layered)
    assert rc != 0 and b'need the layered flavour' in why and b'this unit: layered' in why
    rc, why = call(12, unit=general)       # a crystal on the general unit: accepted so far
    assert rc != 0 and b'surface' not in why and b'crystals' not in why, why
    rc, why = call(12, unit=layered, mat=_structs.MAT_MULTILAYER)
    assert rc != 0 and b'flavour' not in why, why
    rc, why = call(12)
    assert rc != 0 and b'without its compiled unit' in why
    for kind in (4, 5, 6, 9, 10):       # accepted: the call fails later, on the empty records
        rc, why = call(kind)
        assert rc != 0 and b'surface' not in why and b'crystals' not in why, (kind, why)
