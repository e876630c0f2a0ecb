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
