"""TEST INFRASTRUCTURE ONLY — makes the read-only reference importable in the
BUILD container (never on the GPU box): puts /root/reference on sys.path plus a
throw-away stub for its one missing import (colorama,
xrt/backends/raycing/singletons.py:2)."""
import os
import sys
import tempfile

REFERENCE = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REFERENCE, 'xrt'))


def activate():
    if not available():
        raise RuntimeError('reference tree not present (fixtures can only be '
                           'regenerated in the build container)')
    os.environ.setdefault('MPLBACKEND', 'Agg')
    try:
        import colorama  # noqa: F401
    except ImportError:
        stub = os.path.join(tempfile.gettempdir(), 'xrt_amd_colorama_stub')
        os.makedirs(os.path.join(stub, 'colorama'), exist_ok=True)
        with open(os.path.join(stub, 'colorama', '__init__.py'), 'w') as f:
            f.write("class _C:\n    def __getattr__(self, k):\n        return ''\n"
                    "Fore = Back = Style = _C()\n\n\ndef init(*a, **k):\n    pass\n")
        sys.path.insert(0, stub)
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
