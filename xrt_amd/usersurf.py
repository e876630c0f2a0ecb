"""User-defined surfaces on the GPU.

In xrt a new surface is an ``OE`` subclass with its own ``local_z`` / ``local_n``; for the
accelerated path the reference takes the same two functions as OpenCL source strings on the
class (``cl_local_z``, ``cl_local_n``, parameters in ``cl_plist``: oes/base.py:69-90) and
splices them into its kernel (oes/base.py:552-564, ``find_intersection_CL`` :887-931). Here:

    class Saddle(roe.OE):
        hip_plist = property(lambda self: (self.cx, self.cy))       # up to 12 numbers -> p[]
        # (a grating may add hip_local_g = 'g[0] = 0.; g[1] = p[2] * (1 + p[3] * y); g[2] = 0.;')
        hip_local_z = 'return p[0] * x * x - p[1] * y * y;'
        hip_local_n = '''double a = -2. * p[0] * x, b = 2. * p[1] * y;
                         double r = 1. / sqrt(a * a + b * b + 1.);
                         n[0] = a * r; n[1] = b * r; n[2] = r;'''

The two snippets are the bodies of ``double local_z(double x, double y, const double* p)`` and
``void local_n(double x, double y, const double* p, double* n)`` (n = the unit normal) in HIP
C++. Once per class they are compiled (``hipcc``, ~40 s, cached on disk by content) into a
unit that holds the ray kernels instantiated around them (csrc/user_unit.hip.in), and
libxrt_hip.so opens it (``xrt_hip_user_surface_load``); the element then runs the same fused
solve + reflect pass as the built-in surfaces. A subclass that only overrides the numpy
methods still raises: Python code cannot run in a kernel.
"""
import ctypes
import hashlib
import os
import shutil
import subprocess
import threading

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
from .csrc.build import FLAGS as _BUILD_FLAGS     # the library's own flags: one place

_FLAGS = [f for f in _BUILD_FLAGS if f != '-Wall'] + ['-shared']
_lock = threading.Lock()
_handles = {}       # unit path -> handle


def cache_dir():
    d = os.environ.get('XRT_HIP_USER_CACHE') or os.path.join(
        os.path.expanduser('~'), '.cache', 'xrt_amd', 'units')
    os.makedirs(d, exist_ok=True)
    return d


def _headers_digest():
    h = hashlib.sha256()
    for name in ('reflect_impl.h', 'screen_impl.h', 'source_impl.h', 'reflect_multi_impl.h', 'reflect_tu.h', 'reflect.h', 'fp64_math.h', 'user_unit.hip.in',
                 os.path.join('..', '..', 'include', 'xrt_hip.h')):
        with open(os.path.join(_CSRC, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


_NO_GROOVES = 'g[0] = 0.; g[1] = 0.; g[2] = 0.;'


def unit_source(local_z, local_n, local_g=None, layered=False):
    """The HIP source of the unit around the snippets (*local_g*: the groove vector of a
    grating, optional; *layered*: the flavour compiled around Parratt's recursion, for
    Multilayer / Coated materials)."""
    with open(os.path.join(_CSRC, 'user_unit.hip.in')) as f:
        text = f.read()
    for marker, body in (('@LOCAL_Z@', local_z), ('@LOCAL_N@', local_n),
                         ('@LOCAL_G@', _NO_GROOVES if local_g is None else local_g)):
        if not isinstance(body, str) or not body.strip():
            raise ValueError('hip_local_z / hip_local_n / hip_local_g must be non-empty source '
                             'strings')
        text = text.replace(marker, body)
    return text.replace('@CSRC@', _CSRC).replace('@LAYERED@', '1' if layered else '0')


def build_unit(local_z, local_n, verbose=False, local_g=None, layered=False):
    """Compiles (or finds in the cache) the unit of the snippets -> path of its .so."""
    source = unit_source(local_z, local_n, local_g, layered)
    key = hashlib.sha256((source + _headers_digest() + ' '.join(_FLAGS)).encode()).hexdigest()[:24]
    out = os.path.join(cache_dir(), 'surface_%s.so' % key)
    with _lock:
        if os.path.exists(out):
            return out
        hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
        if not os.path.exists(hipcc):
            raise _lib.XrtHipError('hipcc not found: a user-defined surface is compiled at run '
                                   'time (there is no CPU fallback)')
        # several ranks may build the same class at once: each writes its own source and object
        # (the pid in the names) and moves the result into place; whoever comes second
        # overwrites the same bytes (one shared .hip was truncated under a reader)
        src = out[:-3] + '.%d.hip' % os.getpid()
        with open(src, 'w') as f:
            f.write(source)
        tmp = out + '.%d.tmp' % os.getpid()
        cmd = [hipcc] + _FLAGS + [src, '-o', tmp]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        try:
            os.remove(src)
        except OSError:
            pass
        if r.returncode != 0:
            raise _lib.XrtHipError('the surface snippets do not compile:\n' + r.stderr[-4000:])
        os.replace(tmp, out)
        return out


def load_unit(path):
    """-> the handle libxrt_hip.so gives for the unit at *path* (one per process and path)."""
    with _lock:
        if path in _handles:
            return _handles[path]
        lib = _lib.load()
        handle = ctypes.c_void_p()
        _lib.check(lib.xrt_hip_user_surface_load(path.encode(), ctypes.byref(handle)),
                   'xrt_hip_user_surface_load')
        _handles[path] = handle.value
        return handle.value


def snippets_of(oe):
    """(hip_local_z, hip_local_n) of an element's class, or None."""
    z, n = getattr(oe, 'hip_local_z', None), getattr(oe, 'hip_local_n', None)
    if z is None and n is None:
        return None
    if not (isinstance(z, str) and isinstance(n, str)):
        raise ValueError('%s: hip_local_z and hip_local_n come together, as source strings'
                         % type(oe).__name__)
    return z, n


def groove_snippet_of(oe):
    """hip_local_g of an element's class (the body of
    ``void local_g(double x, double y, const double* p, double* g)``), or None."""
    g = getattr(oe, 'hip_local_g', None)
    if g is not None and not isinstance(g, str):
        raise ValueError('%s: hip_local_g is a source string' % type(oe).__name__)
    return g


def parameters_of(oe):
    """p[0..11] of an element: its hip_plist (attribute, property or method)."""
    plist = getattr(oe, 'hip_plist', ())
    if callable(plist):
        plist = plist()
    values = [float(v) for v in plist]
    if len(values) > 12:
        raise ValueError('hip_plist holds %d numbers, the kernels take 12' % len(values))
    return values + [0.] * (12 - len(values))


def unit_for(oe, layered=False):
    """The loaded unit of the element's class (compiled on first use) -> handle. *layered*:
    the flavour for Multilayer / Coated materials (a second unit of the same snippets)."""
    z, n = snippets_of(oe)
    return load_unit(build_unit(z, n, local_g=groove_snippet_of(oe), layered=layered))
