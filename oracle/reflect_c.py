"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper and build recipe of oracle/reflect_c.c, the
C/OpenMP restatement of OE.reflect for flat / toroidal Fresnel mirrors (the all-cores CPU
baseline of bench.py's ray-tracing leg). Takes the oracle's parameter dictionary
(oracle/adapters.oracle_params) and an oracle Beam; checked against oracle/reflect_np.py
by tests/test_oracle_reflect_c.py."""
import ctypes
import os
import subprocess

import numpy as np

from . import reflect_np as rn

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'reflect_c.c')
LIB = os.path.join(HERE, '_build', 'libxrt_oracle_reflect.so')
MAXROT, MAXELEM = 8, 4
_lib = None


class Rot(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int), ('axis', ctypes.c_int * MAXROT),
                ('c', ctypes.c_double * MAXROT), ('s', ctypes.c_double * MAXROT)]


class OE(ctypes.Structure):
    _fields_ = [('center', ctypes.c_double * 3), ('sin_az', ctypes.c_double),
                ('cos_az', ctypes.c_double), ('to_local', Rot), ('to_virgin', Rot),
                ('dx', ctypes.c_double), ('surf', ctypes.c_int), ('R', ctypes.c_double),
                ('r', ctypes.c_double), ('phys_x', ctypes.c_double * 2),
                ('phys_y', ctypes.c_double * 2), ('has_opt_x', ctypes.c_int),
                ('has_opt_y', ctypes.c_int), ('opt_x', ctypes.c_double * 2),
                ('opt_y', ctypes.c_double * 2), ('over_mask', ctypes.c_int),
                ('lost_num', ctypes.c_int), ('roll', ctypes.c_double),
                ('nelem', ctypes.c_int), ('Z', ctypes.c_int * MAXELEM),
                ('tab_n', ctypes.c_int * MAXELEM), ('quantity', ctypes.c_double * MAXELEM),
                ('tab_E', ctypes.c_void_p * MAXELEM), ('tab_f1', ctypes.c_void_p * MAXELEM),
                ('tab_f2', ctypes.c_void_p * MAXELEM), ('rho', ctypes.c_double),
                ('mass', ctypes.c_double)]


class BeamRec(ctypes.Structure):
    _fields_ = [(f, ctypes.c_void_p) for f in
                ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'state')]


def build(force=False):
    """gcc -O2 -fopenmp; the .so is git-ignored but travels with the snapshot."""
    if not force and os.path.exists(LIB) and \
            os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(['gcc', '-O2', '-fopenmp', '-shared', '-fPIC', SRC, '-o', LIB,
                           '-lm'])
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.xrt_oracle_reflect_max_threads.restype = ctypes.c_int
        _lib.xrt_oracle_reflect.restype = ctypes.c_int
    return _lib


def max_threads():
    return int(load().xrt_oracle_reflect_max_threads())


def _fill_rot(rec, steps):
    rec.n = len(steps)
    for k, (axis, c, s) in enumerate(steps):
        rec.axis[k], rec.c[k], rec.s[k] = 'xyz'.index(axis), c, s


def _record(p):
    """oracle parameter dictionary -> (C struct, arrays to keep alive)."""
    surf = p['surface']
    if surf['kind'] not in ('flat', 'toroid') or surf.get('alpha'):
        raise NotImplementedError('reflect_c restates flat and toroidal mirrors only')
    m = p.get('material')
    if m is None or m['kind'] != 'mirror':
        raise NotImplementedError('reflect_c restates Fresnel mirror coatings only')
    if any(p.get(k, 0) for k in ('extraPitch', 'extraRoll', 'extraYaw')) or \
            not p.get('shape', 'rect').startswith('re'):
        raise NotImplementedError('extra rotations / round shapes')
    o = OE()
    keep = []
    for k in range(3):
        o.center[k] = p['center'][k]
    o.sin_az, o.cos_az = p['azimuth_sc']
    roll = p['roll'] + p['positionRoll']
    seq = p.get('rotationSequence', 'RzRyRx')
    _fill_rot(o.to_local, rn.rotation_steps(seq, -p['pitch'], -roll, -p['yaw']))
    _fill_rot(o.to_virgin, rn.rotation_steps('-' + seq, p['pitch'], roll, p['yaw']))
    o.dx = float(p.get('dx', 0) or 0.)
    o.surf = 1 if surf['kind'] == 'toroid' else 0
    o.R, o.r = float(surf.get('R', 0.)), float(surf.get('r', 0.))
    for name, lim in (('phys_x', 'surfPhysX'), ('phys_y', 'surfPhysY')):
        getattr(o, name)[0], getattr(o, name)[1] = float(p[lim][0]), float(p[lim][1])
    for axis, lim in (('x', 'surfOptX'), ('y', 'surfOptY')):
        opt = p.get(lim)
        setattr(o, 'has_opt_' + axis, 0 if opt is None else 1)
        if opt is not None:
            getattr(o, 'opt_' + axis)[0], getattr(o, 'opt_' + axis)[1] = map(float, opt)
    edges = str(p.get('overEdge', 'yMax')).lower()
    o.over_mask = sum(bit for word, bit in (('xmin', 1), ('xmax', 2), ('ymin', 4),
                                            ('ymax', 8)) if word in edges)
    o.lost_num, o.roll = int(p['lostNum']), float(roll)
    o.nelem = len(m['elements'])
    for e, (elem, q) in enumerate(zip(m['elements'], m['quantities'])):
        tabs = [np.ascontiguousarray(elem[t], dtype=np.float64) for t in ('E', 'f1', 'f2')]
        keep += tabs
        o.Z[e], o.tab_n[e], o.quantity[e] = int(elem['Z']), len(tabs[0]), float(q)
        o.tab_E[e], o.tab_f1[e], o.tab_f2[e] = (t.ctypes.data for t in tabs)
    o.rho, o.mass = float(m['rho']), float(m['mass'])
    return o, keep


def _beam_record(arrays):
    rec = BeamRec()
    for name in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'state'):
        setattr(rec, name, arrays[name].ctypes.data)
    return rec


def _arrays_of(beam):
    out = {f: np.ascontiguousarray(getattr(beam, f), dtype=np.float64)
           for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp')}
    out['Jsp'] = np.ascontiguousarray(beam.Jsp, dtype=np.complex128)
    out['state'] = np.ascontiguousarray(beam.state, dtype=np.int32)
    return out


def oe_reflect(params, beam):
    """-> (gb, lb) oracle Beams like reflect_np.oe_reflect (lb.theta included)."""
    lib = load()
    o, keep = _record(params)
    n = len(beam.x)
    src = _arrays_of(beam)
    outs = []
    for _ in range(2):
        b = rn.Beam(n)
        arrays = _arrays_of(b)
        outs.append((b, arrays))
    theta = np.zeros(n)
    recs = [_beam_record(src)] + [_beam_record(a) for _, a in outs]
    rc = lib.xrt_oracle_reflect(ctypes.byref(o), ctypes.c_int64(n), ctypes.byref(recs[0]),
                                ctypes.byref(recs[1]), ctypes.byref(recs[2]),
                                ctypes.c_void_p(theta.ctypes.data))
    if rc == -2:
        raise NotImplementedError('this batch takes Brent\'s method (not restated in C)')
    if rc != 0:
        raise RuntimeError('xrt_oracle_reflect returned %d' % rc)
    for b, arrays in outs:
        for name, values in arrays.items():
            setattr(b, name, values)
    gb, lb = outs[0][0], outs[1][0]
    lb.theta = theta
    return gb, lb
