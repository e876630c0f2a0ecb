"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the DEVICE ray generator of
``GeometricSource(rng='device')`` (csrc/source.hip), never imported by the product.

What is sampled and how it becomes a beam follows the reference
(/root/reference/xrt/backends/raycing/sources/geoms.py: ``_apply_distribution`` :370-407,
``_set_annulus`` :409-418, ``shine`` :420-535, ``make_polarization`` :63-179, ``make_energy``
:16-60); WHERE the random numbers come from does not: the reference consumes numpy's global
Mersenne twister ray after ray, array after array, which no parallel generator can reproduce.
The device path uses the counter-based Philox4x32-10 of Salmon et al., "Parallel random
numbers: as easy as 1, 2, 3" (SC'11; Random123 v1.14 `philox.h`, not vendored by the
reference and absent here): its published constants and known-answer vectors are restated
below and checked by tests/test_geosource_oracle.py. SURVEY 8c pins row a2 by distribution
moments for exactly this reason; the parity of the kernel with THIS file is bit-for-bit on
the integers and the uniforms and within a few ulp on the transcendental laws.

Stream layout (one Philox block = four 32-bit words = two 53-bit uniforms Ua, Ub):
  counter = (ray index low, ray index high, slot, call number), key = 64-bit seed
  slot 0  random |Ep| of an unpolarised beam with amplitudes      Ua
  slot 1  y                                                        one number
  slot 2  x and z when both are plain normals, or an annulus (r from Ua, phi from Ub);
          otherwise x alone, and
  slot 3  z alone
  slot 4  x' and z' (same rule), slot 5 z' alone
  slot 6  energy (a filament beam takes ray 0's)
  one number = Ua for 'flat' (and the uniform ray density form of 'normal'),
               sqrt(-2 ln(1 - Ua)) cos(2 pi Ub) for 'normal' (Box-Muller; the pair form
               gives the sine to the second coordinate)
"""
import numpy as np

PI2 = 6.283185307179586476925286766559          # physconsts.py:34

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
SLOT_PHASE, SLOT_Y, SLOT_XZ, SLOT_Z, SLOT_AC, SLOT_C, SLOT_E = range(7)
LAW_NONE, LAW_NORMAL, LAW_FLAT, LAW_NORMAL_UNIFORM = 0, 1, 2, 3


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten rounds of Philox-4x32 on arrays of uint32 counters -> four uint32 arrays."""
    c = [np.asarray(v, dtype=np.uint64) & MASK for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & MASK,
             (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & MASK]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def uniforms(index, slot, call, seed):
    """(Ua, Ub) in [0, 1): 53 bits each, from the high 27 and 26 bits of two words."""
    index = np.asarray(index, dtype=np.uint64)
    r = philox4x32_10(index & MASK, index >> np.uint64(32), slot, call,
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    r = [v.astype(np.float64) for v in
         (r[0] >> np.uint32(5), r[1] >> np.uint32(6), r[2] >> np.uint32(5), r[3] >> np.uint32(6))]
    return (r[0] * 67108864. + r[1]) * 2.**-53, (r[2] * 67108864. + r[3]) * 2.**-53


def box_muller(ua, ub):
    radius = np.sqrt(-2. * np.log(1. - ua))
    angle = PI2 * ub
    return radius * np.cos(angle), radius * np.sin(angle)


def _law(dist, size, uniform_density):
    """(law, p0, p1) of one coordinate as the kernel takes it (geoms.py:370-407)."""
    if dist == 'normal' and uniform_density:
        w = np.atleast_1d(size)
        return LAW_NORMAL_UNIFORM, float(w[0]), float(w[-1] if len(w) > 1 else 5 * abs(w[0]))
    if dist == 'normal':
        sigma = float(size[0] if isinstance(size, (list, tuple)) else size)
        return (LAW_NORMAL, sigma, 0.) if sigma >= 0 else (LAW_NONE, 0., 0.)   # ValueError -> zeros
    if dist == 'flat':
        if isinstance(size, (list, tuple, np.ndarray)):
            return LAW_FLAT, float(size[0]), float(size[1])
        if size > 0:
            return LAW_FLAT, -size * 0.5, size * 0.5
    return LAW_NONE, 0., 0.


class Spec(object):
    """The parameters of one source as plain numbers (what xrt_hip_geosource carries)."""

    def __init__(self, nrays, seed=0, call=0, distx='normal', dx=0.32, disty=None, dy=0,
                 distz='normal', dz=0.018, distxprime='normal', dxprime=1e-3,
                 distzprime='normal', dzprime=1e-4, distE='lines', energies=(9000.,),
                 energyWeights=None, polarization=(1., 0., 0j, 1., 0.), filamentBeam=False,
                 uniformRayDensity=False, withAmplitudes=False, steps=(), azimuth=None,
                 center=None):
        self.__dict__.update(locals())
        del self.__dict__['self']


def shine(s):
    """dict of the 13 (15) beam arrays the kernel writes for Spec *s*."""
    n = int(s.nrays)
    i = np.arange(n, dtype=np.uint64)
    amp = s.withAmplitudes or s.uniformRayDensity
    jss, jpp, jsp, es, ep = s.polarization
    o = dict(Jss=np.full(n, float(jss)), Jpp=np.full(n, float(jpp)),
             Jsp=np.full(n, complex(jsp)), state=np.ones(n, np.int32),
             path=np.zeros(n), E=np.full(n, 9000.))
    if amp:
        o['Es'] = np.full(n, complex(0. if es is None else es))
        if isinstance(ep, str):
            o['Ep'] = (uniforms(i, SLOT_PHASE, s.call, s.seed)[0] * 2**(-0.5)).astype(complex)
        else:
            o['Ep'] = np.full(n, complex(0. if ep is None else ep))

    def weigh(axis, sigma, cut):
        w = np.exp(-axis**2 / sigma**2 / 2) / PI2**0.5 / sigma * 2 * cut
        for f in ('Jss', 'Jpp', 'Jsp'):
            o[f] = o[f] * w
        for f in ('Es', 'Ep'):
            o[f] = o[f] * w**0.5

    def one(law, slot, second=False):
        kind, p0, p1 = law
        ua, ub = uniforms(i, slot, s.call, s.seed)
        if kind == LAW_NORMAL:
            return box_muller(ua, ub)[0] * p0
        if kind == LAW_FLAT:
            return p0 + (p1 - p0) * ua
        if kind == LAW_NORMAL_UNIFORM:
            v = -p1 + (p1 - (-p1)) * ua
            weigh(v, p0, p1)
            return v
        return np.zeros(n)

    def pair(first, second, slot):
        d1, s1, d2, s2 = first + second
        if 'annulus' in (d1, d2) and isinstance(s1, (list, tuple, np.ndarray)):
            rmin, rmax = s1
            pmin, pmax = s2 if isinstance(s2, (list, tuple, np.ndarray)) else (0, PI2)
            ua, ub = uniforms(i, slot, s.call, s.seed)
            r = np.sqrt(2 * ua / (2. / (rmax**2 - rmin**2)) + rmin**2) if rmax > rmin \
                else np.full(n, float(rmax))
            phi = pmin + (pmax - pmin) * ub
            return r * np.cos(phi), r * np.sin(phi)
        l1, l2 = _law(d1, s1, s.uniformRayDensity), _law(d2, s2, s.uniformRayDensity)
        if l1[0] == LAW_NORMAL and l2[0] == LAW_NORMAL:
            g1, g2 = box_muller(*uniforms(i, slot, s.call, s.seed))
            return g1 * l1[1], g2 * l2[1]
        return one(l1, slot), one(l2, slot + 1)

    o['y'] = one(_law(s.disty, s.dy, s.uniformRayDensity), SLOT_Y)
    o['x'], o['z'] = pair((s.distx, s.dx), (s.distz, s.dz), SLOT_XZ)
    o['a'], o['c'] = pair((s.distxprime, s.dxprime), (s.distzprime, s.dzprime), SLOT_AC)
    ac = o['a']**2 + o['c']**2
    if (ac > 1).any():
        b = (ac + 1)**0.5
        o['a'], o['c'], o['b'] = o['a'] / b, o['c'] / b, 1.0 / b
    else:
        o['b'] = (1 - ac)**0.5
    if s.distE is not None:
        ie = np.zeros(n, np.uint64) if s.filamentBeam else i
        ua, ub = uniforms(ie, SLOT_E, s.call, s.seed)
        v = np.atleast_1d(np.asarray(s.energies, dtype=float))
        if s.distE == 'normal':
            spread = abs(v[1]) if len(v) == 2 else 0.
            spread = 0. if spread > 0.1 * abs(v[0]) else spread
            o['E'] = v[0] + spread * box_muller(ua, ub)[0]
        elif s.distE == 'flat':
            top = (v[1] or v[0]) if len(v) == 2 else v[0]
            o['E'] = v[0] + (top - v[0]) * ua
        else:
            if 0 in v:
                v = v[v > 0]
            w = None if s.energyWeights is None else np.atleast_1d(s.energyWeights).astype(float)
            if w is None or len(w) != len(v):
                w = np.ones(len(v))
            cdf = np.cumsum(w) / np.sum(w)
            cdf[-1] = 1.
            o['E'] = v[np.minimum(np.searchsorted(cdf, ua, side='right'), len(v) - 1)]
    for name in ('x', 'y', 'z', 'a', 'b', 'c'):
        o[name] = np.array(o[name], dtype=float)
    plane = ((1, 2, 1.), (0, 2, -1.), (0, 1, 1.))       # _rotate.py:5-20
    for names in (('x', 'y', 'z'), ('a', 'b', 'c')):
        for axis, cs, sn in s.steps:
            p, q, sense = plane[axis]
            u, v = o[names[p]], o[names[q]]
            o[names[p]], o[names[q]] = u * cs - v * (sense * sn), u * (sense * sn) + v * cs
    if s.azimuth is not None:                            # beamline.py:266-287
        cs, sn = s.azimuth
        if sn != 0:
            for p, q in (('a', 'b'), ('x', 'y')):
                u, v = o[p], o[q]
                o[p], o[q] = u * cs - v * (-sn), u * (-sn) + v * cs
    if s.center is not None:
        for name, c0 in zip('xyz', s.center):
            o[name] = o[name] + c0
    return o
