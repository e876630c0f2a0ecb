"""TEST INFRASTRUCTURE ONLY — numpy restatement of the amplitude functions on
xrt's ray-surface hot path. Materials are plain dictionaries (see
``make_material`` / ``make_crystal_si``); element tables come from
tests/golden/g6_element_tables.npz (extracted from the reference's data files by
oracle/gen_fixtures_p1.py).

Reference anchors, relative to xrt/backends/raycing/materials/:

* interp_f1f2        <- element.py:252-263 (np.interp on the tabulated f1, f2)
* f0                 <- element.py:203-207 (Waasmaier-Kirfel 5 Gaussians + c)
* refractive_index   <- material.py:348-378
* material_amplitude <- material.py:415-493 (Fresnel rs, rp / ts, tp)
* structure_factor   <- crystals_basic.py:22-31 (fcc), 76-80 (diamond)
* crystal_amplitude  <- crystal.py:492-645 (Belyakov-Dmitrienko), :297-306
                        (get_F_chi), :1105-1120 (Bragg angle)
* si_lattice_a       <- crystals_basic.py:99-142 (Swenson thermal expansion)
* make_multilayer / multilayer_amplitude <- multilayer.py:167-191 (depth grading),
                        :257-566 (Parratt recursion, Nevot-Croce factors; Coated :577-625)
"""
import numpy as np

from .consts import AVOGADRO, CH, CHBAR, PI, PI2, R0


# --------------------------------------------------------------------------
# elements
# --------------------------------------------------------------------------
def load_element(tables, name):
    """tables: the npz/dict of g6_element_tables; returns an element dict."""
    return dict(name=name, Z=int(tables[name + '_Z']),
                mass=float(tables[name + '_mass']),
                f0coeffs=np.array(tables[name + '_f0']),
                E=np.array(tables[name + '_E']),
                f1=np.array(tables[name + '_f1']),
                f2=np.array(tables[name + '_f2']))


def interp_f1f2(elem, E):
    if np.any(E < elem['E'][0]) or np.any(E > elem['E'][-1]):
        raise ValueError('E is out of the data table range')
    f1 = np.interp(E, elem['E'], elem['f1'])
    f2 = np.interp(E, elem['E'], elem['f2'])
    return f1 + 1j*f2


def f0(elem, qOver4pi=0):
    c = elem['f0coeffs']
    return c[5] + sum(a * np.exp(-b * qOver4pi**2)
                      for a, b in zip(c[:5], c[6:]))


# --------------------------------------------------------------------------
# amorphous materials (mirror / thin mirror / plate)
# --------------------------------------------------------------------------
def make_material(elements, quantities=None, kind='mirror', rho=0., t=None):
    if quantities is None:
        quantities = [1. for _ in elements]
    mass = 0.
    for elem, xi in zip(elements, quantities):
        mass += xi * elem['mass']
    return dict(kind=kind, elements=list(elements), quantities=list(quantities),
                rho=rho, t=t, mass=mass)


def refractive_index(m, E):
    given = m.get('refractiveIndex')
    if isinstance(given, (list, tuple)):             # material.py:364-371: [energies, spline]
        if np.min(E) > given[0][0] and np.max(E) < given[0][-1]:
            return given[1](E)
    elif given is not None:                          # :372-373: a constant
        return given
    xf = np.zeros_like(E) * 0j
    for elem, xi in zip(m['elements'], m['quantities']):
        xf += (elem['Z'] + interp_f1f2(elem, E)) * xi
    return 1 - 1e-24 * AVOGADRO * R0 / PI2 * (CH/E)**2 * m['rho'] * \
        xf / m['mass']


def material_amplitude(m, E, beamInDotNormal, fromVacuum=True):
    kind = m['kind']
    if kind in ('FZP'):                                # material.py:457-459
        return 1, 1, 0
    n = refractive_index(m, E)
    if fromVacuum:
        n1 = 1.
        n2 = n
    else:
        n1 = n
        n2 = 1.
    cosAlpha = abs(beamInDotNormal)
    sinAlpha2 = 1 - beamInDotNormal**2
    if isinstance(sinAlpha2, np.ndarray):
        sinAlpha2[sinAlpha2 < 0] = 0
    n1cosAlpha = n1 * cosAlpha
    cosBeta = np.sqrt(1 - (n1/n2)**2*sinAlpha2)
    n2cosBeta = n2 * cosBeta
    if kind in ('mirror', 'thin mirror', 'grating'):   # material.py:476
        rs = (n1cosAlpha - n2cosBeta) / (n1cosAlpha + n2cosBeta)
        rp = (n2*cosAlpha - n1*cosBeta) / (n2*cosAlpha + n1*cosBeta)
        if kind == 'thin mirror':
            p2 = np.exp(2j * E / CHBAR * n2cosBeta * m['t'] * 1e7)
            rs *= (1 - p2) / (1 - rs**2*p2)
            rp *= (1 - p2) / (1 - rp**2*p2)
    elif kind in ('plate', 'lens'):
        tf = np.sqrt(
            (n2cosBeta * np.conjugate(n1)).real / cosAlpha) / abs(n1)
        rs = 2 * n1cosAlpha / (n1cosAlpha + n2cosBeta) * tf
        rp = 2 * n1cosAlpha / (n2*cosAlpha + n1*cosBeta) * tf
    else:
        raise ValueError('Unknown kind of material')
    return (rs, rp, abs(n.imag) * E / CHBAR * 2e8, n.real * E / CHBAR * 1e8)


# --------------------------------------------------------------------------
# multilayers and coated mirrors
# --------------------------------------------------------------------------
def _graded(high, low, nPairs, power):                 # multilayer.py:167-191
    if low:
        layers = np.arange(1, nPairs+1)
        qRoot = (high/low)**(1./power)
        qB = (nPairs-qRoot) / (qRoot-1.)
        qA = high * (qB+1)**power
        return qA * (qB+layers)**(-power)
    return np.ones(nPairs) * float(high)


def make_multilayer(tLayer=None, tThickness=0., bLayer=None, bThickness=0., nPairs=0,
                    substrate=None, tThicknessLow=0., bThicknessLow=0., idThickness=0.,
                    power=2., substRoughness=0., substThickness=np.inf, geom='reflected',
                    kind='multilayer'):
    """Layers are material dicts (make_material) or None = vacuum. kind 'mirror' is the
    reference's Coated (one period, no top layer)."""
    return dict(kind=kind, layered=True, tLayer=tLayer, bLayer=bLayer, substrate=substrate,
                nPairs=int(nPairs), dti=_graded(float(tThickness), float(tThicknessLow),
                                                int(nPairs), power),
                dbi=_graded(float(bThickness), float(bThicknessLow), int(nPairs), power),
                idThickness=idThickness, substRoughness=float(substRoughness),
                substThickness=substThickness, geom=geom,
                d=float(tThickness + bThickness))


def make_coated(coating, cThickness, substrate, surfaceRoughness=0., substRoughness=0.):
    return make_multilayer(bLayer=coating, bThickness=cThickness, idThickness=surfaceRoughness,
                           nPairs=1, substrate=substrate, substRoughness=substRoughness,
                           kind='mirror')


def multilayer_amplitude(ml, E, beamInDotNormal):
    k = E / CHBAR
    nt = refractive_index(ml['tLayer'], E).conjugate() if ml['tLayer'] else 1.
    nb = refractive_index(ml['bLayer'], E).conjugate() if ml['bLayer'] else 1.
    ns = refractive_index(ml['substrate'], E).conjugate() if ml['substrate'] else 1.
    tran = 'tran' in ml['geom']
    Q = 2 * k * abs(beamInDotNormal)
    Q2 = Q**2
    k28 = 8 * k**2
    Qt = (Q2 + (nt-1)*k28)**0.5
    Qb = (Q2 + (nb-1)*k28)**0.5
    Qs = (Q2 + (ns-1)*k28)**0.5
    id2 = ml['idThickness']**2

    def interface(Qa, na, Qb_, nb_, rough):
        """(r_s, r_p, t_s, t_p) from medium a into medium b."""
        A, B = Qa/na*nb_, Qb_/nb_*na
        return (np.complex128((Qa-Qb_) / (Qa+Qb_) * rough),
                np.complex128((A-B) / (A+B) * rough),
                np.complex128(2*Qa / (Qa+Qb_) * rough),
                np.complex128(2*A / (A+B) * rough))
    vt = interface(Q, 1., Qt, nt, np.exp(-0.5 * Q * Qt * id2))
    roughtb = np.exp(-0.5 * Qt * Qb * id2)
    tb = interface(Qt, nt, Qb, nb, roughtb)
    bt = interface(Qb, nb, Qt, nt, roughtb)
    rmsbs = id2 if ml['tLayer'] else ml['substRoughness']**2
    roughbs = np.exp(-0.5 * Qb * Qs * rmsbs)
    bs = interface(Qb, nb, Qs, ns, roughbs)
    sv = interface(Qs, ns, Q, 1., roughbs)
    nPairs = ml['nPairs']
    if tran:
        rj_s, rj_p, tj_s, tj_p = sv
        extraLayer = 1
    else:
        rj_s, rj_p = bs[0], bs[1]
        tj_s = tj_p = 0.
        extraLayer = 0
    for i in reversed(range(2*nPairs+extraLayer)):
        if i % 2 == 0:
            if i == 0:
                f = vt
                iQT = Qt * ml['dti'][0]
            elif i == 2*nPairs:
                f = bs
                iQT = Qs * ml['substThickness']
            else:
                f = bt
                iQT = Qt * ml['dti'][i//2]
        else:
            f = tb
            iQT = Qb * ml['dbi'][i//2]
        p1i = np.complex128(np.exp(0.5j*iQT))
        p2i = p1i**2
        rj2i_s = rj_s * p2i
        rj2i_p = rj_p * p2i
        ri_s = (f[0] + rj2i_s) / (1 + f[0]*rj2i_s)
        ri_p = (f[1] + rj2i_p) / (1 + f[1]*rj2i_p)
        if tran:
            tj_s = f[2] * tj_s * p1i / (1 + f[0]*rj2i_s)
            tj_p = f[3] * tj_p * p1i / (1 + f[1]*rj2i_p)
        rj_s, rj_p = ri_s, ri_p
    if tran:
        return tj_s, tj_p
    nn = nt[0] if isinstance(nt, np.ndarray) else nt
    if (nn - 1) > 0:
        return rj_s.conjugate(), rj_p.conjugate()
    return rj_s, rj_p


# --------------------------------------------------------------------------
# crystals
# --------------------------------------------------------------------------
def _dl_l(t):
    if t >= 0.0 and t < 30.0:
        return -2.154537e-004
    elif t >= 30.0 and t < 130.0:
        return -2.303956e-014 * t**4 + 7.834799e-011 * t**3 - \
            1.724143e-008 * t**2 + 8.396104e-007 * t - 2.276144e-004
    elif t >= 130.0 and t < 293.0:
        return -1.223001e-011 * t**3 + 1.532991e-008 * t**2 - \
            3.263667e-006 * t - 5.217231e-005
    elif t >= 293.0 and t <= 1000.0:
        return -1.161022e-012 * t**3 + 3.311476e-009 * t**2 + \
            1.124129e-006 * t - 5.844535e-004
    else:
        return 1.0e+100


def si_lattice_a(tK=297.15):
    a0 = 5.430710
    return a0 * (_dl_l(tK) - _dl_l(273.15 + 19.9) + 1)


def make_crystal(elem, hkl, d, structure='diamond', geom='Bragg reflected',
                 t=None, factDW=1., V=None):
    if len(geom) < 6:
        geom = geom.strip() + ' reflected'
    sqrthkl2 = (sum(i**2 for i in hkl))**0.5
    if V is None:
        V = (d * sqrthkl2)**3
    chiToF = -R0 / PI / V
    return dict(kind='crystal', structure=structure, elements=[elem],
                hkl=tuple(hkl), d=d, V=V, chiToF=chiToF, geom=geom, t=t,
                factDW=factDW)


def make_crystal_si(elem_si, hkl=(1, 1, 1), tK=297.15, **kw):
    sqrthkl2 = (sum(i**2 for i in hkl))**0.5
    d = si_lattice_a(tK) / sqrthkl2
    return make_crystal(elem_si, hkl, d, 'diamond', **kw)


def make_crystal_from_cell(elems, atomsXYZ, hkl, a, b=None, c=None, alpha=90, beta=90,
                           gamma=90, atomsFraction=None, geom='Bragg reflected', t=None,
                           factDW=1.):
    """CrystalFromCell (crystals_basic.py:157-440): *elems* = one element dict per atom of
    the cell (the same dict object for atoms of one element)."""
    b, c = b or a, c or a
    fractions = [1 for _ in elems] if atomsFraction is None else atomsFraction
    ca, cb, cg = np.cos(np.radians((alpha, beta, gamma)))
    sa, sb, sg = np.sin(np.radians((alpha, beta, gamma)))
    V = a * b * c * (1 - ca**2 - cb**2 - cg**2 + 2*ca*cb*cg)**0.5
    h, k, l = hkl   # noqa: E741
    d = V / (a * b * c) *\
        ((h*sa/a)**2 + (k*sb/b)**2 + (l*sg/c)**2 +
         2*h*k * (ca*cb - cg) / (a*b) +
         2*h*l * (ca*cg - cb) / (a*c) +
         2*k*l * (cb*cg - ca) / (b*c))**(-0.5)
    if len(geom) < 6:
        geom = geom.strip() + ' reflected'
    return dict(kind='crystal', structure='cell', elements=list(elems),
                atomsXYZ=[list(r) for r in atomsXYZ], atomsFraction=list(fractions),
                hkl=tuple(hkl), d=d, V=V, chiToF=-R0 / PI / V, geom=geom, t=t,
                factDW=factDW)


def structure_factor(cr, E, sinThetaOverLambda=0):
    if cr['structure'] == 'cell':                  # crystals_basic.py:424-440
        F0, Fhkl, Fhkl_ = 0, 0, 0
        unique = {}
        for el, xyz, af in zip(cr['elements'], cr['atomsXYZ'], cr['atomsFraction']):
            if el['Z'] in unique:
                f0v, anomalousPart = unique[el['Z']]
            else:
                f0v = f0(el, sinThetaOverLambda)
                anomalousPart = interp_f1f2(el, E)
                unique[el['Z']] = f0v, anomalousPart
            F0 += af * (el['Z']+anomalousPart) * cr['factDW']
            fact = af * (f0v+anomalousPart) * cr['factDW']
            expiHr = np.exp(2j * np.pi * np.dot(xyz, cr['hkl']))
            Fhkl += fact * expiHr
            Fhkl_ += fact / expiHr
        return F0, Fhkl, Fhkl_
    elem = cr['elements'][0]
    anomalousPart = interp_f1f2(elem, E)
    F0 = 4 * (elem['Z']+anomalousPart) * cr['factDW']
    residue = sum(i % 2 for i in cr['hkl'])
    if residue == 0 or residue == 3:
        f0v = f0(elem, sinThetaOverLambda)
        Fhkl = 4 * (f0v+anomalousPart) * cr['factDW']
    else:
        Fhkl = 0.
    if cr['structure'] == 'fcc':
        return F0, Fhkl, Fhkl
    diamondToFcc = 1 + np.exp(0.5j * PI * sum(cr['hkl']))
    return F0 * 2, Fhkl * diamondToFcc, Fhkl * diamondToFcc.conjugate()


def F_chi(cr, E, sinThetaOverLambda):
    F0, Fhkl, Fhkl_ = structure_factor(cr, E, sinThetaOverLambda)
    waveLength = CH / E
    lambdaSquare = waveLength**2
    chiToFlambdaSquare = cr['chiToF'] * lambdaSquare
    chi0 = np.conjugate(F0) * chiToFlambdaSquare
    chih = np.conjugate(Fhkl) * chiToFlambdaSquare
    chih_ = np.conjugate(Fhkl_) * chiToFlambdaSquare
    return F0, Fhkl, Fhkl_, chi0, chih, chih_


def sin_bragg_angle(cr, E, order=1):
    a = order * CH / (2*cr['d']*E)
    try:
        a[a > 1] = 1 - 1e-16
        a[a < -1] = -1 + 1e-16
    except TypeError:
        if a > 1:
            a = 1 - 1e-16
        elif a < -1:
            a = -1 + 1e-16
    return a


def bragg_angle(cr, E, order=1):
    return np.arcsin(sin_bragg_angle(cr, E, order))


def crystal_amplitude(cr, E, beamInDotNormal, beamOutDotNormal=None,
                      beamInDotHNormal=None):
    geom = cr['geom']
    tmm = cr['t']

    def for_one_polarization(polFactor):
        delta = np.sqrt((alpha**2 + polFactor**2 * chih * chih_ / b))
        if tmm is None:                                   # thick Bragg
            with np.errstate(divide='ignore', invalid='ignore'):
                ra = chih * polFactor / (alpha+delta)
            ad = alpha - delta
            ad[ad == 0] = 1e-100
            rb = chih * polFactor / ad
            indB = np.where(np.isnan(ra))
            ra[indB] = rb[indB]
            indB = np.where(abs(rb) < abs(ra))
            ra[indB] = rb[indB]
            return ra / np.sqrt(abs(b))
        t = tmm * 1e7
        l = t * delta * k02 / 2. / kHs  # noqa: E741
        with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
            if geom.startswith('Bragg'):
                if geom.endswith('transmitted'):
                    ra = 1 / (np.cos(l) - 1j * alpha * np.sin(l) / delta) * \
                        np.exp(1j * k02 * t * (chi0 - alpha*b) / 2 / k0s)
                else:
                    ra = chih * polFactor / (alpha + 1j*delta / np.tan(l))
            else:
                if geom.endswith('transmitted'):
                    ra = (np.cos(l) + 1j * alpha * np.sin(l) / delta) *\
                        np.exp(1j * k02 * t * (chi0 - alpha*b) / 2 / k0s)
                else:
                    ra = chih * polFactor * np.sin(l) / delta *\
                        np.exp(1j * k02 * t * (chi0 - alpha*b) / 2 / k0s)
        if not geom.endswith('transmitted'):
            ra /= np.sqrt(abs(b))
        return ra

    waveLength = CH / E
    k = PI2 / waveLength
    k0s = -beamInDotNormal * k
    if beamOutDotNormal is None:
        beamOutDotNormal = -beamInDotNormal
    kHs = -beamOutDotNormal * k
    if beamInDotHNormal is None:
        beamInDotHNormal = beamInDotNormal
    crystd = cr['d']
    HH = PI2 / crystd
    k0H = abs(beamInDotHNormal) * HH * k
    k02 = k**2
    H2 = HH**2
    kHs0 = kHs == 0
    kHs[kHs0] = 1
    b = k0s / kHs
    b[kHs0] = -1
    F0, Fhkl, Fhkl_, chi0, chih, chih_ = F_chi(cr, E, 0.5/crystd)
    thetaB = bragg_angle(cr, E)
    alpha = (H2/2 - k0H) / k02 + chi0/2 * (1/b - 1)
    curveS = for_one_polarization(1.)
    polFactor = np.cos(2. * thetaB)
    curveP = for_one_polarization(polFactor)
    return curveS, curveP
