"""TEST INFRASTRUCTURE ONLY — converts xrt_amd (product) objects into the
oracle's parameter dictionaries / Beam records, so that tests and bench.py's
cpu_baseline leg can run the numpy restatement on the very same configuration."""
import numpy as np

from . import fixture_io, materials_np as mn, reflect_np as rn


def oracle_params(oe):
    """xrt_amd OE -> the oracle's parameter dictionary."""
    p = dict(
        center=[float(c) for c in oe.center],
        azimuth_sc=(oe.bl.sinAzimuth, oe.bl.cosAzimuth), pitch=oe.pitch,
        roll=oe.roll, yaw=oe.yaw, positionRoll=oe.positionRoll,
        rotationSequence=oe.rotationSequence, extraPitch=oe.extraPitch,
        extraRoll=oe.extraRoll, extraYaw=oe.extraYaw,
        extraRotationSequence=oe.extraRotationSequence, dx=oe.dx, shape=oe.shape,
        overEdge=oe.overEdge, lostNum=oe.lostNum, surfPhysX=list(oe.limPhysX),
        surfPhysY=list(oe.limPhysY), surfOptX=oe.limOptX, surfOptY=oe.limOptY)
    if hasattr(oe, 'invertNormal'):
        p['invertNormal'] = int(oe.invertNormal)
    tb = fixture_io.tables()
    if hasattr(oe, 'parabParam'):
        p['surface'] = dict(
            kind='parabola_param', isCylindrical=bool(oe.isCylindrical),
            isClosed=bool(oe.isClosed), cosGamma=oe.cosGamma, sinGamma=oe.sinGamma,
            y0=oe.y0, z0=oe.z0, parabParam=oe.parabParam)
    elif hasattr(oe, 'hyperbolaA'):
        p['surface'] = dict(
            kind='hyperbola_param', isCylindrical=bool(oe.isCylindrical),
            isClosed=bool(oe.isClosed), cosGamma=oe.cosGamma, sinGamma=oe.sinGamma,
            y0=oe.y0, z0=oe.z0, hyperbolaA=oe.hyperbolaA, hyperbolaB=oe.hyperbolaB)
    elif hasattr(oe, 'ellipseA'):
        p['surface'] = dict(
            kind='ellipse_param', isCylindrical=bool(oe.isCylindrical),
            isClosed=bool(oe.isClosed), p=oe.p, q=oe.q, cosGamma=oe.cosGamma,
            sinGamma=oe.sinGamma, y0=oe.y0, z0=oe.z0, ellipseA=oe.ellipseA,
            ellipseB=oe.ellipseB)
    elif hasattr(oe, 'tanBlaze'):
        p['surface'] = rn.make_blazed(oe.blaze, oe.rho0, oe.antiblaze)
    elif hasattr(oe, 'R') and hasattr(oe, 'r'):
        p['surface'] = dict(kind='toroid', R=oe.R, r=oe.r)
    elif hasattr(oe, 'R'):
        p['surface'] = dict(kind='bentflat', R=oe.R, y0=oe.limPhysY[0])
    else:
        p['surface'] = dict(kind='flat', alpha=oe.alpha)

    def mat(m):
        if m is None:
            return None
        if m.kind == 'crystal':
            return mn.make_crystal(mn.load_element(tb, m.elements[0].name), m.hkl,
                                   m.d, 'diamond', m.geom, m.t, m.factDW, m.V)
        return mn.make_material([mn.load_element(tb, e.name) for e in m.elements],
                                list(m.quantities),
                                ('grating' if 'order' in p else 'mirror')
                                if m.kind == 'auto' else m.kind, m.rho, m.t)
    if hasattr(oe, '_is_grating') and oe._is_grating():
        p['order'] = int(oe.order)
        if oe.gratingDensity is not None and type(oe).local_g is type(oe).__mro__[-2].local_g:
            p['gratingDensity'] = list(oe.gratingDensity)
        else:
            g = oe.local_g(np.zeros(1), np.zeros(1))
            p['gVector'] = tuple(float(np.ravel(v)[0]) for v in g)
    material = oe.material
    if isinstance(material, (list, tuple)):
        material = material[0]
    p['material'] = mat(material)
    if hasattr(oe, 'cryst2pitch'):
        p.update(bragg=oe.bragg, cryst1roll=oe.cryst1roll, cryst2roll=oe.cryst2roll,
                 cryst2pitch=oe.cryst2pitch, cryst2finePitch=oe.cryst2finePitch,
                 cryst2perpTransl=oe.cryst2perpTransl,
                 cryst2longTransl=oe.cryst2longTransl,
                 surfPhysX2=list(oe.limPhysX2), surfPhysY2=list(oe.limPhysY2),
                 surfOptX2=oe.limOptX2, surfOptY2=oe.limOptY2,
                 surface2=dict(kind='flat', alpha=oe.alpha, flip_n_y=True),
                 is_plate=hasattr(oe, 't'),
                 material2=mat(oe.material2))
    return p


def to_oracle_beam(b):
    o = rn.Beam(len(b), with_amplitudes=b.has_amplitudes())
    for f in o.fields():
        setattr(o, f, np.array(b.peek(f)))
    return o
