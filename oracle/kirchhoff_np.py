"""TEST INFRASTRUCTURE ONLY — numpy restatement of xrt's Fresnel-Kirchhoff
diffraction integral (the P2 oracle).

Follows, expression by expression (same operation order, so that ``r`` and
``k*r`` round identically to the reference):

* ``kirchhoff_conv``      <- xrt/backends/raycing/waves.py:834-851
                             (_diffraction_integral_conv, the numpy path);
* ``to_cl_convention``    <- xrt/backends/raycing/cl/diffract.cl:119-148
                             (what the OpenCL kernel returns instead);
* ``diffract_post``       <- xrt/backends/raycing/waves.py:707-749
                             (accumulate, phase strip, normalise, flux norm).

Parity pinned: tests/golden/g4_*.npz hold inputs and outputs produced by the
imported reference (oracle/gen_fixtures_p2.py); tests/test_oracle_p2_golden.py checks
this module against them.
"""
import numpy as np
from .consts import CHBAR


def kirchhoff_conv(px, py, pz, sx, sy, sz, n, nl, E, Es, Ep,
                   max_pairs=4_000_000):
    """Raw integrals (Es, Ep, aE, bE, cE) per pixel, numpy sign convention.

    px,py,pz: pixel coordinates [Np] in the diffracting element's local frame.
    sx,sy,sz,nl,E,Es,Ep: good samples [Ns]; n: 3 scalars or 3 arrays [Ns].
    Pixels are processed in row chunks (rows are independent: ``sum(axis=1)``
    reduces each row on its own) to bound the O(Np*Ns) temporaries.
    """
    px = np.asarray(px, dtype=np.float64)
    npix = len(px)
    ns = len(sx)
    rows = max(1, int(max_pairs // max(ns, 1)))
    outs = [np.zeros(npix, dtype=np.complex128) for _ in range(5)]
    k = E / CHBAR * 1e7                                    # waves.py:841
    for i0 in range(0, npix, rows):
        sl = slice(i0, min(npix, i0 + rows))
        a = px[sl, np.newaxis] - sx                        # waves.py:836
        b = py[sl, np.newaxis] - sy                        # :837
        c = pz[sl, np.newaxis] - sz                        # :838
        pathAfter = (a**2 + b**2 + c**2)**0.5              # :839
        cosn = (a*n[0] + b*n[1] + c*n[2]) / pathAfter      # :840
        U = k*1j/(4*np.pi) * (nl+cosn) * np.exp(1j*k*(pathAfter)) / pathAfter
        outs[0][sl] = (Es * U).sum(axis=1)                 # :845
        outs[1][sl] = (Ep * U).sum(axis=1)                 # :846
        abcU = k**2/(4*np.pi) * (Es+Ep) * U / pathAfter    # :847
        outs[2][sl] = (abcU * a).sum(axis=1)               # :848
        outs[3][sl] = (abcU * b).sum(axis=1)
        outs[4][sl] = (abcU * c).sum(axis=1)
    return tuple(outs)


def to_cl_convention(Es, Ep, aE, bE, cE):
    """Map numpy-convention raw integrals to what the OpenCL kernel
    integrate_kirchhoff returns (diffract.cl:136-148): fields negated
    (-i/4pi vs +i k/4pi), direction integrals scaled by (1+i)*4pi/i."""
    f = (1 + 1j) * 4*np.pi / 1j
    return -Es, -Ep, aE*f, bE*f, cE*f


def diffract_post(acc, raw, is_oe, dS, area, sumJ, sumJnl, nrays, repeats):
    """waves.py:707-749. ``acc``: dict of the 5 accumulators (updated in
    place); ``raw``: the 5 integrals of this repeat. Returns dict with
    Es, Ep, Jss, Jpp, Jsp, a, b, c (normalised)."""
    for key, val in zip(('EsAcc', 'EpAcc', 'aEacc', 'bEacc', 'cEacc'), raw):
        acc[key] = acc[key] + val
    Es = acc['EsAcc'].copy()
    Ep = acc['EpAcc'].copy()
    Jss = (Es * np.conj(Es)).real
    Jpp = (Ep * np.conj(Ep)).real
    Jsp = Es * np.conj(Ep)
    if is_oe:                                              # waves.py:719-722
        comp = acc['cEacc'] if abs(acc['cEacc'][0]) > abs(acc['bEacc'][0]) \
            else acc['bEacc']
    else:
        comp = acc['bEacc']
    toReal = np.exp(-1j * np.angle(comp))
    a = (acc['aEacc'] * toReal).real
    b = (acc['bEacc'] * toReal).real
    c = (acc['cEacc'] * toReal).real
    norm = (a**2 + b**2 + c**2)**0.5
    norm[norm == 0] = 1.
    a /= norm
    b /= norm
    c /= norm
    fnorm = dS * area * sumJ                               # waves.py:739-744
    de = nrays * sumJnl * repeats
    fnorm = fnorm / de if de > 0 else 0
    return dict(Es=Es * fnorm**0.5, Ep=Ep * fnorm**0.5, Jss=Jss * fnorm,
                Jpp=Jpp * fnorm, Jsp=Jsp * fnorm, a=a, b=b, c=c)
