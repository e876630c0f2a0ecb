"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement (numpy, fp64) of the reference's undulator field integral — the
per-ray sum over the quadrature nodes of one undulator period (far field) or of
all Np periods (tapered / near field):

    Undulator._sp_sum            xrt/backends/raycing/sources/synchr.py:1930-2038
    Undulator._build_I_map_conv  synchr.py:2050-2108 (pre-factors wu, ww1, ab and
                                 the Amp2Flux scaling around the sum)
    Undulator._build_integration_grid  synchr.py:1789-1801 (node tables)

The reference's OpenCL kernels `undulator`, `undulator_taper`, `undulator_nf`
(cl/undulator.cl:54-300) evaluate the same integrals; they differ from the numpy
path in three documented places: the far-field kernel divides
betaP.z by beta.z (relative 2e-8), the near-field kernel uses r.y = -Kx sin/γ
and sin(w/wu * R0z) where `_sp_sum` has +Kx sin/γ and sin(R0z). This file (and
the HIP kernel it checks) follows the numpy path `_sp_sum`, operation by
operation, because that is the path the reference can run here.

Pinned against fixtures made by calling the reference's own `_sp_sum` and
`_build_I_map_conv` (oracle/gen_fixtures_undulator.py → tests/golden/g9_*.npz).
"""
import numpy as np

from .consts import PI, PI2

E2WC = 5067.7309392068091       # synchr.py / sybase.py module constant [1/(eV mm)]
FINE_STR = 1 / 137.03599976
SIE0 = 1.602176565e-19

MODE_FAR, MODE_TAPER, MODE_NF = 0, 1, 2


def clenshaw_curtis(n):
    """n-point Clenshaw–Curtis rule on [-1, 1] (the reference's default
    quadrature, sybase.py:1106-1140 via FFT; here the classical closed form
    w_k = c_k/N · [1 − Σ_j b_j/(4j²−1) · cos(2jkπ/N)], N = n−1), equal to the
    reference's weights to rounding."""
    N = n - 1
    k = np.arange(n)
    x = -np.cos(np.pi * k / N)
    j = np.arange(1, N // 2 + 1)
    b = np.where(2 * j == N, 1., 2.)
    c = np.where((k == 0) | (k == N), 1., 2.)
    s = (b / (4. * j * j - 1.) *
         np.cos(2. * np.pi * np.outer(k, j) / N)).sum(axis=1)
    return x, c / N * (1. - s)


def node_tables(quadm, g_intervals, phase, use_gauleg=False):
    """Node/weight tables of synchr.py:1789-1801: `g_intervals` equal panels of
    one undulator period [-π, π], `quadm` nodes in each."""
    if use_gauleg:
        tg_n, ag_n = np.polynomial.legendre.leggauss(quadm)
    else:
        tg_n, ag_n = clenshaw_curtis(quadm)
    dstep = 2 * PI / float(g_intervals)
    dI = np.arange(-PI + 0.5 * dstep, PI, dstep)
    tg = (dI[:, None] + 0.5 * dstep * tg_n).ravel()
    ag = (dI[:, None] * 0 + ag_n).ravel()
    return dict(tg=tg, ag=ag, sintg=np.sin(tg), costg=np.cos(tg),
                sintgph=np.sin(tg + phase), costgph=np.cos(tg + phase),
                dstep=dstep)


def sp_sum(mode, Kx, Ky, Np, tab, ww1, w, wu, gamma, ddphi, ddpsi,
           taper_val=None, r0z=None):
    """Is, Ip = wu/γ · Σ_nodes ag · e^{iφ} · [n × ((n−β) × β')]_{x,y} / (1−n·β)²

    synchr.py:1930-2038. `tab` = node_tables(); per-ray arrays ww1, w, wu,
    gamma, ddphi, ddpsi. mode TAPER needs taper_val (= Undulator._taperVal),
    mode NF needs r0z (= R0·2π/L0, synchr.py:2077)."""
    tg, ag = tab['tg'], tab['ag']
    sintg, costg = tab['sintg'], tab['costg']
    sintgph, costgph = tab['sintgph'], tab['costgph']
    taperC = 1
    alphaS = 0
    sin2x = 2. * sintg * costg
    sin2xph = 2. * sintgph * costgph
    revg = 1. / gamma
    revg2 = revg**2
    betam = 1. - (1. + 0.5 * Kx**2 + 0.5 * Ky**2) * 0.5 * revg2
    wwu = w / wu
    Bs = np.zeros(len(w), dtype=np.complex128)
    Bp = np.zeros(len(w), dtype=np.complex128)
    dirx = ddphi
    diry = ddpsi
    dirz = 1. - 0.5 * (ddphi**2 + ddpsi**2)
    nper = Np if mode != MODE_FAR else 1
    if mode == MODE_NF:
        R0 = np.array((np.tan(ddphi), np.tan(ddpsi), np.ones_like(ddpsi)))
        R0 *= r0z
        sinr0z = np.sin(R0[-1])     # sic: no w/wu here (synchr.py:1950)
        cosr0z = np.cos(R0[-1])
    for ip in range(nper):
        for i in range(len(tg)):
            if mode == MODE_TAPER:
                zloc = -(nper - 1) * np.pi + ip * PI2 + tg[i]
                alphaS = taper_val / E2WC
                taperC = 1. - alphaS * zloc / wu
                ucos = ww1 * zloc + wwu * revg * (
                    -Ky * dirx * (sintg[i] + alphaS / wu *
                                  (1 - costg[i] - zloc * sintg[i])) +
                    Kx * diry * sintg[i] + 0.125 * revg *
                    (Kx**2 * sin2xph[i] + Ky**2 * (
                        sin2x[i] - 2 * alphaS / wu *
                        (zloc**2 + costg[i]**2 + zloc * sin2x[i]))))
                eucos = np.cos(ucos) + 1j * np.sin(ucos)
            elif mode == MODE_NF:
                zterm = 0.5 * (Ky**2 * sin2x[i] + Kx**2 * sin2xph[i]) * revg
                zloc = -(nper - 1) * np.pi + ip * PI2 + tg[i]
                rx = Ky * sintg[i] * revg
                ry = Kx * sintgph[i] * revg
                rz = betam * zloc - 0.25 * zterm * revg
                drx, dry, drz = R0[0] - rx, R0[1] - ry, R0[2] - rz
                dist = np.sqrt(drx * drx + dry * dry + drz * drz)
                drs = 0.5 * (drx**2 + dry**2) / drz
                a1 = wwu * zloc * (1. - betam)
                a2 = wwu * (drs + 0.25 * zterm * revg)
                sinzloc, coszloc = np.sin(a1), np.cos(a1)
                sindrs, cosdrs = np.sin(a2), np.cos(a2)
                ex = (-sinr0z * sinzloc * cosdrs - sinr0z * coszloc * sindrs -
                      cosr0z * sinzloc * sindrs + cosr0z * coszloc * cosdrs)
                ey = (-sinr0z * sinzloc * sindrs + sinr0z * coszloc * cosdrs +
                      cosr0z * sinzloc * cosdrs + cosr0z * coszloc * sindrs)
                eucos = ex + 1j * ey
                dirx, diry, dirz = drx / dist, dry / dist, drz / dist
            else:
                ucos = ww1 * tg[i] + wwu * revg * (
                    -Ky * ddphi * sintg[i] + Kx * ddpsi * sintgph[i] +
                    0.125 * revg * (Ky**2 * sin2x[i] + Kx**2 * sin2xph[i]))
                eucos = np.cos(ucos) + 1j * np.sin(ucos)
            betax = taperC * Ky * revg * costg[i]
            betay = -Kx * revg * costgph[i]
            betaz = 1. - 0.5 * (revg2 + betax * betax + betay * betay)
            betaPx = -Ky * (alphaS * costg[i] + taperC * sintg[i])
            betaPy = Kx * sintgph[i]
            betaPz = 0.5 * revg * (
                Ky**2 * taperC * (alphaS * costg[i]**2 + taperC * sin2x[i]) +
                Kx**2 * sin2xph[i])
            rkrel = 1. / (1. - dirx * betax - diry * betay - dirz * betaz)
            eucos = eucos * (ag[i] * rkrel**2)
            bnx, bny, bnz = dirx - betax, diry - betay, dirz - betaz
            nbp = dirx * betaPx + diry * betaPy + dirz * betaPz
            nbn = dirx * bnx + diry * bny + dirz * bnz
            Bs += eucos * (bnx * nbp - betaPx * nbn)
            Bp += eucos * (bny * nbp - betaPy * nbn)
    return wu * revg * Bs, wu * revg * Bp


def prefactors(Kx, Ky, Np, L0, gamma, w, ddtheta, ddpsi, single_period):
    """wu, ww1, ab of synchr.py:2060-2068. `single_period` = far-field case
    (the Np periods enter through the analytic sin(πNp·ww1)/sin(π·ww1))."""
    gamma = gamma * np.ones(len(w))
    gamma2 = gamma**2
    wu = PI / L0 / gamma2 * np.ones_like(w) * \
        (2 * gamma2 - 1 - 0.5 * Kx**2 - 0.5 * Ky**2) / E2WC
    ww1 = w * ((1. + 0.5 * Kx**2 + 0.5 * Ky**2) +
               gamma2 * (ddtheta**2 + ddpsi**2)) / (2. * gamma2 * wu)
    if single_period:
        ab = 1. / PI2 / wu * np.sin(PI * Np * ww1) / np.sin(PI * ww1)
    else:
        ab = 1. / PI2 / wu
    return gamma, wu, ww1, ab


def intensity_map(mode, Kx, Ky, Np, L0, gamma0, eI, dist_e_bw, tab, w, ddtheta,
                  ddpsi, taper_val=None, R0=None, harmonic=None):
    """(I, Es, Ep) of Undulator._build_I_map_conv (synchr.py:2050-2108) for
    eEspread = 0."""
    gamma, wu, ww1, ab = prefactors(Kx, Ky, Np, L0, gamma0, w, ddtheta, ddpsi,
                                    mode == MODE_FAR)
    r0z = None if R0 is None else R0 * np.pi * 2 / L0
    Is, Ip = sp_sum(mode, Kx, Ky, Np, tab, ww1, w, wu, gamma, ddtheta, ddpsi,
                    taper_val, r0z)
    bw = 0.001 if dist_e_bw else 1. / w
    amp2flux = FINE_STR * bw * eI / SIE0
    if harmonic is not None:
        for a in (Is, Ip):
            a[ww1 > harmonic + 0.5] = 0
            a[ww1 < harmonic - 0.5] = 0
    dstep = tab['dstep']
    field = np.abs(Is)**2 + np.abs(Ip)**2
    return (amp2flux * ab**2 * 0.25 * dstep**2 * field,
            np.sqrt(amp2flux) * ab * Is * 0.5 * dstep,
            np.sqrt(amp2flux) * ab * Ip * 0.5 * dstep)


# --------------------------------------------------------------------------
# custom (tabulated) magnetic field: SourceFromField._sp_sum, synchr.py:888-973
# (OpenCL twins `custom_field`, `custom_field_filament`, cl/undulator.cl:822-1103)
# --------------------------------------------------------------------------
EMC = 0.5866791802416487        # physconsts.py:24


def custom_sp_sum(filament, tab, emcg, w, gamma, ddphi, ddpsi, betam, R0=None):
    """Is, Ip of SourceFromField._sp_sum. `tab`: dict of the node tables tg, ag,
    Bx, By, Bz, betax, betay, trajx, trajy, trajz on the integration grid;
    per-ray arrays emcg, w, gamma, ddphi, ddpsi; betam = betazav[-1]; R0: None
    or the scalar screen distance [mm] (near field).

    Reproduces the reference as written, including its choice of the carrier
    wc, whose two branches are swapped relative to the vectorised `_sp`
    (synchr.py:901-902 vs :813-816)."""
    tg, ag = tab['tg'], tab['ag']
    Bx, By, Bz = tab['Bx'], tab['By'], tab['Bz']
    Bs = np.zeros(len(w), dtype=np.complex128)
    Bp = np.zeros(len(w), dtype=np.complex128)
    gamma_ = gamma[0] if filament else gamma
    dirx = ddphi
    diry = ddpsi
    dirz = np.sqrt(1. - ddphi**2 - ddpsi**2)
    revgamma2 = 1. / gamma_**2
    wc = w * E2WC / (1. + (betam*EMC**2 - 0.5)*revgamma2) if filament else \
        w * E2WC / betam
    if R0 is not None:
        R0v = np.array((np.tan(ddphi), np.tan(ddpsi), np.ones_like(ddpsi)))
        R0v *= R0
        sinr0z, cosr0z = np.sin(wc*R0v[2, :]), np.cos(wc*R0v[2, :])
    for i in range(len(tg)):
        if filament:
            betax_, betay_ = tab['betax'][i], tab['betay'][i]
            trajx_, trajy_, trajz_ = tab['trajx'][i], tab['trajy'][i], tab['trajz'][i]
        else:
            betax_ = emcg*tab['betax'][i]
            betay_ = emcg*tab['betay'][i]
            trajx_ = emcg*tab['trajx'][i]
            trajy_ = emcg*tab['trajy'][i]
            trajz_ = tg[i]*(1.-0.5*revgamma2) + EMC**2*revgamma2*tab['trajz'][i]
        if R0 is not None:
            drx, dry, drz = R0v[0] - trajx_, R0v[1] - trajy_, R0v[2] - trajz_
            dist = np.sqrt(drx*drx + dry*dry + drz*drz)
            rdrz = 1./drz
            drs = (drx**2+dry**2)*rdrz
            LRS = 0.5*drs - 0.125*drs**2*rdrz + 0.0625*drs**3*rdrz**2
            a1 = wc * (tg[i] - trajz_)
            a2 = wc * LRS
            sinzloc, coszloc = np.sin(a1), np.cos(a1)
            sindrs, cosdrs = np.sin(a2), np.cos(a2)
            ex = (-sinr0z*sinzloc*cosdrs - sinr0z*coszloc*sindrs -
                  cosr0z*sinzloc*sindrs + cosr0z*coszloc*cosdrs)
            ey = (-sinr0z*sinzloc*sindrs + sinr0z*coszloc*cosdrs +
                  cosr0z*sinzloc*cosdrs + cosr0z*coszloc*sindrs)
            dirx, diry, dirz = drx/dist, dry/dist, drz/dist
        else:
            phz = wc*(tg[i] - dirz*trajz_)
            phxy = wc*(dirx*trajx_ + diry*trajy_)
            sinphz, cosphz = np.sin(phz), np.cos(phz)
            sinphxy, cosphxy = np.sin(phxy), np.cos(phxy)
            ex = sinphz*cosphxy - cosphz*sinphxy
            ey = cosphz*cosphxy + sinphz*sinphxy
        eucos = ex + 1j*ey
        smTerm = revgamma2 + betax_**2 + betay_**2
        betaz = 1. - 0.5*smTerm - 0.125*smTerm**2 - 0.0625*smTerm**3
        betaPx = betay_*Bz[i] - betaz*By[i]
        betaPy = -betax_*Bz[i] + betaz*Bx[i]
        betaPz = betax_*By[i] - betay_*Bx[i]
        rkrel = 1./(1. - dirx*betax_ - diry*betay_ - dirz*betaz)
        eucos = eucos * (ag[i] * rkrel**2)
        bnx, bny, bnz = dirx - betax_, diry - betay_, dirz - betaz
        nbp = dirx*betaPx + diry*betaPy + dirz*betaPz
        nbn = dirx*bnx + diry*bny + dirz*bnz
        Bs += eucos*(bnx*nbp - betaPx*nbn)
        Bp += eucos*(bny*nbp - betaPy*nbn)
    return Bs*emcg, Bp*emcg


# --------------------------------------------------------------------------
# electron trajectory in a tabulated field: SourceFromField._build_trajectory_conv,
# synchr.py:1049-1147 (OpenCL twins get_trajectory[_filament], cl/undulator.cl:733, 918)
# --------------------------------------------------------------------------
SIM0, C_LIGHT = 9.109383701528e-31, 2.99792458e10        # physconsts.py:17, 14


def trajectory(wtGrid, Bx, By, Bz, gamma=None):
    """-> (betax, betay, betam_int, trajx, trajy, trajz) ON THE GRID wtGrid (the
    reference then splines them onto the integration nodes). *gamma* None: the
    non-filament form (velocities per unit emcg); else the filament electron's gamma.
    Field arrays on the half-step grid (2 len(wtGrid) - 1 points)."""
    filamentBeam = gamma is not None

    def f_beta(B, beta):
        return emcg*np.array((beta[1]*B[2]-B[1], B[0] - beta[0]*B[2]))

    def f_traj(beta):
        if filamentBeam:
            smTerm = 1./gamma**2 + beta[0]**2 + beta[1]**2
            betaz = 1. - 0.5*smTerm - 0.125*smTerm**2 - 0.0625*smTerm**3
        else:
            betaz = -0.5*(beta[0]**2 + beta[1]**2)
        return np.array((beta[0], beta[1], betaz))

    def next_beta_rk(iB, beta):
        k1beta = rkStep * f_beta([Bx[iB], By[iB], Bz[iB]], beta)
        k2beta = rkStep * f_beta([Bx[iB+1], By[iB+1], Bz[iB+1]], beta + 0.5*k1beta)
        k3beta = rkStep * f_beta([Bx[iB+1], By[iB+1], Bz[iB+1]], beta + 0.5*k2beta)
        k4beta = rkStep * f_beta([Bx[iB+2], By[iB+2], Bz[iB+2]], beta + k3beta)
        return beta + (k1beta + 2*k2beta + 2*k3beta + k4beta)/6.

    def next_traj_rk(iB, beta, traj):
        k1beta = rkStep * f_beta([Bx[iB], By[iB], Bz[iB]], beta)
        k1traj = rkStep * f_traj(beta)
        k2beta = rkStep * f_beta([Bx[iB+1], By[iB+1], Bz[iB+1]], beta + 0.5*k1beta)
        k2traj = rkStep * f_traj(beta + 0.5*k1beta)
        k3beta = rkStep * f_beta([Bx[iB+1], By[iB+1], Bz[iB+1]], beta + 0.5*k2beta)
        k3traj = rkStep * f_traj(beta + 0.5*k2beta)
        k4beta = rkStep * f_beta([Bx[iB+2], By[iB+2], Bz[iB+2]], beta + k3beta)
        k4traj = rkStep * f_traj(beta + k3beta)
        return (beta + (k1beta + 2*k2beta + 2*k3beta + k4beta)/6.,
                traj + (k1traj + 2*k2traj + 2*k3traj + k4traj)/6.)

    if filamentBeam:
        gamma = np.array(gamma)
    emcg = SIE0 / SIM0 / C_LIGHT / 10. / gamma if filamentBeam else 1.
    beta_next = np.zeros(2)
    beta0 = np.zeros(2)
    betam_int = 0
    for i in range(len(wtGrid)-1):
        rkStep = wtGrid[i+1] - wtGrid[i]
        beta_next = next_beta_rk(2*i, beta_next)
        beta0 += rkStep * beta_next
    beta0 /= -(wtGrid[-1] - wtGrid[0])
    beta_next = np.copy(beta0)
    traj_next = np.zeros(3)
    traj0 = np.zeros(3)
    for i in range(len(wtGrid)-1):
        rkStep = wtGrid[i+1] - wtGrid[i]
        beta_next, traj_next = next_traj_rk(2*i, beta_next, traj_next)
        traj0 += rkStep * traj_next
        if filamentBeam:
            betam_int += rkStep * np.sqrt(
                1. - 1./gamma**2 - beta_next[0]**2 - beta_next[1]**2)
        else:
            betam_int += beta_next[0]**2 + beta_next[1]**2
    traj0 /= -(wtGrid[-1] - wtGrid[0])
    beta_next = np.copy(beta0)
    traj_next = np.copy(traj0)
    if filamentBeam:
        betam_int /= -(wtGrid[-1] - wtGrid[0])
    else:
        betam_int *= -0.5/(len(wtGrid)-1)
    betax, betay = [beta0[0]], [beta0[1]]
    trajx, trajy, trajz = [traj0[0]], [traj0[1]], [traj0[2]]
    for i in range(len(wtGrid)-1):
        rkStep = wtGrid[i+1] - wtGrid[i]
        beta_next, traj_next = next_traj_rk(2*i, beta_next, traj_next)
        betax.append(beta_next[0])
        betay.append(beta_next[1])
        trajx.append(traj_next[0])
        trajy.append(traj_next[1])
        trajz.append(traj_next[2])
    return (np.array(betax), np.array(betay), float(betam_int), np.array(trajx),
            np.array(trajy), np.array(trajz))
