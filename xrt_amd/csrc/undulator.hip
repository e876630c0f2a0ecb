// Undulator field integral for gfx950.
//
// What the reference computes (xrt/backends/raycing/sources/synchr.py:1930-2038,
// Undulator._sp_sum; OpenCL twins cl/undulator.cl:54-300): for every ray
// (photon energy w, observation angles ddphi/ddpsi, electron gamma) the sum over
// the quadrature nodes of one undulator period (far field) or over all Np
// periods (tapered gap / near field) of
//     ag · e^{iφ} · [ n × ((n − β) × β') ]_{x,y} / (1 − n·β)²
// One ray per lane; the node tables are the same for all rays, so a small pack
// kernel turns them into 16-double records that the main loop reads through
// the scalar unit (wave-uniform address → s_load), leaving the vector ALU the
// fp64 arithmetic only. The arithmetic follows the numpy path operation by
// operation (this library is built with -ffp-contract=off), because
// krel = 1 − n·β cancels eight digits and every rounding upstream of it shows.
#include "undulator.h"

#include "fp64_math.h"

namespace xrt {

namespace {

constexpr double PI_ = 3.1415926535897932384626433832795;
constexpr double PI2_ = 6.283185307179586476925286766559;
constexpr double E2WC_ = 5067.7309392068091;          // synchr.py module constant
constexpr double FINE_STR_ = 1 / 137.03599976;       // physconsts.py
constexpr double SIE0_ = 1.602176565e-19;

// record layout (doubles)
enum { N_TG, N_AG, N_S, N_C, N_SPH, N_CPH, N_S2X, N_S2XPH, N_SUM2, N_BPX, N_BPY, N_C2,
       N_KX2S2XPH, N_PAD0, N_PAD1, N_PAD2, N_REC };
static_assert(N_REC == UND_NODE_DOUBLES, "node record size");

// the workspace's sincos table (behind the jend node records): entry k = (cos, sin) of k steps
__device__ __forceinline__ void tab_pack(double* __restrict__ rec, int64_t jend) {
  double2* gtab = reinterpret_cast<double2*>(rec + jend * N_REC);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < SINCOS_TAB_N;
       k += gridDim.x * blockDim.x) {
    const double qt = (double)k * (4.0 / SINCOS_TAB_N);   // quarter turns, exact
    const double n = __builtin_rint(qt);
    double sn, cs;
    sincos_quarter_turns(qt - n, (unsigned)(int)n, sn, cs);
    gtab[k] = make_double2(cs, sn);
  }
}
// ... and its copy into a block's LDS
__device__ __forceinline__ void tab_fetch(double2* tab, const double* __restrict__ rec,
                                          int64_t jend) {
  const double2* __restrict__ gtab = reinterpret_cast<const double2*>(rec + jend * N_REC);
  for (int k = threadIdx.x; k < SINCOS_TAB_N; k += blockDim.x) tab[k] = gtab[k];
  __syncthreads();
}

__global__ void und_pack(UndulatorArgs a, double* __restrict__ rec) {
  tab_pack(rec, a.jend);
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= a.jend) return;
  double s = a.sintg[j], c = a.costg[j], sp = a.sintgph[j], cp = a.costgph[j];
  double kx2 = a.Kx * a.Kx, ky2 = a.Ky * a.Ky;
  double s2x = (2. * s) * c;
  double s2xph = (2. * sp) * cp;
  double* r = rec + j * N_REC;
  r[N_TG] = a.tg[j];
  r[N_AG] = a.ag[j];
  r[N_S] = s;
  r[N_C] = c;
  r[N_SPH] = sp;
  r[N_CPH] = cp;
  r[N_S2X] = s2x;
  r[N_S2XPH] = s2xph;
  r[N_SUM2] = ky2 * s2x + kx2 * s2xph;   // Ky²·sin2x + Kx²·sin2xph (synchr.py:2003, :2017)
  r[N_BPX] = (-a.Ky) * s;                 // betaPx for taperC = 1, alphaS = 0
  r[N_BPY] = a.Kx * sp;
  r[N_C2] = c * c;
  r[N_KX2S2XPH] = kx2 * s2xph;
  r[N_PAD0] = r[N_PAD1] = r[N_PAD2] = 0.;
}

// a / b, correctly rounded, for a divisor whose correctly rounded reciprocal y
// is already known (same construction as reflect.hip's div_const)
__device__ __forceinline__ double div_known(double a, double b, double y) {
  double q = a * y;
  double r = fma_(-q, b, a);
  return fma_(r, y, q);
}

// sincos of a node phase: SAFE = the caller has bounded every phase of the wave's rays below
// 2^42 (sincos_tab's range), so the loop body has no branch
template <bool SAFE>
__device__ __forceinline__ void sincos_node(double phi, const double2* tab,
                                            const SinCosTabRegs<>& k, double& sn, double& cs) {
  if (SAFE)
    sincos_tab<SINCOS_TAB_N, true>(phi, tab, k, sn, cs);
  else
    sincos_any(phi, tab, k, sn, cs);
}

// The per-ray sum; returns wu/γ·Σ (synchr.py:2038). The record of node j + 1 is requested
// (scalar loads: the address is wave-uniform) before node j is worked on: the loop used to
// wait for its record at the top of every node.
// PLANAR: Kx = 0 (the usual undulator). Every term with Kx in it is then an exact zero that is
// added to or subtracted from something: leaving those operations out gives the same bits
// (x + 0 = x, fma(0, y, x) = x) with nine instructions per node less.
template <int MODE, bool SAFE, bool PLANAR>
__device__ __forceinline__ void und_ray(const UndulatorArgs& a, const double* __restrict__ rec,
                                        const double2* tab, double g, double wu, double w,
                                        double ww1, double phi, double psi, double2& out_s,
                                        double2& out_p) {
  const SinCosTabRegs<> kreg;
  const double Kx = a.Kx, Ky = a.Ky;
  const double kx2 = Kx * Kx, ky2 = Ky * Ky;
  const double revg = 1. / g;
  const double revg2 = revg * revg;
  const double wwu = w / wu;
  const double wwug = wwu * revg;
  const double e8 = 0.125 * revg;
  const double h2 = 0.5 * revg;
  const double kyg = Ky * revg;
  const double nkxg = (-Kx) * revg;
  double dirx = phi, diry = psi;
  double dirz = 1. - 0.5 * (phi * phi + psi * psi);
  const double nky_dx = (-Ky) * dirx;
  const double kx_dy = Kx * diry;

  // mode-specific per-ray constants
  double aw = 0., aw2 = 0., rwu = 0.;            // taper
  double betam = 0., omb = 0., r0x = 0., r0y = 0., r0zv = 0., sr0 = 0., cr0 = 0.;
  if (MODE == UND_TAPER) {
    rwu = 1. / wu;
    aw = a.alpha_s / wu;
    aw2 = (2 * a.alpha_s) / wu;
  }
  if (MODE == UND_NF) {
    betam = 1. - (((1. + 0.5 * kx2) + 0.5 * ky2) * 0.5) * revg2;
    omb = 1. - betam;
    r0x = tan(phi) * a.r0z;
    r0y = tan(psi) * a.r0z;
    r0zv = a.r0z;
    sincos_phase(r0zv, sr0, cr0);   // sic: no w/wu factor (synchr.py:1950-1951)
  }

  double bsr = 0., bsi = 0., bpr = 0., bpi = 0.;
  const int nper = (MODE == UND_FAR) ? 1 : a.nper;
  for (int ip = 0; ip < nper; ++ip) {
    const double z0 = (double)(-(nper - 1)) * PI_ + (double)ip * PI2_;
    const double* __restrict__ ahead = rec;
    double r[N_REC], rn[N_REC];
#pragma unroll
    for (int k = 0; k < N_PAD0; ++k) r[k] = ahead[k];
#ifdef UND_UNROLL
#pragma unroll UND_UNROLL
#endif
    // (a 32-bit node counter: the 64-bit comparison has no scalar form and cost every node two
    // vector instructions, tools/kisa_audit.py)
    const int jend = (int)a.jend;
    for (int j = 0; j < jend; ++j) {
      if (j + 1 < jend) ahead += N_REC;
#pragma unroll
      for (int k = 0; k < N_PAD0; ++k) rn[k] = ahead[k];
      const double tg = r[N_TG], ag = r[N_AG], s = r[N_S], c = r[N_C];
      const double sp = PLANAR ? 0. : r[N_SPH], cp = PLANAR ? 0. : r[N_CPH];
      double er, ei, betax, bPx, bPz;
      const double bPy = PLANAR ? 0. : r[N_BPY];
      if (MODE == UND_FAR) {
#pragma clang fp contract(fast)
        double A = PLANAR ? nky_dx * s + e8 * r[N_SUM2]
                          : (nky_dx * s + kx_dy * sp) + e8 * r[N_SUM2];
        double ucos = ww1 * tg + wwug * A;
        sincos_node<SAFE>(ucos, tab, kreg, ei, er);
        betax = kyg * c;
        bPx = r[N_BPX];
        bPz = h2 * r[N_SUM2];
      } else if (MODE == UND_TAPER) {
        const double zloc = z0 + tg;
        const double s2x = r[N_S2X];
        double taperC = 1. - div_known(a.alpha_s * zloc, wu, rwu);
        double u1 = (1 - c) - zloc * s;                       // wave-uniform
        double u2 = (zloc * zloc + r[N_C2]) + zloc * s2x;     // wave-uniform
        double T1 = nky_dx * (s + aw * u1);
        double T2 = kx_dy * s;                                // sic: sintg, not sintgph
        double T3 = PLANAR ? e8 * (ky2 * (s2x - aw2 * u2))
                           : e8 * (r[N_KX2S2XPH] + ky2 * (s2x - aw2 * u2));
        double ucos = ww1 * zloc + wwug * (PLANAR ? T1 + T3 : (T1 + T2) + T3);
        sincos_node<SAFE>(ucos, tab, kreg, ei, er);
        betax = ((taperC * Ky) * revg) * c;
        bPx = (-Ky) * (a.alpha_s * c + taperC * s);
        bPz = PLANAR ? h2 * ((ky2 * taperC) * (a.alpha_s * r[N_C2] + taperC * s2x))
                     : h2 * ((ky2 * taperC) * (a.alpha_s * r[N_C2] + taperC * s2x) +
                             r[N_KX2S2XPH]);
      } else {
        const double zloc = z0 + tg;
        double zterm = (0.5 * r[N_SUM2]) * revg;
        double q4 = (0.25 * zterm) * revg;
        double rx = (Ky * s) * revg;
        double ry = (Kx * sp) * revg;
        double rz = betam * zloc - q4;
        double drx = r0x - rx, dry = PLANAR ? r0y : r0y - ry, drz = r0zv - rz;
        double dxy = drx * drx + dry * dry;
        double dist = __builtin_sqrt(dxy + drz * drz);
        double drs = (0.5 * dxy) / drz;
        double sz, cz, sd, cd;
        sincos_node<SAFE>((wwu * zloc) * omb, tab, kreg, sz, cz);
        sincos_node<SAFE>(wwu * (drs + q4), tab, kreg, sd, cd);
        er = ((((-sr0) * sz) * cd - (sr0 * cz) * sd) - (cr0 * sz) * sd) + (cr0 * cz) * cd;
        ei = ((((-sr0) * sz) * sd + (sr0 * cz) * cd) + (cr0 * sz) * cd) + (cr0 * cz) * sd;
        dirx = drx / dist;
        diry = dry / dist;
        dirz = drz / dist;
        betax = kyg * c;
        bPx = r[N_BPX];
        bPz = h2 * r[N_SUM2];
      }
      const double betay = PLANAR ? 0. : nkxg * cp;
      // krel = 1 - n.beta ~ 3e-8: eight digits cancel and it enters squared -- its sum keeps
      // the reference's roundings. krel is an ordinary number (1e-9 .. 2): the division
      // without its range scaling, same bits (fp64_math.h).
      const double betaz = PLANAR ? 1. - 0.5 * (revg2 + betax * betax)
                                  : 1. - 0.5 * ((revg2 + betax * betax) + betay * betay);
      const double krel = PLANAR ? (1. - dirx * betax) - dirz * betaz
                                 : ((1. - dirx * betax) - diry * betay) - dirz * betaz;
      // 1/krel: the hardware seed (2^-24) and ONE Newton step = 2e-15 relative (profiles/
      // r03_probe_fp64_seeds.txt); the correctly rounded quotient took four more instructions
      // per node for digits that the 1e-9 the fields are held to does not see.
      double rkrel = __builtin_amdgcn_rcp(krel);
      rkrel = fma_(fma_(-krel, rkrel, 1.0), rkrel, rkrel);
      {
        // Fields are compared at 1e-5 (observed 1e-16 when every rounding of the reference
        // is reproduced): from here on products feeding sums are fused, a quarter of the
        // node loop's instructions less.
#pragma clang fp contract(fast)
        const double fac = ag * (rkrel * rkrel);
        er = er * fac;
        ei = ei * fac;
        const double bnx = dirx - betax, bny = PLANAR ? diry : diry - betay, bnz = dirz - betaz;
        const double nbp = PLANAR ? dirx * bPx + dirz * bPz : (dirx * bPx + diry * bPy) + dirz * bPz;
        const double nbn = (dirx * bnx + diry * bny) + dirz * bnz;
        const double ts = bnx * nbp - bPx * nbn;
        const double tp = PLANAR ? bny * nbp : bny * nbp - bPy * nbn;
        bsr += er * ts;
        bsi += ei * ts;
        bpr += er * tp;
        bpi += ei * tp;
      }
#pragma unroll
      for (int k = 0; k < N_PAD0; ++k) r[k] = rn[k];
    }
  }
  const double f = wu * revg;
  out_s = make_double2(f * bsr, f * bsi);
  out_p = make_double2(f * bpr, f * bpi);
}

// An upper bound of every node phase of a ray (far field, tapered gap), from |sin|, |cos| <= 1
// and |tg| <= pi: below 2^42 the table sincos serves the whole loop. The near field's second
// phase has a quotient in it: no bound, the general form.
template <int MODE>
__device__ __forceinline__ bool phases_small(const UndulatorArgs& a, double g, double wu,
                                             double w, double ww1, double phi, double psi) {
  if (MODE == UND_NF) return false;
  const double kx2 = a.Kx * a.Kx, ky2 = a.Ky * a.Ky;
  const double Z = (MODE == UND_FAR ? 1. : (double)a.nper) * PI_;
  const double aw = MODE == UND_TAPER ? __builtin_fabs(a.alpha_s / wu) : 0.;
  const double A = __builtin_fabs(a.Ky * phi) * (1. + aw * (2. + Z)) + __builtin_fabs(a.Kx * psi) +
                   (0.125 / g) * (kx2 + ky2 * (1. + 2. * aw * ((Z * Z + 1.) + Z)));
  const double bound = __builtin_fabs(ww1) * Z + __builtin_fabs((w / wu) / g) * A;
  return bound < 0x1p42;      // (false for NaN)
}

// the per-ray sum with the loop form its wave qualifies for
template <int MODE>
__device__ __forceinline__ void und_ray_any(const UndulatorArgs& a,
                                            const double* __restrict__ rec, const double2* tab,
                                            double g, double wu, double w, double ww1, double phi,
                                            double psi, double2& s, double2& p) {
  // (the general sincos only with the general field: one loop less to keep in registers)
  if (a.Kx == 0. && __all(phases_small<MODE>(a, g, wu, w, ww1, phi, psi)))
    und_ray<MODE, true, true>(a, rec, tab, g, wu, w, ww1, phi, psi, s, p);
  else if (__all(phases_small<MODE>(a, g, wu, w, ww1, phi, psi)))
    und_ray<MODE, true, false>(a, rec, tab, g, wu, w, ww1, phi, psi, s, p);
  else
    und_ray<MODE, false, false>(a, rec, tab, g, wu, w, ww1, phi, psi, s, p);
}

// The kernels are persistent: UND_WAVES waves per SIMD (the launch is sized by the occupancy),
// every block copies the sincos table once and walks over tiles of 256 rays. Four waves per
// SIMD divide the 16 per SIMD of a 2^20-ray map evenly (five left a fifth of the last round).
#ifndef UND_WAVES
#define UND_WAVES 4
#endif
#ifndef UND_BLOCK          /* lanes per block (A/B: 1024 = one table copy per CU instead of four) */
#define UND_BLOCK 256
#endif
#define UND_PER_CU(waves) ((waves) * 256 / UND_BLOCK > 0 ? (waves) * 256 / UND_BLOCK : 1)                   // (the multi-period modes: one less, their three loop forms need the registers)
template <int MODE>
__global__ void __launch_bounds__(UND_BLOCK, UND_PER_CU(MODE == UND_FAR ? UND_WAVES : UND_WAVES - 1))
und_sum(UndulatorArgs a, const double* __restrict__ rec, int64_t n,
        const double* __restrict__ gamma, const double* __restrict__ wu,
        const double* __restrict__ w, const double* __restrict__ ww1,
        const double* __restrict__ ddphi, const double* __restrict__ ddpsi,
        double2* __restrict__ Is, double2* __restrict__ Ip) {
  // (cos, sin) of 2048 steps per turn for the in-loop sincos, fp64_math.h
  __shared__ double2 tab[SINCOS_TAB_N];
  tab_fetch(tab, rec, a.jend);
  const int64_t tiles = (n + blockDim.x - 1) / blockDim.x;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t i = t * blockDim.x + threadIdx.x;
    const int64_t k = i < n ? i : n - 1;      // (idle lanes repeat the last ray)
    double2 s, p;
    und_ray_any<MODE>(a, rec, tab, gamma[k], wu[k], w[k], ww1[k], ddphi[k], ddpsi[k], s, p);
    if (i < n) {
      Is[i] = s;
      Ip[i] = p;
    }
  }
}

// Whole Undulator._build_I_map_conv (synchr.py:2050-2108) in one kernel: the
// pre-factors wu, ww1, ab from (w, theta, psi, gamma), the sum, the harmonic
// window and the Amp2Flux scaling. numpy's operation order throughout.
template <int MODE>
__global__ void __launch_bounds__(UND_BLOCK, UND_PER_CU(MODE == UND_FAR ? UND_WAVES : UND_WAVES - 1))
und_imap(UndulatorArgs a, UndulatorMap m, const double* __restrict__ rec, int64_t n,
         const double* __restrict__ w_, const double* __restrict__ theta,
         const double* __restrict__ psi_, const double* __restrict__ gamma_,
         double* __restrict__ I, double2* __restrict__ Es, double2* __restrict__ Ep) {
  __shared__ double2 tab[SINCOS_TAB_N];
  tab_fetch(tab, rec, a.jend);
  const int64_t tiles = (n + blockDim.x - 1) / blockDim.x;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t i0 = t * blockDim.x + threadIdx.x;
    const int64_t i = i0 < n ? i0 : n - 1;    // (idle lanes repeat the last ray)
    const double w = w_[i], th = theta[i], ps = psi_[i];
    const double g = gamma_ ? gamma_[i] : m.gamma0;
    const double kx2 = a.Kx * a.Kx, ky2 = a.Ky * a.Ky;
    const double g2 = g * g;
    const double wu =
        ((((PI_ / m.L0) / g2) * 1.0) * (((2 * g2 - 1) - 0.5 * kx2) - 0.5 * ky2)) / E2WC_;
    const double ww1 = (w * (((1. + 0.5 * kx2) + 0.5 * ky2) + g2 * (th * th + ps * ps))) /
                       ((2. * g2) * wu);
    double ab = (1. / PI2_) / wu;
    if (MODE == UND_FAR) {
      double s1, c1, s2, c2;
      sincos_phase((PI_ * m.Np) * ww1, s1, c1);
      sincos_phase(PI_ * ww1, s2, c2);
      ab = (ab * s1) / s2;
    }
    double2 s, p;
    und_ray_any<MODE>(a, rec, tab, g, wu, w, ww1, th, ps, s, p);
    if (m.has_harmonic && (ww1 > m.harmonic + 0.5 || ww1 < m.harmonic - 0.5)) {
      s = make_double2(0., 0.);
      p = make_double2(0., 0.);
    }
    const double bw = m.dist_bw ? 0.001 : 1. / w;
    const double a2f = ((FINE_STR_ * bw) * m.eI) / SIE0_;
    // |Es|^2 + |Ep|^2 (numpy: abs(.)**2 through hypot; the squares themselves agree to an ulp)
    const double field = (s.x * s.x + s.y * s.y) + (p.x * p.x + p.y * p.y);
    const double f = __builtin_sqrt(a2f) * ab;
    if (i0 < n) {
      I[i] = (((a2f * (ab * ab)) * 0.25) * (m.dstep * m.dstep)) * field;
      Es[i] = make_double2(((f * s.x) * 0.5) * m.dstep, ((f * s.y) * 0.5) * m.dstep);
      Ep[i] = make_double2(((f * p.x) * 0.5) * m.dstep, ((f * p.y) * 0.5) * m.dstep);
    }
  }
}

// blocks of a persistent launch: what fits the chip at once, or one per tile if that is less
template <class K>
static unsigned persistent_grid(K kernel, int64_t n) {
  int dev = 0, cus = 256, per_cu = UND_WAVES;
  if (hipGetDevice(&dev) == hipSuccess)
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, UND_BLOCK, 0) != hipSuccess ||
      per_cu < 1) {
    (void)hipGetLastError();
    per_cu = UND_PER_CU(UND_WAVES);
  }
  if (per_cu > UND_PER_CU(UND_WAVES)) per_cu = UND_PER_CU(UND_WAVES);
  const int64_t tiles = (n + UND_BLOCK - 1) / UND_BLOCK, fit = (int64_t)cus * per_cu;
  return (unsigned)(tiles < fit ? tiles : fit);
}

}  // namespace

// ---------------------------------------------------------------------------
// Custom (tabulated) magnetic field: SourceFromField._sp_sum, synchr.py:888-973;
// OpenCL twins custom_field / custom_field_filament (cl/undulator.cl:822-1103).
// Node record: tg, ag, Bx, By, Bz, betax, betay, trajx, trajy, trajz.
// ---------------------------------------------------------------------------
namespace {
enum { C_TG, C_AG, C_BX, C_BY, C_BZ, C_BETAX, C_BETAY, C_TRAJX, C_TRAJY, C_TRAJZ, C_REC = 16 };
constexpr double EMC_ = 0.5866791802416487;   // physconsts.py:24

__global__ void cust_pack(xrt_hip_custom_field a, double* __restrict__ rec) {
  tab_pack(rec, a.jend);
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= a.jend) return;
  double* r = rec + j * C_REC;
  r[C_TG] = a.tg[j];
  r[C_AG] = a.ag[j];
  r[C_BX] = a.Bx[j];
  r[C_BY] = a.By[j];
  r[C_BZ] = a.Bz[j];
  r[C_BETAX] = a.betax[j];
  r[C_BETAY] = a.betay[j];
  r[C_TRAJX] = a.trajx[j];
  r[C_TRAJY] = a.trajy[j];
  r[C_TRAJZ] = a.trajz[j];
  for (int k = C_TRAJZ + 1; k < C_REC; ++k) r[k] = 0.;
}

template <bool FIL, bool NF>
__global__ void __launch_bounds__(256)
cust_sum(xrt_hip_custom_field a, const double* __restrict__ rec, int64_t n,
         const double* __restrict__ emcg_, const double* __restrict__ gamma,
         const double* __restrict__ w_, const double* __restrict__ ddphi,
         const double* __restrict__ ddpsi, double2* __restrict__ Is,
         double2* __restrict__ Ip) {
  __shared__ double2 tab[SINCOS_TAB_N];
  tab_fetch(tab, rec, a.jend);
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SinCosTabRegs<> kreg;
  const double emcg = emcg_[i], w = w_[i], phi = ddphi[i], psi = ddpsi[i];
  const double g = FIL ? gamma[0] : gamma[i];
  double dirx = phi, diry = psi;
  double dirz = __builtin_sqrt((1. - phi * phi) - psi * psi);
  const double revg2 = 1. / (g * g);
  const double emc2 = EMC_ * EMC_;
  // sic: the two carrier formulas are swapped w.r.t. the vectorised _sp
  const double by_mean = (w * E2WC_) / a.betam;
  const double by_gamma = (w * E2WC_) / (1. + (a.betam * emc2 - 0.5) * revg2);
  const double wc = a.carrier_form ? (FIL ? by_mean : by_gamma)   // the vectorised _sp
                                   : (FIL ? (a.wc > 0. ? a.wc : by_gamma) : by_mean);
  const double zfac = 1. - 0.5 * revg2;     // non-filament trajz_
  const double tfac = emc2 * revg2;
  double r0x = 0., r0y = 0., r0z = 0., sr0 = 0., cr0 = 0.;
  if (NF) {
    r0x = tan(phi) * a.R0;
    r0y = tan(psi) * a.R0;
    r0z = a.R0;
    sincos_phase(wc * r0z, sr0, cr0);
  }
  double bsr = 0., bsi = 0., bpr = 0., bpi = 0.;
  const double* __restrict__ r = rec;
  for (int64_t j = 0; j < a.jend; ++j, r += C_REC) {
    const double tg = r[C_TG], ag = r[C_AG];
    double bx, by, tx, ty, tz;
    if (FIL) {
      bx = r[C_BETAX];
      by = r[C_BETAY];
      tx = r[C_TRAJX];
      ty = r[C_TRAJY];
      tz = r[C_TRAJZ];
    } else {
      bx = emcg * r[C_BETAX];
      by = emcg * r[C_BETAY];
      tx = emcg * r[C_TRAJX];
      ty = emcg * r[C_TRAJY];
      tz = tg * zfac + tfac * r[C_TRAJZ];
    }
    double er, ei;
    if (NF) {
      const double drx = r0x - tx, dry = r0y - ty, drz = r0z - tz;
      const double dxy = drx * drx + dry * dry;
      const double dist = __builtin_sqrt(dxy + drz * drz);
      const double rdrz = 1. / drz;
      const double drs = dxy * rdrz;
      const double LRS = (0.5 * drs - (0.125 * (drs * drs)) * rdrz) +
                         (0.0625 * pow(drs, 3.)) * (rdrz * rdrz);
      double sz, cz, sd, cd;
      sincos_any(wc * (tg - tz), tab, kreg, sz, cz);
      sincos_any(wc * LRS, tab, kreg, sd, cd);
      er = ((((-sr0) * sz) * cd - (sr0 * cz) * sd) - (cr0 * sz) * sd) + (cr0 * cz) * cd;
      ei = ((((-sr0) * sz) * sd + (sr0 * cz) * cd) + (cr0 * sz) * cd) + (cr0 * cz) * sd;
      dirx = drx / dist;
      diry = dry / dist;
      dirz = drz / dist;
    } else {
      double s1, c1, s2, c2;
      sincos_any(wc * (tg - dirz * tz), tab, kreg, s1, c1);
      sincos_any(wc * (dirx * tx + diry * ty), tab, kreg, s2, c2);
      er = s1 * c2 - c1 * s2;
      ei = c1 * c2 + s1 * s2;
    }
    const double sm = (revg2 + bx * bx) + by * by;
    const double bz = ((1. - 0.5 * sm) - 0.125 * (sm * sm)) - 0.0625 * pow(sm, 3.);
    const double Bx = r[C_BX], By = r[C_BY], Bz = r[C_BZ];
    const double bPx = by * Bz - bz * By;
    const double bPy = (-bx) * Bz + bz * Bx;
    const double bPz = bx * By - by * Bx;
    const double krel = ((1. - dirx * bx) - diry * by) - dirz * bz;
    const double rkrel = 1. / krel;
    const double fac = ag * (rkrel * rkrel);
    er *= fac;
    ei *= fac;
    const double bnx = dirx - bx, bny = diry - by, bnz = dirz - bz;
    const double nbp = (dirx * bPx + diry * bPy) + dirz * bPz;
    const double nbn = (dirx * bnx + diry * bny) + dirz * bnz;
    const double ts = bnx * nbp - bPx * nbn;
    const double tp = bny * nbp - bPy * nbn;
    bsr += er * ts;
    bsi += ei * ts;
    bpr += er * tp;
    bpi += ei * tp;
  }
  Is[i] = make_double2(bsr * emcg, bsi * emcg);
  Ip[i] = make_double2(bpr * emcg, bpi * emcg);
}
}  // namespace

hipError_t custom_field_launch(const xrt_hip_custom_field& a, int64_t n, const double* emcg,
                               const double* gamma, const double* w, const double* ddphi,
                               const double* ddpsi, double* Is_ri, double* Ip_ri,
                               void* workspace, hipStream_t st, hipEvent_t e0,
                               hipEvent_t e1) {
  double* rec = reinterpret_cast<double*>(workspace);
  if (a.jend > 0) {
    hipLaunchKernelGGL(cust_pack, dim3((unsigned)((a.jend + 127) / 128)), dim3(128), 0, st, a,
                       rec);
  }
  if (n <= 0) return hipGetLastError();
  if (e0) (void)hipEventRecord(e0, st);
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  double2* is = reinterpret_cast<double2*>(Is_ri);
  double2* ip = reinterpret_cast<double2*>(Ip_ri);
  const bool nf = a.near_field != 0, fil = a.filament != 0;
#define XRT_CUST(FIL, NF)                                                                   \
  hipLaunchKernelGGL((cust_sum<FIL, NF>), grid, block, 0, st, a, rec, n, emcg, gamma, w, \
                     ddphi, ddpsi, is, ip)
  if (fil && nf)
    XRT_CUST(true, true);
  else if (fil)
    XRT_CUST(true, false);
  else if (nf)
    XRT_CUST(false, true);
  else
    XRT_CUST(false, false);
#undef XRT_CUST
  if (e1) (void)hipEventRecord(e1, st);
  return hipGetLastError();
}

hipError_t undulator_pack_launch(const UndulatorArgs& a, void* workspace, hipStream_t st) {
  if (a.jend <= 0) return hipSuccess;
  int pb = (int)((a.jend + 127) / 128);
  hipLaunchKernelGGL(und_pack, dim3(pb), dim3(128), 0, st, a, reinterpret_cast<double*>(workspace));
  return hipGetLastError();
}

hipError_t undulator_sum_launch(const UndulatorArgs& a, int64_t n, const double* gamma,
                                const double* wu, const double* w, const double* ww1,
                                const double* ddphi, const double* ddpsi, double* Is_ri,
                                double* Ip_ri, const void* workspace, hipStream_t st) {
  const double* rec = reinterpret_cast<const double*>(workspace);
  if (n <= 0) return hipSuccess;
  dim3 block(UND_BLOCK);
  double2* is = reinterpret_cast<double2*>(Is_ri);
  double2* ip = reinterpret_cast<double2*>(Ip_ri);
  switch (a.mode) {
    case UND_FAR:
      hipLaunchKernelGGL(und_sum<UND_FAR>, dim3(persistent_grid(und_sum<UND_FAR>, n)), block, 0,
                         st, a, rec, n, gamma, wu, w, ww1, ddphi, ddpsi, is, ip);
      break;
    case UND_TAPER:
      hipLaunchKernelGGL(und_sum<UND_TAPER>, dim3(persistent_grid(und_sum<UND_TAPER>, n)), block,
                         0, st, a, rec, n, gamma, wu, w, ww1, ddphi, ddpsi, is, ip);
      break;
    case UND_NF:
      hipLaunchKernelGGL(und_sum<UND_NF>, dim3(persistent_grid(und_sum<UND_NF>, n)), block, 0, st,
                         a, rec, n, gamma, wu, w, ww1, ddphi, ddpsi, is, ip);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t undulator_imap_launch(const UndulatorArgs& a, const UndulatorMap& m, int64_t n,
                                 const double* w, const double* theta, const double* psi,
                                 const double* gamma, double* I, double* Es_ri, double* Ep_ri,
                                 const void* workspace, hipStream_t st) {
  const double* rec = reinterpret_cast<const double*>(workspace);
  if (n <= 0) return hipSuccess;
  dim3 block(UND_BLOCK);
  double2* es = reinterpret_cast<double2*>(Es_ri);
  double2* ep = reinterpret_cast<double2*>(Ep_ri);
  switch (a.mode) {
    case UND_FAR:
      hipLaunchKernelGGL(und_imap<UND_FAR>, dim3(persistent_grid(und_imap<UND_FAR>, n)), block, 0,
                         st, a, m, rec, n, w, theta, psi, gamma, I, es, ep);
      break;
    case UND_TAPER:
      hipLaunchKernelGGL(und_imap<UND_TAPER>, dim3(persistent_grid(und_imap<UND_TAPER>, n)), block,
                         0, st, a, m, rec, n, w, theta, psi, gamma, I, es, ep);
      break;
    case UND_NF:
      hipLaunchKernelGGL(und_imap<UND_NF>, dim3(persistent_grid(und_imap<UND_NF>, n)), block, 0, st,
                         a, m, rec, n, w, theta, psi, gamma, I, es, ep);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Electron trajectory in a tabulated field: SourceFromField._build_trajectory_conv
// (sources/synchr.py:1049-1147), the job of the OpenCL one-work-item kernels
// get_trajectory / get_trajectory_filament (cl/undulator.cl:733, 918). Fourth-order
// Runge-Kutta along the grid wt[n] with the field on the half-step grid B[2n-1]; three
// sweeps: (1) velocity from rest -> its mean is removed, (2) velocity + position -> the
// mean position is removed and the mean longitudinal term is collected, (3) the final
// tables. A recurrence: ONE lane walks it (a step is ~90 dependent fp64 operations, the
// grid has thousands of points: milliseconds, where the reference's Python loop takes
// seconds per call), in the reference's operation order. FIL: filamentBeam (velocities in
// units of c with the electron's own gamma) / else per unit emcg.
// ---------------------------------------------------------------------------
namespace {
struct Vel {
  double x, y;
};
struct Pos {
  double x, y, z;
};
struct Field {
  double x, y, z;
};

__device__ __forceinline__ Vel beta_rate(double emcg, const Field& B, const Vel& b) {
  return {emcg * (b.y * B.z - B.y), emcg * (B.x - b.x * B.z)};
}

template <bool FIL>
__device__ __forceinline__ Pos traj_rate(double gamma, const Vel& b) {
  double bz;
  if (FIL) {
    const double sm = 1. / (gamma * gamma) + b.x * b.x + b.y * b.y;
    bz = 1. - 0.5 * sm - 0.125 * (sm * sm) - 0.0625 * (sm * sm * sm);
  } else {
    bz = -0.5 * (b.x * b.x + b.y * b.y);
  }
  return {b.x, b.y, bz};
}

__device__ __forceinline__ Vel vel_at(const Vel& b, double f, const Vel& k) {
  return {b.x + f * k.x, b.y + f * k.y};
}

__device__ __forceinline__ double rk_mix(double k1, double k2, double k3, double k4) {
  return (k1 + 2. * k2 + 2. * k3 + k4) / 6.;
}

// one Runge-Kutta step of the velocity (and, WITH_POS, of the position)
template <bool FIL, bool WITH_POS>
__device__ __forceinline__ void rk_step(double h, double emcg, double gamma, const Field& B0,
                                        const Field& B1, const Field& B2, Vel& b, Pos& p) {
  const Vel r1 = beta_rate(emcg, B0, b);
  const Vel k1 = {h * r1.x, h * r1.y};
  const Vel b2 = vel_at(b, 0.5, k1);
  const Vel r2 = beta_rate(emcg, B1, b2);
  const Vel k2 = {h * r2.x, h * r2.y};
  const Vel b3 = vel_at(b, 0.5, k2);
  const Vel r3 = beta_rate(emcg, B1, b3);
  const Vel k3 = {h * r3.x, h * r3.y};
  const Vel b4 = {b.x + k3.x, b.y + k3.y};
  const Vel r4 = beta_rate(emcg, B2, b4);
  const Vel k4 = {h * r4.x, h * r4.y};
  if (WITH_POS) {
    const Pos t1 = traj_rate<FIL>(gamma, b), t2 = traj_rate<FIL>(gamma, b2);
    const Pos t3 = traj_rate<FIL>(gamma, b3), t4 = traj_rate<FIL>(gamma, b4);
    p.x = p.x + rk_mix(h * t1.x, h * t2.x, h * t3.x, h * t4.x);
    p.y = p.y + rk_mix(h * t1.y, h * t2.y, h * t3.y, h * t4.y);
    p.z = p.z + rk_mix(h * t1.z, h * t2.z, h * t3.z, h * t4.z);
  }
  b.x = b.x + rk_mix(k1.x, k2.x, k3.x, k4.x);
  b.y = b.y + rk_mix(k1.y, k2.y, k3.y, k4.y);
}

template <bool FIL>
__global__ void trajectory_kernel(int64_t n, const double* __restrict__ wt,
                                  const double* __restrict__ Bx, const double* __restrict__ By,
                                  const double* __restrict__ Bz, double gamma, double emcg,
                                  double* __restrict__ betax, double* __restrict__ betay,
                                  double* __restrict__ trajx, double* __restrict__ trajy,
                                  double* __restrict__ trajz, double* __restrict__ betam) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const double span = -(wt[n - 1] - wt[0]);
  auto field = [&](int64_t j) { return Field{Bx[j], By[j], Bz[j]}; };
  // sweep 1: velocity from rest; beta0 = -(its integral) / length
  Vel b = {0., 0.}, b0 = {0., 0.};
  Pos none = {0., 0., 0.};
  for (int64_t i = 0; i + 1 < n; ++i) {
    const double h = wt[i + 1] - wt[i];
    rk_step<FIL, false>(h, emcg, gamma, field(2 * i), field(2 * i + 1), field(2 * i + 2), b,
                        none);
    b0.x += h * b.x;
    b0.y += h * b.y;
  }
  b0.x /= span;
  b0.y /= span;
  // sweep 2: position from the origin; traj0 likewise, and the mean longitudinal term
  b = b0;
  Pos p = {0., 0., 0.}, p0 = {0., 0., 0.};
  double bm = 0.;
  for (int64_t i = 0; i + 1 < n; ++i) {
    const double h = wt[i + 1] - wt[i];
    rk_step<FIL, true>(h, emcg, gamma, field(2 * i), field(2 * i + 1), field(2 * i + 2), b, p);
    p0.x += h * p.x;
    p0.y += h * p.y;
    p0.z += h * p.z;
    if (FIL)
      bm += h * sqrt(1. - 1. / (gamma * gamma) - b.x * b.x - b.y * b.y);
    else
      bm += b.x * b.x + b.y * b.y;
  }
  p0.x /= span;
  p0.y /= span;
  p0.z /= span;
  if (FIL)
    bm /= span;
  else
    bm *= -0.5 / (double)(n - 1);
  *betam = bm;
  // sweep 3: the tables
  b = b0;
  p = p0;
  betax[0] = b.x;
  betay[0] = b.y;
  trajx[0] = p.x;
  trajy[0] = p.y;
  trajz[0] = p.z;
  for (int64_t i = 0; i + 1 < n; ++i) {
    const double h = wt[i + 1] - wt[i];
    rk_step<FIL, true>(h, emcg, gamma, field(2 * i), field(2 * i + 1), field(2 * i + 2), b, p);
    betax[i + 1] = b.x;
    betay[i + 1] = b.y;
    trajx[i + 1] = p.x;
    trajy[i + 1] = p.y;
    trajz[i + 1] = p.z;
  }
}
}  // namespace

hipError_t trajectory_launch(int filament, int64_t n, const double* wt, const double* Bx,
                             const double* By, const double* Bz, double gamma, double emcg,
                             double* betax, double* betay, double* trajx, double* trajy,
                             double* trajz, double* betam, hipStream_t st) {
  if (n < 2) return hipErrorInvalidValue;
  if (filament)
    hipLaunchKernelGGL(trajectory_kernel<true>, dim3(1), dim3(64), 0, st, n, wt, Bx, By, Bz,
                       gamma, emcg, betax, betay, trajx, trajy, trajz, betam);
  else
    hipLaunchKernelGGL(trajectory_kernel<false>, dim3(1), dim3(64), 0, st, n, wt, Bx, By, Bz,
                       gamma, emcg, betax, betay, trajx, trajy, trajz, betam);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Bending magnet / wiggler: BendingMagnet.build_I_map (sources/synchr.py:185-227), an
// elementwise map per (E, theta, psi) with the modified Bessel functions K_{1/3}, K_{2/3}
// (scipy.special.kv in the reference).
// ---------------------------------------------------------------------------
namespace {
constexpr double kPI = 3.1415926535897932384626433832795;
// K_{1/3}(x) and K_{2/3}(x), x > 0: Temme's series below x = 2, Steed's continued fraction
// above (the classical pair of methods for fractional order, with mu = -1/3 so that one
// upward recurrence step gives the order 2/3); 1/Gamma(1 -+ mu) and their two combinations
// are constants for this mu.
__device__ __forceinline__ void bessel_k13_k23(double x, double& k13, double& k23) {
  constexpr double MU = -1. / 3., MU2 = MU * MU;
  constexpr double GAMPL = 0.7384881116216483129, GAMMI = 1.1198465217221856850;
  constexpr double GAM1 = -0.5720376151508060581, GAM2 = 0.9291673166719169990;
  constexpr double EPS = 1e-16;
  if (!(x > 0.)) {
    k13 = k23 = x == 0. ? INFINITY : NAN;
    return;
  }
  if (x > 705.) {  // exp(-x) underflows what the prefactors can bring back
    k13 = k23 = 0.;
    return;
  }
  const double xi = 1. / x, xi2 = 2. * xi;
  if (x < 2.) {
    const double x2 = 0.5 * x, pimu = kPI * MU;
    const double fact = pimu / sin(pimu);
    double d = -log(x2);
    double e = MU * d;
    const double fact2 = fabs(e) < EPS ? 1. : sinh(e) / e;
    double ff = fact * (GAM1 * cosh(e) + GAM2 * fact2 * d);
    double sum = ff;
    e = exp(e);
    double p = 0.5 * e / GAMPL, q = 0.5 / (e * GAMMI), c = 1.;
    d = x2 * x2;
    double sum1 = p;
    for (int i = 1; i <= 500; ++i) {
      ff = (i * ff + p + q) / (i * (double)i - MU2);
      c *= d / i;
      p /= i - MU;
      q /= i + MU;
      const double del = c * ff;
      sum += del;
      sum1 += c * (p - i * ff);
      if (fabs(del) < fabs(sum) * EPS) break;
    }
    k13 = sum;
    k23 = sum1 * xi2;
    return;
  }
  double b = 2. * (1. + x), d = 1. / b, h = d, delh = d, q1 = 0., q2 = 1.;
  const double a1 = 0.25 - MU2;
  double q = a1, c = a1, a = -a1, sc = 1. + q * delh;
  for (int i = 2; i <= 500; ++i) {
    a -= 2 * (i - 1);
    c = -a * c / i;
    const double qnew = (q1 - b * q2) / a;
    q1 = q2;
    q2 = qnew;
    q += c * qnew;
    b += 2.;
    d = 1. / (b + a * d);
    delh = (b * d - 1.) * delh;
    h += delh;
    const double dels = q * delh;
    sc += dels;
    if (fabs(dels / sc) < EPS) break;
  }
  h = a1 * h;
  k13 = sqrt(kPI / (2. * x)) * exp(-x) / sc;
  k23 = k13 * (MU + x + 0.5 - h) * xi;
}

__global__ __launch_bounds__(256) void bend_imap(xrt_hip_bend m, int64_t n,
                                                 const double* __restrict__ E,
                                                 const double* __restrict__ theta,
                                                 const double* __restrict__ psi,
                                                 const double* __restrict__ gamma_ray,
                                                 double* __restrict__ I,
                                                 double2* __restrict__ Es,
                                                 double2* __restrict__ Ep) {
  constexpr double SQ3 = 1.7320508075688772935, E2W = 1519267514747457.9195;
  constexpr double SIE0 = 1.602176565e-19, SIM0 = 9.109383701528e-31;
  constexpr double FINE_STR = 1 / 137.03599976;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double gamma = gamma_ray ? gamma_ray[i] : m.gamma;
  const double e = E[i];
  double w_cr = 1.5 * (gamma * gamma) * m.B * SIE0 / SIM0;
  if (m.wiggler) {  // the field the electron sees where it points along theta
    const double arg = theta[i] * gamma / m.K;
    const double under = 1. - arg * arg;
    w_cr *= sqrt(under > 0. ? under : 0.);
  }
  if (!isfinite(w_cr)) w_cr = 0.;
  const double gpsi = gamma * psi[i];
  const double g2p1 = gpsi * gpsi + 1.;
  const double eta = 0.5 * e * E2W / w_cr * (g2p1 * sqrt(g2p1));
  const double pref = -0.5 * SQ3 / kPI * gamma * e * E2W / w_cr * g2p1;  // ampSP = i pref
  double k13, k23;
  bessel_k13_k23(eta, k13, k23);
  double as = pref * k23;                        // ampS = i as
  double ap = -gpsi * pref * k13 / sqrt(g2p1);   // ampP = ap
  if (!isfinite(as)) as = 0.;
  if (!isfinite(ap)) ap = 0.;
  const double flux = FINE_STR * (m.per_bandwidth ? 0.001 : 1. / e) * m.eI / SIE0 * m.poles;
  const double root = sqrt(flux);
  I[i] = flux * (as * as + ap * ap);
  Es[i] = make_double2(0., root * as);
  Ep[i] = make_double2(root * ap, 0.);
}

__global__ __launch_bounds__(256) void bessel_k_probe(int64_t n, const double* __restrict__ x,
                                                      double* __restrict__ k13,
                                                      double* __restrict__ k23) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) bessel_k13_k23(x[i], k13[i], k23[i]);
}
}  // namespace

hipError_t bend_imap_launch(const xrt_hip_bend& m, int64_t n, const double* E,
                            const double* theta, const double* psi, const double* gamma,
                            double* I, double* Es_ri, double* Ep_ri, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(bend_imap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m, n, E,
                     theta, psi, gamma, I, reinterpret_cast<double2*>(Es_ri),
                     reinterpret_cast<double2*>(Ep_ri));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// GaussianBeam / LaguerreGaussianBeam / HermiteGaussianBeam.shine on the points of a wave
// (sources/geoms.py:684-810): the analytical mode field per point, times sqrt(dS), and the
// ray direction along the local wavefront normal. The carrier phase k y is ~1e11 rad: its
// terms are rounded in numpy's order.
// ---------------------------------------------------------------------------
namespace {
__device__ __forceinline__ double hermite_phys(int n, double x) {   // H_n, physicists'
  double h0 = 1., h1 = 2. * x;
  if (n == 0) return h0;
  for (int k = 1; k < n; ++k) {
    const double h2 = 2. * x * h1 - 2. * k * h0;
    h0 = h1;
    h1 = h2;
  }
  return h1;
}
__device__ __forceinline__ double laguerre_gen(int p, double alpha, double x) {   // L_p^alpha
  double l0 = 1., l1 = 1. + alpha - x;
  if (p == 0) return l0;
  for (int k = 1; k < p; ++k) {
    const double l2 = ((2. * k + 1. + alpha - x) * l1 - (k + alpha) * l0) / (k + 1.);
    l0 = l1;
    l1 = l2;
  }
  return l1;
}
__device__ __forceinline__ double ipow(double b, int e) {
  double r = 1.;
  for (int k = 0; k < e; ++k) r *= b;
  return r;
}

__global__ __launch_bounds__(256) void gauss_beam_kernel(
    xrt_hip_gauss G, int64_t n, const double* __restrict__ xs, const double* __restrict__ ys,
    const double* __restrict__ zs, const double* __restrict__ Es,
    const double* __restrict__ dS, double dS_scalar, double2* __restrict__ amp_out,
    double* __restrict__ oa, double* __restrict__ ob, double* __restrict__ oc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double kCH = 6.626069573e-27 * 2.99792458e10 / 1.602176565e-12 * 1e8;
  const double kCHBAR = kCH / 6.283185307179586476925286766559;
  const double c0 = sqrt(2. / 3.141592653589793);
  const double x = xs[i], y = ys[i], z = zs[i];
  const double k = Es[i] / kCHBAR * 1e7;
  const int gouy = G.mode == 1 ? abs(G.l) + 2 * G.p : G.mode == 2 ? G.m + G.n : 0;
  double re, im;        // the field
  double invR, wx, wz, rsq;
  if (G.astigmatic) {
    double s, c;
    sincos_phase(k * y, s, c);
    re = c0 * c;
    im = c0 * s;
    for (int ax = 0; ax < 2; ++ax) {
      const double w0 = ax == 0 ? G.w0x : G.w0z;
      const double yR = k / 2. * (w0 * w0);
      invR = y / (y * y + yR * yR);
      const double psi = (gouy + 1) * atan2(y, yR) * 0.5;
      const double q = y / yR;
      const double w = w0 * sqrt(1. + q * q);
      rsq = ax == 0 ? x * x : z * z;
      if (ax == 0) wx = w; else wz = w;
      const double mag = exp(-rsq / (w * w)) / sqrt(w);
      sincos(((0.5 * k) * rsq) * invR - psi, &s, &c);
      const double r2 = (re * c - im * s) * mag, i2 = (re * s + im * c) * mag;
      re = r2;
      im = i2;
    }
  } else {
    const double w0 = G.w0x;
    const double yR = k / 2. * (w0 * w0);
    invR = y / (y * y + yR * yR);
    const double psi = (gouy + 1) * atan2(y, yR);
    const double q = y / yR;
    const double w = w0 * sqrt(1. + q * q);
    wx = wz = w;
    rsq = x * x + z * z;
    const double theta = k * (y + (0.5 * rsq) * invR) - psi;
    double s, c;
    sincos_phase(theta, s, c);
    const double mag = c0 / w * exp(-rsq / (w * w));
    re = mag * c;
    im = mag * s;
  }
  if (G.mode == 1) {          // Laguerre-Gauss, :758-765
    const int al = abs(G.l);
    double f = G.clp * ipow(sqrt(rsq * 2.) / wx, al);
    if (G.p > 0) f *= laguerre_gen(G.p, (double)al, 2. * rsq / (wx * wx));
    double s, c;
    sincos((double)G.l * atan2(z, x), &s, &c);
    const double r2 = (re * c - im * s) * f, i2 = (re * s + im * c) * f;
    re = r2;
    im = i2;
  } else if (G.mode == 2) {   // Hermite-Gauss, :766-775
    double f = G.clp;
    const double r2 = sqrt(2.);
    if (G.m > 0) f *= hermite_phys(G.m, r2 * x / wx);
    if (G.n > 0) f *= hermite_phys(G.n, r2 * z / wz);
    re *= f;
    im *= f;
  }
  const double area = sqrt(dS ? dS[i] : dS_scalar);
  amp_out[i] = make_double2(re * area, im * area);
  // direction: along the radius of curvature (the last axis's, as the reference), :785-795
  double b = invR == 0. ? 1e20 : 1. / invR;
  b = sqrt(b * b - x * x - z * z);
  const double norm = sqrt(x * x + b * b + z * z);
  oa[i] = x / norm;
  ob[i] = b / norm;
  oc[i] = z / norm;
}
}  // namespace

hipError_t gauss_beam_launch(const xrt_hip_gauss& G, int64_t n, const double* x,
                             const double* y, const double* z, const double* E,
                             const double* dS, double dS_scalar, double* amp_ri, double* a,
                             double* b, double* c, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gauss_beam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, G,
                     n, x, y, z, E, dS, dS_scalar, reinterpret_cast<double2*>(amp_ri), a, b, c);
  return hipGetLastError();
}

hipError_t bessel_k_probe_launch(int64_t n, const double* x, double* k13, double* k23,
                                 hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(bessel_k_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, x,
                     k13, k23);
  return hipGetLastError();
}

}  // namespace xrt
