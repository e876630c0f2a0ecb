// Ray-surface intersection + reflect / refract with Fresnel / Bragg amplitudes
// for gfx950 (MI355X), fp64. One "pass" = OE._reflect_local of the reference
// (oes/reflect.py:551-1139) with the frame transforms of OE.reflect /
// DCM.double_reflect fused in.
//
// The reference takes four batch-global decisions that every ray depends on:
//   (1) the bracketing axis = argmax of max|a|, max|b|, max|c| over the rays
//       with state 1 (oes/base.py:1257-1270);
//   (2) the bracket formula = sign of that direction component of the FIRST
//       entering ray (base.py:1239);
//   (3) t1.min(), t2.max() clamps and the Brent-vs-secant choice
//       max|dz2| > 20 max|dz1| (base.py:861-878);
//   (4) for crystals, the sign of mean(beamInDotNormal) over the rays that hit
//       (reflect.py:573-574).
// They stay on the device: tiny reduction kernels write them into a GStat
// record in the workspace, the next kernel reads them. Kernel sequence:
//   init -> K1 stats_dir -> decide_axis -> K2 stats_bracket ->
//   K3 fused solve+finish            (mirror / plate / no material), or
//   K3a solve -> K3b finish          (crystal: needs decision (4) in between).
// One lane = one ray; ray fields are SoA so every load/store is coalesced
// (8 B/lane, 512 B per wave instruction). All arithmetic that decides the ray
// state follows numpy's operation order with no FMA contraction
// (-ffp-contract=off), IEEE division and correctly rounded sqrt.
#include <hip/hip_runtime.h>
//
// This header holds the device code (templates); the kernels are instantiated by the
// translation units reflect_hot / _xtal / _generic / _layered_* / _exact*.hip (built in
// parallel, see reflect_tu.h), the launch logic and the utility kernels are in reflect.hip.
#pragma once
#include <type_traits>
#include <math.h>
#include <stdint.h>

#include "../../include/xrt_hip.h"
#include "fp64_math.h"
#include "kernarg.h"
#include "reflect.h"
#include "screen_impl.h"
#include "plot_tail.h"
#include "source_impl.h"

namespace xrt {

// constants, restated from xrt/backends/raycing/physconsts.py (same FP expressions)
__device__ constexpr double kPI = 3.1415926535897932384626433832795;
__device__ constexpr double kPI2 = 6.283185307179586476925286766559;
__device__ constexpr double kCH = 6.626069573e-27 * 2.99792458e10 / 1.602176565e-12 * 1e8;
__device__ constexpr double kCHBAR = kCH / kPI2;
__device__ constexpr double kR0 = 2.817940285e-5;
__device__ constexpr double kAVOGADRO = 6.02214199e23;
__device__ constexpr double kZEps = 1e-12;        // raycing/__init__.py:86
__device__ constexpr int kMaxIteration = 100;     // :88
__device__ constexpr double kDt = 1e-5;           // :90
__device__ constexpr double kMaxHalfSize = 1000.; // :92
__device__ constexpr double kMaxDepth = 100.;     // :94

// ---------------------------------------------------------------------------
// Which chunk of the beam a block works on. Hardware hands consecutive block ids to the
// eight XCDs in turn; with XRT_XCD_CHUNKS every XCD walks ONE contiguous eighth of the beam,
// so that the dirty lines of an output array in its L2 are neighbours in memory.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned beam_block() {
#ifdef XRT_XCD_CHUNKS
  const unsigned nb = gridDim.x, b = blockIdx.x;
  const unsigned q = nb >> 3, rem = nb & 7u, x = b & 7u, j = b >> 3;
  return x * q + (x < rem ? x : rem) + j;
#else
  return blockIdx.x;
#endif
}

// ---------------------------------------------------------------------------
// Development probe (tools/build_variant.sh NAME -DXRT_PROBE_TIMING): shader-clock stamps
// between the sections of a ray's life in the lean fused kernel, summed per wave into
// g_probe_ticks (read back through xrt_probe_ticks of reflect_hot.hip). Not compiled into
// the regular library.
// ---------------------------------------------------------------------------
#ifdef XRT_PROBE_TIMING
#define XRT_PROBE_WAVES (1 << 18)
__device__ unsigned long long g_probe_ticks[XRT_PROBE_WAVES * 8];   // one slot per wave
struct ProbeClock {
  unsigned long long t[8];
  __device__ __forceinline__ void tick(int k, double dep) {
    unsigned long long v;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "v"(dep) : "memory");
    t[k] = v;
  }
  __device__ __forceinline__ void flush(int n) {
    const unsigned long long w = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0 && w < XRT_PROBE_WAVES) {
      for (int k = 0; k + 1 < n; ++k) g_probe_ticks[w * 8 + k] = t[k + 1] - t[k];
      g_probe_ticks[w * 8 + 7] = t[0];
    }
  }
};
#define XRT_TICK(pc, k, dep) (pc).tick(k, dep)
#else
struct ProbeClock {};
#define XRT_TICK(pc, k, dep) ((void)0)
#endif

// ---------------------------------------------------------------------------
// complex helpers (numpy's algorithms where the choice is visible at 1e-16)
// ---------------------------------------------------------------------------
struct cplx {
  double re, im;
};
__device__ __forceinline__ cplx C(double r, double i) { return cplx{r, i}; }
__device__ __forceinline__ cplx operator+(cplx a, cplx b) { return C(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cplx operator-(cplx a, cplx b) { return C(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cplx operator-(cplx a) { return C(-a.re, -a.im); }
__device__ __forceinline__ cplx operator*(cplx a, cplx b) {
  return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
__device__ __forceinline__ cplx operator*(cplx a, double s) { return C(a.re * s, a.im * s); }
__device__ __forceinline__ cplx operator*(double s, cplx a) { return C(a.re * s, a.im * s); }
// numpy divides complex by real through its complex loop (Smith): x * (1/s)
__device__ __forceinline__ double frcp(double x);
__device__ __forceinline__ cplx operator/(cplx a, double s) {
  const double scl = frcp(s);
  return C(a.re * scl, a.im * scl);
}
__device__ __forceinline__ cplx conj(cplx a) { return C(a.re, -a.im); }
// The amplitude half of the pipeline only needs ~1e-16 relative accuracy (it is
// compared at 1e-10, bar 1e-5), not IEEE-exact quotients: reciprocal by
// v_rcp_f64 + two Newton steps (<= 1 ulp) instead of the 14-instruction IEEE
// division, and an unscaled hypot (magnitudes are O(1e-10..1e3) here).
__device__ __forceinline__ double frcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma_(fma_(-x, r, 1.0), r, r);
  r = fma_(fma_(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fhypot(double a, double b) {
  return __builtin_sqrt(fma_(a, a, b * b));
}
__device__ __forceinline__ double cnorm2(double re, double im) { return fma_(re, re, im * im); }
__device__ __forceinline__ double cabs_(cplx a) { return fhypot(a.re, a.im); }
__device__ __forceinline__ bool cisnan(cplx a) { return isnan(a.re) || isnan(a.im); }
// a / b = a conj(b) / |b|^2: one reciprocal and no branch. (numpy's loop is Smith's
// algorithm, two divisions; the operands of this pipeline are far from over- or
// underflowing when squared, and ~1e-16 relative is all that is compared.) b = 0 gives
// inf / NaN components like numpy's.
__device__ __forceinline__ cplx operator/(cplx a, cplx b) {
  const double scl = frcp(cnorm2(b.re, b.im));
  return C(fma_(a.re, b.re, a.im * b.im) * scl, fma_(a.im, b.re, -(a.re * b.im)) * scl);
}
// principal root (C99 csqrt); both square roots through the unscaled Goldschmidt
// sequence of fp64_math.h, whose second result 1/sqrt replaces the division by 2t
__device__ __forceinline__ cplx csqrt_(cplx z) {
  if (z.re == 0. && z.im == 0.) return C(0., z.im);
  double unused, it;
  const double m = sqrt_rn_rinv(cnorm2(z.re, z.im), unused);
  const double t = sqrt_rn_rinv((fabs(z.re) + m) * 0.5, it);   // it = 1/t
  const double other = fabs(z.im) * (0.5 * it);
  if (z.re >= 0.) return C(t, copysign(other, z.im));
  return C(other, copysign(t, z.im));
}
__device__ __forceinline__ cplx cexp_(cplx z) {
  double s, c;
  sincos(z.im, &s, &c);
  const double e = exp(z.re);
  return C(e * c, e * s);
}
__device__ __forceinline__ cplx ccos_(cplx z) {
  double s, c;
  sincos(z.re, &s, &c);
  return C(c * cosh(z.im), -s * sinh(z.im));
}
__device__ __forceinline__ cplx csin_(cplx z) {
  double s, c;
  sincos(z.re, &s, &c);
  return C(s * cosh(z.im), c * sinh(z.im));
}
__device__ __forceinline__ cplx ctan_(cplx z) {
  double s2, c2;
  sincos(2. * z.re, &s2, &c2);
  if (fabs(z.im) > 20.) {  // cosh(2y) dwarfs cos(2x): tan -> +-i
    const double e = exp(-2. * fabs(z.im));
    return C(2. * s2 * e, copysign(1., z.im));
  }
  const double den = c2 + cosh(2. * z.im);
  return C(s2 / den, sinh(2. * z.im) / den);
}

// ---------------------------------------------------------------------------
// geometry helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ void rotate3(const xrt_hip_rotation& R, double& x, double& y,
                                        double& z) {
  for (int i = 0; i < R.n; ++i) {
    const double c = R.cosa[i], s = R.sina[i];
    const int ax = R.axis[i];
    if (ax == 2) {  // rotate_z, _rotate.py:17-20
      const double xn = c * x - s * y, yn = s * x + c * y;
      x = xn;
      y = yn;
    } else if (ax == 1) {  // rotate_y, :11-14
      const double xn = c * x + s * z, zn = -s * x + c * z;
      x = xn;
      z = zn;
    } else {  // rotate_x, :5-8
      const double yn = c * y - s * z, zn = s * y + c * z;
      y = yn;
      z = zn;
    }
  }
}

__device__ __forceinline__ bool entering(const xrt_hip_pass& P, int st) {
  return P.good_mode == 0 ? (st > 0) : (st == 1 || st == 2);
}

// direction of ray i in the true local frame (beamline.py:243-252, reflect.py:617-629)
__device__ __forceinline__ void local_dir(const xrt_hip_pass& P, double& a, double& b,
                                          double& c) {
  if (P.in_is_global && P.sin_az != 0.) {
    const double an = P.cos_az * a - P.sin_az * b, bn = P.sin_az * a + P.cos_az * b;
    a = an;
    b = bn;
  }
  rotate3(P.to_local, a, b, c);
}

__device__ __forceinline__ void local_pos(const xrt_hip_pass& P, double& x, double& y,
                                          double& z) {
  if (P.in_is_global) {
    x = x - P.center[0];
    y = y - P.center[1];
    z = z - P.center[2];
    if (P.sin_az != 0.) {
      const double xn = P.cos_az * x - P.sin_az * y, yn = P.sin_az * x + P.cos_az * y;
      x = xn;
      y = yn;
    }
  }
  rotate3(P.to_local, x, y, z);
  x -= P.shift[0];
  y -= P.shift[1];
  z -= P.shift[2];
}

// a / b for a divisor known in advance, y = RN(1/b) from the host: the tail of
// the IEEE division sequence (quotient estimate, exact remainder by fma, one
// correction) - 3 instructions instead of ~14, same correctly rounded result for
// normal-range operands (tests/test_gpu_math.py checks it against '/').
__device__ __forceinline__ double div_const(double a, double b, double y) {
  const double q0 = a * y;
  const double r = fma_(-q0, b, a);
  return fma_(r, y, q0);
}

// correctly rounded sqrt for x in {0} U [2^-60, 2^60] without the library's
// range scaling (the toroid radicand 1 - (x/r)^2 is 0 or >= 2^-53)
__device__ __forceinline__ double sqrt_unit(double x) {
  double dummy;
  return x == 0. ? 0. : sqrt_rn_rinv(x, dummy);
}

// numpy's floor_divide / remainder for doubles (npy_divmod): the facet
// bookkeeping of the blazed grating is written with `//` and `%`
__device__ __forceinline__ void np_divmod(double a, double b, double& fdiv, double& mod) {
  mod = fmod(a, b);
  double div = (a - mod) / b;
  if (mod != 0.) {
    if ((b < 0.) != (mod < 0.)) {
      mod += b;
      div -= 1.0;
    }
  } else {
    mod = copysign(0., b);
  }
  if (div != 0.) {
    fdiv = floor(div);
    if (div - fdiv > 0.5) fdiv += 1.0;
  } else {
    fdiv = copysign(0., a / b);
  }
}

// blazed grating, gratings.py:461-490: is (x, y) on the blaze facet (the one
// facing the source) of its groove?
__device__ __forceinline__ bool blazed_front(const xrt_hip_pass& P, double y, double& y1,
                                             double& yL) {
  const double rho_1 = P.surf_p[0];
  double fdiv;
  np_divmod(y, rho_1, fdiv, yL);
  const double y0 = fdiv * rho_1;
  y1 = y0 + rho_1;
  const double yC = (y1 - y0) / P.surf_p[7];
  return yL > yC;
}

// ---------------------------------------------------------------------------
// Compile-time specialisation. K = Spec<F, SK, MK, PLAIN>:
//   F      surface family (0 = flat / toroid / bent-flat, 1 = + blazed / conics)
//   SK, MK surface kind / material kind fixed at compile time, or -1 = read from
//          the pass / material records at run time
//   PLAIN  no grating equation, no asymmetric cut, intersection search on
// The generic kernel carries every branch (13 k instructions, 128 VGPRs); with the
// kinds known the common cases shrink to a quarter of that and ~90 VGPRs
// (measured on cfg2: fused kernel 0.84 -> 0.77 ms).
// ---------------------------------------------------------------------------
#ifndef XRT_LEAN_WAVES
#define XRT_LEAN_WAVES 4
#endif
// waves per SIMD of the kernels for layered materials (measured on 1e7 rays, W/Si x40 /
// Rh coating: 2 waves 3.36 / 1.71 ms, 3 waves 2.93 / 1.49, 4 waves 3.03 / 1.56)
#ifndef XRT_LAYERED_WAVES
#define XRT_LAYERED_WAVES 3
#endif
template <int F_, int SK_, int MK_, bool PLAIN_>
struct Spec {
  static constexpr int F = F_, SK = SK_, MK = MK_;
  static constexpr bool PLAIN = PLAIN_;
  // waves per SIMD the fused kernel is compiled for
  // (layered materials: the Parratt recursion keeps ~30 complex numbers per ray alive and
  // is compute bound -- 168 VGPRs, three waves)
  static constexpr int WAVES = MK_ == XRT_HIP_MAT_MULTILAYER ? XRT_LAYERED_WAVES
                               : (PLAIN_ && SK_ >= 0 && MK_ >= 0) ? XRT_LEAN_WAVES
                                                                  : REFLECT_FUSED_WAVES;
  // crystal known to be thick (crystal.py:571-584): the thin-crystal forms with their
  // complex exp / cos / sin / tan are not compiled in
  static constexpr bool XTHICK = false;
  // crystals given by their unit cell (structure code 2: up to four f1/f2 look-ups per ray)
  // are compiled in
  static constexpr bool XCELL = true;
  // zones and groove vectors per ray from the caller (general zone plate)
  static constexpr bool RAYG = false;
  // the height map of OE(figureError=...) is evaluated (xrt_hip_pass.fe_*): only the Figured
  // kernels below carry the spline code
  static constexpr bool FE = false;
};
// a thick (semi-infinite) crystal: what a DCM is made of
template <int SK_>
struct ThickXtal : Spec<0, SK_, XRT_HIP_MAT_CRYSTAL, false> {
  static constexpr bool XTHICK = true;
  // one-element lattices only (the launcher sends cell crystals to the other crystal
  // kernels): with the four-element look-ups compiled in, the fused DCM kernel spilled 194
  // SGPRs instead of 79 and lost 2 %
  static constexpr bool XCELL = false;
};
// Bragg crystals, and multilayers (geom_bragg set; a Coated mirror has it cleared): the
// direction comes from the grating equation with the batch's sign of beamInDotNormal
__host__ __device__ inline bool deflects_as_crystal(const xrt_hip_material& M) {
  return M.kind == XRT_HIP_MAT_CRYSTAL || (M.kind == XRT_HIP_MAT_MULTILAYER && M.geom_bragg);
}
// The general zone plate's kernels: only they read xrt_hip_pass.state_ray / g_ray_*. (In the
// generic kernels the two loads of a groove component -- from the caller's array or from
// the pass record -- were merged into one load through a selected pointer, the by-value pass
// record went to scratch for it, 1 KB per lane in the exact kernel, and EVERY pass paid 12 us
// more launch overhead.)
struct PerRayZones : Spec<0, -1, -1, false> {
  static constexpr bool RAYG = true;
};
// OE(figureError = ...): surface family F with the figure-error spline in find_dz and in the
// normal, surface and material kinds read at run time
template <int F_>
struct Figured : Spec<F_, -1, -1, false> {
  static constexpr bool FE = true;
  // (the 4 x 4 coefficient block and two sets of basis functions on top of the generic pass:
  // 22 VGPRs spilled at four waves per SIMD, none at three)
  static constexpr int WAVES = XRT_LAYERED_WAVES;
};
using Layered0 = Spec<0, -1, XRT_HIP_MAT_MULTILAYER, false>;
using Layered1 = Spec<1, -1, XRT_HIP_MAT_MULTILAYER, false>;
using Layered2 = Spec<2, -1, XRT_HIP_MAT_MULTILAYER, false>;
using Generic0 = Spec<0, -1, -1, false>;
using Generic1 = Spec<1, -1, -1, false>;
// Surface families: 0 = flat / toroid / bent flat (+ sagittal, bent and diced under a crystal);
// 1 = blazed grating, parametric conics, lenses, cone; 2 = bent-crystal shapes, diced elements,
// VFM / DualVFM under a non-crystal material; 3 = all of them (the utility kernels). Keeping
// family 2 out of family 1 keeps the parametric-mirror kernel at 128 VGPRs without spills (with
// everything in one kernel it spilled 126 VGPRs and kept the pass record in scratch).
using Generic2 = Spec<2, -1, -1, false>;
using GenericAll = Spec<3, -1, -1, false>;
#define PSURF(P) (K::SK >= 0 ? K::SK : (P).surf_kind)
#define MKIND(M) (K::MK >= 0 ? K::MK : (M).kind)
#define PGRATING(P) (K::PLAIN ? 0 : (P).grating)
#define PASYM(P) (K::PLAIN ? 0 : (P).asymmetric)
#define PNIS(P) (K::PLAIN ? 0 : (P).no_intersection_search)

// F = surface family, a compile-time switch: 0 = flat / toroid / bent-flat (the
// bulk ray-tracing kernels stay free of the code below), 1 = blazed grating and
// parametric ellipse (fmod / atan2 / sincos in the solve)
// kernels compiled for layered materials (Multilayer, Coated); no other kernel holds the
// recursion
template <class K>
__device__ __forceinline__ constexpr bool layered() {
  return K::MK == XRT_HIP_MAT_MULTILAYER;
}
// Surface kinds beyond the blazed grating are dispatched to the family-1 kernels -- except
// under a crystal, whose kernels are family 0 (bent analysers, the sagittal DCM crystal).
// Keeping them out of the other family-0 kernels keeps those lean: the generic exact kernel
// of a plain mirror pass went from 176 B to 1 KB of scratch (and the pass from 21 to 32 us of
// launch overhead) with the new kinds compiled in.
// A parameter of the pass record as an opaque value. `c ? P.a : P.b` on two plain loads is
// turned into ONE load through a selected address -- an address INTO the by-value record,
// which then has to be copied to scratch for the whole kernel (1 KB per lane, and 12 us of
// launch overhead per dispatch). Values that went through readfirstlane are selected as
// values.
__device__ __forceinline__ double pinned(double v) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)bits);
  const int hi = __builtin_amdgcn_readfirstlane((int)(bits >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// passes whose local beam may be left out (xrt_hip_reflect_pass_f64_dev with out_local NULL):
// not the layered kernels (since round 5 the crystal kernels, the fused DCM among them, take the
// test: a scalar compare per store)
template <class K>
__device__ __forceinline__ constexpr bool optional_local() {
#ifdef XRT_NO_OPTIONAL_LOCAL      /* A/B: what the test costs the hot kernel */
  return false;
#else
  return K::MK != XRT_HIP_MAT_MULTILAYER;
#endif
}
template <class K>
__device__ __forceinline__ constexpr bool wide_surfaces() {
  return K::F == 2 || K::F == 3;
}
template <class K>
__device__ __forceinline__ constexpr bool bent_crystal_surfaces() {
  return K::F == 2 || K::F == 3 || (K::F != 4 && K::MK == XRT_HIP_MAT_CRYSTAL);
}
// family 4: a user-defined surface. Only a unit compiled around the user's two functions
// (csrc/user_unit.hip.in defines XRT_USER_SURFACE and xrt_user::local_z / local_n before it
// includes this file) instantiates it.
template <class K>
__device__ __forceinline__ constexpr bool user_surface() {
  return K::F == 4;
}
template <class K>
__device__ __forceinline__ bool surf_is_param(const xrt_hip_pass& P) {
  return (K::F & 1) && PSURF(P) == XRT_HIP_SURF_ELLIPSE_PARAM;
}
template <class K>
__device__ __forceinline__ bool surf_is_blazed(const xrt_hip_pass& P) {
  return (K::F & 1) && PSURF(P) == XRT_HIP_SURF_BLAZED;
}

template <class K>
__device__ __forceinline__ bool surf_is_lens(const xrt_hip_pass& P) {
  return (K::F & 1) && PSURF(P) == XRT_HIP_SURF_PARABOLOID;
}
template <class K>
__device__ __forceinline__ bool surf_is_cone(const xrt_hip_pass& P) {
  return (K::F & 1) && PSURF(P) == XRT_HIP_SURF_CONE;
}
// the paraboloid of a refractive lens before its cut-off, refractive.py:396, 411
__device__ __forceinline__ double lens_parabola(const xrt_hip_pass& P, double& x, double y) {
  if (P.surf_p[4] != 0.) x = 0.;  // parabolic cylinder: local_z1(0, y), :613-617
  return (x * x + y * y) / P.surf_p[0];
}

// EllipticalMirrorParam, parametric.py:213-231. rotate_x(y, z, c, s) =
// (c y - s z, s y + c z) (_rotate.py:5-6)
__device__ __forceinline__ void ell_xyz_to_param(const xrt_hip_pass& P, double x, double y,
                                                 double z, double& s, double& phi, double& r) {
  const double yy = y - P.surf_p[0], zz = z - P.surf_p[1];
  const double cg = P.surf_p[2], sg = P.surf_p[3];
  const double yN = cg * yy - sg * zz;
  const double zN = sg * yy + cg * zz;
  s = yN;
  phi = atan2_np(x, zN);
  r = sqrt(x * x + zN * zN);
}

__device__ __forceinline__ void ell_param_to_xyz(const xrt_hip_pass& P, double s, double phi,
                                                 double r, double& x, double& y, double& z) {
  double sn, cs;
  sincos(phi, &sn, &cs);
  x = r * sn;
  const double zz = r * cs;
  const double cg = P.surf_p[2], msg = -P.surf_p[3];
  const double yN = cg * s - msg * zz;
  const double zN = msg * s + cg * zz;
  y = yN + P.surf_p[0];
  z = zN + P.surf_p[1];
}

__device__ __forceinline__ double ell_local_r(const xrt_hip_pass& P, double s, double phi) {
  const double A = P.surf_p[4], B = P.surf_p[5];
  const int conic = (int)P.surf_p[8];
  s = P.surf_p[9] + s;   // capillaries measure s from their middle (ctd), parametric.py:878, 975
  double r;
  if (conic == 3) {  // paraboloid capillary, :780-781: A = s0, B = focus
    r = 2. * sqrt((A - s) * B);
  } else if (conic == 1) {  // parabola, parametric.py:450-453: A = parabParam
    double r2 = A * s + A * A;
    if (r2 < 0.) r2 = 0.;
    r = 2. * sqrt(r2);
  } else if (conic == 2) {  // hyperbola, :690-691
    r = B * sqrt(fabs((s * s) / (A * A) - 1.));
  } else {
    r = B * sqrt(fabs(1. - (s * s) / (A * A)));
  }
  if (P.surf_p[6] != 0.) r /= fabs(cos_np(phi));
  if (P.surf_p[7] != 0.) return r;
  if (conic == 2) return fabs(phi) < kPI / 2. ? r : 1e20;
  return fabs(phi) > kPI / 2. ? r : 1e20;
}

// Diced elements (oes/bragg.py:8-101, 345-375): the facet a point lies on -- its centre
// (numpy's round: half to even), the point in facet coordinates, the height of the base
// surface and the base normals at the centre (cn[0..2] atomic planes, cn[3..5] surface).
struct Facet {
  double fx, fy, cz;
  double cn[6];
};
__device__ __forceinline__ Facet diced_facet(const xrt_hip_pass& P, double x, double y) {
  Facet f;
  const double cx = rint(x / P.surf_p[7]) * P.surf_p[7];
  const double cy = rint(y / P.surf_p[8]) * P.surf_p[8];
  f.fx = x - cx;
  f.fy = y - cy;
  if (P.surf_p[0] == 0.) {  // flat
    f.cz = 0.;
    f.cn[0] = f.cn[3] = 0.;
    f.cn[1] = f.cn[4] = 0.;
    f.cn[2] = f.cn[5] = 1.;
    return f;
  }
  const double Rm = P.surf_p[2], Rs = P.surf_p[3];
  const double root = sqrt(Rm * Rm - cy * cy);
  // JohannToroid.local_z / local_n_toroid at the centre, bragg.py:236-269
  const double z = (Rm - Rs) - root;
  f.cz = (sqrt(z * z - cx * cx) / fabs(z)) * z + Rs;
  const double b = -cy / Rm, c0 = root / Rm;
  const double r = Rs - (Rm - root);
  const double cosang = sqrt(r * r - cx * cx) / r, sinang = -cx / r;
  f.cn[3] = sinang * c0;
  f.cn[4] = b;
  f.cn[5] = cosang * c0;
  if (P.surf_p[1] == 0.) {  // Johann: the planes follow the base surface
    f.cn[0] = f.cn[3];
    f.cn[1] = f.cn[4];
    f.cn[2] = f.cn[5];
    if (P.surf_p[6] != 0.) {   // (alpha of the base class, :254-266)
      const double ca = P.surf_p[4], sa = P.surf_p[5];
      f.cn[1] = ca * b + sa * c0;
      const double cA = -sa * b + ca * c0;
      f.cn[0] = sinang * cA;
      f.cn[2] = cosang * cA;
    }
  } else {  // Johansson, :279-295
    double pb = -cy, pc = root + Rm;
    const double norm = sqrt(pb * pb + pc * pc);
    pb /= norm;
    pc /= norm;
    if (P.surf_p[6] != 0.) {
      const double ca = P.surf_p[4], sa = P.surf_p[5];
      const double b1 = ca * pb + sa * pc;
      pc = -sa * pb + ca * pc;
      pb = b1;
    }
    double pa = sinang * pc;
    pc = cosang * pc;
    if (P.surf_p[6] != 0.) {
      const double a1 = cosang * pa + sinang * pc;
      pc = -sinang * pa + cosang * pc;
      pa = a1;
    }
    f.cn[0] = pa;
    f.cn[1] = pb;
    f.cn[2] = pc;
  }
  return f;
}

// surface height, oes/base.py:675-679 (flat), oes/__init__.py:398-401 (toroid)
// ---------------------------------------------------------------------------
// Figure error: scipy's RectBivariateSpline.ev (FITPACK bispev / parder) on the device.
// fe_interval = fpbisp's argument clamp and knot search (a guess by scaling, exact for the
// uniform interior of a linspace grid, corrected against the knots), fe_basis = fpbspl's
// recurrence for the K + 1 B-splines that are not zero on the interval, fe_spline_k = fpbisp's
// double sum in its order.
// What bounds it is the number of scattered loads -- every lane is somewhere else on the map,
// a gather of 64 addresses occupies the CU's address unit for 16 cycles, and the root search
// evaluates the map ~9 times per ray one after the other. The first version read 12 knots and
// 16 coefficients per evaluation (3.2 ms per 1e7 rays). Now:
//   * knots are COMPUTED where the map's grid is a numpy linspace (every generated map): the
//     not-a-knot sequence is x[0] four times, x[2] .. x[N - 3], x[N - 1] four times with
//     x[j] = j * step + lo in linspace's own two roundings -- the same bits, no load (FeKnots;
//     the host checks the knots against the formula before it sets fe_grid; maps read from a
//     file keep the loads);
//   * coefficients come as PAIRS of rows, pc[i][j] = (c[i][j], c[i + 1][j]): a 4 x 4 block is
//     8 loads of 16 B instead of 16 of 8 B.
// ---------------------------------------------------------------------------
struct FeKnots {
  const double* t;      // the knots (used when !grid)
  int n;                // their number
  int off;              // index shift: the derivative axes drop the first knot (off = 1)
  int grid;             // computed: t[i] = x[clamp(i - 2)] of a linspace
  double lo, step, hi;  // x[0], linspace's step, x[N - 1]; N = n0 - 4 nodes (n0 = n + 2 off)
  double inv[3];        // 1 / (j step), j = 1..3: the knot differences of the uniform interior
  // the 2 K knots fpbspl reads on interval l all lie in the uniform interior
  __device__ __forceinline__ bool inner(int l, int K) const {
    return grid && l + 1 - K + off >= 4 && l + K + off < n + 2 * off - 4;
  }
  __device__ __forceinline__ double at(int i) const {
    if (!grid) return t[i];
    const int i0 = i + off, n0 = n + 2 * off;
    if (i0 < 4) return lo;
    if (i0 >= n0 - 4) return hi;
    return (double)(i0 - 2) * step + lo;
  }
};
template <int K>
__device__ __forceinline__ int fe_interval(const FeKnots& T, double& arg) {
  const double tb = T.at(K), te = T.at(T.n - K - 1);
  if (arg < tb) arg = tb;
  if (arg > te) arg = te;
  const int last = T.n - K - 2;                 // l <= last: t[l] <= arg <= t[l + 1]
  int l = K + (int)((arg - tb) * frcp(te - tb) * (double)(last - K + 1));
  l = l < K ? K : (l > last ? last : l);
  while (l > K && arg < T.at(l)) --l;
  while (l < last && arg >= T.at(l + 1)) ++l;
  return l;
}
template <int K>
__device__ __forceinline__ void fe_basis(const FeKnots& T, double x, int l, double (&h)[K + 1]) {
  // (compile-time degree: the recurrence unrolls and h stays in registers)
  double kn[2 * K > 0 ? 2 * K : 1];
#pragma unroll
  for (int i = 0; i < 2 * K; ++i) kn[i] = T.at(l + 1 - K + i);      // t[l + 1 - K .. l + K]
  double hh[K + 1];
  h[0] = 1.;
#pragma unroll
  for (int j = 1; j <= K; ++j) {
#pragma unroll
    for (int i = 0; i < j; ++i) hh[i] = h[i];
    h[0] = 0.;
#pragma unroll
    for (int i = 0; i < j; ++i) {
      const double hi = kn[K + i], lo = kn[K + i - j];     // t[l + 1 + i], t[l + 1 + i - j]
      const double f = hh[i] / (hi - lo);
      h[i] = h[i] + f * (hi - x);
      h[i + 1] = f * (x - lo);
    }
  }
}
// One axis of a map on a linspace grid, everything from the ray's coordinate: clamp, interval by
// scaling, the 2 K knots from ONE integer conversion (x[j] = j step + lo), the recurrence with
// products by 1 / (j step) where FITPACK divides by a difference of j knots (the quotient by the
// rounded difference and the product differ in the last bit or two -- of nanometres). false:
// the interval touches the ends of the knot sequence, or the scaled guess missed it by one --
// the wave then takes fe_interval / fe_basis. (Through T.at() each of the 20 knots an
// evaluation looks at cost a conversion, two comparisons and two selects: 240 of its 440
// issue slots; this form takes ~60 per axis.)
template <int K>
__device__ __forceinline__ bool fe_axis_grid(const FeKnots& T, double& arg, int& l,
                                             double (&h)[K + 1]) {
  if (arg < T.lo) arg = T.lo;
  if (arg > T.hi) arg = T.hi;
  const int last = T.n - K - 2;
  // (node j = floor((arg - lo) / step) lies at knot j + 2 of the full sequence: two nodes are
  // not knots of a not-a-knot spline)
  l = 2 - T.off + (int)((arg - T.lo) * T.inv[0]);
  l = l < K ? K : (l > last ? last : l);
  const double d0 = (double)(l + T.off - K - 1);
  double kn[2 * K > 0 ? 2 * K : 2];
#pragma unroll
  for (int i = 0; i < 2 * K; ++i) kn[i] = (d0 + (double)i) * T.step + T.lo;
  if (K == 0) {
    kn[0] = d0 * T.step + T.lo;
    kn[1] = (d0 + 1.) * T.step + T.lo;
  }
  constexpr int LO = K > 0 ? K - 1 : 0;
  const bool ok = T.inner(l, K > 0 ? K : 1) && arg >= kn[LO] && arg < kn[LO + 1];
  double hh[K + 1];
  h[0] = 1.;
#pragma unroll
  for (int j = 1; j <= K; ++j) {
#pragma unroll
    for (int i = 0; i < j; ++i) hh[i] = h[i];
    h[0] = 0.;
#pragma unroll
    for (int i = 0; i < j; ++i) {
      const double f = hh[i] * T.inv[j - 1];
      h[i] = h[i] + f * (kn[K + i] - arg);
      h[i + 1] = f * (arg - kn[K + i - j]);
    }
  }
  return ok;
}
// The same for a wave with lanes at the ENDS of the knot sequence (a ray that misses the mirror
// is clamped to the edge of the map; one such lane in a wave is common: 2 % of missing rays are
// 78 % of the waves): still no loads -- every knot from its index by two comparisons, the
// reciprocal of each knot difference by v_rcp + two Newton steps (differences there are 1..5
// steps, never zero between t[l + 1 + i] and t[l + 1 + i - j]) -- and the interval corrected
// against the knots like fe_interval does.
template <int K>
__device__ __forceinline__ void fe_axis_ends(const FeKnots& T, double& arg, int& l,
                                             double (&h)[K + 1]) {
  if (arg < T.lo) arg = T.lo;
  if (arg > T.hi) arg = T.hi;
  const int last = T.n - K - 2, n0 = T.n + 2 * T.off;
  l = 2 - T.off + (int)((arg - T.lo) * T.inv[0]);
  l = l < K ? K : (l > last ? last : l);
  constexpr int NK = K > 0 ? 2 * K : 2, LO = K > 0 ? K - 1 : 0, K1 = K > 0 ? K : 1;
  double kn[NK];
  for (int tries = 0; tries < 4; ++tries) {
    const int i0 = l + 1 - K1 + T.off;                  // index of kn[0] in the full sequence
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int q = i0 + i;
      const double inner = (double)(q - 2) * T.step + T.lo;
      kn[i] = q < 4 ? T.lo : (q >= n0 - 4 ? T.hi : inner);
    }
    if (l > K && arg < kn[LO])
      --l;
    else if (l < last && arg >= kn[LO + 1])
      ++l;
    else
      break;
  }
  double hh[K + 1];
  h[0] = 1.;
#pragma unroll
  for (int j = 1; j <= K; ++j) {
#pragma unroll
    for (int i = 0; i < j; ++i) hh[i] = h[i];
    h[0] = 0.;
#pragma unroll
    for (int i = 0; i < j; ++i) {
      const double hi = kn[K + i], lo = kn[K + i - j];
      const double f = hh[i] * frcp(hi - lo);
      h[i] = h[i] + f * (hi - arg);
      h[i + 1] = f * (arg - lo);
    }
  }
}
// the spline of degrees (KU, KV) on knots U, V with paired coefficient rows pc (row = u) at (u, v)
template <int KU, int KV>
__device__ __forceinline__ double fe_spline_k(const FeKnots& U, const FeKnots& V,
                                              const double* __restrict__ pc, double u, double v) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  int lu, lv;
  double hu[KU + 1], hv[KV + 1];
  bool have = false;
  if (U.grid && V.grid) {
    const bool oku = fe_axis_grid<KU>(U, u, lu, hu);
    const bool okv = fe_axis_grid<KV>(V, v, lv, hv);
    if (__builtin_amdgcn_ballot_w64(!(oku && okv)) != 0ull) {    // (the whole wave, or none of it)
      fe_axis_ends<KU>(U, u, lu, hu);
      fe_axis_ends<KV>(V, v, lv, hv);
    }
    have = true;
  }
  if (!have) {
    lu = fe_interval<KU>(U, u);
    lv = fe_interval<KV>(V, v);
  }
  const int ncv = V.n - KV - 1;
  const v2d* row = reinterpret_cast<const v2d*>(pc) + (int64_t)(lu - KU) * ncv + (lv - KV);
  // (requested before the basis is worked out) rows i and i + 1 arrive together
  double cf[KU + 2][KV + 1];
#pragma unroll
  for (int i = 0; i <= KU; i += 2) {
#pragma unroll
    for (int j = 0; j <= KV; ++j) {
      const v2d q = row[(int64_t)i * ncv + j];
      cf[i][j] = q.x;
      cf[i + 1][j] = q.y;
    }
  }
  if (!have) {
    fe_basis<KU>(U, u, lu, hu);
    fe_basis<KV>(V, v, lv, hv);
  }
  double sp = 0.;
#pragma unroll
  for (int i = 0; i <= KU; ++i) {
#pragma unroll
    for (int j = 0; j <= KV; ++j) sp += cf[i][j] * hu[i] * hv[j];
  }
  return sp;
}
// DU, DV: 1 = the partial derivative along that axis (one degree less, FITPACK's parder)
template <int DU, int DV>
__device__ __forceinline__ double fe_spline(int k, const FeKnots& U, const FeKnots& V,
                                            const double* __restrict__ pc, double u, double v) {
  if (k == 3) return fe_spline_k<3 - DU, 3 - DV>(U, V, pc, u, v);
  if (k == 2) return fe_spline_k<2 - DU, 2 - DV>(U, V, pc, u, v);
  return fe_spline_k<1 - DU, 1 - DV>(U, V, pc, u, v);
}
// the knots of axis 0 (y) / 1 (x) of the pass's map; deriv: without the first and the last one
__device__ __forceinline__ FeKnots fe_knots(const xrt_hip_pass& P, int axis, int deriv) {
  FeKnots T;
  T.t = (axis ? P.fe_tx : P.fe_ty) + deriv;
  T.n = (axis ? P.fe_ntx : P.fe_nty) - 2 * deriv;
  T.off = deriv;
  T.grid = P.fe_grid[axis];
  T.lo = P.fe_lo[axis];
  T.step = P.fe_step[axis];
  T.hi = P.fe_hi[axis];
  T.inv[0] = P.fe_inv[axis][0];
  T.inv[1] = P.fe_inv[axis][1];
  T.inv[2] = P.fe_inv[axis][2];
  return T;
}
// local_z_distorted, figure_error.py:214-235 [mm]
__device__ __forceinline__ double figure_height(const xrt_hip_pass& P, double x, double y) {
#ifdef XRT_PROBE_FE_NULL          /* A/B: the Figured pass without its spline evaluations */
  return 0. * (x + y);
#endif
  return fe_spline<0, 0>(P.fe_k, fe_knots(P, 0, 0), fe_knots(P, 1, 0), P.fe_c,
                         y + P.fe_shift[1], x + P.fe_shift[0]) * 1e-6;
}
// local_n_distorted -> [d_pitch, d_roll] (figure_error.py:237-265) applied to the surface
// normal as reflect.py:767-775 does: rotate_x by d_pitch, then rotate_y by d_roll
__device__ __forceinline__ void figure_turn_normal(const xrt_hip_pass& P, double x, double y,
                                                   double& nx, double& ny, double& nz) {
  const double u = y + P.fe_shift[1], v = x + P.fe_shift[0];
  const double a = fe_spline<0, 1>(P.fe_k, fe_knots(P, 0, 0), fe_knots(P, 1, 1), P.fe_cx, u, v) * 1e-6;
  const double b = fe_spline<1, 0>(P.fe_k, fe_knots(P, 0, 1), fe_knots(P, 1, 0), P.fe_cy, u, v) * 1e-6;
  double sX, cX, sY, cY;
  sincos(atan(b), &sX, &cX);
  sincos(-atan(a), &sY, &cY);
  const double y1 = cX * ny - sX * nz, z1 = sX * ny + cX * nz;      // _rotate.py:5-8
  const double x2 = cY * nx + sY * z1, z2 = -sY * nx + cY * z1;     // _rotate.py:11-14
  nx = x2;
  ny = y1;
  nz = z2;
}

template <class K>
__device__ __forceinline__ double surf_z(const xrt_hip_pass& P, double x, double y) {
#ifdef XRT_USER_SURFACE
  if (user_surface<K>()) return xrt_user::local_z(x, y, P.surf_p);
#endif
  if (PSURF(P) == XRT_HIP_SURF_TOROID) {
    const double R = P.surf_p[0], r = P.surf_p[1];
    double q, h;
    const double yy = y * y;
    if (P.surf_p[4] != 0.) {  // reciprocals usable (finite, normal radii)
      q = div_const(x, r, P.surf_p[3]);
      h = div_const(yy * 0.5, R, P.surf_p[2]);
    } else {
      q = x / r;
      h = yy / 2.0 / R;
    }
    double rx = 1. - q * q;
    if (rx < 0.) rx = 0.;
    return h + r * (1. - sqrt_unit(rx));
  }
  if (PSURF(P) == XRT_HIP_SURF_BENTFLAT) {  // (y**2 - limPhysY[0]**2) / 2.0 / R
    const double num = (y * y - P.surf_p[1]) * 0.5;
    return P.surf_p[4] != 0. ? div_const(num, P.surf_p[0], P.surf_p[2]) : num / P.surf_p[0];
  }
  if (surf_is_blazed<K>(P)) {  // gratings.py:475-480
    double y1, yL;
    return blazed_front(P, y, y1, yL) ? -(y1 - y) * pinned(P.surf_p[1])
                                      : -yL * pinned(P.surf_p[2]);
  }
  if (surf_is_lens<K>(P)) {  // refractive.py:394-399
    const double z = lens_parabola(P, x, y);
    return P.surf_p[3] != 0. && z > P.surf_p[2] ? P.surf_p[2] : z;
  }
  if (PSURF(P) == XRT_HIP_SURF_SAGITTAL)  // oes/__init__.py:655-656 (crystals: family 0 too)
    return P.surf_p[0] - sqrt(P.surf_p[1] - x * x);
  if (wide_surfaces<K>() && PSURF(P) == XRT_HIP_SURF_VFM) {  // oes/__init__.py:458-467
    double z = P.surf_p[0] - sqrt(P.surf_p[1] - x * x);
    if (z > P.surf_p[2]) z = P.surf_p[2];
    return z + (y * y - P.surf_p[3]) / 2.0 / P.surf_p[4];
  }
  if (wide_surfaces<K>() && PSURF(P) == XRT_HIP_SURF_DUALVFM) {  // oes/__init__.py:532-550
    const bool left = x < 0.;
    const double u = x - (left ? pinned(P.surf_p[5]) : pinned(P.surf_p[2]));
    double z = (left ? pinned(P.surf_p[3]) : pinned(P.surf_p[0])) -
               sqrt((left ? pinned(P.surf_p[4]) : pinned(P.surf_p[1])) - u * u);
    if (isnan(z) || z > 0.) z = 0.;
    return z + (y * y - P.surf_p[6]) / 2.0 / P.surf_p[7];
  }
  if (bent_crystal_surfaces<K>() && PSURF(P) == XRT_HIP_SURF_DICED) {  // bragg.py:52-65
    const Facet f = diced_facet(P, x, y);
    // Johansson facets are ground to the meridional radius (:365-366)
    const double dz = P.surf_p[1] == 1. ? f.fy * f.fy / 2.0 / P.surf_p[2] : 0.;
    return f.cz + ((dz - f.cn[3] * f.fx) - f.cn[4] * f.fy) / f.cn[5];
  }
  if (bent_crystal_surfaces<K>() && PSURF(P) == XRT_HIP_SURF_BENT_BRAGG) {  // oes/bragg.py:138-144, 236-241
    const double Rm = P.surf_p[2], Rs = P.surf_p[3];
    if (P.surf_p[0] == 1.) return y * y / 2.0 / Rm;
    if (P.surf_p[0] == 3.) return Rm - sqrt((Rm * Rm - x * x) - y * y);   // laue.py:488-489
    if (P.surf_p[0] == 4.) return (x * x + y * y) / 2.0 / Rm;             // :491
    if (P.surf_p[0] == 5.) return 0.5 * (x * x) / Rs + 0.5 * (y * y) / Rm;   // BentLaue2D, :365
    const double root = sqrt(Rm * Rm - y * y);
    if (P.surf_p[0] == 0.) return Rm - root;
    // toroid: the meridional circle turned about the sagittal axis
    const double z = (Rm - Rs) - root;
    const double cosangle = sqrt(z * z - x * x) / fabs(z);
    return cosangle * z + Rs;
  }
  if (surf_is_cone<K>(P)) {  // oes/__init__.py:623-627
    const double u = y - P.surf_p[0];
    const double root = sqrt(P.surf_p[1] * (u * u) - P.surf_p[2] * (x * x));
    return P.surf_p[3] * u - P.surf_p[4] * root;
  }
  return 0.;
}

// find_dz, oes/base.py:801-846
template <class K>
__device__ __forceinline__ double find_dz(const xrt_hip_pass& P, double t, double x0,
                                          double y0, double z0, double a, double b,
                                          double c, double& x, double& y, double& z) {
  x = x0 + a * t;
  y = y0 + b * t;
  z = z0 + c * t;
  if (surf_is_param<K>(P)) {  // base.py:822-841: (x, y, z) become (s, phi, r), diffSign = -1
    double sp, phi, rr;
    ell_xyz_to_param(P, x, y, z, sp, phi, rr);
    x = sp;
    y = phi;
    z = rr;
    double s = ell_local_r(P, sp, phi);
    if (isnan(s)) s = 0.;
    return (z - s) * -1. * (double)P.invert_normal;
  }
  double s = surf_z<K>(P, x, y);
  if constexpr (K::FE) {                  // base.py:826-830: surf += z_distorted
    if (P.fe_c) s += figure_height(P, x, y);
  }
  if (isnan(s)) s = 0.;
  return (z - s) * (double)P.invert_normal;
}

// _set_t + clamp, oes/base.py:1231-1245, 1275
__device__ __forceinline__ void bracket(const xrt_hip_pass& P, int axis, int positive,
                                        double x, double y, double z, double a, double b,
                                        double c, double& tMin, double& tMax) {
  double limMin, limMax, xyz, abc;
  if (axis == 0) {
    limMin = P.phys_x[0] > -INFINITY ? P.phys_x[0] : -kMaxHalfSize;
    limMax = P.phys_x[1] < INFINITY ? P.phys_x[1] : kMaxHalfSize;
    xyz = x;
    abc = a;
  } else if (axis == 1) {
    limMin = P.phys_y[0] > -INFINITY ? P.phys_y[0] : -kMaxHalfSize;
    limMax = P.phys_y[1] < INFINITY ? P.phys_y[1] : kMaxHalfSize;
    xyz = y;
    abc = b;
  } else {
    limMin = -kMaxDepth;
    limMax = kMaxDepth;
    xyz = z;
    abc = c;
  }
  if (positive) {
    tMin = (limMin - xyz) / abc - kDt;
    tMax = (limMax - xyz) / abc + kDt;
  } else {
    tMin = (limMax - xyz) / abc - kDt;
    tMax = (limMin - xyz) / abc + kDt;
  }
  if (tMin < -1e6 * kZEps) tMin = -1e6 * kZEps;
}

__device__ __forceinline__ int sgn(double v) { return (v > 0.) - (v < 0.); }
// sgn(a) == sgn(b) for two numbers that are not NaN (false if either is)
__device__ __forceinline__ bool same_sign(double a, double b) {
  const bool ordered = !(a != a) && !(b != b);
  const bool az = a == 0., bz = b == 0.;
  const bool bits = (__double2hiint(a) ^ __double2hiint(b)) >= 0;
  return ordered && (az == bz) && (az || bits);
}

// ---------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------
template <class T, class F>
__device__ __forceinline__ T block_reduce(T v, F f, T* lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = f(v, __shfl_xor(v, off));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  T r = lds[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = f(r, lds[w]);
  return r;
}

__device__ __forceinline__ void atomic_min_double(double* p, double v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  unsigned long long old = *q;
  while (v < __longlong_as_double((long long)old)) {
    const unsigned long long prev = atomicCAS(q, old, (unsigned long long)__double_as_longlong(v));
    if (prev == old) break;
    old = prev;
  }
}
__device__ __forceinline__ void atomic_max_double(double* p, double v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  unsigned long long old = *q;
  while (v > __longlong_as_double((long long)old)) {
    const unsigned long long prev = atomicCAS(q, old, (unsigned long long)__double_as_longlong(v));
    if (prev == old) break;
    old = prev;
  }
}

// Loads that another workgroup of the SAME launch may have written (reflect_exact's
// phases): agent scope = served by L2, past this CU's vector L1 and the scalar cache,
// neither of which another CU's stores ever refresh.
__device__ __forceinline__ double ld_agent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(
      reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
      __HIP_MEMORY_SCOPE_AGENT));
}

// ---------------------------------------------------------------------------
// K0 / K1 / decide / K2. The statistics passes and their folds are device functions:
// they run as phases of reflect_exact (one launch, grid barriers in between); blockIdx /
// gridDim are that kernel's.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gstat_reset(GStat* g, int redo) {
  g->optimistic = 0;
  g->redo = redo;
  g->maxa = 0.;
  g->maxb = 0.;
  g->maxc = 0.;
  g->first_good = ~0ull;
  g->n_enter = 0;
  g->n_main = 0;
  g->axis = 1;
  g->positive = 1;
  g->t1min = INFINITY;
  g->t2max = -INFINITY;
  g->maxdz1 = 0.;
  g->maxdz2 = 0.;
  g->bracket_valid = 0;
  g->any_neg = 0;
  g->any_pos = 0;
  g->n_good1 = 0;
  g->sum_bdn = 0.;
  g->emin = -INFINITY;
  g->emax = INFINITY;
  for (int e = 0; e < XRT_HIP_MAX_ELEM; ++e) {
    g->tab_lo[e] = 0;
    g->tab_hi[e] = 0x7fffffff;
  }
  g->win_lo = -INFINITY;
  g->win_hi = INFINITY;
  g->tab_fast = nullptr;
}

__device__ __forceinline__ void stats_dir_body(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                               double* __restrict__ part) {
  // two-level reduction: every block writes one 64-byte partial record, a
  // one-block kernel folds them (no same-address atomics, deterministic)
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  double ma = 0., mb = 0., mc = 0., emin = INFINITY, emax = -INFINITY;
  unsigned long long first = ~0ull, nent = 0, nmain = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < in.n; i += stride) {
    // all five loads are issued before the state is looked at: one memory round
    // trip per iteration instead of up to three dependent ones
    const int st = in.state[i];
    const double E = in.E[i];
    double a = in.a[i], b = in.b[i], c = in.c[i];
    if (entering(P, st)) {
      if ((unsigned long long)i < first) first = (unsigned long long)i;
      ++nent;
      emin = E < emin ? E : emin;
      emax = E > emax ? E : emax;
      if (st == 1) {  // mainPartForBracketing, reflect.py:644
        local_dir(P, a, b, c);
        ma = fmax(ma, fabs(a));
        mb = fmax(mb, fabs(b));
        mc = fmax(mc, fabs(c));
        ++nmain;
      }
    }
  }
  auto fmaxd = [](double u, double v) { return u > v ? u : v; };
  auto fminu = [](unsigned long long u, unsigned long long v) { return u < v ? u : v; };
  auto faddu = [](unsigned long long u, unsigned long long v) { return u + v; };
  ma = block_reduce(ma, fmaxd, lds_d);
  mb = block_reduce(mb, fmaxd, lds_d);
  mc = block_reduce(mc, fmaxd, lds_d);
  first = block_reduce(first, fminu, lds_u);
  nent = block_reduce(nent, faddu, lds_u);
  nmain = block_reduce(nmain, faddu, lds_u);
  auto fmind = [](double u, double v) { return u < v ? u : v; };
  emin = block_reduce(emin, fmind, lds_d);
  emax = block_reduce(emax, fmaxd, lds_d);
  if (threadIdx.x == 0) {
    double* o = part + (int64_t)blockIdx.x * 8;
    o[0] = ma;
    o[1] = mb;
    o[2] = mc;
    o[3] = __longlong_as_double((long long)first);
    o[4] = (double)nent;
    o[5] = (double)nmain;
    o[6] = emin;
    o[7] = emax;
  }
}

// The TabFast records of a material around energy E0 (one thread). ub[e] = upper_bound of
// E0 in element e's table. false if a table is too short there or a slope is not finite
// (equal neighbouring energies at an absorption edge): the rays then search the tables.
__device__ __forceinline__ bool tab_fast_build(const xrt_hip_material& M, const int* ub,
                                               TabFast* tf) {
  if (M.kind == XRT_HIP_MAT_NONE || M.n_fixed) return false;
  for (int e = 0; e < M.nelem; ++e) {
    const int n = M.tab_n[e], j0 = ub[e] - 2;
    if (j0 < 0 || j0 + 3 > n - 2) return false;    // (j >= n - 1 is np.interp's end rule)
    const double* tE = M.tab_E[e];
    TabFast t;
    for (int k = 0; k < 4; ++k) t.x[k] = tE[j0 + k];
    for (int k = 0; k < 3; ++k) {
      const double dx = t.x[k + 1] - t.x[k];
      t.f1[k] = M.tab_f1[e][j0 + k];
      t.f2[k] = M.tab_f2[e][j0 + k];
      t.s1[k] = (M.tab_f1[e][j0 + k + 1] - t.f1[k]) / dx;
      t.s2[k] = (M.tab_f2[e][j0 + k + 1] - t.f2[k]) / dx;
      if (!(fabs(t.s1[k]) < INFINITY) || !(fabs(t.s2[k]) < INFINITY)) return false;
    }
    tf[e] = t;
  }
  return true;
}
// where they live: behind the report slots in the partial-record area of the pass
__host__ __device__ inline TabFast* tab_fast_of(void* part) {
  return reinterpret_cast<TabFast*>(reinterpret_cast<OptStat*>(part) + REFLECT_OPT_SLOTS);
}

// f1/f2 table window of a batch: upper_bound(E table, emin / emax) per element
__device__ __forceinline__ void table_windows(const xrt_hip_material& M, double emin,
                                              double emax, GStat* g) {
  g->emin = emin;
  g->emax = emax;
  if (M.kind != XRT_HIP_MAT_NONE && emin <= emax) {
    for (int e = 0; e < M.nelem; ++e) {
      const double* tE = M.tab_E[e];
      const int n = M.tab_n[e];
      int lo = 0, hi = n;
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (emin >= tE[mid]) lo = mid + 1; else hi = mid;
      }
      g->tab_lo[e] = lo;
      hi = n;
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (emax >= tE[mid]) lo = mid + 1; else hi = mid;
      }
      g->tab_hi[e] = lo;
    }
  }
}

// the same by the whole block: upper_bound(x) in a sorted table = number of entries <= x,
// counted in parallel (a one-thread binary search is 20 dependent memory round trips)
// (ub_lo / ub_hi: the same numbers in every thread's registers, so that the thread that goes on
// with them does not read them back from memory)
__device__ __forceinline__ void table_windows_block(const xrt_hip_material& M, double emin,
                                                    double emax, GStat* g,
                                                    unsigned long long* lds_u,
                                                    int (&ub_lo)[XRT_HIP_MAX_ELEM],
                                                    int (&ub_hi)[XRT_HIP_MAX_ELEM]) {
#pragma unroll
  for (int e = 0; e < XRT_HIP_MAX_ELEM; ++e) ub_lo[e] = ub_hi[e] = 0;
  if (M.kind == XRT_HIP_MAT_NONE || !(emin <= emax)) {
    if (threadIdx.x == 0) {
      g->emin = emin;
      g->emax = emax;
    }
    return;
  }
  auto faddu = [](unsigned long long u, unsigned long long v) { return u + v; };
#pragma unroll
  for (int e = 0; e < XRT_HIP_MAX_ELEM; ++e) {
    if (e >= M.nelem) break;
    const double* tE = M.tab_E[e];
    const int n = M.tab_n[e];
    unsigned long long lo = 0, hi = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const double v = tE[j];
      lo += v <= emin;
      hi += v <= emax;
    }
    // both counts in one reduction (they are below 2^32)
    const unsigned long long both = block_reduce(lo << 32 | hi, faddu, lds_u);
    ub_lo[e] = (int)(both >> 32);
    ub_hi[e] = (int)(both & 0xffffffffu);
    if (threadIdx.x == 0) {
      g->tab_lo[e] = ub_lo[e];
      g->tab_hi[e] = ub_hi[e];
    }
  }
  if (threadIdx.x == 0) {
    g->emin = emin;
    g->emax = emax;
  }
}

// stride = doubles per partial record: 8 (reflect_stats_dir) or 16
// (reflect_stats_dir_y, which also carries the bracket statistics of the y axis
// for both signs: [8..11] positive, [12..15] negative)
__device__ __forceinline__ void decide_axis_body(const xrt_hip_pass& P,
                                                 const xrt_hip_material& M,
                                                 const xrt_hip_beam& in, const double* part,
                                                 int nblocks, int stride, GStat* g) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  if (threadIdx.x == 0) {
    g->any_neg = 0;       // (a crystal's sign flags are re-raised by the exact pass)
    g->any_pos = 0;
    g->bracket_valid = 0;
    g->win_lo = -INFINITY;   // the windows below hold for the whole batch
    g->win_hi = INFINITY;
    g->tab_fast = nullptr;   // (the exact statistics overwrite the area they lived in)
  }
  double ma = 0., mb = 0., mc = 0., nent = 0., nmain = 0., emin = INFINITY, emax = -INFINITY;
  unsigned long long first = ~0ull;
  double yb[8] = {INFINITY, -INFINITY, 0., 0., INFINITY, -INFINITY, 0., 0.};
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    const double* o = part + (int64_t)b * stride;
    if (stride == 16) {
      for (int v = 0; v < 2; ++v) {
        const double q0 = ld_agent(o + 8 + 4 * v), q1 = ld_agent(o + 9 + 4 * v);
        yb[4 * v] = q0 < yb[4 * v] ? q0 : yb[4 * v];
        yb[4 * v + 1] = q1 > yb[4 * v + 1] ? q1 : yb[4 * v + 1];
        yb[4 * v + 2] = fmax(yb[4 * v + 2], ld_agent(o + 10 + 4 * v));
        yb[4 * v + 3] = fmax(yb[4 * v + 3], ld_agent(o + 11 + 4 * v));
      }
    }
    ma = fmax(ma, ld_agent(o));
    mb = fmax(mb, ld_agent(o + 1));
    mc = fmax(mc, ld_agent(o + 2));
    const unsigned long long f = (unsigned long long)__double_as_longlong(ld_agent(o + 3));
    first = f < first ? f : first;
    nent += ld_agent(o + 4);
    nmain += ld_agent(o + 5);
    const double e0 = ld_agent(o + 6), e1 = ld_agent(o + 7);
    emin = e0 < emin ? e0 : emin;
    emax = e1 > emax ? e1 : emax;
  }
  auto fmaxd = [](double u, double v) { return u > v ? u : v; };
  auto fmind = [](double u, double v) { return u < v ? u : v; };
  auto fminu = [](unsigned long long u, unsigned long long v) { return u < v ? u : v; };
  auto faddd = [](double u, double v) { return u + v; };
  emin = block_reduce(emin, fmind, lds_d);
  emax = block_reduce(emax, fmaxd, lds_d);
  ma = block_reduce(ma, fmaxd, lds_d);
  mb = block_reduce(mb, fmaxd, lds_d);
  mc = block_reduce(mc, fmaxd, lds_d);
  first = block_reduce(first, fminu, lds_u);
  nent = block_reduce(nent, faddd, lds_d);
  nmain = block_reduce(nmain, faddd, lds_d);
  if (stride == 16) {
    for (int v = 0; v < 2; ++v) {
      yb[4 * v] = block_reduce(yb[4 * v], fmind, lds_d);
      yb[4 * v + 1] = block_reduce(yb[4 * v + 1], fmaxd, lds_d);
      yb[4 * v + 2] = block_reduce(yb[4 * v + 2], fmaxd, lds_d);
      yb[4 * v + 3] = block_reduce(yb[4 * v + 3], fmaxd, lds_d);
    }
  }
  if (threadIdx.x != 0) return;
  g->maxa = ma;
  g->maxb = mb;
  g->maxc = mc;
  g->first_good = first;
  g->n_enter = (unsigned long long)nent;
  g->n_main = (unsigned long long)nmain;
  if (nent == 0.) return;
  table_windows(M, emin, emax, g);
  double maxa = ma, maxb = mb, maxc = mc;
  if (nmain == 0.) {  // np.max of an empty selection -> (0, 1, 0), base.py:1261-1262
    maxa = 0.;
    maxb = 1.;
    maxc = 0.;
  }
  const double mm = fmax(fmax(maxa, maxb), maxc);
  int axis = 2;
  if (mm == maxa)
    axis = 0;
  else if (mm == maxb)
    axis = 1;
  const int64_t i0 = (int64_t)first;
  double a = in.a[i0], b = in.b[i0], c = in.c[i0];
  local_dir(P, a, b, c);
  const double comp = axis == 0 ? a : (axis == 1 ? b : c);
  g->axis = axis;
  g->positive = comp > 0. ? 1 : 0;
  if (stride == 16 && axis == 1) {  // the usual case: the brackets are known already
    const int v = comp > 0. ? 0 : 1;
    g->t1min = yb[4 * v];
    g->t2max = yb[4 * v + 1];
    g->maxdz1 = yb[4 * v + 2];
    g->maxdz2 = yb[4 * v + 3];
    g->bracket_valid = 1;
  }
}

// One thread: the f1/f2 window of the pass (the table interval of the head ray's energy and its
// two neighbours; rays outside it search the whole table, interp_f1f2) and the TabFast records,
// from the upper bounds the block has counted. The window edges are fetched before the records
// are built and stored, so that all table loads are one trip.
__device__ __forceinline__ void decide_windows(const xrt_hip_material& M,
                                               const int (&ub_lo)[XRT_HIP_MAX_ELEM],
                                               const int (&ub_hi)[XRT_HIP_MAX_ELEM], GStat* g,
                                               TabFast* tf) {
  double wlo = -INFINITY, whi = INFINITY;
#pragma unroll
  for (int e = 0; e < XRT_HIP_MAX_ELEM; ++e) {
    if (e >= M.nelem || M.kind == XRT_HIP_MAT_NONE) break;
    const int n = M.tab_n[e];
    const int lo = ub_lo[e] > 0 ? ub_lo[e] - 1 : 0;
    const int hi = ub_hi[e] + 1 < n ? ub_hi[e] + 1 : n;
    g->tab_lo[e] = lo;
    g->tab_hi[e] = hi;
    if (lo > 0) wlo = fmax(wlo, M.tab_E[e][lo - 1]);
    if (hi < n) whi = fmin(whi, M.tab_E[e][hi]);
  }
  g->win_lo = wlo;     // one energy interval in which every element's window holds
  g->win_hi = whi;
  g->tab_fast = tab_fast_build(M, ub_lo, tf) ? tf : nullptr;
}

// ---------------------------------------------------------------------------
// The optimistic single pass. The reference takes four decisions from the whole
// batch before it moves a single ray: the bracketing axis (largest direction
// cosine), which of two bracket formulas (sign of the FIRST ray's component), the
// clamp range of the iterates ([min t1, max t2]) and secant-or-Brent (max |dz| at
// the bracket ends). For a beam that travels along the beamline they come out the
// same every time: axis y, the sign of ray 0, a clamp that never bites (a
// bracket-keeping secant iterate stays inside its own bracket, which lies inside
// the global range) and secant. So the fused kernel is first run ON those
// assumptions (more precisely: on ray 0's largest direction cosine as the axis, so
// that normal-incidence elements take the single pass as well) and every ray checks
// them for itself:
//   * a state-1 ray whose own largest cosine is another one (the axis might differ),
//   * an iterate outside its own bracket (the clamp might have acted),
//   * ray 0 not entering (the first entering ray is somebody else),
// raise `viol`; the bracket-end |dz| maxima are collected on the way. A one-thread
// kernel then decides: if anything was contradicted, or the maxima ask for Brent,
// `redo` goes up and the exact sequence (statistics, decisions, fused kernel again)
// that follows in the stream does the pass properly; otherwise those kernels return
// at once. Results are bit-identical to the exact sequence either way.
// ---------------------------------------------------------------------------
// rays looked at for the first entering ray and the first entering ray with state 1
#define REFLECT_SCAN 1024

// what the optimistic pass assumes, from the head of the beam (one block of
// REFLECT_BLOCK lanes). Returns false (and raises g->redo) when no ray of the head
// enters: the exact sequence then does the pass.
// ub_lo / ub_hi, dir0: what the DCM's decide kernel reuses (the counted upper bounds of the
// head ray's energy in M's tables; the head ray's direction in the beam's frame)
__device__ __forceinline__ bool decide_opt_body(const xrt_hip_pass& P, const xrt_hip_material& M,
                                                const xrt_hip_beam& in, OptStat* slots,
                                                GStat* g, unsigned long long* lds_u,
                                                int (&ub_lo)[XRT_HIP_MAX_ELEM],
                                                int (&ub_hi)[XRT_HIP_MAX_ELEM],
                                                double (&dir0)[3]) {
  for (int k = threadIdx.x; k < REFLECT_OPT_SLOTS; k += blockDim.x) {
    slots[k].maxdz1 = 0;
    slots[k].maxdz2 = 0;
    slots[k].viol = 0;
  }
  // (this opens the pass: no separate init launch)
  if (threadIdx.x == 0) {
    gstat_reset(g, 0);
    g->bar = 0;
    g->hang = 0;
  }
  // the first entering ray decides the bracket formula (base.py:1268-1283), the state-1
  // rays the axis (base.py:1257-1263): the first of each kind within the head
  unsigned long long i0 = ~0ull, i1 = ~0ull;
  const int64_t nscan = in.n < REFLECT_SCAN ? in.n : REFLECT_SCAN;
  for (int64_t i = threadIdx.x; i < nscan; i += blockDim.x) {
    const int st = in.state[i];
    if (entering(P, st)) {
      if ((unsigned long long)i < i0) i0 = (unsigned long long)i;
      if (st == 1 && (unsigned long long)i < i1) i1 = (unsigned long long)i;
    }
  }
  auto fminu = [](unsigned long long u, unsigned long long v) { return u < v ? u : v; };
  i0 = block_reduce(i0, fminu, lds_u);
  i1 = block_reduce(i1, fminu, lds_u);
  if (i0 == ~0ull) {
    if (threadIdx.x == 0) g->redo = 1;   // nothing assumed: the exact sequence handles it
    return false;
  }
  __syncthreads();                        // the reset precedes the window stores
  // the two head rays the last step needs: requested now, used after the table has been counted
  double a0 = 0., b0 = 0., c0 = 0., a1 = 0., b1 = 0., c1 = 0.;
  if (threadIdx.x == 0) {
    a0 = in.a[i0];
    b0 = in.b[i0];
    c0 = in.c[i0];
    dir0[0] = a0;
    dir0[1] = b0;
    dir0[2] = c0;
    if (i1 != ~0ull) {
      a1 = in.a[i1];
      b1 = in.b[i1];
      c1 = in.c[i1];
    }
  }
  // f1/f2 window: the table interval of that ray's energy and its two neighbours; rays
  // outside it search the whole table (interp_f1f2)
  const double E0 = in.E[i0];
  table_windows_block(M, E0, E0, g, lds_u, ub_lo, ub_hi);
  if (threadIdx.x != 0) return true;
  decide_windows(M, ub_lo, ub_hi, g, tab_fast_of(slots));
  // axis: the largest direction cosine of the first state-1 ray (y along a beamline, z at
  // normal incidence); every state-1 ray then checks that the same cosine strictly
  // dominates its own. No state-1 ray in the head: y, which is also what the reference
  // falls back to when the batch has none at all (base.py:1261-1262) -- any state-1 ray
  // further down still has to agree.
  int axis = 1;
  if (i1 != ~0ull) {
    local_dir(P, a1, b1, c1);
    const double m1 = fmax(fmax(fabs(a1), fabs(b1)), fabs(c1));
    axis = m1 == fabs(a1) ? 0 : (m1 == fabs(b1) ? 1 : 2);
  }
  local_dir(P, a0, b0, c0);
  const double comp0 = axis == 0 ? a0 : (axis == 1 ? b0 : c0);
  g->first_good = i0;
  g->axis = axis;
  g->positive = comp0 > 0. ? 1 : 0;
  g->t1min = -INFINITY;    // no clamp: escapes are reported instead
  g->t2max = INFINITY;
  g->maxdz1 = 1.;          // secant
  g->maxdz2 = 0.;
  g->optimistic = 1;
  if (P.method_hint && *P.method_hint) {   // this element's batches ask for Brent (the exit
    g->maxdz1 = 0.;                        // surface of a plate does): assume that instead
    g->maxdz2 = 1.;
    g->optimistic = 2;
  }
  return true;
}

#ifdef XRT_REFLECT_MAIN_TU
__global__ __launch_bounds__(REFLECT_BLOCK) void reflect_decide_opt(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_beam in, double* part, GStat* g) {
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  int ub_lo[XRT_HIP_MAX_ELEM], ub_hi[XRT_HIP_MAX_ELEM];
  double dir0[3] = {0., 0., 0.};
  // the (idle) partial-record area holds the report slots of the fused kernel
  decide_opt_body(P, M, in, reinterpret_cast<OptStat*>(part), g, lds_u, ub_lo, ub_hi, dir0);
}
// the same for a beam that exists only as its source's record: ray 0 is made here (into the
// one-ray beam `head`, scratch of the pass) and stands for the head of the beam -- every ray of
// a source enters with the same state, so the first entering ray IS ray 0
__global__ __launch_bounds__(REFLECT_BLOCK) void reflect_decide_opt_gen(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_geosource G, xrt_hip_beam head, double* part,
    GStat* g) {
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  if (threadIdx.x == 0) {
    const gen::GenRay r = gen::make_ray(G, gen::call_of(G), 0, false);
    head.state[0] = G.state;
    head.a[0] = r.a;
    head.b[0] = r.b;
    head.c[0] = r.c;
    head.E[0] = r.E;
  }
  __threadfence_block();
  __syncthreads();
  int ub_lo[XRT_HIP_MAX_ELEM], ub_hi[XRT_HIP_MAX_ELEM];
  double dir0[3] = {0., 0., 0.};
  decide_opt_body(P, M, head, reinterpret_cast<OptStat*>(part), g, lds_u, ub_lo, ub_hi, dir0);
}
#endif

// What the optimistic pass reported, folded by every block that needs the verdict
// (reflect_exact's gate; 256 slots): true = an assumption was contradicted or the
// bracket-end |dz| maxima ask for Brent, the pass has to be redone exactly.
__device__ __forceinline__ bool fold_opt(const OptStat* slots, double* lds_d, double& m1o,
                                         double& m2o, bool assumed_brent = false) {
  double m1 = 0., m2 = 0., viol = 0.;
  for (int k = threadIdx.x; k < REFLECT_OPT_SLOTS; k += blockDim.x) {
    m1 = fmax(m1, ld_agent(reinterpret_cast<const double*>(&slots[k].maxdz1)));
    m2 = fmax(m2, ld_agent(reinterpret_cast<const double*>(&slots[k].maxdz2)));
    viol = fmax(viol, (double)__hip_atomic_load(&slots[k].viol, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT));
  }
  auto fmaxd = [](double u, double v) { return u > v ? u : v; };
  m1 = block_reduce(m1, fmaxd, lds_d);
  m2 = block_reduce(m2, fmaxd, lds_d);
  viol = block_reduce(viol, fmaxd, lds_d);
  m1o = m1;
  m2o = m2;
  return viol != 0. || (m2 > m1 * 20.) != assumed_brent;
}


struct LocalRay {
  double x, y, z, a, b, c;
};

__device__ __forceinline__ LocalRay load_local(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                               int64_t i, LocalRay* raw = nullptr) {
  LocalRay r;
  r.x = in.x[i];
  r.y = in.y[i];
  r.z = in.z[i];
  r.a = in.a[i];
  r.b = in.b[i];
  r.c = in.c[i];
  if (raw) *raw = r;   // as it came, for a ray that leaves as it came
  local_pos(P, r.x, r.y, r.z);
  local_dir(P, r.a, r.b, r.c);
  return r;
}

template <class K>
__device__ __forceinline__ void stats_bracket_body(const xrt_hip_pass& P,
                                                   const xrt_hip_beam& in, int axis,
                                                   int positive, double* __restrict__ part) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  double t1m = INFINITY, t2m = -INFINITY, d1m = 0., d2m = 0.;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < in.n; i += stride) {
    const int st = in.state[i];
    const LocalRay r = load_local(P, in, i);   // loads issued together with the state
    if (!entering(P, st)) continue;
    double t1, t2, x, y, z;
    bracket(P, axis, positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
    const double dz1 = find_dz<K>(P, t1, r.x, r.y, r.z, r.a, r.b, r.c, x, y, z);
    double dz2 = find_dz<K>(P, t2, r.x, r.y, r.z, r.a, r.b, r.c, x, y, z);
    if (dz1 <= 0. || dz2 >= 0.) dz2 = 0.;  // base.py:863-865
    t1m = t1 < t1m ? t1 : t1m;
    t2m = t2 > t2m ? t2 : t2m;
    d1m = fmax(d1m, fabs(dz1));
    d2m = fmax(d2m, fabs(dz2));
  }
  auto fmaxd = [](double u, double v) { return u > v ? u : v; };
  auto fmind = [](double u, double v) { return u < v ? u : v; };
  t1m = block_reduce(t1m, fmind, lds_d);
  t2m = block_reduce(t2m, fmaxd, lds_d);
  d1m = block_reduce(d1m, fmaxd, lds_d);
  d2m = block_reduce(d2m, fmaxd, lds_d);
  if (threadIdx.x == 0) {
    double* o = part + (int64_t)blockIdx.x * 8;
    o[0] = t1m;
    o[1] = t2m;
    o[2] = d1m;
    o[3] = d2m;
  }
}

// First statistics pass that also anticipates the second one. The bracketing axis
// is y whenever max|b| is the largest direction cosine - always, for a beam that
// travels along the beamline. The brackets of that axis are evaluated here for
// both signs of the first ray's b (the reference's _set_t takes either formula for
// the whole batch); if the decision comes out as "y", reflect_stats_bracket finds
// bracket_valid set and returns at once. One pass over the beam saved.
template <class K>
__device__ __forceinline__ void stats_dir_y_body(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                                 double* __restrict__ part) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  double ma = 0., mb = 0., mc = 0., emin = INFINITY, emax = -INFINITY;
  unsigned long long first = ~0ull, nent = 0, nmain = 0;
  double t1m[2] = {INFINITY, INFINITY}, t2m[2] = {-INFINITY, -INFINITY};
  double d1m[2] = {0., 0.}, d2m[2] = {0., 0.};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // Which of the two formulas applies is the sign of the FIRST entering ray's b. If
  // ray 0 enters it is that ray, and only its variant is worth evaluating.
  int only = -1;
  if (in.n > 0 && entering(P, in.state[0])) {
    double a0 = in.a[0], b0 = in.b[0], c0 = in.c[0];
    local_dir(P, a0, b0, c0);
    only = b0 > 0. ? 0 : 1;
  }
  only = __builtin_amdgcn_readfirstlane(only);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < in.n; i += stride) {
    const int st = in.state[i];
    const double E = in.E[i];
    const LocalRay r = load_local(P, in, i);
    if (!entering(P, st)) continue;
    if ((unsigned long long)i < first) first = (unsigned long long)i;
    ++nent;
    emin = E < emin ? E : emin;
    emax = E > emax ? E : emax;
    if (st == 1) {  // mainPartForBracketing, reflect.py:644
      ma = fmax(ma, fabs(r.a));
      mb = fmax(mb, fabs(r.b));
      mc = fmax(mc, fabs(r.c));
      ++nmain;
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      if (only >= 0 && only != v) continue;   // wave-uniform
      double t1, t2, x, y, z;
      bracket(P, 1, v == 0 ? 1 : 0, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
      const double dz1 = find_dz<K>(P, t1, r.x, r.y, r.z, r.a, r.b, r.c, x, y, z);
      double dz2 = find_dz<K>(P, t2, r.x, r.y, r.z, r.a, r.b, r.c, x, y, z);
      if (dz1 <= 0. || dz2 >= 0.) dz2 = 0.;  // base.py:863-865
      t1m[v] = t1 < t1m[v] ? t1 : t1m[v];
      t2m[v] = t2 > t2m[v] ? t2 : t2m[v];
      d1m[v] = fmax(d1m[v], fabs(dz1));
      d2m[v] = fmax(d2m[v], fabs(dz2));
    }
  }
  auto fmaxd = [](double u, double v) { return u > v ? u : v; };
  auto fmind = [](double u, double v) { return u < v ? u : v; };
  auto fminu = [](unsigned long long u, unsigned long long v) { return u < v ? u : v; };
  auto faddu = [](unsigned long long u, unsigned long long v) { return u + v; };
  ma = block_reduce(ma, fmaxd, lds_d);
  mb = block_reduce(mb, fmaxd, lds_d);
  mc = block_reduce(mc, fmaxd, lds_d);
  first = block_reduce(first, fminu, lds_u);
  nent = block_reduce(nent, faddu, lds_u);
  nmain = block_reduce(nmain, faddu, lds_u);
  emin = block_reduce(emin, fmind, lds_d);
  emax = block_reduce(emax, fmaxd, lds_d);
  for (int v = 0; v < 2; ++v) {
    t1m[v] = block_reduce(t1m[v], fmind, lds_d);
    t2m[v] = block_reduce(t2m[v], fmaxd, lds_d);
    d1m[v] = block_reduce(d1m[v], fmaxd, lds_d);
    d2m[v] = block_reduce(d2m[v], fmaxd, lds_d);
  }
  if (threadIdx.x == 0) {
    double* o = part + (int64_t)blockIdx.x * 16;
    o[0] = ma;
    o[1] = mb;
    o[2] = mc;
    o[3] = __longlong_as_double((long long)first);
    o[4] = (double)nent;
    o[5] = (double)nmain;
    o[6] = emin;
    o[7] = emax;
    for (int v = 0; v < 2; ++v) {
      o[8 + 4 * v] = t1m[v];
      o[9 + 4 * v] = t2m[v];
      o[10 + 4 * v] = d1m[v];
      o[11 + 4 * v] = d2m[v];
    }
  }
}

__device__ __forceinline__ void reduce_bracket_body(const double* part, int nblocks, GStat* g) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  double t1m = INFINITY, t2m = -INFINITY, d1m = 0., d2m = 0.;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    const double* o = part + (int64_t)b * 8;
    const double q0 = ld_agent(o), q1 = ld_agent(o + 1);
    t1m = q0 < t1m ? q0 : t1m;
    t2m = q1 > t2m ? q1 : t2m;
    d1m = fmax(d1m, ld_agent(o + 2));
    d2m = fmax(d2m, ld_agent(o + 3));
  }
  auto fmaxd = [](double u, double v) { return u > v ? u : v; };
  auto fmind = [](double u, double v) { return u < v ? u : v; };
  t1m = block_reduce(t1m, fmind, lds_d);
  t2m = block_reduce(t2m, fmaxd, lds_d);
  d1m = block_reduce(d1m, fmaxd, lds_d);
  d2m = block_reduce(d2m, fmaxd, lds_d);
  if (threadIdx.x == 0) {
    g->t1min = t1m;
    g->t2max = t2m;
    g->maxdz1 = d1m;
    g->maxdz2 = d2m;
  }
}

// sum(beamInDotNormal) and count over the rays that hit (crystal path)
__device__ __forceinline__ void reduce_bdn_body(const double* part, int nblocks, GStat* g) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  double sum = 0., cnt = 0.;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    sum += ld_agent(part + (int64_t)b * 8);
    cnt += ld_agent(part + (int64_t)b * 8 + 1);
  }
  auto faddd = [](double u, double v) { return u + v; };
  sum = block_reduce(sum, faddd, lds_d);
  cnt = block_reduce(cnt, faddd, lds_d);
  if (threadIdx.x == 0) {
    g->sum_bdn = sum;
    g->n_good1 = (unsigned long long)cnt;
  }
}

// ---------------------------------------------------------------------------
// root solve, oes/base.py:848-1048, one ray
// ---------------------------------------------------------------------------
struct Hit {
  double t, x, y, z;     // local Cartesian hit point
  double px, py;         // what local_n takes: (x, y), or (s, phi) on a parametric surface
  int lost;  // ind1 of the reference: dz1 <= 0
};

// end of the solve: on a parametric surface the solver worked in (s, phi, r); the
// reference keeps those for local_n and converts back for everything else
// (reflect.py:701-704, 1066-1071)
template <class K>
__device__ __forceinline__ void hit_done(const xrt_hip_pass& P, Hit& h) {
  h.px = h.x;
  h.py = h.y;
  if (surf_is_param<K>(P)) {
    double x, y, z;
    ell_param_to_xyz(P, h.x, h.y, h.z, x, y, z);
    h.x = x;
    h.y = y;
    h.z = z;
  }
}

// what a ray of the optimistic pass reports (see reflect_decide_opt)
struct SolveAux {
  double adz1 = 0., adz2 = 0.;   // |dz| at the bracket ends, as find_intersection maximises them
  int escaped = 0;               // an iterate left the ray's own bracket
};

template <class K, bool OPT = false>
__device__ __forceinline__ Hit solve_ray(const xrt_hip_pass& P, const GStat& g,
                                         const LocalRay& r, SolveAux* aux = nullptr) {
  Hit h;
#ifdef XRT_PROBE_NULL_COMPUTE
  if (true) {
#else
  if (PNIS(P)) {  // reflect.py:676-682
#endif
    h.t = 0.;
    h.x = r.x;
    h.y = r.y;
    h.z = r.z;
    h.lost = 0;
    if (surf_is_param<K>(P)) ell_xyz_to_param(P, r.x, r.y, r.z, h.x, h.y, h.z);
    hit_done<K>(P, h);
    return h;
  }
  if (surf_is_blazed<K>(P)) {
    // first illuminated facet in closed form, gratings.py:492-522 (the bracket
    // is not used). A ray above both facets makes the reference raise; here it
    // is marked lost.
    const double rho_1 = P.surf_p[0], tanB = P.surf_p[1], tanAB = P.surf_p[2];
    const double b_c = r.b / r.c;
    const double v = r.y - b_c * r.z;
    const double n = floor(v / rho_1);
    const double y0 = rho_1 * n;
    const double y1 = y0 + rho_1;
    double zabl = P.surf_p[9] != 0. ? (y0 - r.y) / b_c + r.z
                                    : ((-tanAB) * (v - y0)) / (1. + tanAB * b_c);
    double zbl = P.surf_p[8] != 0. ? (y1 - r.y) / b_c + r.z
                                   : (tanB * (v - y1)) / (1. - tanB * b_c);
    h.lost = (zabl > 0. && zbl > 0.) ? 1 : 0;
    if (zabl > 0.) zabl = zbl - 1.;
    if (zbl > 0.) zbl = zabl - 1.;
    h.z = zbl;
    h.y = b_c * (h.z - r.z) + r.y;
    h.t = (h.y - r.y) / r.b;
    h.x = r.x + h.t * r.a;
    hit_done<K>(P, h);
    return h;
  }
  double t1, t2;
  bracket(P, g.axis, g.positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
  double x1, y1, z1, x2, y2, z2;
  double dz1 = find_dz<K>(P, t1, r.x, r.y, r.z, r.a, r.b, r.c, x1, y1, z1);
  double dz2 = find_dz<K>(P, t2, r.x, r.y, r.z, r.a, r.b, r.c, x2, y2, z2);
  const bool ind1 = dz1 <= 0.;
  const bool ind2 = dz2 >= 0.;
  if (OPT) {
    aux->adz1 = fabs(dz1);
    aux->adz2 = (ind1 || ind2) ? 0. : fabs(dz2);   // base.py:863-865
  }
  const double t1own = t1, t2own = t2;
  h.lost = ind1 ? 1 : 0;
  if (ind1) {
    h.t = t1;
    h.x = x1;
    h.y = y1;
    h.z = z1;
    hit_done<K>(P, h);
    return h;
  }
  if (ind2) {
    h.t = t2;
    h.x = x2;
    h.y = y2;
    h.z = z2;
    hit_done<K>(P, h);
    return h;
  }
  const double tMinG = g.t1min, tMaxG = g.t2max;
  const bool use_brent = g.maxdz2 > g.maxdz1 * 20.;
  int numit = 2;
  if (!use_brent) {
    // bracket-keeping secant, base.py:933-959. The first step is taken
    // unconditionally (the reference filters on |dz2| only after it).
    bool active = true;
#ifdef XRT_PROBE_ITERS
    while (active && numit < 2 + XRT_PROBE_ITERS) {
#else
    while (active && numit < kMaxIteration) {
#endif
      const double t = t1, dz = dz1;
      t1 = t2;
      dz1 = dz2;
      t2 = t - (t1 - t) * dz / (dz1 - dz);
      if (OPT) {
        // the optimistic pass runs without the batch's clamp (t1.min(), t2.max() are not
        // known yet): an iterate that leaves the ray's own bracket is reported instead
        aux->escaped |= (t2 < t1own) || (t2 > t2own);
      } else {
        if (t2 < tMinG) t2 = tMinG;
        if (t2 > tMaxG) t2 = tMaxG;
      }
      dz2 = find_dz<K>(P, t2, r.x, r.y, r.z, r.a, r.b, r.c, x2, y2, z2);
      // np.sign(dz2) == np.sign(dz1) with both ordered: equal sign bits and both zero or
      // both non-zero
      if (same_sign(dz2, dz1)) {
        t1 = t;
        dz1 = dz;
      }
      active = fabs(dz2) > kZEps;
      ++numit;
    }
  } else {
    // Brent, base.py:961-1048
    if (fabs(dz1) < fabs(dz2)) {
      double tmp = t1;
      t1 = t2;
      t2 = tmp;
      tmp = dz1;
      dz1 = dz2;
      dz2 = tmp;
    }
    double t3 = t1, dz3 = dz1, t4 = 0.;
    bool mflag = true;
    bool active = fabs(dz2) > kZEps;
    while (active && numit < kMaxIteration) {
      double xa = t1, xb = t2, xc = t3, xd = t4;
      double fa = dz1, fb = dz2, fc = dz3;
      double xs;
      if (fa != fc && fb != fc) {
        xs = xa * fb * fc / (fa - fb) / (fa - fc) + fa * xb * fc / (fb - fa) / (fb - fc) +
             fa * fb * xc / (fc - fa) / (fc - fb);
      } else {
        xs = xb - fb * (xb - xa) / (fb - fa);
      }
      const double q = (3. * xa + xb) / 4.;
      const bool cond1 = ((xs < q) && (xs < xb)) || ((xs > q) && (xs > xb));
      const bool cond2 = mflag && (fabs(xs - xb) >= (fabs(xb - xc) / 2.));
      const bool cond3 = (!mflag) && (fabs(xs - xb) >= (fabs(xc - xd) / 2.));
      const bool cond4 = mflag && (fabs(xb - xc) < kZEps);
      const bool cond5 = (!mflag) && (fabs(xc - xd) < kZEps);
      const bool conds = cond1 || cond2 || cond3 || cond4 || cond5;
      if (conds) xs = (xa + xb) / 2.;
      mflag = conds;
      const double fs = find_dz<K>(P, xs, r.x, r.y, r.z, r.a, r.b, r.c, x2, y2, z2);
      xd = xc;
      xc = xb;
      fc = fb;
      const bool neg = ((fa < 0.) && (fs > 0.)) || ((fa > 0.) && (fs < 0.));
      if (neg) {
        xb = xs;
        fb = fs;
      } else {
        xa = xs;
        fa = fs;
      }
      if (fabs(fa) < fabs(fb)) {
        double tmp = xa;
        xa = xb;
        xb = tmp;
        tmp = fa;
        fa = fb;
        fb = tmp;
      }
      t1 = xa;
      t2 = xb;
      t3 = xc;
      t4 = xd;
      dz1 = fa;
      dz2 = fb;
      dz3 = fc;
      active = fabs(dz2) > kZEps;
      ++numit;
    }
  }
  h.t = t2;
  h.x = x2;
  h.y = y2;
  h.z = z2;
  hit_done<K>(P, h);
  return h;
}

// The zone of a Fresnel zone plate a point falls into, NormalFZP.rays_good_gn
// (gratings.py:120-137): i = int(r_to_i(r)) with r_to_i = scipy's interp1d(rn, zones,
// bounds_error=False, fill_value=0), which for this table is np.interp (its slope form,
// exact table values at the knots) with 0 outside [rn[0], rn[N]]. -> transparent or not;
// rho = the local zone density 1 / (i_to_r(i+1) - i_to_r(i-1)), table ends giving 0.
__device__ __forceinline__ bool fzp_zone(const xrt_hip_pass& P, double x, double y, double& r,
                                         double& rho) {
  const double* rn = P.zone_r;
  const int N = P.zone_n;
  r = sqrt(x * x + y * y);
  double zi = 0.;
  if (r >= rn[0] && r <= rn[N]) {
    int lo = 0, hi = N;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (r >= rn[mid])
        lo = mid;
      else
        hi = mid;
    }
    if (r == rn[N]) {
      zi = (double)N;
    } else if (r == rn[lo]) {
      zi = (double)lo;
    } else {
      const double slope = 1. / (rn[lo + 1] - rn[lo]);
      zi = slope * (r - rn[lo]) + (double)lo;
    }
  }
  const int i = (int)zi;
  const double above = i + 1 <= N ? rn[i + 1] : 0.;
  const double below = i - 1 >= 0 ? rn[i - 1] : 0.;
  rho = 1. / (above - below);
  return (i % 2 == P.zone_black) && (r < rn[N]);
}

// rays_good, oes/base.py:1094-1163
__device__ __forceinline__ int rays_good_outline(const xrt_hip_pass& P, double x, double y);
template <class K>
__device__ __forceinline__ int rays_good(const xrt_hip_pass& P, double x, double y) {
  const int st = rays_good_outline(P, x, y);
  if (bent_crystal_surfaces<K>() && PSURF(P) == XRT_HIP_SURF_DICED) {  // bragg.py:92-101
    const Facet f = diced_facet(P, x, y);
    return fabs(f.fx) > P.surf_p[9] || fabs(f.fy) > P.surf_p[10] ? P.lost_num : st;
  }
  if (PGRATING(P) == 2 && P.zone_r) {  // gratings.py:123-129: opaque zones absorb
    double r, rho;
    return fzp_zone(P, x, y, r, rho) && st == 1 ? 1 : P.lost_num;
  }
  return st;
}
__device__ __forceinline__ int rays_good_outline(const xrt_hip_pass& P, double x, double y) {
  int st = 1;
  if (P.shape == XRT_HIP_SHAPE_POLYGON) {
    // matplotlib's Path.contains_points (src/_path.h, point_in_path_impl, radius 0): an
    // edge v0 -> v1 of the closed outline whose ends lie on different sides of the
    // horizontal through the point toggles `inside` when the crossing is to its right,
    // with matplotlib's tie rules on both comparisons
    bool inside = false;
    const double* v = P.poly_xy;
    double x0 = v[2 * (P.poly_n - 1)], y0 = v[2 * (P.poly_n - 1) + 1];
    for (int k = 0; k < P.poly_n; ++k) {
      const double x1 = v[2 * k], y1 = v[2 * k + 1];
      const bool up0 = y0 >= y, up1 = y1 >= y;
      if (up0 != up1 && (((y1 - y) * (x0 - x1) >= (x1 - x) * (y0 - y1)) == up1)) inside = !inside;
      x0 = x1;
      y0 = y1;
    }
    if (!(isfinite(x) && isfinite(y))) inside = false;
    if (P.poly_n < 3) inside = false;
    // base.py:1158-1160
    if (inside) return 1;
    return y < P.phys_y[0] ? P.lost_num : 3;
  }
  if (P.shape == XRT_HIP_SHAPE_RECT) {
    if (P.has_opt_x &&
        (((P.phys_x[0] <= x) && (x < P.opt_x[0])) || ((P.opt_x[1] <= x) && (x < P.phys_x[1]))))
      st = 2;
    if (P.has_opt_y &&
        (((P.phys_y[0] <= y) && (y < P.opt_y[0])) || ((P.opt_y[1] <= y) && (y < P.phys_y[1]))))
      st = 2;
    const bool outside =
        (x < P.phys_x[0]) || (x > P.phys_x[1]) || (y < P.phys_y[0]) || (y > P.phys_y[1]);
    bool over = false;
    if (P.over_mask & XRT_HIP_OVER_XMIN) over |= x < P.phys_x[0];
    if (P.over_mask & XRT_HIP_OVER_XMAX) over |= x > P.phys_x[1];
    if (P.over_mask & XRT_HIP_OVER_YMIN) over |= y < P.phys_y[0];
    if (P.over_mask & XRT_HIP_OVER_YMAX) over |= y > P.phys_y[1];
    if (outside) st = P.lost_num;
    if (over) st = 3;
  } else {
    double cx = (P.phys_x[0] + P.phys_x[1]) * 0.5;
    if (isnan(cx)) cx = 0.;
    const double rx = (P.phys_x[1] - P.phys_x[0]) * 0.5;
    double cy = (P.phys_y[0] + P.phys_y[1]) * 0.5;
    const double ry = (P.phys_y[1] - P.phys_y[0]) * 0.5;
    if (isnan(cy)) cy = 0.;
    if (!isinf(rx)) {
      const double u = (x - cx) / rx, v = (y - cy) / ry;
      if (u * u + v * v > 1.) st = P.lost_num;
    }
  }
  return st;
}

// ---------------------------------------------------------------------------
// amplitudes
// ---------------------------------------------------------------------------
// np.interp on the element table (element.py:252-263): upper_bound - 1, then
// slope*(x - xp[j]) + fp[j]
struct TabWin {
  int lo[XRT_HIP_MAX_ELEM], hi[XRT_HIP_MAX_ELEM];
  double elo, ehi;   // energies it is valid for
  const TabFast* fast;   // the pass's interval records, or null
};
__device__ __forceinline__ TabWin full_window() {
  TabWin w;
  for (int e = 0; e < XRT_HIP_MAX_ELEM; ++e) {
    w.lo[e] = 0;
    w.hi[e] = 0x7fffffff;
  }
  w.elo = -INFINITY;
  w.ehi = INFINITY;
  w.fast = nullptr;
  return w;
}
__device__ __forceinline__ TabWin window_of(const GStat& g) {
  TabWin w;
  for (int e = 0; e < XRT_HIP_MAX_ELEM; ++e) {
    w.lo[e] = g.tab_lo[e];
    w.hi[e] = g.tab_hi[e];
  }
  w.elo = g.win_lo;
  w.ehi = g.win_hi;
  w.fast = g.tab_fast;
  return w;
}

// np.interp from the pass's TabFast record (see reflect.h); ok = false: E is outside its
// knots. The record is the same for every lane and was written by an earlier kernel: it is
// read through the constant address space, i.e. by scalar loads into SGPRs.
__device__ __forceinline__ cplx interp_fast(const TabFast* rec, double E, bool& ok) {
  typedef const double __attribute__((address_space(4))) kdouble;
  kdouble* t = (kdouble*)(unsigned long long)rec;
  const double x0 = t[0], x1 = t[1], x2 = t[2], x3 = t[3];
  ok = E >= x0 && E < x3;
  const bool k1 = E >= x1, k2 = E >= x2;
  const double x = k2 ? x2 : (k1 ? x1 : x0);
  const double f1 = k2 ? t[6] : (k1 ? t[5] : t[4]);
  const double s1 = k2 ? t[9] : (k1 ? t[8] : t[7]);
  const double f2 = k2 ? t[12] : (k1 ? t[11] : t[10]);
  const double s2 = k2 ? t[15] : (k1 ? t[14] : t[13]);
  const double dE = E - x;
  return C(s1 * dE + f1, s2 * dE + f2);
}

__device__ __forceinline__ cplx interp_f1f2(const xrt_hip_material& M, int e, double E,
                                            const TabWin& w) {
  if (w.fast) {
    bool ok;
    const cplx f = interp_fast(w.fast + e, E, ok);
    if (ok) return f;
  }
  const double* __restrict__ tE = M.tab_E[e];
  const int n = M.tab_n[e];
  // upper_bound(E) lies in [w.lo, w.hi] for tE[w.lo - 1] <= E < tE[w.hi] (all elements:
  // for E in [w.elo, w.ehi)).
  // The exact sequence hands over the window of the batch's energy range (valid for
  // every ray); the optimistic pass a window around ray 0's energy, which a ray of a
  // different energy simply does not use.
  int lo = w.lo[e], hi = w.hi[e] < n ? w.hi[e] : n;
  if (!(E >= w.elo && E < w.ehi)) {
    lo = 0;
    hi = n;
    // no batch window (stand-alone calls, the layers of a multilayer, a ray of another
    // energy): the coarse index, if the caller supplied one -- two dependent loads in
    // place of ten
    const int32_t* __restrict__ coarse = M.tab_bucket[e];
    const long long k = (__double_as_longlong(E) >> XRT_HIP_BUCKET_SHIFT) - XRT_HIP_BUCKET_KEY0;
    if (coarse && k >= 0 && k < XRT_HIP_BUCKETS) {
      lo = coarse[k];
      hi = coarse[k + 1];
    }
  }
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (E >= tE[mid])
      lo = mid + 1;
    else
      hi = mid;
  }
  int j = lo - 1;
  if (j < 0) j = 0;
  double f1, f2;
  if (j >= n - 1) {
    f1 = M.tab_f1[e][n - 1];
    f2 = M.tab_f2[e][n - 1];
  } else if (tE[j] == E) {
    f1 = M.tab_f1[e][j];
    f2 = M.tab_f2[e][j];
  } else {
    const double dx = tE[j + 1] - tE[j];
    const double s1 = div_rn(M.tab_f1[e][j + 1] - M.tab_f1[e][j], dx);
    const double s2 = div_rn(M.tab_f2[e][j + 1] - M.tab_f2[e][j], dx);
    f1 = s1 * (E - tE[j]) + M.tab_f1[e][j];
    f2 = s2 * (E - tE[j]) + M.tab_f2[e][j];
  }
  return C(f1, f2);
}

// np.interp(E, tE, tI) for one ray (grating efficiency from a file, material.py:408-410)
__device__ __forceinline__ double efficiency_at(const double* __restrict__ tE,
                                             const double* __restrict__ tI, int n, double E) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (E >= tE[mid])
      lo = mid + 1;
    else
      hi = mid;
  }
  const int j = lo - 1;
  if (j < 0) return tI[0];
  if (j >= n - 1) return tI[n - 1];
  if (tE[j] == E) return tI[j];
  return div_rn(tI[j + 1] - tI[j], tE[j + 1] - tE[j]) * (E - tE[j]) + tI[j];
}

// material.py:348-378
__device__ __forceinline__ cplx refractive_index(const xrt_hip_material& M, double E,
                                                 const TabWin& w) {
  if (M.n_fixed) return C(M.n_re, M.n_im);
  cplx xf = C(0., 0.);
#pragma unroll
  for (int e = 0; e < XRT_HIP_MAX_ELEM; ++e) {
    if (e >= M.nelem) break;
    cplx f = interp_f1f2(M, e, E, w);
    f.re += (double)M.Z[e];
    xf = xf + f * M.quantity[e];
  }
  const double wl = div_rn(kCH, E);
  const double pre = 1e-24 * kAVOGADRO * kR0 / kPI2 * (wl * wl) * M.rho;
  const cplx v = (xf * pre) / M.mass;
  return C(1. - v.re, -v.im);
}

struct Ampl {
  cplx rs, rp;
  double mu, nk;
};

// Fresnel, material.py:415-493
// npre: the refractive index at E (for a crystal: f1 + i f2 of its element), if the
// caller looked it up already (the fused
// kernels do so BEFORE the root solve: the table search is two dependent trips to L2,
// which then overlap with the solve instead of standing between it and the amplitudes)
__device__ __forceinline__ Ampl material_amplitude_n(const xrt_hip_material& M, int kind,
                                                     double E, double bdn, const TabWin& w,
                                                     bool have_n, cplx n_given);
__device__ __forceinline__ Ampl material_amplitude(const xrt_hip_material& M, int kind,
                                                   double E, double bdn, const TabWin& w,
                                                   const cplx* npre = nullptr) {
  return material_amplitude_n(M, kind, E, bdn, w, npre != nullptr, npre ? *npre : C(1., 0.));
}
// (the index by VALUE: a pointer that may point at the caller's copy or at a local one kept
// that local in scratch memory)
__device__ __forceinline__ Ampl material_amplitude_n(const xrt_hip_material& M, int kind,
                                                     double E, double bdn, const TabWin& w,
                                                     bool have_n, cplx n_given) {
  Ampl A;
  const cplx n = have_n ? n_given : refractive_index(M, E, w);
  const cplx one = C(1., 0.);
  const double cosAlpha = fabs(bdn);
  double sinAlpha2 = 1. - bdn * bdn;
  if (sinAlpha2 < 0.) sinAlpha2 = 0.;
  if (M.from_vacuum && kind == XRT_HIP_MAT_MIRROR) {
    // a mirror seen from vacuum (the rule): n1 = 1, n2 = n. The same values as the general
    // form below (products with 1 + 0i are exact), without forming them.
    const double inv = frcp(cnorm2(n.re, n.im));
    const cplx rat = C(n.re * inv, -n.im * inv);
    const cplx cosBeta = csqrt_(one - (rat * rat) * sinAlpha2);
    const cplx nb = n * cosBeta, na = n * cosAlpha, ca = C(cosAlpha, 0.);
    A.rs = (ca - nb) / (ca + nb);
    A.rp = (na - cosBeta) / (na + cosBeta);
    A.mu = div_const(fabs(n.im) * E, kCHBAR, 1. / kCHBAR) * 2e8;
    A.nk = div_const(n.re * E, kCHBAR, 1. / kCHBAR) * 1e8;
    return A;
  }
  const cplx n1 = M.from_vacuum ? one : n;
  const cplx n2 = M.from_vacuum ? n : one;
  const cplx n1cosAlpha = n1 * cosAlpha;
  const cplx rat = n1 / n2;
  const cplx cosBeta = csqrt_(one - (rat * rat) * sinAlpha2);
  const cplx n2cosBeta = n2 * cosBeta;
  if (kind == XRT_HIP_MAT_MIRROR || kind == XRT_HIP_MAT_THIN_MIRROR) {
    A.rs = (n1cosAlpha - n2cosBeta) / (n1cosAlpha + n2cosBeta);
    A.rp = (n2 * cosAlpha - n1 * cosBeta) / (n2 * cosAlpha + n1 * cosBeta);
    if (kind == XRT_HIP_MAT_THIN_MIRROR) {
      // p2 = exp(2j E/CHBAR n2cosBeta t 1e7)
      const double f = 2. * E * (1.0 / kCHBAR);
      const cplx arg = ((C(0., f) * n2cosBeta) * M.t) * 1e7;
      const cplx p2 = cexp_(arg);
      A.rs = A.rs * ((one - p2) / (one - (A.rs * A.rs) * p2));
      A.rp = A.rp * ((one - p2) / (one - (A.rp * A.rp) * p2));
    }
  } else {  // plate
    const double tf = sqrt((n2cosBeta * conj(n1)).re / cosAlpha) / cabs_(n1);
    A.rs = ((2. * n1cosAlpha) / (n1cosAlpha + n2cosBeta)) * tf;
    A.rp = ((2. * n1cosAlpha) / (n2 * cosAlpha + n1 * cosBeta)) * tf;
  }
  // (the exact quotients: mu, nk are compared bitwise)
  A.mu = div_const(fabs(n.im) * E, kCHBAR, 1. / kCHBAR) * 2e8;
  A.nk = div_const(n.re * E, kCHBAR, 1. / kCHBAR) * 1e8;
  return A;
}

// Bragg / Laue dynamical-diffraction amplitudes, crystal.py:492-645.
// Amplitudes are compared at 1e-5 (observed ~1e-10), only ray STATES are bit-exact and they
// do not depend on anything below: the arithmetic here is arranged for few issue slots
// (one reciprocal per complex division, fused products, quantities that depend on the
// photon energy only computed once per ray -- for BOTH crystals of a DCM).
//
// What the amplitude needs of (crystal, photon energy): get_F_chi (crystal.py:297-306),
// the structure factors (crystals_basic.py:22-31, 76-80, 424-440), get_Bragg_angle (:1105-1120)
struct XtalEnergy {
  cplx chi0, chih, chih_;   // chi_0, chi_h, chi_hbar
  cplx chihh;               // chi_h chi_hbar
  double a1;                // H / k  (= 2 sin(thetaB))
  double a0;                // H^2 / (2 k^2)
  double cos2thetaB;        // the polarisation factor of p, crystal.py:643
  double k;                 // the wavenumber [1/A] (thin crystals)
};

// apre: f1 + i f2 of the crystal's element at E, if the caller looked it up already
template <bool CELL = true>
__device__ __forceinline__ XtalEnergy xtal_energy(const xrt_hip_material& M, double E,
                                                  const TabWin& w, const cplx* apre = nullptr) {
  XtalEnergy X;
  // a1 = H / k = (2 pi / d) (CH / (2 pi E)) = CH / (d E); the wavelength is a1 d
  X.a1 = kCH * frcp(M.d * E);
  const double waveLength = X.a1 * M.d;
  X.k = E * (kPI2 / kCH);
  X.a0 = 0.5 * (X.a1 * X.a1);
  cplx F0 = C(0., 0.), Fh = C(0., 0.), Fh_ = C(0., 0.);
  if (CELL && M.structure == 2) {
    // from the unit cell, crystals_basic.py:424-440: the sums over the atoms of each
    // element are constants of the reflection, only f1 + i f2 depends on the ray
    // (constant indices after unrolling: a run-time index into the by-value material record
    // would make the compiler keep a copy of it in scratch)
    const xrt_hip_cell& cell = *M.cell;
#pragma unroll
    for (int e = 0; e < XRT_HIP_MAX_ELEM; ++e) {
      if (e >= M.nelem) break;
      const cplx anom = interp_f1f2(M, e, E, w);
      F0 = F0 + (C((double)M.Z[e], 0.) + anom) * cell.w[e];
      const cplx f = C(cell.f0[e], 0.) + anom;
      Fh = Fh + f * C(cell.s[e][0], cell.s[e][1]);
      Fh_ = Fh_ + f * C(cell.sm[e][0], cell.sm[e][1]);
    }
    F0 = F0 * M.fact_dw;
    Fh = Fh * M.fact_dw;
    Fh_ = Fh_ * M.fact_dw;
  } else {
    const cplx anom = apre ? *apre : interp_f1f2(M, 0, E, w);
    F0 = (C((double)M.Z[0], 0.) + anom) * 4. * M.fact_dw;
    const int residue = (abs(M.hkl[0]) % 2) + (abs(M.hkl[1]) % 2) + (abs(M.hkl[2]) % 2);
    if (residue == 0 || residue == 3) Fh = (C(M.f0_hkl, 0.) + anom) * 4. * M.fact_dw;
    Fh_ = Fh;
    if (M.structure == 1) {
      const cplx d2f = C(M.d2f_re, M.d2f_im);
      F0 = F0 * 2.;
      Fh_ = Fh * conj(d2f);
      Fh = Fh * d2f;
    }
  }
  const double c2l = M.chi_to_f * (waveLength * waveLength);
  X.chi0 = conj(F0) * c2l;
  X.chih = conj(Fh) * c2l;
  X.chih_ = conj(Fh_) * c2l;
  X.chihh = X.chih * X.chih_;
  // Bragg angle, crystal.py:1105-1120: only cos(2 thetaB) = 1 - 2 sin^2(thetaB) is used
  double sb = 0.5 * X.a1;
  if (sb > 1.) sb = 1. - 1e-16;
  if (sb < -1.) sb = -1. + 1e-16;
  X.cos2thetaB = 1. - 2. * (sb * sb);
  return X;
}

// one polarisation of a thin crystal / a Laue case, crystal.py:586-616
__device__ __forceinline__ cplx crystal_thin_pol(const xrt_hip_material& M, double polFactor,
                                                 cplx alpha, cplx delta, cplx chih, cplx chi0,
                                                 double b, double k02, double k0s, double kHs) {
  const double sqb = sqrt(fabs(b));
  const double t = M.t_crystal * 1e7;
  const cplx l = ((delta * t) * k02) / 2. / kHs;
  const cplx I = C(0., 1.);
  cplx ra;
  // exp(1j k02 t (chi0 - alpha b) / 2 / k0s)
  const cplx ph = cexp_(((I * (k02 * t)) * (chi0 - alpha * b)) / 2. / k0s);
  if (M.geom_bragg) {
    if (M.geom_transmitted)
      ra = (C(1., 0.) / (ccos_(l) - ((I * alpha) * csin_(l)) / delta)) * ph;
    else
      ra = (chih * polFactor) / (alpha + (I * delta) / ctan_(l));
  } else {
    if (M.geom_transmitted)
      ra = (ccos_(l) + ((I * alpha) * csin_(l)) / delta) * ph;
    else
      ra = (((chih * polFactor) * csin_(l)) / delta) * ph;
  }
  if (!M.geom_transmitted) ra = ra / sqb;
  return ra;
}

// the thick Bragg case, crystal.py:571-584: chi_h C / (alpha +- delta), the root of smaller
// modulus, over sqrt|b|. The smaller quotient has the LARGER denominator, and
// |alpha + delta|^2 - |alpha - delta|^2 = 4 Re(alpha conj(delta)): one complex division.
// (rsqb_pol = polFactor / sqrt|b|)
__device__ __forceinline__ cplx crystal_thick_pol(cplx alpha, cplx delta, cplx chih,
                                                  double rsqb_pol) {
  const bool plus = fma_(alpha.re, delta.re, alpha.im * delta.im) >= 0.;
  const cplx den = plus ? alpha + delta : alpha - delta;
  const double scl = rsqb_pol * frcp(cnorm2(den.re, den.im));
  return C(fma_(chih.re, den.re, chih.im * den.im) * scl,
           fma_(chih.im, den.re, -(chih.re * den.im)) * scl);
}

template <bool THICK = false>
__device__ __forceinline__ Ampl crystal_amplitude_at(const xrt_hip_material& M,
                                                     const XtalEnergy& X, double bdsn,
                                                     double bosn, double bdhn) {
  Ampl A;
  A.mu = 0.;
  A.nk = 0.;
  // b = k0s / kHs = (in . n_s) / (out . n_s); kHs == 0 -> b = -1 (crystal.py:634-637)
  double b = -1., invb = -1.;
  if (bosn != 0.) {
    const double t = frcp(bdsn * bosn);
    b = (bdsn * bdsn) * t;
    invb = (bosn * bosn) * t;
  }
  // alpha = (H^2/2 - k0.H) / k^2 + chi0/2 (1/b - 1)
  const double h = 0.5 * (invb - 1.);
  const cplx alpha = C(fma_(X.chi0.re, h, fma_(-fabs(bdhn), X.a1, X.a0)), X.chi0.im * h);
  const cplx alpha2 = C(fma_(alpha.re, alpha.re, -(alpha.im * alpha.im)),
                        2. * (alpha.re * alpha.im));
  const double c2 = X.cos2thetaB;
  const cplx xb = X.chihh * invb;
  const cplx delta_s = csqrt_(alpha2 + xb);
  const cplx delta_p = csqrt_(alpha2 + xb * (c2 * c2));
  if (THICK || M.thick) {
    double rsqb;
    (void)sqrt_rn_rinv(fabs(b), rsqb);
    A.rs = crystal_thick_pol(alpha, delta_s, X.chih, rsqb);
    A.rp = crystal_thick_pol(alpha, delta_p, X.chih, rsqb * c2);
    return A;
  }
  const double k02 = X.k * X.k;
  const double k0s = -bdsn * X.k;
  const double kHs = bosn != 0. ? -bosn * X.k : 1.;
  A.rs = crystal_thin_pol(M, 1., alpha, delta_s, X.chih, X.chi0, b, k02, k0s, kHs);
  A.rp = crystal_thin_pol(M, c2, alpha, delta_p, X.chih, X.chi0, b, k02, k0s, kHs);
  return A;
}

template <bool THICK = false, bool CELL = true>
__device__ __forceinline__ Ampl crystal_amplitude(const xrt_hip_material& M, double E,
                                                  double bdsn, double bosn, double bdhn,
                                                  const TabWin& w, const cplx* apre = nullptr) {
  return crystal_amplitude_at<THICK>(M, xtal_energy<CELL>(M, E, w, apre), bdsn, bosn, bdhn);
}

// Multilayer / Coated, materials/multilayer.py:257-566: Parratt's recursion from the
// substrate up, separately for s and p, every interface weakened by its Nevot-Croce
// factor. Q_j = sqrt(Q^2 + 8 k^2 (n_j - 1)) with the conjugated tabulated index
// (:335-345). Amplitudes are compared at 1e-5, so the arithmetic is free to be cheap:
// one reciprocal per complex division, the phase factor of a layer by one exp + sincos.
struct Interface {   // from medium a down into medium b
  cplx rs, rp, ts, tp;
};
__device__ __forceinline__ Interface interface_ab(cplx Qa, cplx na, cplx Qb, cplx nb,
                                                  double sigma2, bool want_t) {
  Interface f;
  cplx rough = C(1., 0.);
  if (sigma2 != 0.) rough = cexp_((Qa * Qb) * (-0.5 * sigma2));
  const cplx is = rough / (Qa + Qb);
  f.rs = (Qa - Qb) * is;
  const cplx A = (Qa / na) * nb, B = (Qb / nb) * na;
  const cplx ip = rough / (A + B);
  f.rp = (A - B) * ip;
  f.ts = f.tp = C(0., 0.);
  if (want_t) {
    f.ts = (Qa * 2.) * is;
    f.tp = (A * 2.) * ip;
  }
  return f;
}

__device__ __forceinline__ cplx layer_index(const xrt_hip_material& m, double E) {
  if (m.nelem == 0) return C(1., 0.);
  return conj(refractive_index(m, E, full_window()));
}

struct StackState {
  cplx rs, rp, ts, tp;   // net amplitudes of everything below the present layer
};
// one layer of phase thickness phi = Q_layer * thickness on top of the state, under the
// interface `f` (multilayer.py:417-429)
template <bool TRAN>
__device__ __forceinline__ void add_layer(StackState& s, cplx rs, cplx rp, cplx ts, cplx tp,
                                          cplx p1, cplx p2) {
  const cplx us = s.rs * p2, up = s.rp * p2;
  const cplx ds = C(1., 0.) / (C(1., 0.) + rs * us), dp = C(1., 0.) / (C(1., 0.) + rp * up);
  s.rs = (rs + us) * ds;
  s.rp = (rp + up) * dp;
  if (TRAN) {
    s.ts = ((ts * s.ts) * p1) * ds;
    s.tp = ((tp * s.tp) * p1) * dp;
  }
}
__device__ __forceinline__ cplx half_phase(cplx Q, double thickness) {
  if (isinf(thickness)) return C(0., 0.);   // an absorbing half space lets nothing through
  return cexp_(C(-0.5 * Q.im * thickness, 0.5 * Q.re * thickness));
}

template <bool TRAN>
__device__ __forceinline__ void multilayer_stack(const xrt_hip_multilayer& L, double E,
                                                 double bdn, cplx& out_s, cplx& out_p) {
  const double k = E / kCHBAR;
  const cplx nt = layer_index(L.top, E), nb = layer_index(L.bottom, E),
             ns = layer_index(L.substrate, E);
  const double Q = 2. * k * fabs(bdn);
  const double Q2 = Q * Q, k28 = 8. * (k * k);
  const cplx Qv = C(Q, 0.), one = C(1., 0.);
  const cplx Qt = csqrt_(C(Q2, 0.) + (nt - one) * k28);
  const cplx Qb = csqrt_(C(Q2, 0.) + (nb - one) * k28);
  const cplx Qs = csqrt_(C(Q2, 0.) + (ns - one) * k28);
  // A coating (Coated: one period, no top layer, zero top thickness): the vacuum "layer" on
  // top reflects nothing and has no phase -- r = (Q - Q) / 2Q = 0, p = 1, the step leaves the
  // stack as it is (bit for bit) -- so its interface, roughness factor and step are skipped.
  const bool bare_top = !TRAN && L.top.nelem == 0 && L.npairs == 1 && L.dti[0] == 0.;
  if (bare_top) {   // (a path of its own: the loop below stays as the compiler likes it)
    const Interface tb1 = interface_ab(Qt, nt, Qb, nb, L.id2, false);
    const Interface bs1 = interface_ab(Qb, nb, Qs, ns, L.bs_rough2, false);
    StackState c;
    c.rs = bs1.rs;
    c.rp = bs1.rp;
    c.ts = c.tp = C(0., 0.);
    const cplx p1 = half_phase(Qb, L.dbi[0]);
    add_layer<false>(c, tb1.rs, tb1.rp, tb1.ts, tb1.tp, p1, p1 * p1);
    out_s = c.rs;   // (n_t = 1: no sign flip)
    out_p = c.rp;
    return;
  }
  const Interface vt = interface_ab(Qv, one, Qt, nt, L.id2, TRAN);
  const Interface tb = interface_ab(Qt, nt, Qb, nb, L.id2, TRAN);
  const Interface bs = interface_ab(Qb, nb, Qs, ns, L.bs_rough2, TRAN);
  StackState s;
  cplx bt_ts = C(0., 0.), bt_tp = C(0., 0.);
  if (TRAN) {
    // the far side of the substrate carries the substrate's roughness factor (:366-371)
    const Interface sv = interface_ab(Qs, ns, Qv, one, 0., true);
    const cplx rough = L.bs_rough2 != 0. ? cexp_((Qb * Qs) * (-0.5 * L.bs_rough2)) : one;
    s.rs = sv.rs * rough;
    s.rp = sv.rp * rough;
    s.ts = sv.ts * rough;
    s.tp = sv.tp * rough;
    const cplx p1 = half_phase(Qs, L.subst_thickness);
    add_layer<true>(s, bs.rs, bs.rp, bs.ts, bs.tp, p1, p1 * p1);
    // bottom -> top interface: r = -r(top -> bottom), t with the roles swapped (:349-352)
    const Interface bt = interface_ab(Qb, nb, Qt, nt, L.id2, true);
    bt_ts = bt.ts;
    bt_tp = bt.tp;
  } else {
    s.rs = bs.rs;
    s.rp = bs.rp;
    s.ts = s.tp = C(0., 0.);
  }
  cplx p1t = C(1., 0.), p2t = p1t, p1b = p1t, p2b = p1t;
  if (L.uniform) {
    p1t = half_phase(Qt, L.dti[0]);
    p2t = p1t * p1t;
    p1b = half_phase(Qb, L.dbi[0]);
    p2b = p1b * p1b;
  }
  for (int pair = L.npairs - 1; pair >= 0; --pair) {
    if (!L.uniform) {
      p1b = half_phase(Qb, L.dbi[pair]);
      p2b = p1b * p1b;
      p1t = half_phase(Qt, L.dti[pair]);
      p2t = p1t * p1t;
    }
    add_layer<TRAN>(s, tb.rs, tb.rp, tb.ts, tb.tp, p1b, p2b);
    if (pair == 0)
      add_layer<TRAN>(s, vt.rs, vt.rp, vt.ts, vt.tp, p1t, p2t);
    else
      add_layer<TRAN>(s, -tb.rs, -tb.rp, bt_ts, bt_tp, p1t, p2t);
  }
  if (TRAN) {
    out_s = s.ts;
    out_p = s.tp;
  } else {
    // a tabulated delta < 0 of the top layer turns the sign convention (:558-562; the
    // reference looks at the first ray of the batch, here every ray at its own energy)
    const bool flip = nt.re - 1. > 0.;
    out_s = flip ? conj(s.rs) : s.rs;
    out_p = flip ? conj(s.rp) : s.rp;
  }
}

__device__ __forceinline__ Ampl multilayer_amplitude(const xrt_hip_multilayer& L, double E,
                                                     double bdn) {
  Ampl A;
  if (L.transmitted)
    multilayer_stack<true>(L, E, bdn, A.rs, A.rp);
  else
    multilayer_stack<false>(L, E, bdn, A.rs, A.rp);
  A.mu = 0.;
  A.nk = 0.;
  return A;
}

// ---------------------------------------------------------------------------
// per-ray record and the "finish" half of _reflect_local (reflect.py:715-1110)
// ---------------------------------------------------------------------------
struct RayIn {
  double path, E, Jss, Jpp, Jsr, Jsi, Esr, Esi, Epr, Epi;
};

struct Rec {   // one whole ray in registers (the beam between the crystals of a DCM)
  double x, y, z, a, b, c;
  RayIn f;
  int st;
};

__device__ __forceinline__ void rot_coherency(double c, double s, double& Jss, double& Jpp,
                                              double& Jsr, double Jsi) {
  // sources/beams.py:448-479 with c = cos(roll), s = sin(roll) (imaginary part of
  // Jsp is unchanged)
  const double c2 = c * c, s2 = s * s, cs = c * s;
  const double ss = Jss * c2 + Jpp * s2 + 2. * Jsr * cs;
  const double pp = Jss * s2 + Jpp * c2 - 2. * Jsr * cs;
  const double sr = (Jpp - Jss) * cs + Jsr * (c2 - s2);
  Jss = ss;
  Jpp = pp;
  Jsr = sr;
  (void)Jsi;
}

struct Finished {
  double a, b, c;           // outgoing direction, local
  double theta, bdn;        // grazing angle; beamInDotNormal (clamped to [-1, 1])
  RayIn lo;                 // local beam fields (path, J, E-fields)
  double vJss, vJpp, vJsr, vJsi, vEsr, vEsi, vEpr, vEpi;  // rotated back for vlb
};

// The coherency matrix and the field amplitudes of the incoming ray (8 doubles)
// are loaded only after the geometry and the Fresnel / Bragg amplitudes are done:
// keeping them live across that code was what spilled to scratch.
__device__ __forceinline__ void load_fields(const xrt_hip_beam& in, int64_t i, bool has_amp,
                                            RayIn& q) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  q.Jss = __builtin_nontemporal_load(&in.Jss[i]);
  q.Jpp = __builtin_nontemporal_load(&in.Jpp[i]);
  const v2d js = __builtin_nontemporal_load(&reinterpret_cast<const v2d*>(in.Jsp_ri)[i]);
  q.Jsr = js.x;
  q.Jsi = js.y;
  q.Esr = q.Esi = q.Epr = q.Epi = 0.;
  if (has_amp) {
    const v2d es = __builtin_nontemporal_load(&reinterpret_cast<const v2d*>(in.Es_ri)[i]);
    const v2d ep = __builtin_nontemporal_load(&reinterpret_cast<const v2d*>(in.Ep_ri)[i]);
    q.Esr = es.x;
    q.Esi = es.y;
    q.Epr = ep.x;
    q.Epi = ep.y;
  }
}

// kernels compiled for fixed kinds need ~90 VGPRs: they can afford to have the 8
// doubles of the coherency matrix / amplitudes in flight during the whole finish
template <class K>
__device__ __forceinline__ constexpr bool early_fields() {
  return K::PLAIN && K::MK >= 0 && K::SK >= 0;
}

// What a fused kernel asks memory for as its very first act -- before it has looked at the
// pass's decisions, the ray's state or anything else that would put a round trip in front of
// these loads: state, position, direction, energy, and (FIELDS) the rest of the record. One
// trip to HBM per ray; the wave's prologue used to make three (state + position, then the
// energy for the f1/f2 look-up, then the fields), each behind the other.
struct RayRequest {
  int st0;
  LocalRay raw;
  RayIn q;      // E always; path, J, E fields only with FIELDS
};
template <bool FIELDS>
__device__ __forceinline__ RayRequest request_ray(const xrt_hip_beam& in, int64_t i,
                                                  bool has_amp) {
  RayRequest R;
  const bool live = i < in.n;
  const int64_t ii = live ? i : 0;
  // (non-temporal: read once; -1 % at 1e7 rays, same-box A/B)
  const int st = __builtin_nontemporal_load(&in.state[ii]);
  R.raw.x = __builtin_nontemporal_load(&in.x[ii]);
  R.raw.y = __builtin_nontemporal_load(&in.y[ii]);
  R.raw.z = __builtin_nontemporal_load(&in.z[ii]);
  R.raw.a = __builtin_nontemporal_load(&in.a[ii]);
  R.raw.b = __builtin_nontemporal_load(&in.b[ii]);
  R.raw.c = __builtin_nontemporal_load(&in.c[ii]);
  R.q = RayIn();
  R.q.E = __builtin_nontemporal_load(&in.E[ii]);
  if (FIELDS) {
    R.q.path = __builtin_nontemporal_load(&in.path[ii]);
    load_fields(in, ii, has_amp, R.q);
  }
  R.st0 = live ? st : 0;
  return R;
}

// local_n at a hit point: n[0..2] = n_H (Bragg planes), n[3..5] = the surface normal
// (the same unless the crystal is cut asymmetrically). (x, y): Cartesian hit point;
// (px, py): what the reference hands to local_n -- the same, or (s, phi) on a
// parametric surface.
// Bent crystal analysers (oes/bragg.py:146-343): surface normal n[3..5] and the normal
// of the atomic planes n[0..2]. rotate_x(y, z, c, s) = (c y - s z, s y + c z) is called
// with -sin(alpha); rotate_y(x, z, c, s) = (c x + s z, -s x + c z).
__device__ __forceinline__ void bent_toroid_normal(double x, double y, double Rm, double Rs,
                                                   double& a, double& b, double& c,
                                                   double& cosang, double& sinang) {
  const double root = sqrt(Rm * Rm - y * y);
  b = -y / Rm;
  const double c0 = root / Rm;
  const double r = Rs - (Rm - root);
  cosang = sqrt(r * r - x * x) / r;
  sinang = -x / r;
  a = sinang * c0;
  c = cosang * c0;
}
__device__ __forceinline__ void bent_bragg_normals(const xrt_hip_pass& P, double x, double y,
                                                   double (&n)[6]) {
  const int shape = (int)P.surf_p[0], planes = (int)P.surf_p[1];
  const double Rm = P.surf_p[2], Rs = P.surf_p[3];
  const double ca = P.surf_p[4], sa = P.surf_p[5];
  const bool tilted = P.surf_p[6] != 0.;
  const double root = sqrt(Rm * Rm - y * y);
  double cosang = 1., sinang = 0.;
  if (shape == 5) {  // BentLaue2D.local_n, laue.py:424-452
    const double a0 = -x / Rs, b0 = -y / Rm;
    const double norm = sqrt(a0 * a0 + b0 * b0 + 1.);
    const double a = a0 / norm, b = b0 / norm, c = 1. / norm;
    const double sinpitch = -b, cospitch = sqrt(1. - b * b);
    const double sinroll = -a, cosroll = sqrt(1. - a * a);
    double aB = 0., bB = 1., cB = 0.;
    if (tilted) {
      bB = ca;
      cB = -sa;
    }
    const double a1 = cosroll * aB - sinroll * cB;
    cB = sinroll * aB + cosroll * cB;
    aB = a1;
    const double b1 = cospitch * bB - sinpitch * cB;
    cB = sinpitch * bB + cospitch * cB;
    bB = b1;
    const double normB = sqrt(bB * bB + cB * cB + aB * aB);
    n[0] = aB / normB;
    n[1] = bB / normB;
    n[2] = cB / normB;
    n[3] = a / norm;   // (the reference divides the unit surface normal by its old norm once
    n[4] = b / norm;   // more, :452)
    n[5] = c / norm;
    return;
  }
  if (shape >= 3) {  // BentLaueSphere, laue.py:493-507: planes across the surface
    double a, b;
    if (shape == 3) {
      const double inv = 1. / sqrt((Rm * Rm - x * x) - y * y);
      a = -x * inv;
      b = -y * inv;
    } else {
      a = -x / Rm;
      b = -y / Rm;
    }
    const double norm = sqrt(a * a + b * b + 1.), normB = sqrt(b * b + 1.);
    n[0] = 0.;
    n[1] = 1. / normB;
    n[2] = -b / normB;
    n[3] = a / norm;
    n[4] = b / norm;
    n[5] = 1. / norm;
    return;
  }
  if (shape == 2) {
    bent_toroid_normal(x, y, Rm, Rs, n[3], n[4], n[5], cosang, sinang);
  } else {  // cylinder, :146-166
    n[3] = 0.;
    n[4] = -y / Rm;
    if (shape == 0) {
      n[5] = root / Rm;
    } else {
      const double norm = sqrt(n[4] * n[4] + 1.);
      n[4] /= norm;
      n[5] = 1. / norm;
    }
  }
  if (planes == 0) {  // Johann: the planes follow the surface, turned by alpha about x
    n[0] = n[3];
    n[1] = n[4];
    n[2] = n[5];
    if (tilted) {
      // (the toroid tilts its meridional normal first and turns it sagittally after, :254-266)
      const double b0 = shape == 2 ? -y / Rm : n[4];
      const double c0 = shape == 2 ? root / Rm : n[5];
      n[1] = ca * b0 + sa * c0;
      const double cA = -sa * b0 + ca * c0;
      n[0] = shape == 2 ? sinang * cA : 0.;
      n[2] = shape == 2 ? cosang * cA : cA;
    }
  } else if (planes == 1) {  // Johansson: planes bent to 2 Rm, :185-196, 279-295
    double b = -y, c = root + Rm;
    if (shape != 2 && tilted) {
      const double b1 = ca * b + sa * c;
      c = -sa * b + ca * c;
      b = b1;
    }
    const double norm = sqrt(b * b + c * c);
    b /= norm;
    c /= norm;
    double a = 0.;
    if (shape == 2) {
      if (tilted) {
        const double b1 = ca * b + sa * c;
        c = -sa * b + ca * c;
        b = b1;
      }
      a = sinang * c;
      c = cosang * c;
      if (tilted) {  // the reference turns a tilted plane normal a second time (:292-293)
        const double a1 = cosang * a + sinang * c;
        c = -sinang * a + cosang * c;
        a = a1;
      }
    }
    n[0] = a;
    n[1] = b;
    n[2] = c;
  } else if (planes == 2) {  // planes with their own radii, :336-342
    double cb, sb;
    bent_toroid_normal(x, y, P.surf_p[7], P.surf_p[8], n[0], n[1], n[2], cb, sb);
  } else {
    // Laue: the planes stand across the surface -- the surface normal (planes == 3,
    // BentLaueCylinder, laue.py:153-173) or the radial direction of twice the radius
    // (planes == 4, GroundBentLaueCylinder, :457-470) turned by 90 deg + alpha about x:
    // rotate_x(b, c, -sin(alpha), -cos(alpha)), without alpha (c, -b)
    double b = n[4], c = n[5];
    if (planes == 4) {
      b = -y;
      c = root + Rm;
    }
    double bB = c, cB = -b;
    if (tilted) {
      bB = -sa * b + ca * c;
      cB = -ca * b - sa * c;
    }
    const double norm = planes == 4 ? sqrt(bB * bB + cB * cB) : 1.;
    n[0] = 0.;
    n[1] = bB / norm;
    n[2] = cB / norm;
  }
}

template <class K>
__device__ __forceinline__ void surface_normal(const xrt_hip_pass& P, double x, double y,
                                               double px, double py, double (&n)[6]) {
#ifdef XRT_USER_SURFACE
  if (user_surface<K>()) {
    double v[3] = {0., 0., 1.};
    xrt_user::local_n(x, y, P.surf_p, v);
    n[0] = n[3] = v[0];
    n[1] = n[4] = v[1];
    n[2] = n[5] = v[2];
    return;
  }
#endif
  if (PSURF(P) == XRT_HIP_SURF_TOROID) {  // oes/__init__.py:403-411
    const double R = P.surf_p[0], rr = P.surf_p[1];
    const double qx = x * frcp(rr);
    const double rx = 1. - qx * qx;
    const double ax = rx < 0. ? 0. : frcp(sqrt(rx));
    const double na = -qx * ax;
    const double nb = -y * frcp(R);
    const double inorm = frcp(sqrt(na * na + nb * nb + 1.));
    n[0] = n[3] = na * inorm;
    n[1] = n[4] = nb * inorm;
    n[2] = n[5] = inorm;
  } else if (PSURF(P) == XRT_HIP_SURF_BENTFLAT) {  // oes/__init__.py:296-303
    const double nb = -y * frcp(P.surf_p[0]);
    const double inorm = frcp(sqrt(nb * nb + 1.));
    n[0] = n[3] = 0.;
    n[1] = n[4] = nb * inorm;
    n[2] = n[5] = inorm;
  } else if (surf_is_blazed<K>(P)) {  // gratings.py:482-490
    double y1, yL;
    const bool front = blazed_front(P, py, y1, yL);
    n[0] = n[3] = 0.;
    n[1] = n[4] = front ? -pinned(P.surf_p[3]) : pinned(P.surf_p[5]);
    n[2] = n[5] = front ? pinned(P.surf_p[4]) : pinned(P.surf_p[6]);
  } else if (PSURF(P) == XRT_HIP_SURF_SAGITTAL) {  // oes/__init__.py:658-662
    n[0] = n[3] = -x / P.surf_p[0];
    n[1] = n[4] = 0.;
    n[2] = n[5] = sqrt(P.surf_p[1] - x * x) / P.surf_p[0];
  } else if (bent_crystal_surfaces<K>() && PSURF(P) == XRT_HIP_SURF_DICED) {  // bragg.py:67-90
    Facet f = diced_facet(P, x, y);
    const bool six = P.surf_p[1] == 1. || (P.surf_p[0] != 0. && P.surf_p[6] != 0.);
    if (P.surf_p[1] == 1.) {   // the ground facet's own slope joins the surface normal
      const double db = -f.fy / P.surf_p[2];
      const double dnorm = sqrt(db * db + 1.);
      f.cn[5] += 1. / dnorm;
      f.cn[4] += db / dnorm;
      const double norm = sqrt(f.cn[5] * f.cn[5] + f.cn[4] * f.cn[4] + f.cn[3] * f.cn[3]);
      f.cn[5] /= norm;
      f.cn[4] /= norm;
      f.cn[3] /= norm;
    }
    // a three-component base normal is the surface normal and the plane normal at once
    double pa = six ? f.cn[0] : f.cn[3], pb = six ? f.cn[1] : f.cn[4], pc = six ? f.cn[2] : f.cn[5];
    if (P.surf_p[6] != 0.) {   // the asymmetric cut turns the plane normal (again), :84-88
      const double ca = P.surf_p[4], sa = P.surf_p[5];
      const double b1 = ca * pb + sa * pc;
      pc = -sa * pb + ca * pc;
      pb = b1;
    }
    n[0] = pa;
    n[1] = pb;
    n[2] = pc;
    n[3] = f.cn[3];
    n[4] = f.cn[4];
    n[5] = f.cn[5];
  } else if (bent_crystal_surfaces<K>() && PSURF(P) == XRT_HIP_SURF_BENT_BRAGG) {
    bent_bragg_normals(P, x, y, n);
  } else if (wide_surfaces<K>() && PSURF(P) == XRT_HIP_SURF_VFM) {  // oes/__init__.py:469-477
    double na = -x / sqrt(P.surf_p[1] - x * x);
    if (!isinf(P.surf_p[2]) && (x < P.surf_p[5] || x > P.surf_p[6])) na = 0.;
    const double nb = -y / P.surf_p[4];
    const double norm = sqrt(na * na + nb * nb + 1.);
    n[0] = n[3] = na / norm;
    n[1] = n[4] = nb / norm;
    n[2] = n[5] = 1. / norm;
  } else if (wide_surfaces<K>() && PSURF(P) == XRT_HIP_SURF_DUALVFM) {  // oes/__init__.py:552-571
    const bool left = x < 0.;
    const double u = x - (left ? pinned(P.surf_p[5]) : pinned(P.surf_p[2]));
    const double under = (left ? pinned(P.surf_p[4]) : pinned(P.surf_p[1])) - u * u;
    const double rise = (left ? pinned(P.surf_p[3]) : pinned(P.surf_p[0])) - sqrt(under);
    double na = -u / sqrt(under);
    if (isnan(na) || isnan(rise)) na = 0.;
    // the flat land between and beside the cylinders: local_z > 0 there (its meridional
    // term included, as the reference tests the full height)
    const double zfull = ((isnan(rise) || rise > 0.) ? 0. : rise) +
                         (y * y - P.surf_p[6]) / 2.0 / P.surf_p[7];
    if (zfull > 0.) na = 0.;
    const double nb = -y / P.surf_p[7];
    const double norm = sqrt(na * na + nb * nb + 1.);
    n[0] = n[3] = na / norm;
    n[1] = n[4] = nb / norm;
    n[2] = n[5] = 1. / norm;
  } else if (surf_is_cone<K>(P)) {  // oes/__init__.py:629-636
    const double u = y - P.surf_p[0];
    const double root =
        P.surf_p[4] * sqrt(P.surf_p[1] * (u * u) - ((P.surf_p[5] * x) * x) * P.surf_p[6]);
    const double na = (((-x) * P.surf_p[5]) * P.surf_p[6]) / root;
    const double nb = P.surf_p[7] + (P.surf_p[1] * u) / root;
    const double norm = sqrt(na * na + nb * nb + 1.);
    n[0] = n[3] = na / norm;
    n[1] = n[4] = nb / norm;
    n[2] = n[5] = 1. / norm;
  } else if (surf_is_lens<K>(P)) {  // refractive.py:405-419
    const double z = lens_parabola(P, x, y);
    const bool rim = P.surf_p[3] != 0. && z > P.surf_p[2];
    const double na = rim || P.surf_p[4] != 0. ? 0. : -x / P.surf_p[1];  // cylinder: -0 / (2 f) of an int 0
    const double nb = rim ? 0. : -y / P.surf_p[1];
    const double norm = sqrt(na * na + nb * nb + 1.);
    n[0] = n[3] = na / norm;
    n[1] = n[4] = nb / norm;
    n[2] = n[5] = 1. / norm;
  } else if (surf_is_param<K>(P)) {  // parametric.py:233-247, 460-472, 698-713
    const double A = P.surf_p[4], B = P.surf_p[5];
    const int conic = (int)P.surf_p[8];
    const double sp = P.surf_p[9] + px, phi = py;
    double nr, sg = -1.;
    if (conic == 3) {  // paraboloid capillary, parametric.py:783-788 (its own norm)
      double sn, cs;
      sincos(phi, &sn, &cs);
      const double na = -sn, nb = -sqrt(B / (A - sp)), nc = -cs;
      const double norm = sqrt(na * na + nb * nb + nc * nc);
      n[0] = n[3] = na / norm;
      n[1] = n[4] = nb / norm;
      n[2] = n[5] = nc / norm;
      return;
    }
    if (conic == 1) {
      nr = A / sqrt(A * sp + A * A);
    } else if (conic == 2) {
      double A2s2 = sp * sp - A * A;
      if (A2s2 <= 0.) A2s2 = 1e22;
      nr = (((-B) / A) * sp) / sqrt(A2s2);
      sg = 1.;
    } else {
      double A2s2 = A * A - sp * sp;
      if (A2s2 <= 0.) A2s2 = 1e22;
      nr = (((-B) / A) * sp) / sqrt(A2s2);
    }
    const double norm = sqrt(nr * nr + 1.);
    const double nb = nr / norm;
    double na, nc;
    if (P.surf_p[6] != 0.) {
      na = 0.;
      nc = 1. / norm;
    } else {
      double sn, cs;
      sincos(phi, &sn, &cs);
      na = (sg * sn) / norm;
      nc = (sg * cs) / norm;
    }
    const double cg = P.surf_p[2], msg = -P.surf_p[3];
    n[0] = n[3] = na;
    n[1] = n[4] = cg * nb - msg * nc;
    n[2] = n[5] = msg * nb + cg * nc;
  } else {
    for (int j = 0; j < 6; ++j) n[j] = P.n_const[j];
  }
}

// QREADY: q already holds the ray's fields (they came in registers, not from `in`)
template <class K, bool QREADY = false>
__device__ __forceinline__ Finished finish_ray(const xrt_hip_pass& P,
                                               const xrt_hip_material& M, const GStat& g,
                                               const LocalRay& r, const Hit& h, RayIn q,
                                               const xrt_hip_beam& in, int64_t i,
                                               bool has_amp, int own_sign = 0,
                                               const cplx* npre = nullptr,
                                               XtalEnergy* xe = nullptr,
                                               bool xe_ready = false) {
  // Material(refractiveIndex = table): the caller evaluated the spline at every ray's energy.
  // Only the kernels that read the material kind at run time carry the test (the launcher
  // keeps such materials away from the lean kernels).
  bool have_n = npre != nullptr;
  cplx n_here = have_n ? *npre : C(1., 0.);
  if (K::MK < 0 && M.n_fixed == 2) {
    n_here = C(M.n_ray[2 * i], M.n_ray[2 * i + 1]);
    have_n = true;
  }
  Finished out;
  q.path += h.t;
  // normals: n[0..2] = n_H (Bragg planes), n[3..5] = surface
  double n[6];
  surface_normal<K>(P, h.x, h.y, h.px, h.py, n);
  if constexpr (K::FE) {
    // reflect.py:767-775 turns oeNormal[-3:]: the surface normal -- which is also the normal
    // of the atomic planes unless the cut is asymmetric (six components)
    if (P.fe_c) {
      figure_turn_normal(P, h.x, h.y, n[3], n[4], n[5]);
      if (!PASYM(P)) {
        n[0] = n[3];
        n[1] = n[4];
        n[2] = n[5];
      }
    }
  }
  double bdn = r.a * n[0] + r.b * n[1] + r.c * n[2];
  if (bdn < -1.) bdn = -1.;
  if (bdn > 1.) bdn = 1.;
  out.theta = acos_np(bdn) - kPI / 2.;
  out.bdn = bdn;
  const double bdsn = PASYM(P) ? (r.a * n[3] + r.b * n[4] + r.c * n[5]) : bdn;

  int took_order = 0;   // the diffraction order this ray takes (gratings, zone plates)
  int toWhere = 0;  // reflect.py:723-752
  if (MKIND(M) == XRT_HIP_MAT_PLATE)
    toWhere = 1;
  else if ((MKIND(M) == XRT_HIP_MAT_CRYSTAL || layered<K>()) && M.geom_transmitted)
    toWhere = 2;
  // a multilayer deflects like a Bragg crystal of its period (reflect.py:865-872), a
  // coated mirror like a mirror
  bool as_crystal = MKIND(M) == XRT_HIP_MAT_CRYSTAL;
  if constexpr (layered<K>()) as_crystal = M.geom_bragg != 0;

  double ao = r.a, bo = r.b, co = r.c;  // a_out of the reference
  out.a = r.a;
  out.b = r.b;
  out.c = r.c;
  if (toWhere == 0 || toWhere == 2) {
    if (as_crystal && toWhere == 0) {
      // crystal as a grating, reflect.py:568-612 + 451-469
      const double ndsn = n[0] * n[3] + n[1] * n[4] + n[2] * n[5];
      // sign of the batch mean of beamInDotNormal (reflect.py:573-574). In the
      // optimistic single pass (own_sign) every ray uses its own sign: identical
      // whenever all rays of the batch agree, which the any_neg / any_pos flags verify.
      const double bdnMean = own_sign ? bdn : g.sum_bdn / (double)g.n_good1;
      const double sgbdn = bdnMean < 0. ? 1. : -1.;
      const double wHd = div_rn(1., M.d * 1e-7);
      const double g0 = (n[0] - ndsn * n[3]) * wHd * sgbdn;
      const double g1 = (n[1] - ndsn * n[4]) * wHd * sgbdn;
      const double g2 = (n[2] - ndsn * n[5]) * wHd * sgbdn;
      const double sig = M.geom_bragg ? -1. : 1.;
      const double bdg = r.a * g0 + r.b * g1 + r.c * g2;
      const double G2 = g0 * g0 + g1 * g1 + g2 * g2;
      const double ol = div_rn(kCH, q.E) * 1e-7;
      const double u = bdsn * bdsn - 2. * bdg * ol - G2 * (ol * ol);
      const double dn = bdsn + sig * sqrt_unit(fabs(u));
      ao = r.a - n[3] * dn + g0 * ol;
      bo = r.b - n[4] * dn + g1 * ol;
      co = r.c - n[5] * dn + g2 * ol;
      // (directions are compared at 1e-12: the correctly rounded root and quotients)
      const double nm = sqrt_unit(ao * ao + bo * bo + co * co);
      ao = div_rn(ao, nm);
      bo = div_rn(bo, nm);
      co = div_rn(co, nm);
    } else if (PGRATING(P)) {
      // grating equation, reflect.py:840-861 + 451-469 (sign -1); the groove
      // vector of OE.local_g (base.py:688-717)
      double g0 = P.g_const[0], g1 = P.g_const[1], g2 = P.g_const[2];
      double gsig = -1.;
      took_order = P.order_ray ? P.order_ray[i] : P.grating_order;
      if (K::RAYG && P.grating == 2 && P.g_ray_x) {   // general zone plate: the caller's vectors
        g0 = P.g_ray_x[i];
        g1 = P.g_ray_y[i];
        g2 = 0.;
        gsig = 1.;
      } else if (P.grating == 2) {  // zone plate: gn of rays_good_gn, sign +1 (reflect.py:857)
        double rad, rho;
        fzp_zone(P, h.x, h.y, rad, rho);
        g0 = -h.x / rad * rho;
        g1 = -h.y / rad * rho;
        g2 = 0.;
        gsig = 1.;
#ifdef XRT_USER_SURFACE
      } else if (user_surface<K>() && P.grating_axis == 2) {   // the class's hip_local_g
        double gv[3] = {0., 0., 0.};
        xrt_user::local_g(h.x, h.y, P.surf_p, gv);
        g0 = gv[0];
        g1 = gv[1];
        g2 = gv[2];
#endif
      } else if (P.grating_axis >= 0) {
        const double coord = P.grating_axis == 0 ? h.x : h.y;
        double poly = 0.;
        // coord**ic: exact for ic <= 2 like numpy's; for higher powers numpy
        // calls pow() (one rounding), here repeated products (ic-1 roundings,
        // <= 3 ulp on terms that are ~1e-6 of the line density)
        double pw = 1.;
        for (int ic = 0; ic < P.g_ncoef; ++ic) {
          poly += ((double)(ic + 1) * P.g_coef[ic]) * pw;
          pw *= coord;
        }
        const double N = P.g_rho0 * poly;
        g0 = P.grating_axis == 0 ? N : 0.;
        g1 = P.grating_axis == 0 ? 0. : N;
        g2 = 0.;
      }
      const double bdg = r.a * g0 + r.b * g1 + r.c * g2;
      const double G2 = g0 * g0 + g1 * g1 + g2 * g2;
      // one order for all rays, or the caller's per-ray draw (reflect.py:455-459)
      const double ord = P.order_ray ? (double)P.order_ray[i] : (double)P.grating_order;
      const double ol = div_rn(ord * kCH, q.E) * 1e-7;
      const double u = bdsn * bdsn - 2. * bdg * ol - G2 * (ol * ol);
      const double dn = bdsn + gsig * sqrt_unit(fabs(u));
      ao = r.a - n[3] * dn + g0 * ol;
      bo = r.b - n[4] * dn + g1 * ol;
      co = r.c - n[5] * dn + g2 * ol;
      const double nm = sqrt_unit(ao * ao + bo * bo + co * co);
      ao = div_rn(ao, nm);
      bo = div_rn(bo, nm);
      co = div_rn(co, nm);
    } else {  // specular, reflect.py:875-877
      ao = r.a - n[0] * 2. * bdn;
      bo = r.b - n[1] * 2. * bdn;
      co = r.c - n[2] * 2. * bdn;
    }
    if (toWhere == 0) {
      out.a = ao;
      out.b = bo;
      out.c = co;
    }
  } else {  // refraction, reflect.py:894-919
    const double nre = have_n ? n_here.re : refractive_index(M, q.E, window_of(g)).re;
    const double n1overn2 = M.from_vacuum ? 1. / nre : nre;
    const double signN = (double)sgn(-bdn);
    const double n1c = -n1overn2 * bdn;
    const double cosTheta2 = signN * sqrt(1. - n1overn2 * n1overn2 + n1c * n1c);
    const double dn = n1c - cosTheta2;
    out.a = r.a * n1overn2 + n[0] * dn;
    out.b = r.b * n1overn2 + n[1] * dn;
    out.c = r.c * n1overn2 + n[2] * dn;
  }

  // coherency matrix into the local s/p frame, reflect.py:948-953:
  // rollAngle = roll + atan2(n_x, n_z); only its cos and sin are ever used, so
  // they come from the angle-addition formulas with cos/sin(atan2) = (n_z, n_x)/hyp
  double cosY = P.cos_roll, sinY = P.sin_roll;
  if (n[3] != 0.) {
    double ih;   // 1 / hypot(n_x, n_z) falls out of the root's iteration
    (void)sqrt_rn_rinv(fma_(n[3], n[3], n[5] * n[5]), ih);
    const double cphi = n[5] * ih, sphi = n[3] * ih;
    cosY = P.cos_roll * cphi - P.sin_roll * sphi;
    sinY = P.sin_roll * cphi + P.cos_roll * sphi;
  } else if (n[5] < 0.) {  // atan2(0, negative) = pi
    cosY = -P.cos_roll;
    sinY = -P.sin_roll;
  }
  // amplitudes, reflect.py:955-1035
  Ampl A;
  A.rs = C(1., 0.);
  A.rp = C(1., 0.);
  A.mu = 0.;
  A.nk = 0.;
  if constexpr (layered<K>()) {
    // reflect.py:999-1003 hands over beamInDotSurfaceNormal, :1031-1032 (Coated, kind
    // 'mirror') beamInDotNormal
    A = multilayer_amplitude(*M.layers, q.E, M.geom_bragg ? bdsn : bdn);
  } else if (MKIND(M) == XRT_HIP_MAT_CRYSTAL) {
    const double bosn = ao * n[3] + bo * n[4] + co * n[5];
    // xe: the energy-dependent part is shared between the two crystals of a DCM
    XtalEnergy X;
    if (xe && xe_ready) {
      X = *xe;
    } else {
      X = xtal_energy<K::XCELL>(M, q.E, window_of(g), npre);
      if (xe) *xe = X;
    }
    A = crystal_amplitude_at<K::XTHICK>(M, X, bdsn, bosn, bdn);
#ifndef XRT_PROBE_NO_AMPL
  } else if (MKIND(M) != XRT_HIP_MAT_NONE) {
    A = material_amplitude_n(M, MKIND(M), q.E, bdn, window_of(g), have_n, n_here);
#endif
  }
  if (PGRATING(P) && P.eff_n > 0) {  // tabulated efficiency of the order, material.py:391-413
    double amp = 0.;
    int row = -1;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < P.eff_n && P.eff_order[k] == took_order) {
        amp = P.eff_amp[k];
        row = k;
      }
    // efficiency file: np.interp on (E, row), :408-410 (not compiled into the kernels of
    // layered materials and crystals: a grating's material is neither, capi.hip refuses it)
    if (K::MK != XRT_HIP_MAT_MULTILAYER && K::MK != XRT_HIP_MAT_CRYSTAL && P.eff_tab_n > 0 &&
        row >= 0)
      amp = sqrt(efficiency_at(P.eff_tab_E, P.eff_tab_I + (int64_t)row * P.eff_tab_n,
                               P.eff_tab_n, q.E));
    A.rs = A.rp = C(amp, 0.);
    A.mu = A.nk = 0.;
  }
  if (cisnan(A.rs)) A.rs = C(0., 0.);
  if (cisnan(A.rp)) A.rp = C(0., 0.);
  // now the fields of the incoming ray, rotated into the local s/p frame (the lean
  // specialised kernels have registers to spare and issued these loads up front)
  if (!QREADY && !early_fields<K>()) load_fields(in, i, has_amp, q);
  double Jss = q.Jss, Jpp = q.Jpp, Jsr = q.Jsr, Jsi = q.Jsi;
  rot_coherency(cosY, -sinY, Jss, Jpp, Jsr, Jsi);
  cplx Es = C(q.Esr, q.Esi), Ep = C(q.Epr, q.Epi);
  if (has_amp) {
    const cplx e1 = Es * cosY + Ep * (-sinY);
    const cplx e2 = Es * sinY + Ep * cosY;
    Es = e1;
    Ep = e2;
  }
  // J' and E', reflect.py:1038-1064
  const double as2 = A.rs.re * A.rs.re + A.rs.im * A.rs.im;
  const double ap2 = A.rp.re * A.rp.re + A.rp.im * A.rp.im;
  Jss = Jss * as2;
  Jpp = Jpp * ap2;
  const cplx jsp = (C(Jsr, Jsi) * A.rs) * conj(A.rp);
  Jsr = jsp.re;
  Jsi = jsp.im;
  if (has_amp) {
    Es = Es * A.rs;
    Ep = Ep * A.rp;
  }
  if (!M.from_vacuum && MKIND(M) != XRT_HIP_MAT_NONE && MKIND(M) != XRT_HIP_MAT_CRYSTAL &&
      !layered<K>()) {
    const double att = exp(-A.mu * h.t * 0.1);
    Jss *= att;
    Jpp *= att;
    Jsr *= att;
    Jsi *= att;
    if (has_amp) {
      double s, c;
      sincos(0.1 * A.nk * h.t, &s, &c);
      const double sq = sqrt(att);
      const cplx mPh = C(sq * c, sq * s);
      Es = Es * mPh;
      Ep = Ep * mPh;
    }
  } else if (has_amp) {
    // exp(1e7j E/CHBAR t): numpy divides the complex 1e7j*E by CHBAR with
    // Smith's algorithm = multiply by 1.0/CHBAR; the phase is ~1e12 rad, so
    // the rounding sequence must be the same
    const double ph = ((1e7 * q.E) * (1.0 / kCHBAR)) * h.t;
    double s, c;
    sincos_phase(ph, s, c);
    const cplx mPh = C(c, s);
    Es = Es * mPh;
    Ep = Ep * mPh;
  }
  q.Jss = Jss;
  q.Jpp = Jpp;
  q.Jsr = Jsr;
  q.Jsi = Jsi;
  q.Esr = Es.re;
  q.Esi = Es.im;
  q.Epr = Ep.re;
  q.Epi = Ep.im;
  out.lo = q;
  // rotate back for the virgin-local beam, reflect.py:1106-1110
  out.vJss = Jss;
  out.vJpp = Jpp;
  out.vJsr = Jsr;
  out.vJsi = Jsi;
  rot_coherency(cosY, sinY, out.vJss, out.vJpp, out.vJsr, out.vJsi);
  if (has_amp) {
    const cplx e1 = Es * cosY + Ep * sinY;
    const cplx e2 = Es * (-sinY) + Ep * cosY;
    out.vEsr = e1.re;
    out.vEsi = e1.im;
    out.vEpr = e2.re;
    out.vEpi = e2.im;
  } else {
    out.vEsr = out.vEsi = out.vEpr = out.vEpi = 0.;
  }
  return out;
}

// ---------------------------------------------------------------------------
// stores
// ---------------------------------------------------------------------------
// OPTIONAL: the record may be one nobody asked for (o.x null: OE.reflect(needLocal=False) makes
// no local beam). Only the passes that can be called that way pay the test: with it in the
// kernels of layered materials their spills went from 50 to 72 VGPRs.
template <bool OPTIONAL = false, bool STATE = true>
__device__ __forceinline__ void store_ray(const xrt_hip_beam& o, int64_t i, double x, double y,
                                          double z, double a, double b, double c, double path,
                                          double E, double Jss, double Jpp, double Jsr,
                                          double Jsi, int st, double Esr, double Esi,
                                          double Epr, double Epi, bool has_amp) {
  // Non-temporal stores: the outgoing records are streamed out once, never read again by this
  // kernel, and three times the size of any cache. Same-box A/B at 1e7 rays: cfg2 0.585 ->
  // 0.571 ms, DCM 0.869 -> 0.853 ms. (No effect in round 2, when the kernels were a prologue
  // away from their streaming floor; the floor itself gains 1-4 %, profiles/r03_probe_stream.txt.)
  typedef double v2d __attribute__((ext_vector_type(2)));
  if (OPTIONAL && !o.x) return;
  __builtin_nontemporal_store(x, &o.x[i]);
  __builtin_nontemporal_store(y, &o.y[i]);
  __builtin_nontemporal_store(z, &o.z[i]);
  __builtin_nontemporal_store(a, &o.a[i]);
  __builtin_nontemporal_store(b, &o.b[i]);
  __builtin_nontemporal_store(c, &o.c[i]);
  __builtin_nontemporal_store(path, &o.path[i]);
  __builtin_nontemporal_store(E, &o.E[i]);
  __builtin_nontemporal_store(Jss, &o.Jss[i]);
  __builtin_nontemporal_store(Jpp, &o.Jpp[i]);
  __builtin_nontemporal_store(v2d{Jsr, Jsi}, &reinterpret_cast<v2d*>(o.Jsp_ri)[i]);
  if (STATE) __builtin_nontemporal_store(st, &o.state[i]);
  if (has_amp) {
    __builtin_nontemporal_store(v2d{Esr, Esi}, &reinterpret_cast<v2d*>(o.Es_ri)[i]);
    __builtin_nontemporal_store(v2d{Epr, Epi}, &reinterpret_cast<v2d*>(o.Ep_ri)[i]);
  }
}

template <bool OPTIONAL = false>
__device__ __forceinline__ void copy_ray(const xrt_hip_beam& o, const xrt_hip_beam& s,
                                         int64_t i, int st, bool has_amp, bool zero_xyz) {
  if (OPTIONAL && !o.x) return;
  double2 js = reinterpret_cast<const double2*>(s.Jsp_ri)[i];
  double2 es = make_double2(0., 0.), ep = make_double2(0., 0.);
  if (has_amp) {
    es = reinterpret_cast<const double2*>(s.Es_ri)[i];
    ep = reinterpret_cast<const double2*>(s.Ep_ri)[i];
  }
  store_ray(o, i, zero_xyz ? 0. : s.x[i], zero_xyz ? 0. : s.y[i], zero_xyz ? 0. : s.z[i],
            s.a[i], s.b[i], s.c[i], s.path[i], s.E[i], s.Jss[i], s.Jpp[i], js.x, js.y, st,
            es.x, es.y, ep.x, ep.y, has_amp);
}

// What a pass does with the outgoing ("virgin" / global) record of a ray besides storing it: a
// consumer fused into the producer (N1 of SURVEY 8f). The script hands the global beam of an
// element straight on to a screen more often than not; the screen's image is then made here,
// from the registers, instead of by a pass of its own that reads the beam back (100 B per ray
// less to read, and 100 B less to write when nobody else wants the global beam: vb.x null).
// Round 6: apertures that follow the element directly (aperture.propagate(gb) before anybody
// else looks at gb) ride as well: the state of the outgoing record is what they leave in
// gb.state (apertures.py:373) -- four compares per aperture instead of a launch that reads 52 B
// per ray -- and a screen behind them sees the marked states (`ap`, screen_impl.h).
struct NoConsumer {
  static constexpr bool ON = false;
};
// (kernarg_at<T>(offset): csrc/kernarg.h -- the tail's records are read where they are used)
struct ScreenConsumer {        // [apertures ->] Screen.expose, flat screens (screen_impl.h):
  xrt_hip_screen S;            // the record as the host fills it
  xrt_hip_beam out;            // the image (x null: no screen, or nobody wants the image itself)
  TailApertures ap;
};
// where the tail's records lie in the kernel's argument record
struct TailAt {
  unsigned vb, S, out, ap, Q;
};
#define XRT_TAIL_AT(ARGS, VB)                                                            \
  TailAt {                                                                               \
    (unsigned)offsetof(ARGS, VB), (unsigned)offsetof(ARGS, scr.S),                        \
        (unsigned)offsetof(ARGS, scr.out), (unsigned)offsetof(ARGS, scr.ap),              \
        (unsigned)offsetof(ARGS, Q)                                                      \
  }

struct LateScreen {
  static constexpr bool ON = true;
  static constexpr bool TAKES = true;
  TailAt at;
  __device__ __forceinline__ const xrt_hip_beam& vb() const {
    return kernarg_at<xrt_hip_beam>(at.vb);
  }
  __device__ __forceinline__ int mark(double x, double y, double z, double a, double b, double c,
                                      int st) const {
    const TailApertures& A = kernarg_at<TailApertures>(at.ap);
    return A.n ? apertures_mark(A, x, y, z, a, b, c, st) : st;
  }
  __device__ __forceinline__ void take(int64_t i, double x, double y, double z, double a,
                                       double b, double c, double path, double E, double Jss,
                                       double Jpp, double Jsr, double Jsi, int st, double Esr,
                                       double Esi, double Epr, double Epi, bool has_amp) const {
    const xrt_hip_beam& o = kernarg_at<xrt_hip_beam>(at.out);
    if (o.x)
      expose_flat_store(kernarg_at<xrt_hip_screen>(at.S), o, i, x, y, z, a, b, c, path, E, Jss, Jpp,
                        Jsr, Jsi, st, Esr, Esi, Epr, Epi, has_amp);
  }
};

// ... and with a PLOT behind the screen (run_ray_tracing's accumulate_plot of that image,
// plot_tail.h): the image is stored only if somebody else reads it (out.x), its ray's weight,
// hue and bins are left in *stash* -- registers of the kernel -- for the wave to sort and write
// when all its lanes are back together (plot_tail_emit at the end of the kernel: take() runs
// inside divergent branches).
struct LateScreenPlot {
  static constexpr bool ON = true;
  static constexpr bool TAKES = true;
  TailAt at;
  PlotStash* stash;
  __device__ __forceinline__ const xrt_hip_beam& vb() const {
    return kernarg_at<xrt_hip_beam>(at.vb);
  }
  __device__ __forceinline__ int mark(double x, double y, double z, double a, double b, double c,
                                      int st) const {
    const TailApertures& A = kernarg_at<TailApertures>(at.ap);
    return A.n ? apertures_mark(A, x, y, z, a, b, c, st) : st;
  }
  __device__ __forceinline__ void take(int64_t i, double x, double y, double z, double a,
                                       double b, double c, double path, double E, double Jss,
                                       double Jpp, double Jsr, double Jsi, int st, double Esr,
                                       double Esi, double Epr, double Epi, bool has_amp) const {
    const ImageRay r = expose_flat(kernarg_at<xrt_hip_screen>(at.S), x, y, z, a, b, c, st);
    const xrt_hip_beam& o = kernarg_at<xrt_hip_beam>(at.out);
    if (o.x) store_image(o, i, r, path, E, Jss, Jpp, Jsr, Jsi, Esr, Esi, Epr, Epi, has_amp);
    *stash = plot_tail_take(kernarg_at<PlotTail>(at.Q), r.x, 0., r.z, r.a, r.b, r.c, path + r.path,
                            E, Jss, Jpp, Jsr, Jsi, r.st);
  }
};
// Apertures alone (nothing exposes the beam behind them): the kernel whose registers are all
// taken -- the DCM's, 119 of 128 -- stores the record first and forms the marks from position and
// direction when everything else is on its way (emit_marked), and has no image arithmetic
// compiled in.
struct LateMarks {
  static constexpr bool ON = true;
  static constexpr bool TAKES = false;
  TailAt at;
  __device__ __forceinline__ const xrt_hip_beam& vb() const {
    return kernarg_at<xrt_hip_beam>(at.vb);
  }
  __device__ __forceinline__ int mark(double x, double y, double z, double a, double b, double c,
                                      int st) const {
    const TailApertures& A = kernarg_at<TailApertures>(at.ap);
    return A.n ? apertures_mark(A, x, y, z, a, b, c, st) : st;
  }
  __device__ __forceinline__ void take(int64_t, double, double, double, double, double, double,
                                       double, double, double, double, double, double, int,
                                       double, double, double, double, bool) const {}
};

// Nothing behind the element at all, but the global beam's array pointers read where the record
// is stored (the DCM's plain kernel: 85 SGPRs waited in VGPR lanes through both crystals)
struct LatePlain {
  static constexpr bool ON = true;
  static constexpr bool TAKES = false;
  TailAt at;
  __device__ __forceinline__ const xrt_hip_beam& vb() const {
    return kernarg_at<xrt_hip_beam>(at.vb);
  }
  __device__ __forceinline__ int mark(double, double, double, double, double, double,
                                      int st) const {
    return st;
  }
  __device__ __forceinline__ void take(int64_t, double, double, double, double, double, double,
                                       double, double, double, double, double, double, int,
                                       double, double, double, double, bool) const {}
};

// the outgoing record of a ray with a consumer behind the element: marks, store, hand-over
template <class CONS>
__device__ __forceinline__ void emit_marked(const CONS& cons, int64_t i,
                                            double x, double y, double z, double a, double b,
                                            double c, double path, double E, double Jss,
                                            double Jpp, double Jsr, double Jsi, int st, double Esr,
                                            double Esi, double Epr, double Epi, bool has_amp) {
  const xrt_hip_beam& vb = cons.vb();           // (the global beam's arrays: fetched here)
  if constexpr (!CONS::TAKES) {
    if (vb.x) {
      store_ray<false, false>(vb, i, x, y, z, a, b, c, path, E, Jss, Jpp, Jsr, Jsi, st, Esr, Esi,
                              Epr, Epi, has_amp);
      __builtin_nontemporal_store(cons.mark(x, y, z, a, b, c, st), &vb.state[i]);
    }
  } else {
    st = cons.mark(x, y, z, a, b, c, st);
    if (vb.x)
      store_ray(vb, i, x, y, z, a, b, c, path, E, Jss, Jpp, Jsr, Jsi, st, Esr, Esi, Epr, Epi,
                has_amp);
    cons.take(i, x, y, z, a, b, c, path, E, Jss, Jpp, Jsr, Jsi, st, Esr, Esi, Epr, Epi, has_amp);
  }
}

// ... of a ray that goes through as it came (record `i` of `s`, read again here: a copy kept in
// registers from the head of the kernel would live across both solves)
template <class CONS>
__device__ __forceinline__ void copy_marked(const CONS& cons, const xrt_hip_beam& s, int64_t i,
                                            int st, bool has_amp) {
  const double2 js = reinterpret_cast<const double2*>(s.Jsp_ri)[i];
  double2 es = make_double2(0., 0.), ep = make_double2(0., 0.);
  if (has_amp) {
    es = reinterpret_cast<const double2*>(s.Es_ri)[i];
    ep = reinterpret_cast<const double2*>(s.Ep_ri)[i];
  }
  const double x = s.x[i], y = s.y[i], z = s.z[i], a = s.a[i], b = s.b[i], c = s.c[i];
  const double path = s.path[i], E = s.E[i], Jss = s.Jss[i], Jpp = s.Jpp[i];
  emit_marked(cons, i, x, y, z, a, b, c, path, E, Jss, Jpp, js.x, js.y, st, es.x, es.y, ep.x, ep.y,
              has_amp);
}

__device__ __forceinline__ PlotStash no_plot_ray(const PlotTail& Q) {
  PlotStash s;
  s.w = s.hue = 0.;
  s.word = 0;
  s.tile = Q.T;
  s.st = 0;
  return s;
}

// everything after the solve for one entering ray: state, finish, both stores.
// QREADY: the ray's fields come in qin instead of from `in`. VREC: the outgoing
// ("virgin") record is handed back in registers instead of being stored, with `kept` =
// the ray ended in state 1 or 2 (otherwise the caller restores it).
struct Completed {
  bool kept;
  Rec v;
};
template <class K, bool QREADY = false, bool VREC = false, class CONS = NoConsumer>
__device__ __forceinline__ Completed complete_ray(
    const xrt_hip_pass& P, const xrt_hip_material& M, const GStat& g, const xrt_hip_beam& in,
    const xrt_hip_beam& restore, const xrt_hip_beam& lb, const xrt_hip_beam& vb, double* theta,
    int64_t i, const LocalRay& r, const Hit& h, int st, bool has_amp, int own_sign = 0,
    double* bdn_out = nullptr, RayIn qin = RayIn(), const cplx* npre = nullptr,
    const LocalRay* raw = nullptr, XtalEnergy* xe = nullptr, bool xe_ready = false,
    ProbeClock* pc = nullptr, const CONS& cons = CONS()) {
  Completed res;
  res.kept = false;
  RayIn q;
  if (QREADY) {
    q = qin;
  } else {
    q.path = in.path[i];
    q.E = in.E[i];
    if (early_fields<K>()) load_fields(in, i, has_amp, q);
  }
  double la = r.a, lbb = r.b, lc = r.c, th = 0.;
  RayIn lo;
  double vJss, vJpp, vJsr, vJsi, vEsr, vEsi, vEpr, vEpi;
#ifdef XRT_PROBE_NULL_COMPUTE
  if (false) {
#else
  if (st == 1) {
#endif
    const Finished fin = finish_ray<K, QREADY>(P, M, g, r, h, q, in, i, has_amp, own_sign, npre,
                                               xe, xe_ready);
    if (bdn_out) *bdn_out = fin.bdn;
    la = fin.a;
    lbb = fin.b;
    lc = fin.c;
    th = fin.theta;
    lo = fin.lo;
    vJss = fin.vJss;
    vJpp = fin.vJpp;
    vJsr = fin.vJsr;
    vJsi = fin.vJsi;
    vEsr = fin.vEsr;
    vEsi = fin.vEsi;
    vEpr = fin.vEpr;
    vEpi = fin.vEpi;
  } else {
    if (!QREADY && !early_fields<K>()) load_fields(in, i, has_amp, q);
    lo = q;
    vJss = q.Jss;
    vJpp = q.Jpp;
    vJsr = q.Jsr;
    vJsi = q.Jsi;
    vEsr = q.Esr;
    vEsi = q.Esi;
    vEpr = q.Epr;
    vEpi = q.Epi;
  }
#ifdef XRT_PROBE_TIMING
  if (pc) XRT_TICK(*pc, 3, la + vJss + lo.Jpp);
#endif
  if (theta) theta[i] = th;
  store_ray<optional_local<K>()>(lb, i, h.x, h.y, h.z, la, lbb, lc, lo.path, lo.E, lo.Jss, lo.Jpp,
                                 lo.Jsr, lo.Jsi, st, lo.Esr, lo.Esi, lo.Epr, lo.Epi, has_amp);
  const bool keep = P.only_state1_out ? (st == 1) : (st == 1 || st == 2);
  if (!keep && VREC) return res;
  res.kept = keep;
  double x, y, z;
  int vst = st;
  if (!keep) {
    // reflect.py:131-134: everything but the state comes from `restore`. The record goes
    // into the registers the kept rays use, so that the outgoing beam is written by ONE
    // store per array and wave (77 % of the cfg2 waves hold a lost / over lane; storing
    // those lanes in a branch of their own cost 3.5-5 % of the kernel).
    if (QREADY && raw && restore.x == in.x) {
      // the lean kernels still hold the whole incoming record: nothing to fetch. One lane
      // of a wave asking for its record again pulls 13 lines of 128 B, by then evicted from
      // the L2, out of HBM: the kernel read 1.27 GB per 1e7 rays instead of 1.00 GB (PMC
      // FETCH_SIZE 618 800 -> 489 263 KB with this branch)
      x = raw->x;
      y = raw->y;
      z = raw->z;
      la = raw->a;
      lbb = raw->b;
      lc = raw->c;
    } else {
    x = restore.x[i];
    y = restore.y[i];
    z = restore.z[i];
    la = restore.a[i];
    lbb = restore.b[i];
    lc = restore.c[i];
    lo.path = restore.path[i];
    lo.E = restore.E[i];
    vJss = restore.Jss[i];
    vJpp = restore.Jpp[i];
    const double2 js = reinterpret_cast<const double2*>(restore.Jsp_ri)[i];
    vJsr = js.x;
    vJsi = js.y;
    vEsr = vEsi = vEpr = vEpi = 0.;
    if (has_amp) {
      const double2 es = reinterpret_cast<const double2*>(restore.Es_ri)[i];
      const double2 ep = reinterpret_cast<const double2*>(restore.Ep_ri)[i];
      vEsr = es.x;
      vEsi = es.y;
      vEpr = ep.x;
      vEpi = ep.y;
    }
    }
    if (P.force_lost_out) vst = P.lost_num;
  } else {
    // back to the virgin local frame, reflect.py:1115-1132
    x = h.x + P.shift[0];
    y = h.y + P.shift[1];
    z = h.z + P.shift[2];
    rotate3(P.to_virgin, x, y, z);
    rotate3(P.to_virgin, la, lbb, lc);
    if (P.out_to_global) {  // beamline.py:267-287
      if (P.sin_az != 0.) {
        const double an = P.cos_az * la - (-P.sin_az) * lbb,
                     bn = (-P.sin_az) * la + P.cos_az * lbb;
        la = an;
        lbb = bn;
        const double xn = P.cos_az * x - (-P.sin_az) * y, yn = (-P.sin_az) * x + P.cos_az * y;
        x = xn;
        y = yn;
      }
      x += P.center[0];
      y += P.center[1];
      z += P.center[2];
    }
  }
  if (VREC) {
    res.v.x = x;
    res.v.y = y;
    res.v.z = z;
    res.v.a = la;
    res.v.b = lbb;
    res.v.c = lc;
    res.v.f.path = lo.path;
    res.v.f.E = lo.E;
    res.v.f.Jss = vJss;
    res.v.f.Jpp = vJpp;
    res.v.f.Jsr = vJsr;
    res.v.f.Jsi = vJsi;
    res.v.f.Esr = vEsr;
    res.v.f.Esi = vEsi;
    res.v.f.Epr = vEpr;
    res.v.f.Epi = vEpi;
    res.v.st = st;
    return res;
  }
  if constexpr (CONS::ON)      // apertures and / or a screen behind the element
    emit_marked(cons, i, x, y, z, la, lbb, lc, lo.path, lo.E, vJss, vJpp, vJsr, vJsi, vst, vEsr, vEsi,
                vEpr, vEpi, has_amp);
  else
    store_ray(vb, i, x, y, z, la, lbb, lc, lo.path, lo.E, vJss, vJpp, vJsr, vJsi, vst, vEsr,
              vEsi, vEpr, vEpi, has_amp);
  return res;
}

template <class K, class CONS = NoConsumer>
__device__ __forceinline__ void pass_through(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                             const xrt_hip_beam& restore,
                                             const xrt_hip_beam& lb, const xrt_hip_beam& vb,
                                             double* theta, int64_t i, int st, bool has_amp,
                                             const CONS& cons = CONS()) {
  // not entering: both outputs are copies (reflect.py:104-108); dcm.py:298-303
  // zeroes the local record of rays that never reached the 2nd crystal
  if (P.zero_local_not_entering)
    copy_ray<optional_local<K>()>(lb, in, i, 0, has_amp, true);
  else
    copy_ray<optional_local<K>()>(lb, in, i, st, has_amp, false);
  int vst = P.force_lost_out ? P.lost_num : st;
  if constexpr (CONS::ON) {
    copy_marked(cons, restore, i, vst, has_amp);
  } else {
    copy_ray(vb, restore, i, vst, has_amp, false);
  }
  if (theta) theta[i] = 0.;
}

// ---------------------------------------------------------------------------
// K3 kernels
// ---------------------------------------------------------------------------
// mode: 0 = optimistic single pass (runs if g.optimistic; reports to OptStat),
//       1 = exact (a phase of reflect_exact: the statistics in g are the batch's own),
//       2 = unconditional (no statistics needed)
__device__ __forceinline__ bool fused_skips(const GStat* gp, int mode) {
  return mode == 0 && !gp->optimistic;
}

// |direction cosine| along `axis` strictly larger than the other two
__device__ __forceinline__ bool dominates(int axis, const LocalRay& r) {
  const double fa = fabs(r.a), fb = fabs(r.b), fc = fabs(r.c);
  return axis == 0 ? (fa > fb && fa > fc) : (axis == 1 ? (fb > fa && fb > fc) : (fc > fa && fc > fb));
}

// wave-level fold of the optimistic pass's reports into this block's slot
__device__ __forceinline__ void report_opt(OptStat* slots, const SolveAux& aux, int viol) {
  double m1 = aux.adz1, m2 = aux.adz2;
  // np.max hands a NaN on to the secant-or-Brent comparison, fmax would drop it: let
  // the exact sequence decide
  viol |= isnan(m1) || isnan(m2);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    m1 = fmax(m1, __shfl_xor(m1, off));
    m2 = fmax(m2, __shfl_xor(m2, off));
  }
  const bool any = __any(viol | aux.escaped);
  if ((threadIdx.x & 63) == 0) {
    OptStat* o = slots + (blockIdx.x % REFLECT_OPT_SLOTS);
    // results unused: no-return atomics, the wave does not wait for them
    if (m1 > 0.) (void)atomicMax(&o->maxdz1, (unsigned long long)__double_as_longlong(m1));
    if (m2 > 0.) (void)atomicMax(&o->maxdz2, (unsigned long long)__double_as_longlong(m2));
    if (any) o->viol = 1;
  }
}

// A crystal pass raises GStat::any_neg / any_pos for the signs of beamInDotNormal it saw.
// Per wave, not per block (a block barrier at the end of the kernel keeps every wave's
// registers allocated until the slowest wave of the block is through), and only by waves
// that started before the flag was up (`seen`, read when the wave starts: one store per
// wave and side would be 150 000 stores into one cache line).
__device__ __forceinline__ void raise_sign_flags(int* __restrict__ flags, int seen_neg,
                                                 int seen_pos, int neg, int pos) {
  const bool wneg = __any(neg), wpos = __any(pos);
  if ((threadIdx.x & 63) == 0) {
    if (wneg && !seen_neg) flags[0] = 1;   // same-value racing stores; read only by the
    if (wpos && !seen_pos) flags[1] = 1;   // kernels that follow
  }
}

// one ray of the fused solve + finish; own_sign / neg / pos serve the crystal variant
template <class K, int mode, bool XTAL, class CONS = NoConsumer>
__device__ __forceinline__ void fused_ray(const xrt_hip_pass& P, const xrt_hip_material& M,
                                          const xrt_hip_beam& in, const xrt_hip_beam& restore,
                                          const xrt_hip_beam& lb, const xrt_hip_beam& vb,
                                          double* theta, const GStat& g, OptStat* opt,
                                          int64_t i, const RayRequest& req, int& neg,
                                          int& pos, const CONS& cons = CONS()) {
  const bool has_amp = in.Es_ri != nullptr;
  ProbeClock pc;
  XRT_TICK(pc, 0, 0.);
  const int st0 = req.st0;
  const LocalRay raw = req.raw;
  LocalRay r = raw;
  local_pos(P, r.x, r.y, r.z);
  local_dir(P, r.a, r.b, r.c);
  const bool active = i < in.n && entering(P, st0);
  // Fresnel coatings: the refractive index now, so that its table look-up (dependent
  // loads) is in flight during the root solve
#ifdef XRT_NO_NPRE
  constexpr bool NPRE = false;
#else
  constexpr bool NPRE = K::PLAIN && K::MK == XRT_HIP_MAT_MIRROR;
#endif
  cplx npre = C(1., 0.);
#ifndef XRT_PROBE_NO_AMPL
  if (NPRE && active) npre = refractive_index(M, req.q.E, window_of(g));
#endif
#ifndef XRT_LATE_FIELDS
  // the lean kernels have the registers to hold the WHOLE input record during the solve
  // (measured on cfg2: 0.71 -> 0.68 ms, at four waves per SIMD instead of five)
  const RayIn qpre = req.q;
#endif
  if (i < in.n && !active)
    pass_through<K, CONS>(P, in, restore, lb, vb, theta, i, st0, has_amp, cons);
  XRT_TICK(pc, 1, r.x + r.a + r.z);
  Hit h;
  if (mode == 0) {
    // solve, then report while the wave is convergent and before the amplitude code
    // needs the registers
    SolveAux aux;
    int viol = 0;
    if (active) {
      h = solve_ray<K, true>(P, g, r, &aux);
      // the axis stands only if its max beats the other two maxima over the state-1 rays
      viol = st0 == 1 && !dominates(g.axis, r);
    }
    report_opt(opt, aux, viol);
  } else if (active) {
    h = solve_ray<K>(P, g, r);
  }
  XRT_TICK(pc, 2, active ? h.t : 0.);
  if (active) {
    int st = rays_good<K>(P, h.x, h.y);
    if constexpr (K::RAYG)
      if (P.state_ray && st == 1) st = P.state_ray[i];   // zones of a general FZP
    if (h.lost) st = P.lost_num;
    if (XTAL) {
      double bdn = 0.;
      complete_ray<K, false, false, CONS>(P, M, g, in, restore, lb, vb, theta, i, r, h, st, has_amp,
                                       1, &bdn, RayIn(), nullptr, nullptr, nullptr, false,
                                       nullptr, cons);
      neg |= st == 1 && bdn < 0.;
      pos |= st == 1 && !(bdn < 0.);
    } else {
#ifndef XRT_LATE_FIELDS
      if (early_fields<K>())
        complete_ray<K, true, false, CONS>(P, M, g, in, restore, lb, vb, theta, i, r, h, st,
                                        has_amp, 0, nullptr, qpre, NPRE ? &npre : nullptr, &raw,
                                        nullptr, false, &pc, cons);
      else
#endif
      complete_ray<K, false, false, CONS>(P, M, g, in, restore, lb, vb, theta, i, r, h, st,
                                       has_amp, 0, nullptr, RayIn(), NPRE ? &npre : nullptr,
                                       nullptr, nullptr, false, nullptr, cons);
    }
  }
#ifdef XRT_PROBE_TIMING
  XRT_TICK(pc, 4, 0.);                     // stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  XRT_TICK(pc, 5, 0.);                     // stores acknowledged
  pc.flush(6);
#endif
}

#ifdef XRT_FUSED_EARLY_ARGS     // (A/B: the arguments one by one, all loaded in the entry block)
template <class K, int mode>
__global__ __launch_bounds__(REFLECT_FUSED_BLOCK, K::WAVES) void reflect_fused(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_beam in, xrt_hip_beam restore,
    xrt_hip_beam lb, xrt_hip_beam vb, double* theta, const GStat* __restrict__ gp,
    OptStat* __restrict__ opt) {
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
#ifdef XRT_LATE_FIELDS
  const RayRequest req = request_ray<false>(in, i, in.Es_ri != nullptr);
#else
  const RayRequest req = request_ray<early_fields<K>()>(in, i, in.Es_ri != nullptr);
#endif
  if (fused_skips(gp, mode)) return;
  const GStat g = *gp;
  int neg = 0, pos = 0;
  fused_ray<K, mode, false>(P, M, in, restore, lb, vb, theta, g, opt, i, req, neg, pos);
}
#endif

// The same pass with a screen in its tail: OE.reflect whose global beam goes straight into
// Screen.expose. `vb` with null arrays: the global beam itself is not wanted (nothing but the
// screen reads it). The optimistic form only (mode 0 / 2): a contradicted pass is redone by
// reflect_redo_scr into the real `vb`, which then makes the image from that. ONE record of
// arguments for the four kernels with a tail, so that the tail's members can be read late
// (kernarg_at; `vb`, `scr` and `Q` are never named in the kernels).
struct FusedTailArgs {
  xrt_hip_pass P;
  xrt_hip_material M;
  xrt_hip_geosource G;         // (the kernels with the source in their head)
  xrt_hip_beam in, restore, lb, vb;
  double* theta;
  const GStat* gp;
  OptStat* opt;
  ScreenConsumer scr;
  PlotTail Q;                  // (the kernels with a plot behind the screen)
};
// (what fused_ray is given in vb's place: never looked at when a consumer rides)
__device__ __forceinline__ xrt_hip_beam no_beam_here() {
  xrt_hip_beam b = {};
  return b;
}

// The lean kernels' plain pass on the same record (the headline kernels of reflect_hot.hip; an
// overload of reflect_fused, so that profiles and tools know the kernel by the name it always
// had): nothing behind the element, the global beam's array pointers read where the record is
// stored.
template <class K, int mode>
__global__ __launch_bounds__(REFLECT_FUSED_BLOCK, K::WAVES) void reflect_fused(
    FusedTailArgs A) {
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
#ifdef XRT_LATE_FIELDS
  const RayRequest req = request_ray<false>(A.in, i, A.in.Es_ri != nullptr);
#else
  const RayRequest req = request_ray<early_fields<K>()>(A.in, i, A.in.Es_ri != nullptr);
#endif
  if (fused_skips(A.gp, mode)) return;
  const GStat g = *A.gp;
  int neg = 0, pos = 0;
  const LatePlain cons{XRT_TAIL_AT(FusedTailArgs, vb)};
  fused_ray<K, mode, false, LatePlain>(A.P, A.M, A.in, A.restore, A.lb, no_beam_here(), A.theta, g,
                                       A.opt, i, req, neg, pos, cons);
}

template <class K, int mode>
__global__ __launch_bounds__(REFLECT_FUSED_BLOCK, K::WAVES) void reflect_fused_scr(
    FusedTailArgs A) {
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
  const RayRequest req = request_ray<early_fields<K>()>(A.in, i, A.in.Es_ri != nullptr);
  if (fused_skips(A.gp, mode)) return;
  const GStat g = *A.gp;
  int neg = 0, pos = 0;
  const LateScreen cons{XRT_TAIL_AT(FusedTailArgs, vb)};
  fused_ray<K, mode, false, LateScreen>(A.P, A.M, A.in, A.restore, A.lb, no_beam_here(), A.theta,
                                        g, A.opt, i, req, neg, pos, cons);
}

// the source's ray of this lane as the pass's request (source_impl.h)
__device__ __forceinline__ RayRequest made_request(const xrt_hip_geosource& G, int64_t i,
                                                   int64_t n, bool has_amp) {
  const gen::GenRay made = gen::make_ray(G, gen::call_of(G), i < n ? i : 0, has_amp);
  RayRequest req;
  req.st0 = i < n ? G.state : 0;
  req.raw.x = made.x;
  req.raw.y = made.y;
  req.raw.z = made.z;
  req.raw.a = made.a;
  req.raw.b = made.b;
  req.raw.c = made.c;
  req.q.path = 0.;
  req.q.E = made.E;
  req.q.Jss = made.r.Jss;
  req.q.Jpp = made.r.Jpp;
  req.q.Jsr = made.r.Jre;
  req.q.Jsi = made.r.Jim;
  req.q.Esr = has_amp ? made.r.Esr : 0.;
  req.q.Esi = has_amp ? made.r.Esi : 0.;
  req.q.Epr = has_amp ? made.r.Epr : 0.;
  req.q.Epi = has_amp ? made.r.Epi : 0.;
  return req;
}

// ... and with the SOURCE in its head: GeometricSource.shine -> OE.reflect -> Screen.expose as one
// pass. The ray is made in registers (source_impl.h: counter-based, so it can be made again
// whenever somebody asks for the source's beam); `in` points at a beam-sized scratch that this
// kernel neither reads nor writes -- it is filled only if the pass has to be redone exactly
// (geosource_shine_if_redo). Every ray of a source has the same state (> 0: all enter).
template <class K>
__global__ __launch_bounds__(REFLECT_FUSED_BLOCK, K::WAVES) void reflect_fused_gen_scr(
    FusedTailArgs A) {
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
  if (fused_skips(A.gp, 0)) return;
  const RayRequest req = made_request(A.G, i, A.in.n, A.in.Es_ri != nullptr);
  const GStat g = *A.gp;
  int neg = 0, pos = 0;
  const LateScreen cons{XRT_TAIL_AT(FusedTailArgs, vb)};
  fused_ray<K, 0, false, LateScreen>(A.P, A.M, A.in, A.in, A.lb, no_beam_here(), A.theta, g, A.opt,
                                     i, req, neg, pos, cons);
}

// ... and with the plot of that image behind the screen (LateScreenPlot): OE.reflect ->
// Screen.expose -> accumulate_plot as one pass; `scr.out` with null arrays: the image itself is
// not wanted either. The wave writes its rays' plot records when the pass is through.
template <class K, int mode>
__global__ __launch_bounds__(REFLECT_FUSED_BLOCK, K::WAVES) void reflect_fused_scr_plot(
    FusedTailArgs A) {
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
  const RayRequest req = request_ray<early_fields<K>()>(A.in, i, A.in.Es_ri != nullptr);
  if (fused_skips(A.gp, mode)) return;
  const GStat g = *A.gp;
  int neg = 0, pos = 0;
  const TailAt at = XRT_TAIL_AT(FusedTailArgs, vb);
  PlotStash stash = no_plot_ray(kernarg_at<PlotTail>(at.Q));
  const LateScreenPlot cons{at, &stash};
  fused_ray<K, mode, false, LateScreenPlot>(A.P, A.M, A.in, A.restore, A.lb, no_beam_here(),
                                            A.theta, g, A.opt, i, req, neg, pos, cons);
  plot_tail_emit(kernarg_at<PlotTail>(at.Q), i >> 6, stash);
}

template <class K>
__global__ __launch_bounds__(REFLECT_FUSED_BLOCK, K::WAVES) void reflect_fused_gen_scr_plot(
    FusedTailArgs A) {
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
  if (fused_skips(A.gp, 0)) return;
  const RayRequest req = made_request(A.G, i, A.in.n, A.in.Es_ri != nullptr);
  const GStat g = *A.gp;
  int neg = 0, pos = 0;
  const TailAt at = XRT_TAIL_AT(FusedTailArgs, vb);
  PlotStash stash = no_plot_ray(kernarg_at<PlotTail>(at.Q));
  const LateScreenPlot cons{at, &stash};
  fused_ray<K, 0, false, LateScreenPlot>(A.P, A.M, A.in, A.in, A.lb, no_beam_here(), A.theta, g,
                                         A.opt, i, req, neg, pos, cons);
  plot_tail_emit(kernarg_at<PlotTail>(at.Q), i >> 6, stash);
}

// ---------------------------------------------------------------------------
// Bragg-reflecting crystals in ONE pass. The reference needs a batch-global sign
// (of the mean beamInDotNormal) before it can deflect a single ray, which is why
// the crystal path was solve -> reduce -> finish. For a real beam every ray has
// the same sign, and then each ray's own sign IS the batch sign: this kernel
// assumes so, and raises GStat::any_neg / any_pos for the sides it saw. Only if both
// are up does reflect_exact run the two-pass tail.
// ---------------------------------------------------------------------------
#ifndef XRT_FUSED_EARLY_ARGS
// (on the record of arguments, the outgoing beam's pointers read where the ray is stored; the
// sign flags are GStat's any_neg / any_pos)
template <class K, int mode>
__global__ __launch_bounds__(REFLECT_FUSED_BLOCK, K::WAVES) void reflect_fused_xtal(
    FusedTailArgs A) {
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
  const RayRequest req = request_ray<false>(A.in, i, A.in.Es_ri != nullptr);
  if (fused_skips(A.gp, mode)) return;
  const GStat g = *A.gp;
  int* any_neg_pos = const_cast<int*>(&A.gp->any_neg);
  const int seen_neg = any_neg_pos[0], seen_pos = any_neg_pos[1];
  int neg = 0, pos = 0;
  const LatePlain cons{XRT_TAIL_AT(FusedTailArgs, vb)};
  fused_ray<K, mode, true, LatePlain>(A.P, A.M, A.in, A.restore, A.lb, no_beam_here(), A.theta, g,
                                      A.opt, i, req, neg, pos, cons);
  raise_sign_flags(any_neg_pos, seen_neg, seen_pos, neg, pos);
}
#endif
// A SINGLE crystal with apertures and / or a flat screen in its tail (round 6, last session:
// monochromator crystal -> slit -> fluorescent screen; flat crystals of surface family 0): the
// record of the lean kernels with a tail, the consumer in the crystal branch of fused_ray. A
// contradicted pass -- or a batch with both signs of beamInDotNormal -- is redone by
// reflect_redo_scr into the real global beam, marks and image from that.
template <class K, int mode>
__global__ __launch_bounds__(REFLECT_FUSED_BLOCK, K::WAVES) void reflect_fused_xtal_scr(
    FusedTailArgs A) {
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
  const RayRequest req = request_ray<false>(A.in, i, A.in.Es_ri != nullptr);
  if (fused_skips(A.gp, mode)) return;
  const GStat g = *A.gp;
  int* any_neg_pos = const_cast<int*>(&A.gp->any_neg);
  const int seen_neg = any_neg_pos[0], seen_pos = any_neg_pos[1];
  int neg = 0, pos = 0;
  const LateScreen cons{XRT_TAIL_AT(FusedTailArgs, vb)};
  fused_ray<K, mode, true, LateScreen>(A.P, A.M, A.in, A.restore, A.lb, no_beam_here(), A.theta, g,
                                       A.opt, i, req, neg, pos, cons);
  raise_sign_flags(any_neg_pos, seen_neg, seen_pos, neg, pos);
}
#ifdef XRT_FUSED_EARLY_ARGS
template <class K, int mode>
__global__ __launch_bounds__(REFLECT_FUSED_BLOCK, K::WAVES) void reflect_fused_xtal(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_beam in, xrt_hip_beam restore,
    xrt_hip_beam lb, xrt_hip_beam vb, double* theta, const GStat* __restrict__ gp,
    int* __restrict__ any_neg_pos, OptStat* __restrict__ opt) {
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
  const RayRequest req = request_ray<false>(in, i, in.Es_ri != nullptr);
  if (fused_skips(gp, mode)) return;
  const GStat g = *gp;
  const int seen_neg = any_neg_pos[0], seen_pos = any_neg_pos[1];
  int neg = 0, pos = 0;
  fused_ray<K, mode, true>(P, M, in, restore, lb, vb, theta, g, opt, i, req, neg, pos);
  raise_sign_flags(any_neg_pos, seen_neg, seen_pos, neg, pos);
}
#endif

// crystal path, first half: solve + state; stores t, local hit point and state,
// accumulates sum(beamInDotNormal) over the rays that hit (reflect.py:573)
template <class K>
__device__ __forceinline__ void solve_body(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                           double* ht, double* hx, double* hy, double* hz,
                                           int32_t* hst, const GStat& g,
                                           double* __restrict__ part) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  double bdn_sum = 0.;
  unsigned long long cnt = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < in.n; i += stride) {
    if (!entering(P, in.state[i])) continue;
    const LocalRay r = load_local(P, in, i);
    const Hit h = solve_ray<K>(P, g, r);
    int st = rays_good<K>(P, h.x, h.y);
    if constexpr (K::RAYG)
      if (P.state_ray && st == 1) st = P.state_ray[i];
    if (h.lost) st = P.lost_num;
    ht[i] = h.t;
    hx[i] = h.x;
    hy[i] = h.y;
    hz[i] = h.z;
    hst[i] = st;
    if (st == 1) {
      double n0 = P.n_const[0], n1 = P.n_const[1], n2 = P.n_const[2];
      if (PSURF(P) == XRT_HIP_SURF_TOROID) {
        const double R = P.surf_p[0], rr = P.surf_p[1];
        const double qx = h.x * frcp(rr);
        const double rx = 1. - qx * qx;
        const double ax = rx < 0. ? 0. : frcp(sqrt(rx));
        const double na = -qx * ax, nb = -h.y * frcp(R);
        const double inorm = frcp(sqrt(na * na + nb * nb + 1.));
        n0 = na * inorm;
        n1 = nb * inorm;
        n2 = inorm;
      } else if (PSURF(P) == XRT_HIP_SURF_BENTFLAT) {
        const double nb = -h.y * frcp(P.surf_p[0]);
        const double inorm = frcp(sqrt(nb * nb + 1.));
        n0 = 0.;
        n1 = nb * inorm;
        n2 = inorm;
      }
      double bdn = r.a * n0 + r.b * n1 + r.c * n2;
      if (bdn < -1.) bdn = -1.;
      if (bdn > 1.) bdn = 1.;
      bdn_sum += bdn;
      ++cnt;
    }
  }
  auto faddd = [](double u, double v) { return u + v; };
  auto faddu = [](unsigned long long u, unsigned long long v) { return u + v; };
  bdn_sum = block_reduce(bdn_sum, faddd, lds_d);
  cnt = block_reduce(cnt, faddu, lds_u);
  if (threadIdx.x == 0) {
    part[(int64_t)blockIdx.x * 8] = bdn_sum;
    part[(int64_t)blockIdx.x * 8 + 1] = (double)cnt;
  }
}

// second half: every ray is read completely before its outputs are written, so the
// outputs may share arrays with `in`
template <class K>
__device__ __forceinline__ void finish_body(const xrt_hip_pass& P, const xrt_hip_material& M,
                                            const xrt_hip_beam& in,
                                            const xrt_hip_beam& restore,
                                            const xrt_hip_beam& lb, const xrt_hip_beam& vb,
                                            double* theta, const double* ht, const double* hx,
                                            const double* hy, const double* hz,
                                            const int32_t* hst, const GStat& g) {
  const bool has_amp = in.Es_ri != nullptr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < in.n; i += stride) {
    const int st0 = in.state[i];
    if (!entering(P, st0)) {
      pass_through<K>(P, in, restore, lb, vb, theta, i, st0, has_amp);
      continue;
    }
    LocalRay r;
    r.x = 0.;
    r.y = 0.;
    r.z = 0.;
    r.a = in.a[i];
    r.b = in.b[i];
    r.c = in.c[i];
    local_dir(P, r.a, r.b, r.c);
    Hit h;
    h.t = ht[i];
    h.x = hx[i];
    h.y = hy[i];
    h.z = hz[i];
    h.px = h.x;   // crystals are only combined with non-parametric surfaces (capi check)
    h.py = h.y;
    h.lost = 0;
    complete_ray<K>(P, M, g, in, restore, lb, vb, theta, i, r, h, hst[i], has_amp);
  }
}

// ---------------------------------------------------------------------------
// reflect_exact: the exact sequence of one pass in ONE launch. It follows the
// optimistic kernel in the stream and first folds that kernel's reports: in the usual
// case that nothing was contradicted (and a crystal batch had one sign) every block
// returns -- one 4-us launch instead of the nine small ones the sequence used to be.
// Otherwise its blocks, all resident (<= one 256-lane block per CU), walk through the
// phases statistics -> decisions -> [bracket statistics -> fold] -> solve + finish ->
// [crystals with both signs: solve -> mean -> finish], separated by grid barriers.
// The phases are the batch-global decisions of the reference (base.py:1231-1295,
// :848-885; reflect.py:573-574). Performance matters little here; the arithmetic per
// ray is the same code as in the optimistic kernels, so both routes give the same bits.
// ---------------------------------------------------------------------------
struct PassAux {   // (the pass, material and beam records travel as kernel arguments of
                   // their own: inside one struct they end up copied to scratch memory)
  double* theta;
  GStat* g;
  double* part;                   // partial records (one per block) / report slots
  double *ht, *hx, *hy, *hz;      // crystal tail: hit records between solve and finish
  int32_t* hst;
  int aliased;                    // an output shares arrays with an input
};

// Barrier over all blocks of the launch (Guideline 16's counter form): every wave's
// stores done -> block sync -> one lane releases at agent scope, arrives, polls with
// relaxed agent loads, acquires -> block sync. The counter only grows (zeroed by the
// kernel that opens the pass); `phase` counts the barriers passed. A poll that never
// ends would take the box down with it: it gives up after ~4 s and flags the pass.
__device__ __forceinline__ void grid_barrier(GStat* g, unsigned& phase) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ++phase;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    (void)__hip_atomic_fetch_add(&g->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = phase * gridDim.x;
    unsigned spins = 0;
    while (__hip_atomic_load(&g->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(32);
      if (++spins > (1u << 22)) {
        g->hang = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// the decisions as another block of this launch wrote them (field by field: a struct
// copied through a word pointer stays in scratch memory)
__device__ __forceinline__ int ld_agent_i(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent_u(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ GStat load_gstat(const GStat* g) {
  GStat r;
  r.maxa = ld_agent(&g->maxa);
  r.maxb = ld_agent(&g->maxb);
  r.maxc = ld_agent(&g->maxc);
  r.first_good = ld_agent_u(&g->first_good);
  r.n_enter = ld_agent_u(&g->n_enter);
  r.n_main = ld_agent_u(&g->n_main);
  r.axis = ld_agent_i(&g->axis);
  r.positive = ld_agent_i(&g->positive);
  r.t1min = ld_agent(&g->t1min);
  r.t2max = ld_agent(&g->t2max);
  r.maxdz1 = ld_agent(&g->maxdz1);
  r.maxdz2 = ld_agent(&g->maxdz2);
  r.bracket_valid = ld_agent_i(&g->bracket_valid);
  r.optimistic = ld_agent_i(&g->optimistic);
  r.redo = ld_agent_i(&g->redo);
  r.any_neg = ld_agent_i(&g->any_neg);
  r.any_pos = ld_agent_i(&g->any_pos);
  r.n_good1 = ld_agent_u(&g->n_good1);
  r.sum_bdn = ld_agent(&g->sum_bdn);
  r.emin = ld_agent(&g->emin);
  r.emax = ld_agent(&g->emax);
#pragma unroll
  for (int e = 0; e < XRT_HIP_MAX_ELEM; ++e) {
    r.tab_lo[e] = ld_agent_i(&g->tab_lo[e]);
    r.tab_hi[e] = ld_agent_i(&g->tab_hi[e]);
  }
  r.win_lo = ld_agent(&g->win_lo);
  r.win_hi = ld_agent(&g->win_hi);
  r.bar = 0;
  r.hang = 0;
  r.tab_fast = nullptr;      // (the exact sequence searches the tables)
  return r;
}

// full: statistics -> decisions -> solve + finish (+ crystal tail if the batch has both
// signs); otherwise the crystal tail alone, on the decisions already in g
template <class K>
__device__ __forceinline__ void exact_pass(const xrt_hip_pass& P, const xrt_hip_material& M,
                                           const xrt_hip_beam& in, const xrt_hip_beam& restore,
                                           const xrt_hip_beam& lb, const xrt_hip_beam& vb,
                                           const PassAux& A, bool full, unsigned& phase) {
  GStat* g = A.g;
  const bool need_mean = deflects_as_crystal(M) && !M.geom_transmitted;
  const bool blazed = P.surf_kind == XRT_HIP_SURF_BLAZED;
  bool tail = !full;
  if (full) {
    if (!P.no_intersection_search) {
      // reductions: one partial record per block, folded by block 0
      if (blazed)   // closed-form intersection: no brackets at all
        stats_dir_body(P, in, A.part);
      else
        stats_dir_y_body<K>(P, in, A.part);
      grid_barrier(g, phase);
      if (blockIdx.x == 0)
        decide_axis_body(P, M, in, A.part, (int)gridDim.x, blazed ? 8 : 16, g);
      grid_barrier(g, phase);
      if (!blazed) {
        const GStat gl = load_gstat(g);
        if (!gl.bracket_valid && gl.n_enter > 0) {   // the axis is not y: second pass
          stats_bracket_body<K>(P, in, gl.axis, gl.positive, A.part);
          grid_barrier(g, phase);
          if (blockIdx.x == 0) reduce_bracket_body(A.part, (int)gridDim.x, g);
          grid_barrier(g, phase);
        }
      }
    }
    if (need_mean && A.aliased) {
      // the own-sign kernel would overwrite the rays the tail has to read again
      tail = true;
    } else {
      const GStat gl = load_gstat(g);
      int neg = 0, pos = 0;
      const int64_t stride = (int64_t)gridDim.x * blockDim.x;
      for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < in.n; base += stride) {
        const RayRequest req = request_ray<early_fields<K>()>(in, base + threadIdx.x,
                                                              in.Es_ri != nullptr);
        if (need_mean)
          fused_ray<K, 1, true>(P, M, in, restore, lb, vb, A.theta, gl, nullptr,
                                base + threadIdx.x, req, neg, pos);
        else
          fused_ray<K, 1, false>(P, M, in, restore, lb, vb, A.theta, gl, nullptr,
                                 base + threadIdx.x, req, neg, pos);
      }
      if (need_mean) {
        neg = __syncthreads_or(neg);
        pos = __syncthreads_or(pos);
        if (threadIdx.x == 0) {
          if (neg) (void)atomicOr(&g->any_neg, 1);
          if (pos) (void)atomicOr(&g->any_pos, 1);
        }
        grid_barrier(g, phase);
        const GStat g2 = load_gstat(g);
        tail = g2.any_neg && g2.any_pos;
      }
    }
  }
  if (need_mean && tail) {
    // exact two-pass sign sequence: the mean of beamInDotNormal over the rays that hit
    GStat gl = load_gstat(g);
    solve_body<K>(P, in, A.ht, A.hx, A.hy, A.hz, A.hst, gl, A.part);
    grid_barrier(g, phase);
    if (blockIdx.x == 0) reduce_bdn_body(A.part, (int)gridDim.x, g);
    grid_barrier(g, phase);
    gl = load_gstat(g);
    finish_body<K>(P, M, in, restore, lb, vb, A.theta, A.ht, A.hx, A.hy, A.hz, A.hst,
                   gl);
  }
}

#define REFLECT_EXACT_BLOCK 256
// verdict on the optimistic kernel that ran before (every block folds the 256 report
// slots itself); block 0 leaves it in g for the host's diagnostics
__device__ __forceinline__ bool exact_gate(GStat* g, const OptStat* slots, double* lds_d,
                                           int32_t* method_hint = nullptr) {
  bool full;
  if (g->optimistic) {
    double m1, m2;
    full = fold_opt(slots, lds_d, m1, m2, g->optimistic == 2);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      g->redo = full ? 1 : 0;
      if (method_hint) *method_hint = m2 > m1 * 20. ? 1 : 0;   // for the element's next pass
      if (!full) {
        g->maxdz1 = m1;   // (diagnostics; the clamp range stays open)
        g->maxdz2 = m2;
      }
    }
  } else {
    full = g->redo != 0;
  }
  return full;
}

template <class K>
__global__ __launch_bounds__(REFLECT_EXACT_BLOCK, 1) void reflect_exact(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_beam in, xrt_hip_beam restore, xrt_hip_beam lb,
    xrt_hip_beam vb, PassAux A) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  const bool need_mean = deflects_as_crystal(M) && !M.geom_transmitted;
  const bool mixed = need_mean && A.g->any_neg && A.g->any_pos;
  const bool full = exact_gate(A.g, reinterpret_cast<const OptStat*>(A.part), lds_d,
                               P.method_hint);
  if (!full && !mixed) return;
  unsigned phase = 0;
  exact_pass<K>(P, M, in, restore, lb, vb, A, full, phase);
}

// reflect_exact behind a pass that had a screen in its tail (and, SRC, the source in its head):
// ONE launch that returns at once in the usual case -- the verdict, the source's beam written
// out for the redo, the exact sequence and the image from the real global beam were four
// launches of ~5 us each, a third of a 1e5-ray iteration. Same device functions, same bits.
template <class K, bool SRC>
__global__ __launch_bounds__(REFLECT_EXACT_BLOCK, 1) void reflect_redo_scr(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_geosource G, xrt_hip_beam in, xrt_hip_beam restore,
    xrt_hip_beam lb, xrt_hip_beam vb, PassAux A, xrt_hip_screen S, xrt_hip_beam sb, PlotTail Q,
    TailApertures ap) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  // (a crystal pass that saw both signs of beamInDotNormal: the whole exact sequence as well --
  // its fused form left no hit records behind for the two-pass tail alone; the lean kernels
  // never raise the flags)
  const bool mixed = A.g->any_neg && A.g->any_pos;
  const bool full = exact_gate(A.g, reinterpret_cast<const OptStat*>(A.part), lds_d,
                               P.method_hint);
  if (!full && !mixed) return;
  unsigned phase = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (SRC) {
    // the pass made its rays in registers: now the beam itself, for every block to read
    const bool amp = in.Es_ri != nullptr;
    for (int64_t i = first; i < in.n; i += stride)
      gen::store_gen_ray(in, i, gen::make_ray(G, gen::call_of(G), i, amp), G.state, amp);
    grid_barrier(A.g, phase);
  }
  exact_pass<K>(P, M, in, restore, lb, vb, A, true, phase);
  grid_barrier(A.g, phase);
  // the image from the real global beam (sb.x null: nobody wants the image itself) and, with a
  // plot behind the screen (Q.w), its records: whole waves stay together for plot_tail_emit
  const bool has_amp = vb.Es_ri != nullptr;
  for (int64_t base = first - (threadIdx.x & 63); base < vb.n; base += stride) {
    const int64_t i = base + (threadIdx.x & 63);
    PlotStash stash = no_plot_ray(Q);
    if (i < vb.n) {
      const double2 js = reinterpret_cast<const double2*>(vb.Jsp_ri)[i];
      double2 es = make_double2(0., 0.), ep = make_double2(0., 0.);
      if (has_amp) {
        es = reinterpret_cast<const double2*>(vb.Es_ri)[i];
        ep = reinterpret_cast<const double2*>(vb.Ep_ri)[i];
      }
      const double path = vb.path[i], E = vb.E[i], Jss = vb.Jss[i], Jpp = vb.Jpp[i];
      int vst = vb.state[i];
      if (ap.n) {      // the apertures behind the element mark the real global beam
        const int marked = apertures_mark(ap, vb.x[i], vb.y[i], vb.z[i], vb.a[i], vb.b[i], vb.c[i], vst);
        if (marked != vst) vb.state[i] = marked;
        vst = marked;
      }
      const ImageRay r = expose_flat(S, vb.x[i], vb.y[i], vb.z[i], vb.a[i], vb.b[i], vb.c[i],
                                     vst);
      if (sb.x) store_image(sb, i, r, path, E, Jss, Jpp, js.x, js.y, es.x, es.y, ep.x, ep.y, has_amp);
      if (Q.w)
        stash = plot_tail_take(Q, r.x, 0., r.z, r.a, r.b, r.c, path + r.path, E, Jss, Jpp, js.x,
                               js.y, r.st);
    }
    if (Q.w) plot_tail_emit(Q, i >> 6, stash);
  }
}

// ---------------------------------------------------------------------------
// DCM.double_reflect (dcm.py:248-354) in one kernel: both crystals per ray, the
// virgin-local beam between them (100 B/ray written by the first pass and read back by
// the second when they are separate launches) stays in registers. Flat crystals.
// The second crystal's batch decisions cannot come from its first entering ray (that ray
// exists only after the first crystal): they are GUESSED from the head ray's direction
// mirrored at the first crystal, and every ray entering the second crystal checks that
// its own direction cosine along the guessed axis dominates (if it has state 1) and has
// the guessed sign -- then the first of them has it too, which is all the reference
// looks at. A contradiction on either crystal, Brent, or a crystal batch with both
// signs makes dcm_exact redo both passes exactly.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_rec(Rec& q, const xrt_hip_beam& s, int64_t i,
                                         bool has_amp) {
  q.x = s.x[i];
  q.y = s.y[i];
  q.z = s.z[i];
  q.a = s.a[i];
  q.b = s.b[i];
  q.c = s.c[i];
  q.f.path = s.path[i];
  q.f.E = s.E[i];
  load_fields(s, i, has_amp, q.f);
  q.st = s.state[i];
}

__device__ __forceinline__ void store_rec(const xrt_hip_beam& o, int64_t i, const Rec& q,
                                          int st, bool has_amp, bool zero_xyz) {
  store_ray(o, i, zero_xyz ? 0. : q.x, zero_xyz ? 0. : q.y, zero_xyz ? 0. : q.z, q.a, q.b,
            q.c, q.f.path, q.f.E, q.f.Jss, q.f.Jpp, q.f.Jsr, q.f.Jsi, st, q.f.Esr, q.f.Esi,
            q.f.Epr, q.f.Epi, has_amp);
}

#ifdef XRT_REFLECT_MAIN_TU
__global__ __launch_bounds__(REFLECT_BLOCK) void reflect_decide_dcm(
    xrt_hip_pass P1, xrt_hip_material M1, xrt_hip_pass P2, xrt_hip_material M2,
    xrt_hip_beam in, double* part1, double* part2, GStat* g1, GStat* g2) {
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  OptStat* slots2 = reinterpret_cast<OptStat*>(part2);
  for (int k = threadIdx.x; k < REFLECT_OPT_SLOTS; k += blockDim.x) {
    slots2[k].maxdz1 = 0;
    slots2[k].maxdz2 = 0;
    slots2[k].viol = 0;
  }
  if (threadIdx.x == 0) {
    gstat_reset(g2, 0);
    g2->bar = 0;
    g2->hang = 0;
  }
  int ub_lo[XRT_HIP_MAX_ELEM], ub_hi[XRT_HIP_MAX_ELEM];
  double dir0[3] = {0., 0., 0.};
  // (first_good of the first crystal = the head ray: its index comes back through g1, read by
  // every thread after the barrier below; its direction through dir0 in thread 0)
  const bool ok = decide_opt_body(P1, M1, in, reinterpret_cast<OptStat*>(part1), g1, lds_u, ub_lo,
                                  ub_hi, dir0);
  if (!ok) return;     // g1->redo is up: dcm_exact does both passes
  // the second crystal's tables are usually the first one's (one crystal cut twice): the
  // counts then hold for it as they are
  bool same = M2.kind == M1.kind && M2.nelem == M1.nelem;
  for (int e = 0; e < M2.nelem && same; ++e)
    same = M2.tab_E[e] == M1.tab_E[e] && M2.tab_n[e] == M1.tab_n[e];
  if (!same) {
    __syncthreads();
    const int64_t i0 = (int64_t)g1->first_good;
    const double E0 = in.E[i0];
    table_windows_block(M2, E0, E0, g2, lds_u, ub_lo, ub_hi);
  }
  if (threadIdx.x != 0) return;
  if (same) {
    g2->emin = g1->emin;
    g2->emax = g1->emax;
  }
  const int64_t i0 = (int64_t)g1->first_good;
  decide_windows(M2, ub_lo, ub_hi, g2, tab_fast_of(part2));
  // the head ray mirrored at the first crystal's surface, seen from the second crystal
  double a = dir0[0], b = dir0[1], c = dir0[2];
  local_dir(P1, a, b, c);
  if (M1.kind != XRT_HIP_MAT_PLATE) {     // (a plate lets it through: refraction is not a
    // change of the dominant component, and every ray checks the assumption anyway)
    const double dn = a * P1.n_const[3] + b * P1.n_const[4] + c * P1.n_const[5];
    a -= 2. * dn * P1.n_const[3];
    b -= 2. * dn * P1.n_const[4];
    c -= 2. * dn * P1.n_const[5];
  }
  rotate3(P1.to_virgin, a, b, c);
  local_dir(P2, a, b, c);
  const double m = fmax(fmax(fabs(a), fabs(b)), fabs(c));
  const int axis = m == fabs(a) ? 0 : (m == fabs(b) ? 1 : 2);
  const double comp = axis == 0 ? a : (axis == 1 ? b : c);
  g2->first_good = (unsigned long long)i0;
  g2->axis = axis;
  g2->positive = comp > 0. ? 1 : 0;
  g2->t1min = -INFINITY;
  g2->t2max = INFINITY;
  g2->maxdz1 = 1.;
  g2->maxdz2 = 0.;
  g2->optimistic = 1;
  if (P2.method_hint && *P2.method_hint) {   // (as decide_opt_body: the exit face of a plate
    g2->maxdz1 = 0.;                         // asks for Brent)
    g2->maxdz2 = 1.;
    g2->optimistic = 2;
  }
}
#endif

// (the body of the kernel: CONS = what rides behind the second crystal -- apertures and a screen,
// reflect_fused_dcm_scr -- or nothing)
template <class K, class CONS>
__device__ __forceinline__ void fused_dcm_ray(
    const xrt_hip_pass& P1, const xrt_hip_material& M1, const xrt_hip_pass& P2,
    const xrt_hip_material& M2, const xrt_hip_beam& in, const xrt_hip_beam& lo1,
    const xrt_hip_beam& lo2, const xrt_hip_beam& gb2, double* theta1, double* theta2,
    const GStat* __restrict__ g1p, const GStat* __restrict__ g2p, int* __restrict__ flags1,
    int* __restrict__ flags2, OptStat* __restrict__ opt1, OptStat* __restrict__ opt2,
    const CONS& cons) {
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
  const bool has_amp = in.Es_ri != nullptr;
  const bool live = i < in.n;
  // the record is requested before anything else is looked at (see RayRequest)
  const RayRequest req = request_ray<true>(in, i, has_amp);
  if (!g1p->optimistic) return;      // nothing could be assumed: dcm_exact does the work
  const int seen1n = flags1[0], seen1p = flags1[1], seen2n = flags2[0], seen2p = flags2[1];
  int neg1 = 0, pos1 = 0, neg2 = 0, pos2 = 0;
  Rec v = {};         // the beam between the crystals (virgin local frame), this ray
  // the anomalous scattering factors at this ray's energy, looked up (dependent trips to
  // L2) before any geometry, once for both crystals if they are cut from one table
  cplx anom1 = C(0., 0.), anom2 = C(0., 0.);
  // chi_0, chi_h, ... at this ray's energy: computed at the first crystal, used again at
  // the second if both are the same reflection of the same crystal (the rule)
  XtalEnergy xe;
  bool have_xe = false;
  const bool one_crystal =
      M1.d == M2.d && M1.chi_to_f == M2.chi_to_f && M1.f0_hkl == M2.f0_hkl &&
      M1.fact_dw == M2.fact_dw && M1.structure == M2.structure && M1.d2f_re == M2.d2f_re &&
      M1.d2f_im == M2.d2f_im && M1.Z[0] == M2.Z[0] && M1.hkl[0] == M2.hkl[0] &&
      M1.hkl[1] == M2.hkl[1] && M1.hkl[2] == M2.hkl[2] && M2.tab_E[0] == M1.tab_E[0] &&
      M2.tab_f1[0] == M1.tab_f1[0] && M2.tab_f2[0] == M1.tab_f2[0];
  RayIn q0 = RayIn();
  const int st0 = req.st0;
  const LocalRay r_in = req.raw;
  if (live) {
    const double E0 = req.q.E;
    anom1 = interp_f1f2(M1, 0, E0, window_of(*g1p));
    q0 = req.q;
    anom2 = (M2.tab_E[0] == M1.tab_E[0] && M2.tab_f1[0] == M1.tab_f1[0] &&
             M2.tab_f2[0] == M1.tab_f2[0])
                ? anom1
                : interp_f1f2(M2, 0, E0, window_of(*g2p));
  }
  // ---- first crystal ----
  {
    const GStat g = *g1p;
    LocalRay r = r_in;
    local_pos(P1, r.x, r.y, r.z);
    local_dir(P1, r.a, r.b, r.c);
    const bool active = live && entering(P1, st0);
    SolveAux aux;
    int viol = 0;
    Hit h;
    if (active) {
      h = solve_ray<K, true>(P1, g, r, &aux);
      viol = st0 == 1 && !dominates(g.axis, r);
    }
    report_opt(opt1, aux, viol);
    bool kept = false;
    if (active) {
      int st = rays_good<K>(P1, h.x, h.y);
      if (h.lost) st = P1.lost_num;
      double bdn = 0.;
      const Completed c1 = complete_ray<K, true, true>(P1, M1, g, in, in, lo1, lo1, theta1, i, r,
                                                       h, st, has_amp, 1, &bdn, q0, &anom1,
                                                       nullptr, &xe, false);
      have_xe = one_crystal && st == 1;
      kept = c1.kept;
      neg1 |= st == 1 && bdn < 0.;
      pos1 |= st == 1 && !(bdn < 0.);
      if (kept) {
        v = c1.v;
      } else {       // reflect.py:131-134: everything but the state comes from the input
        load_rec(v, in, i, has_amp);
        v.st = P1.force_lost_out ? P1.lost_num : st;
      }
    } else if (live) {
      load_rec(v, in, i, has_amp);
      if (lo1.x)
        store_rec(lo1, i, v, P1.zero_local_not_entering ? 0 : st0, has_amp,
                  P1.zero_local_not_entering != 0);
      if (theta1) theta1[i] = 0.;
      v.st = P1.force_lost_out ? P1.lost_num : st0;
    }
  }
  // ---- second crystal ----
  {
    const GStat g = *g2p;
    LocalRay r;
    r.x = v.x;
    r.y = v.y;
    r.z = v.z;
    r.a = v.a;
    r.b = v.b;
    r.c = v.c;
    const bool active = live && entering(P2, v.st);
    SolveAux aux;
    int viol = 0;
    Hit h;
    if (active) {
      local_pos(P2, r.x, r.y, r.z);
      local_dir(P2, r.a, r.b, r.c);
      h = solve_ray<K, true>(P2, g, r, &aux);
      const double comp = g.axis == 0 ? r.a : (g.axis == 1 ? r.b : r.c);
      viol = (v.st == 1 && !dominates(g.axis, r)) || ((comp > 0. ? 1 : 0) != g.positive);
    }
    report_opt(opt2, aux, viol);
    if (active) {
      int st = rays_good<K>(P2, h.x, h.y);
      if (h.lost) st = P2.lost_num;
      double bdn = 0.;
      complete_ray<K, true, false, CONS>(P2, M2, g, in, in, lo2, gb2, theta2, i, r, h, st, has_amp,
                                         1, &bdn, v.f, &anom2, nullptr, &xe, have_xe, nullptr, cons);
      neg2 |= st == 1 && bdn < 0.;
      pos2 |= st == 1 && !(bdn < 0.);
    } else if (live) {
      // dcm.py:298-303 zeroes the local record of rays that never reached the crystal;
      // the global beam gets the ORIGINAL ray back (dcm.py:330-335)
      if (lo2.x)
        store_rec(lo2, i, v, P2.zero_local_not_entering ? 0 : v.st, has_amp,
                  P2.zero_local_not_entering != 0);
      if (theta2) theta2[i] = 0.;
      const int vst = P2.force_lost_out ? P2.lost_num : v.st;
      if constexpr (CONS::ON) {
        copy_marked(cons, in, i, vst, has_amp);
      } else {
        copy_ray(gb2, in, i, vst, has_amp, false);
      }
    }
  }
  raise_sign_flags(flags1, seen1n, seen1p, neg1, pos1);
  raise_sign_flags(flags2, seen2n, seen2p, neg2, pos2);
}

#ifdef XRT_DCM_EARLY_ARGS       // (A/B: the arguments one by one, all loaded in the entry block)
template <class K>
__global__ __launch_bounds__(REFLECT_DCM_BLOCK, 4) void reflect_fused_dcm(
    xrt_hip_pass P1, xrt_hip_material M1, xrt_hip_pass P2, xrt_hip_material M2,
    xrt_hip_beam in, xrt_hip_beam lo1, xrt_hip_beam lo2, xrt_hip_beam gb2, double* theta1,
    double* theta2, const GStat* __restrict__ g1p, const GStat* __restrict__ g2p,
    int* __restrict__ flags1, int* __restrict__ flags2, OptStat* __restrict__ opt1,
    OptStat* __restrict__ opt2) {
  fused_dcm_ray<K, NoConsumer>(P1, M1, P2, M2, in, lo1, lo2, gb2, theta1, theta2, g1p, g2p, flags1,
                               flags2, opt1, opt2, NoConsumer());
}
#endif

// DCM.double_reflect with apertures and / or a screen right behind the monochromator in its tail
// (dcm.py:248-354 -> apertures.py:334-413 -> screens.py:226-302): the marks and the image are made
// from the outgoing record in registers; gb2 with null arrays = nobody else wants the global beam.
// A contradicted pass is redone by reflect_dcm_redo_scr into the real gb2, marks and image from that.
// One record of arguments (see FusedTailArgs): `gb2` and `scr` are read where they are used.
struct DcmTailArgs {
  xrt_hip_pass P1;
  xrt_hip_material M1;
  xrt_hip_pass P2;
  xrt_hip_material M2;
  xrt_hip_beam in, lo1, lo2, gb2;
  double *theta1, *theta2;
  const GStat *g1p, *g2p;
  int *flags1, *flags2;
  OptStat *opt1, *opt2;
  ScreenConsumer scr;
  int Q;                       // (no plot behind a DCM's screen: XRT_TAIL_AT's last member)
};
#ifndef XRT_DCM_EARLY_ARGS
// (the plain pass of the pair: the same record, nothing behind the second crystal)
template <class K>
__global__ __launch_bounds__(REFLECT_DCM_BLOCK, 4) void reflect_fused_dcm(DcmTailArgs A) {
  const LatePlain cons{XRT_TAIL_AT(DcmTailArgs, gb2)};
  fused_dcm_ray<K, LatePlain>(A.P1, A.M1, A.P2, A.M2, A.in, A.lo1, A.lo2, no_beam_here(), A.theta1,
                              A.theta2, A.g1p, A.g2p, A.flags1, A.flags2, A.opt1, A.opt2, cons);
}
#endif
template <class K>
__global__ __launch_bounds__(REFLECT_DCM_BLOCK, 4) void reflect_fused_dcm_scr(DcmTailArgs A) {
  const LateScreen cons{XRT_TAIL_AT(DcmTailArgs, gb2)};
  fused_dcm_ray<K, LateScreen>(A.P1, A.M1, A.P2, A.M2, A.in, A.lo1, A.lo2, no_beam_here(),
                               A.theta1, A.theta2, A.g1p, A.g2p, A.flags1, A.flags2, A.opt1, A.opt2,
                               cons);
}

// ... with apertures alone behind it (the slit behind a monochromator whose beam goes on to the
// next element): no image arithmetic in the kernel, the marks after the stores (LateMarks)
template <class K>
__global__ __launch_bounds__(REFLECT_DCM_BLOCK, 4) void reflect_fused_dcm_marks(DcmTailArgs A) {
  const LateMarks cons{XRT_TAIL_AT(DcmTailArgs, gb2)};
  fused_dcm_ray<K, LateMarks>(A.P1, A.M1, A.P2, A.M2, A.in, A.lo1, A.lo2, no_beam_here(), A.theta1,
                              A.theta2, A.g1p, A.g2p, A.flags1, A.flags2, A.opt1, A.opt2, cons);
}

// Plate.double_refract (oes/refractive.py:171-235 = dcm.py:248-354 with the plate's two faces):
// both surfaces of a flat plate -- a filter, a window -- in ONE pass, the beam inside the plate
// in registers: 416 B per ray instead of 616 (308 + 308), 200 instead of 400 without the local
// beams. The structure of reflect_fused_dcm without anything a crystal needs; decisions, reports
// and the redo (reflect_dcm_exact) are the DCM's.
template <class K>
#ifdef XRT_DCM_EARLY_ARGS
__global__ __launch_bounds__(REFLECT_DCM_BLOCK, 4) void reflect_fused_plate2(
    xrt_hip_pass P1, xrt_hip_material M1, xrt_hip_pass P2, xrt_hip_material M2,
    xrt_hip_beam in, xrt_hip_beam lo1, xrt_hip_beam lo2, xrt_hip_beam gb2, double* theta1,
    double* theta2, const GStat* __restrict__ g1p, const GStat* __restrict__ g2p,
    OptStat* __restrict__ opt1, OptStat* __restrict__ opt2) {
  const NoConsumer cons{};
#else
__global__ __launch_bounds__(REFLECT_DCM_BLOCK, 4) void reflect_fused_plate2(DcmTailArgs A) {
  // (the pair's record; the global beam's array pointers are read where the record is stored)
  const xrt_hip_pass &P1 = A.P1, &P2 = A.P2;
  const xrt_hip_material &M1 = A.M1, &M2 = A.M2;
  const xrt_hip_beam &in = A.in, &lo1 = A.lo1, &lo2 = A.lo2;
  const xrt_hip_beam gb2 = no_beam_here();
  double *theta1 = A.theta1, *theta2 = A.theta2;
  const GStat *g1p = A.g1p, *g2p = A.g2p;
  OptStat *opt1 = A.opt1, *opt2 = A.opt2;
  const LatePlain cons{XRT_TAIL_AT(DcmTailArgs, gb2)};
#endif
  typedef typename std::remove_const<decltype(cons)>::type Cons;
  const int64_t i = (int64_t)beam_block() * blockDim.x + threadIdx.x;
  const bool has_amp = in.Es_ri != nullptr;
  const bool live = i < in.n;
  const RayRequest req = request_ray<true>(in, i, has_amp);
  if (!g1p->optimistic) return;      // nothing could be assumed: dcm_exact does the work
  const int st0 = req.st0;
  const LocalRay r_in = req.raw;
  // the incoming record as the beam between the faces has it when the first face does not
  // keep the ray (reflect.py:131-134: everything but the state comes from the input)
  Rec v = {};
  v.x = r_in.x;
  v.y = r_in.y;
  v.z = r_in.z;
  v.a = r_in.a;
  v.b = r_in.b;
  v.c = r_in.c;
  v.f = req.q;
  v.st = st0;
  // the refractive index at this ray's energy: one look-up for both faces (the same material
  // on both sides of the same plate), in flight during the first root solve
#ifdef XRT_PLATE2_NO_NPRE
  const bool one_n = false;
#else
  bool one_n = M1.nelem == M2.nelem && M1.rho == M2.rho && M1.mass == M2.mass &&
               M1.n_fixed == M2.n_fixed && M1.n_re == M2.n_re && M1.n_im == M2.n_im;
  for (int e = 0; e < M1.nelem && one_n; ++e)
    one_n = M2.tab_E[e] == M1.tab_E[e] && M2.tab_f1[e] == M1.tab_f1[e] &&
            M2.tab_f2[e] == M1.tab_f2[e] && M2.tab_n[e] == M1.tab_n[e] &&
            M2.Z[e] == M1.Z[e] && M2.quantity[e] == M1.quantity[e];
#endif
  cplx n_plate = C(1., 0.);
  if (one_n && live) n_plate = refractive_index(M1, req.q.E, window_of(*g1p));
  const cplx* npre = one_n ? &n_plate : nullptr;
  // ---- front face ----
  {
    const GStat g = *g1p;
    LocalRay r = r_in;
    local_pos(P1, r.x, r.y, r.z);
    local_dir(P1, r.a, r.b, r.c);
    const bool active = live && entering(P1, st0);
    SolveAux aux;
    int viol = 0;
    Hit h;
    if (active) {
      h = solve_ray<K, true>(P1, g, r, &aux);
      viol = st0 == 1 && !dominates(g.axis, r);
    }
    report_opt(opt1, aux, viol);
    if (active) {
      int st = rays_good<K>(P1, h.x, h.y);
      if (h.lost) st = P1.lost_num;
      const Completed c1 = complete_ray<K, true, true>(P1, M1, g, in, in, lo1, lo1, theta1, i, r,
                                                       h, st, has_amp, 0, nullptr, req.q, npre);
      if (c1.kept)
        v = c1.v;
      else
        v.st = P1.force_lost_out ? P1.lost_num : st;
    } else if (live) {
      if (lo1.x)
        store_rec(lo1, i, v, P1.zero_local_not_entering ? 0 : st0, has_amp,
                  P1.zero_local_not_entering != 0);
      if (theta1) theta1[i] = 0.;
      v.st = P1.force_lost_out ? P1.lost_num : st0;
    }
  }
  // ---- back face ----
  {
    const GStat g = *g2p;
    LocalRay r;
    r.x = v.x;
    r.y = v.y;
    r.z = v.z;
    r.a = v.a;
    r.b = v.b;
    r.c = v.c;
    const bool active = live && entering(P2, v.st);
    SolveAux aux;
    int viol = 0;
    Hit h;
    if (active) {
      local_pos(P2, r.x, r.y, r.z);
      local_dir(P2, r.a, r.b, r.c);
      h = solve_ray<K, true>(P2, g, r, &aux);
      const double comp = g.axis == 0 ? r.a : (g.axis == 1 ? r.b : r.c);
      viol = (v.st == 1 && !dominates(g.axis, r)) || ((comp > 0. ? 1 : 0) != g.positive);
    }
    report_opt(opt2, aux, viol);
    if (active) {
      int st = rays_good<K>(P2, h.x, h.y);
      if (h.lost) st = P2.lost_num;
      complete_ray<K, true, false, Cons>(P2, M2, g, in, in, lo2, gb2, theta2, i, r, h, st, has_amp, 0,
                                         nullptr, v.f, npre, nullptr, nullptr, false, nullptr, cons);
    } else if (live) {
      // (as the DCM: the local record of a ray that never reached the face is zeroed, the
      // global beam gets the ORIGINAL ray back)
      if (lo2.x)
        store_rec(lo2, i, v, P2.zero_local_not_entering ? 0 : v.st, has_amp,
                  P2.zero_local_not_entering != 0);
      if (theta2) theta2[i] = 0.;
      const int vst = P2.force_lost_out ? P2.lost_num : v.st;
      if constexpr (Cons::ON)
        copy_marked(cons, in, i, vst, has_amp);
      else
        copy_ray(gb2, in, i, vst, has_amp, false);
    }
  }
}

template <class K>
__global__ __launch_bounds__(REFLECT_EXACT_BLOCK, 1) void reflect_dcm_exact(
    xrt_hip_pass P1, xrt_hip_material M1, xrt_hip_pass P2, xrt_hip_material M2, xrt_hip_beam in,
    xrt_hip_beam lo1, xrt_hip_beam lo2, xrt_hip_beam gb2, PassAux A1, PassAux A2) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  GStat *g1 = A1.g, *g2 = A2.g;
  const bool mixed = (g1->any_neg && g1->any_pos) || (g2->any_neg && g2->any_pos);
  const bool forced = !g1->optimistic;
  bool full = forced;
  if (!forced) {
    const bool f1 = exact_gate(g1, reinterpret_cast<const OptStat*>(A1.part), lds_d,
                               P1.method_hint);
    const bool f2 = exact_gate(g2, reinterpret_cast<const OptStat*>(A2.part), lds_d,
                               P2.method_hint);
    full = f1 || f2;
  }
  if (!full && !mixed) return;
  // anything off (rare): both passes as separate exact passes, the beam between them in
  // gb2's arrays (the second pass reads every ray before it overwrites it)
  unsigned phase1 = 0, phase2 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g1->redo = 1;
    g2->redo = 1;
    g2->any_neg = g2->any_pos = 0;
  }
  exact_pass<K>(P1, M1, in, in, lo1, gb2, A1, true, phase1);
  grid_barrier(g1, phase1);
  exact_pass<K>(P2, M2, gb2, in, lo2, gb2, A2, true, phase2);
}

// reflect_dcm_exact behind a pass with apertures / a screen in its tail: the same verdict and the
// same exact passes, then the marks and the image from the real global beam (as reflect_redo_scr).
template <class K>
__global__ __launch_bounds__(REFLECT_EXACT_BLOCK, 1) void reflect_dcm_redo_scr(
    xrt_hip_pass P1, xrt_hip_material M1, xrt_hip_pass P2, xrt_hip_material M2, xrt_hip_beam in,
    xrt_hip_beam lo1, xrt_hip_beam lo2, xrt_hip_beam gb2, PassAux A1, PassAux A2, xrt_hip_screen S,
    xrt_hip_beam sb, TailApertures ap) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  GStat *g1 = A1.g, *g2 = A2.g;
  const bool mixed = (g1->any_neg && g1->any_pos) || (g2->any_neg && g2->any_pos);
  const bool forced = !g1->optimistic;
  bool full = forced;
  if (!forced) {
    const bool f1 = exact_gate(g1, reinterpret_cast<const OptStat*>(A1.part), lds_d,
                               P1.method_hint);
    const bool f2 = exact_gate(g2, reinterpret_cast<const OptStat*>(A2.part), lds_d,
                               P2.method_hint);
    full = f1 || f2;
  }
  if (!full && !mixed) return;
  unsigned phase1 = 0, phase2 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g1->redo = 1;
    g2->redo = 1;
    g2->any_neg = g2->any_pos = 0;
  }
  exact_pass<K>(P1, M1, in, in, lo1, gb2, A1, true, phase1);
  grid_barrier(g1, phase1);
  exact_pass<K>(P2, M2, gb2, in, lo2, gb2, A2, true, phase2);
  grid_barrier(g2, phase2);
  const bool has_amp = gb2.Es_ri != nullptr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < gb2.n; i += stride) {
    int vst = gb2.state[i];
    const double x = gb2.x[i], y = gb2.y[i], z = gb2.z[i], a = gb2.a[i], b = gb2.b[i], c = gb2.c[i];
    if (ap.n) {
      const int marked = apertures_mark(ap, x, y, z, a, b, c, vst);
      if (marked != vst) gb2.state[i] = marked;
      vst = marked;
    }
    if (sb.x) {
      const double2 js = reinterpret_cast<const double2*>(gb2.Jsp_ri)[i];
      double2 es = make_double2(0., 0.), ep = make_double2(0., 0.);
      if (has_amp) {
        es = reinterpret_cast<const double2*>(gb2.Es_ri)[i];
        ep = reinterpret_cast<const double2*>(gb2.Ep_ri)[i];
      }
      expose_flat_store(S, sb, i, x, y, z, a, b, c, gb2.path[i], gb2.E[i], gb2.Jss[i], gb2.Jpp[i],
                        js.x, js.y, vst, es.x, es.y, ep.x, ep.y, has_amp);
    }
  }
}

}  // namespace xrt
