// fp64 building blocks for the gfx950 kernels.
//
// This translation unit is compiled with -ffp-contract=off: every a*b+c below
// is two roundings unless it is spelled fma(). The reference (numpy) never
// fuses, so the values that decide ray states / the Kirchhoff phase must be
// computed with the very same sequence of IEEE operations.
#pragma once
#include <hip/hip_runtime.h>

namespace xrt {

__device__ __forceinline__ double fma_(double a, double b, double c) {
  return __builtin_fma(a, b, c);
}

// ---------------------------------------------------------------------------
// IEEE division without its range scaling. LLVM lowers an f64 '/' on AMDGPU to
//   v_div_scale x2, v_rcp_f64, two Newton steps, q = a y, r = fma(-b, q, a),
//   v_div_fmas (= fma(r, y, q) unless the scaling flag is up), v_div_fixup
// -- 11 instructions. v_div_scale only rescales when an exponent is extreme (denominator or
// quotient near the denormal or overflow range, numerator below 2^-969), v_div_fixup only
// patches zeros / infinities / NaN / denormal results: with operands whose exponents are
// ordinary the remaining core is the same sequence of roundings, i.e. the same correctly
// rounded bits in 8 instructions (tools/probes/probe_fp64_seeds.hip: 1e9 random pairs
// with exponents -60..60 agree with '/' bit for bit). Guarding it with exponent tests costs
// more than it saves, so it is only used where the caller knows the operands are ordinary
// (a = +-0 gives +0, b = 0 gives NaN).
__device__ __forceinline__ double div_rn(double a, double b) {
  double y = __builtin_amdgcn_rcp(b);
  y = fma_(fma_(-b, y, 1.0), y, y);
  y = fma_(fma_(-b, y, 1.0), y, y);
  const double q = a * y;
  const double r = fma_(-b, q, a);
  return fma_(r, y, q);
}
// acos as numpy computes it (libm, < 1 ulp) for the grazing and Bragg angles of a beamline,
// |x| <= 0.5, without ocml's both-branches evaluation: fdlibm's rational form
// acos(x) = pi/2 - (x + x R(x^2)), R = p/q (Sun's coefficients), 25 slots instead of ~95;
// the library call for steeper incidence. Within 1 ulp (2.2e-16) of libm's value, which is
// what theta = acos(.) - pi/2 is compared at.
__device__ __forceinline__ double acos_np(double x) {
  if (__builtin_expect(!(__builtin_fabs(x) <= 0.5), 0)) return acos(x);
  const double z = x * x;
  double p = 3.47933107596021167570e-05;
  p = fma_(p, z, 7.91534994289814532176e-04);
  p = fma_(p, z, -4.00555345006794114027e-02);
  p = fma_(p, z, 2.01212532134862925881e-01);
  p = fma_(p, z, -3.25565818622400915405e-01);
  p = fma_(p, z, 1.66666666666666657415e-01);
  p *= z;
  double q = 7.70381505559019352791e-02;
  q = fma_(q, z, -6.88283971605453293030e-01);
  q = fma_(q, z, 2.02094576023350569471e+00);
  q = fma_(q, z, -2.40339491173441421878e+00);
  q = fma_(q, z, 1.0);
  double y = __builtin_amdgcn_rcp(q);
  y = fma_(fma_(-q, y, 1.0), y, y);
  y = fma_(fma_(-q, y, 1.0), y, y);
  const double r = p * y;
  return 1.57079632679489655800e+00 - (x - (6.12323399573676603587e-17 - x * r));
}

// ---------------------------------------------------------------------------
// arctan2 and cos as libm rounds them, for the arguments of a parametric (cylindrical) conic
// at grazing incidence. The reference's root solve in (s, phi, r) stops at |dz| <= 1e-12 mm: a
// ray whose residual lands within rounding noise of that threshold takes one iteration more or
// less depending on the LAST BIT of phi = arctan2(x, z') and of cos(phi), and its path then
// moves by ~1e-12 / sin(grazing angle) -- several ulp, k dt ~ 1e-5 rad at 280 eV. glibc's
// functions are correctly rounded but for ~1 % of arguments (<= 0.52 / 0.55 ulp); ocml's are
// < 1 ulp, i.e. differ on ~1/4 of them. Here: z' < 0 and |x| <= |z'| / 16 (the surface lies
// around phi = +-pi), resp. |phi| > 3: the function evaluated in double-double and rounded
// once. Everything else goes to the library.
// ---------------------------------------------------------------------------
struct dd_t {
  double hi, lo;
};
__device__ __forceinline__ dd_t dd_two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return dd_t{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd_t dd_fast_sum(double a, double b) {   // |a| >= |b|
  const double s = a + b;
  return dd_t{s, b - (s - a)};
}
__device__ __forceinline__ dd_t dd_prod(double a, double b) {
  const double p = a * b;
  return dd_t{p, fma_(a, b, -p)};
}
__device__ __forceinline__ dd_t dd_add(dd_t a, dd_t b) {
  dd_t s = dd_two_sum(a.hi, b.hi);
  s.lo += a.lo + b.lo;
  return dd_fast_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd_t dd_neg(dd_t a) { return dd_t{-a.hi, -a.lo}; }
__device__ __forceinline__ dd_t dd_mul(dd_t a, dd_t b) {
  dd_t p = dd_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return dd_fast_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd_t dd_mul_d(dd_t a, double b) {
  dd_t p = dd_prod(a.hi, b);
  p.lo += a.lo * b;
  return dd_fast_sum(p.hi, p.lo);
}
// pi = PI_HI + PI_LO to 107 bits
#define XRT_PI_HI 0x1.921fb54442d18p+1
#define XRT_PI_LO 0x1.1a62633145c07p-53

__device__ __forceinline__ double atan2_np(double y, double x) {
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  if (__builtin_expect(!(x < 0. && ay <= ax * 0.0625 && ay > ax * 0x1p-40 && ax < 0x1p500 &&
                         ax > 0x1p-500), 0))
    return atan2(y, x);
  // t = |y| / |x| to double-double: quotient, exact remainder, its quotient
  const double q1 = ay / ax;
  const double q2 = fma_(-q1, ax, ay) / ax;
  const dd_t t = dd_fast_sum(q1, q2);
  // atan t = t - t^3/3 + t^5 (1/5 - t^2/7 + ... ), t <= 1/16: the tail in plain doubles
  const dd_t t2 = dd_mul(t, t);
  const dd_t t3 = dd_mul(t2, t);
  const dd_t third = dd_t{0x1.5555555555555p-2, 0x1.5555555555555p-56};
  const double w = t2.hi;
  double tail = 1. / 19.;
  tail = fma_(-tail, w, 1. / 17.);
  tail = fma_(-tail, w, 1. / 15.);
  tail = fma_(-tail, w, 1. / 13.);
  tail = fma_(-tail, w, 1. / 11.);
  tail = fma_(-tail, w, 1. / 9.);
  tail = fma_(-tail, w, 1. / 7.);
  tail = fma_(-tail, w, 1. / 5.);
  tail *= (w * w) * t.hi;
  dd_t a = dd_add(t, dd_neg(dd_mul(t3, third)));
  a = dd_add(a, dd_t{tail, 0.});
  const dd_t phi = dd_add(dd_t{XRT_PI_HI, XRT_PI_LO}, dd_neg(a));
  const double r = phi.hi + phi.lo;
  return y < 0. ? -r : r;
}

__device__ __forceinline__ double cos_np(double phi) {
  const double ap = __builtin_fabs(phi);
  if (__builtin_expect(!(ap > 3.0 && ap < 3.3), 0)) return cos(phi);
  // d = pi - |phi| (the first difference is exact), |d| < 0.16; cos phi = -cos d
  const dd_t d = dd_fast_sum(XRT_PI_HI - ap, XRT_PI_LO);
  const dd_t d2 = dd_mul(d, d);
  const dd_t u = dd_mul_d(d2, 0.5);                 // d^2 / 2
  const dd_t sixth = dd_t{0x1.5555555555555p-3, 0x1.5555555555555p-57};
  const dd_t u2 = dd_mul(dd_mul(u, u), sixth);      // d^4 / 24
  const double w = d2.hi;
  // - d^6/720 + d^8/40320 - ... in plain doubles
  double tail = 1. / 20922789888000.;               // 1/16!
  tail = fma_(-tail, w, 1. / 87178291200.);         // 1/14!
  tail = fma_(-tail, w, 1. / 479001600.);           // 1/12!
  tail = fma_(-tail, w, 1. / 3628800.);             // 1/10!
  tail = fma_(-tail, w, 1. / 40320.);               // 1/8!
  tail = fma_(-tail, w, 1. / 720.);                 // 1/6!
  tail *= -((w * w) * w);
  dd_t c = dd_add(dd_t{1., 0.}, dd_neg(u));
  c = dd_add(c, u2);
  c = dd_add(c, dd_t{tail, 0.});
  return -(c.hi + c.lo);
}

// Correctly rounded sqrt(x) for normal positive x (no scaling / special cases:
// callers guarantee 2^-700 < x < 2^700) that ALSO hands back 1/sqrt(x) to ~1 ulp
// for free. Same Goldschmidt iteration LLVM emits for f64 sqrt on AMDGPU
// (SIISelLowering lowerFSQRTF64) minus its denormal scaling; h1 of that
// iteration converges to 0.5/sqrt(x), which the plain builtin throws away.
__device__ __forceinline__ double sqrt_rn_rinv(double x, double& rinv) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  double r0 = fma_(-h, g, 0.5);
  g = fma_(g, r0, g);
  h = fma_(h, r0, h);
  double d0 = fma_(-g, g, x);
  g = fma_(d0, h, g);
  double d1 = fma_(-g, g, x);
  g = fma_(d1, h, g);
  rinv = h + h;
  return g;
}

// The Kirchhoff inner loop's form: returns h = 1/(2 sqrt(x)) as it falls out of the
// iteration (callers fold the factor 2 into a per-sample constant) and takes ONE
// correction step. After the Goldschmidt step g and h are good to ~2^-51; the first
// correction leaves an error of ~2^-100, so the second one can only matter for an
// argument whose root lies within that of a rounding boundary. Measured on gfx950
// (tools/probes/probe_sqrt_corrections.hip, profiles/r01_sqrt_corrections.txt): over
// 5.5e11 random arguments in [1,4) and in 2^26..2^28 (r^2 of 8..16 m in mm^2) the
// second correction changed none, and all agree with the compiler's correctly
// rounded sqrt; the hard cases of tests/test_gpu_kirchhoff.py (squares and their
// neighbours) pass as well. Two VALU slots of 59.
__device__ __forceinline__ double sqrt_rn_halfinv(double x, double& hinv) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  double r0 = fma_(-h, g, 0.5);
  g = fma_(g, r0, g);
  h = fma_(h, r0, h);
  double d0 = fma_(-g, g, x);
  g = fma_(d0, h, g);
  hinv = h;
  return g;
}

// The same iteration from a caller-supplied seed y = (1 + e)/sqrt(x), h0 = y/2, instead
// of v_rsq_f64 (a quarter-rate instruction: 16 cycles per wave, as much as four fma).
// The Goldschmidt step leaves g, h with a relative error of 1.5 e^2, the correction
// 2.25 e^4: for |e| <= 2^-26 that is 2^-103, the same as after the hardware seed, so
// the result is the correctly rounded root under the same argument as above
// (tests/test_gpu_math.py compares it with np.sqrt at the seed-error limit). The
// Kirchhoff loop uses it when every distance of a launch is within 2^-27 (relative) of
// a per-sample constant: receiving points on a plane y = const facing the samples
// (paraxial geometry, the rule on a beamline), seed = 1/|y - sy| from the pack kernel,
// held in SGPRs. Six VALU slots, no transcendental.
__device__ __forceinline__ double sqrt_rn_seeded(double x, double y, double h0,
                                                 double& hinv) {
  double g = x * y;
  double r0 = fma_(-h0, g, 0.5);
  double h = fma_(h0, r0, h0);
  g = fma_(g, r0, g);
  double d0 = fma_(-g, g, x);
  g = fma_(d0, h, g);
  hinv = h;
  return g;
}

// sin/cos of (q + u) quarter turns, |u| <= 1/2: minimax polynomials in w = u^2 for
// sin(pi/2 u) and cos(pi/2 u), then the quadrant q (mod 4) swaps / negates.
__device__ __forceinline__ void sincos_quarter_turns(double u, unsigned q, double& sn,
                                                     double& cs) {
  double w = u * u;
  double ps = 0x1.e3f38399551bfp-25;
  ps = fma_(ps, w, -0x1.e30071afc3e59p-19);
  ps = fma_(ps, w, 0x1.50782fda12d96p-13);
  ps = fma_(ps, w, -0x1.32d2cce2e5b19p-8);
  ps = fma_(ps, w, 0x1.466bc677587f8p-4);
  ps = fma_(ps, w, -0x1.4abbce625be41p-1);
  ps = fma_(ps, w, 0x1.921fb54442d18p+0);
  double s = ps * u;
  double pc = 0x1.f3dbcea61b1a4p-22;
  pc = fma_(pc, w, -0x1.a6c9c1be9eb49p-16);
  pc = fma_(pc, w, 0x1.e1f4fb60281f6p-11);
  pc = fma_(pc, w, -0x1.55d3c7dbfd139p-6);
  pc = fma_(pc, w, 0x1.03c1f081b0780p-2);
  pc = fma_(pc, w, -0x1.3bd3cc9be458bp+0);
  double c = fma_(pc, w, 1.0);
  // quadrant: 0:(c,s) 1:(-s,c) 2:(-c,-s) 3:(s,-c)
  double cc = (q & 1u) ? s : c;
  double ss = (q & 1u) ? c : s;
  // sign bit flips through the high dword only
  unsigned long long cb = __double_as_longlong(cc);
  unsigned long long sb = __double_as_longlong(ss);
  cb ^= (unsigned long long)(((q + 1u) >> 1) & 1u) << 63;
  sb ^= (unsigned long long)((q >> 1) & 1u) << 63;
  cs = __longlong_as_double(cb);
  sn = __longlong_as_double(sb);
}

// sin/cos of a LARGE positive-or-negative phase phi [rad] (|phi| < 2^50),
// accurate to ~2e-16 absolute. The Kirchhoff phase k*r is ~4e11 rad, so ocml's
// generic sincos would take its Payne-Hanek path every call; here the
// reduction is two fma against a double-double 2/pi:
//   t = phi*(2/pi) in quarter turns, n = rint(t) via the 1.5*2^52 trick (its
//   low mantissa bits give the quadrant), u = phi*(2/pi) - n exactly (fma),
// then minimax polynomials in w = u^2 for sin(pi/2 u) and cos(pi/2 u), |u|<=1/2.
__device__ __forceinline__ void sincos_phase(double phi, double& sn, double& cs) {
  const double TWO_OVER_PI_HI = 0x1.45f306dc9c883p-1;
  const double TWO_OVER_PI_LO = -0x1.6b01ec5417056p-55;
  const double MAGIC = 0x1.8p52;
  double t = phi * TWO_OVER_PI_HI;
  double m = t + MAGIC;
  double n = m - MAGIC;
  unsigned q = (unsigned)__double2loint(m);
  double u = fma_(phi, TWO_OVER_PI_HI, -n);
  u = fma_(phi, TWO_OVER_PI_LO, u);
  sincos_quarter_turns(u, q, sn, cs);
}

// ---------------------------------------------------------------------------
// Table-driven variant for the Kirchhoff inner loop, |phi| < 2^42. The circle is
// cut into SINCOS_TAB_N steps of delta = 2 pi / N; (cos, sin) of every step sit in
// LDS (filled by the block with the routine above: each entry correctly rounded to
// < 1 ulp), the remainder |theta| <= delta/2 = 1.53e-3 rad needs only
//   sin theta = theta (1 - theta^2/6)           (next term 7e-17)
//   cos theta = 1 - theta^2/2 + theta^4/24       (next term 2e-20)
// and one complex product joins them: 15 VALU slots instead of 31, no quadrant
// selects. Absolute accuracy ~4e-16. The reduction is the same two-fma form with
// N/(2 pi) as a double-double.
// ---------------------------------------------------------------------------
#define SINCOS_TAB_N 2048

template <int N = SINCOS_TAB_N>
__device__ __forceinline__ void sincos_tab_fill(double2* tab) {
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const double qt = (double)j * (4.0 / N);   // quarter turns, exact
    const double n = __builtin_rint(qt);
    double s, c;
    sincos_quarter_turns(qt - n, (unsigned)(int)n, s, c);
    tab[j] = make_double2(c, s);
  }
  __syncthreads();
}

// Two of the polynomial constants, parked in VGPRs. A VOP3 instruction reads at most
// one SGPR / literal: fma(S1, w, S0) and fma(C2, w, C1) need one of their constants in
// a VGPR, and the compiler re-materialises those with a v_mov in every loop iteration
// unless they are opaque to it. One instance per kernel, made before the loop.
template <int N = SINCOS_TAB_N>
struct SinCosTabRegs {
  double s0, c1;
  __device__ __forceinline__ SinCosTabRegs() {
    constexpr double DELTA = 0x1.921fb54442d18p+1 * (2.0 / N);
    const double S0 = DELTA, C1 = -(DELTA * DELTA) / 2.0;
    asm volatile("v_mov_b64 %0, %1" : "=v"(s0) : "s"(S0));
    asm volatile("v_mov_b64 %0, %1" : "=v"(c1) : "s"(C1));
  }
};

// N = 2048 (32 KB of LDS; |phi| < 2^42): remainder |theta| <= 1.53e-3, cos needs the
// theta^4 term; absolute accuracy ~4e-16. N = 4096 (|phi| < 2^41) (64 KB; the kernels that run two blocks per CU
// anyway): |theta| <= 7.7e-4 (8.9e-4 at the largest phases), cos theta = 1 - theta^2/2
// is good to 1.5e-14 (2.6e-14 there) (the phase
// itself, k r ~ 4e11 rad, is only known to 6e-5 rad), and the byte offset of the entry
// is ONE instruction: an SDWA shift whose 16-bit destination keeps exactly the 12 index
// bits times 16.
// LEAN (N = 2048): cos theta without its theta^4 term, 2.3e-13 absolute -- for sums that are
// compared at 1e-9.
template <int N = SINCOS_TAB_N, bool LEAN = false>
__device__ __forceinline__ void sincos_tab(double phi, const double2* tab,
                                           const SinCosTabRegs<N>& k, double& sn,
                                           double& cs) {
  static_assert(N == 2048 || N == 4096, "table sizes the offset arithmetic knows");
  constexpr double STEPS_PER_RAD_HI = 0x1.45f306dc9c883p-1 * (N / 4);
  constexpr double STEPS_PER_RAD_LO = -0x1.6b01ec5417056p-55 * (N / 4);
  constexpr double DELTA = 0x1.921fb54442d18p+1 * (2.0 / N);
  constexpr double S1 = -(DELTA * DELTA * DELTA) / 6.0;
  constexpr double C2 = (DELTA * DELTA) * (DELTA * DELTA) / 24.0;
  const double MAGIC = 0x1.8p52;
  // one rounding instead of numpy-style two: n may differ from rint(t) on exact
  // ties only, which merely lets |u| reach 1/2 + 2^-53
  const double m = fma_(phi, STEPS_PER_RAD_HI, MAGIC);
  const double n = m - MAGIC;
  // byte offset of the entry
  // (asm: left to itself the compiler SLP-packs the shifts of two pairs into five ops)
  unsigned off;
  if (N == 2048)
    asm("v_lshlrev_b32 %0, 4, %1\n\tv_and_b32 %0, 0x7ff0, %0"
        : "=v"(off)
        : "v"((unsigned)__double2loint(m)));
  else
    asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD "
        "src0_sel:DWORD src1_sel:DWORD"
        : "=v"(off)
        : "v"((unsigned)__double2loint(m)));
  const double2 T = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(tab) + off);
  double u = fma_(phi, STEPS_PER_RAD_HI, -n);
  u = fma_(phi, STEPS_PER_RAD_LO, u);
  const double w = u * u;
  const double s = fma_(S1, w, k.s0) * u;
  const double c = N == 2048 && !LEAN ? fma_(fma_(C2, w, k.c1), w, 1.0) : fma_(k.c1, w, 1.0);
  cs = T.x * c;
  cs = fma_(-T.y, s, cs);
  sn = T.y * c;
  sn = fma_(T.x, s, sn);
}

// The same for phi = kr * r with the factor N / (2 pi) folded into kr on the host side of the
// loop: ks = k N / (2 pi) as a double-double (ksh, ksl), the multiplication k r is not formed at
// all (the relaxed Kirchhoff loop: one slot less, and the steps come out more accurately than
// from the rounded product k r -- but not as numpy rounds them).
template <int N = SINCOS_TAB_N>
__device__ __forceinline__ void sincos_tab_scaled(double r, double ksh, double ksl,
                                                  const double2* tab, const SinCosTabRegs<N>& k,
                                                  double& sn, double& cs) {
  static_assert(N == 2048 || N == 4096, "table sizes the offset arithmetic knows");
  constexpr double DELTA = 0x1.921fb54442d18p+1 * (2.0 / N);
  constexpr double S1 = -(DELTA * DELTA * DELTA) / 6.0;
  constexpr double C2 = (DELTA * DELTA) * (DELTA * DELTA) / 24.0;
  const double MAGIC = 0x1.8p52;
  const double m = fma_(r, ksh, MAGIC);
  const double n = m - MAGIC;
  unsigned off;
  if (N == 2048)
    asm("v_lshlrev_b32 %0, 4, %1\n\tv_and_b32 %0, 0x7ff0, %0"
        : "=v"(off)
        : "v"((unsigned)__double2loint(m)));
  else
    asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD "
        "src0_sel:DWORD src1_sel:DWORD"
        : "=v"(off)
        : "v"((unsigned)__double2loint(m)));
  const double2 T = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(tab) + off);
  double u = fma_(r, ksh, -n);
  u = fma_(r, ksl, u);
  const double w = u * u;
  const double s = fma_(S1, w, k.s0) * u;
  const double c = N == 2048 ? fma_(fma_(C2, w, k.c1), w, 1.0) : fma_(k.c1, w, 1.0);
  cs = T.x * c;
  cs = fma_(-T.y, s, cs);
  sn = T.y * c;
  sn = fma_(T.x, s, sn);
}

// table form where its bound holds, the general polynomial form otherwise (the
// branch is taken per lane; a wave whose lanes all qualify skips the slow side)
__device__ __forceinline__ void sincos_any(double phi, const double2* tab,
                                           const SinCosTabRegs<>& k, double& sn, double& cs) {
  if (__builtin_fabs(phi) < 0x1p42)
    sincos_tab(phi, tab, k, sn, cs);
  else
    sincos_phase(phi, sn, cs);
}

}  // namespace xrt
