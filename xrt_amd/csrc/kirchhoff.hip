// Fresnel-Kirchhoff diffraction integral for gfx950 (MI355X), fp64.
//
// Replaces xrt's numpy `_diffraction_integral_conv` (waves.py:834-851) and the
// OpenCL kernel `integrate_kirchhoff` (cl/diffract.cl:80-151): for every
// receiving point p and every source sample s
//     d = p - s, r = |d|, g = (k/r)(d.n/r + nl) e^{ikr}
//     S += g Es, P += g Ep, (A,B,C) += (k/r) g k(Es+Ep) d
// followed by the constant prefactors of either convention.
//
// Design (streaming form, VALU fp64):
//   * one lane = PPT receiving points, 10 fp64 accumulators each, kept in VGPRs;
//   * samples are pre-packed into 128-byte records (pack kernel below) and the
//     inner loop indexes them with a wave-uniform index, so they arrive through
//     the scalar cache into SGPRs (s_load_dwordx16): no LDS traffic, no VGPRs,
//     and VALU takes them as its one scalar operand;
//   * the grid is (pixel tiles) x (sample splits); split = blockIdx % nsplit so
//     that, with the observed block -> XCD round robin, each XCD's L2 streams
//     its own slice of the sample records; partial sums go to a workspace and a
//     tiny finalize kernel adds them in fixed order (deterministic, no atomics);
//   * r and k*r use exactly numpy's operation order with no FMA contraction
//     (k*r ~ 4e11 rad: one ulp is 6e-5 rad), sqrt is correctly rounded and also
//     yields 1/r; sincos uses a 2-fma double-double reduction onto a 2048-step
//     (cos, sin) table in LDS plus a 2-term remainder (fp64_math.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fp64_math.h"
#include "kirchhoff.h"

namespace xrt {

// ---------------------------------------------------------------------------
// pack: sample arrays -> 16-double records. Positions / normals are read with
// an element stride so that both the SoA layout (stride 1) and the reference's
// OpenCL marshalling ns x [x,y,z,0] (stride 4, waves.py:872-879) feed it.
//   [0..2] x,y,z  [3] 2k nl  [4] 4k ny  [5] k  [6,7] Es  |  [8] 2k^2  [9] 4k nx
//   [10] 4k nz  [11,12] Ep  [13,14] k*(Es+Ep)  [15] 2k
// (the first 72 bytes are all the Ep == 0, planar-normal case reads)
// (1/r comes out of the sqrt iteration as h = 1/(2r): the factors 2 and the k of
// (k/r)(d.n/r + nl) = h (4k n.d h + 2k nl) are folded in here, once per sample
// instead of once per pair.)
// The kernel also classifies the sample set so that the main kernel can take a
// shorter instruction stream when it is safe: flags bit 0 = some Ep != 0,
// bit 1 = some normal has an x or z component.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(256) void kirchhoff_pack(
    int64_t ns, const double* __restrict__ sx, const double* __restrict__ sy,
    const double* __restrict__ sz, int pstride, const double* __restrict__ nx,
    const double* __restrict__ ny, const double* __restrict__ nz, int nstride,
    const double* __restrict__ nl, const double* __restrict__ k,
    const double2* __restrict__ Es, const double2* __restrict__ Ep,
    double* __restrict__ rec, unsigned* __restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned f = 0;
  unsigned long long kabs = 0, sabs = 0;
  if (i < ns) {
    double2 es = Es[i], ep = Ep[i];
    double kk = k[i];
    double2* o = reinterpret_cast<double2*>(rec + i * KIRCHHOFF_REC_DOUBLES);
    const int64_t ip = i * pstride, in = i * nstride;
    const double vx = nx[in], vy = ny[in], vz = nz[in];
    const double k2 = 2. * kk;
    o[0] = make_double2(sx[ip], sy[ip]);
    o[1] = make_double2(sz[ip], k2 * nl[i]);
    o[2] = make_double2(k2 * (2. * vy), kk);
    o[3] = es;
    o[4] = make_double2(k2 * kk, k2 * (2. * vx));
    o[5] = make_double2(k2 * (2. * vz), ep.x);
    // numpy: k**2/(4pi) * (Es+Ep) * U / r ; the sum Es+Ep is formed first there too
    o[6] = make_double2(ep.y, kk * (es.x + ep.x));
    o[7] = make_double2(kk * (es.y + ep.y), k2);
    if (ep.x != 0. || ep.y != 0.) f |= KIRCHHOFF_FLAG_EP;
    if (vx != 0. || vz != 0.) f |= KIRCHHOFF_FLAG_NXZ;
    // ingredients of the bound |k r| <= kmax (|p|_1 + |s|_1) the main kernel checks
    // before it trusts the table-driven sincos (non-negative doubles order like
    // their bit patterns)
    kabs = __double_as_longlong(fabs(kk));
    sabs = __double_as_longlong(fabs(sx[ip]) + fabs(sy[ip]) + fabs(sz[ip]));
  }
  f = __builtin_amdgcn_readfirstlane(__reduce_or_sync(~0ull, f));
  kabs = wave_max_u64(kabs);
  sabs = wave_max_u64(sabs);
  if ((threadIdx.x & 63) == 0) {
    if (f) atomicOr(flags, f);
    unsigned long long* bound = reinterpret_cast<unsigned long long*>(flags) + 1;
    atomicMax(bound, kabs);
    atomicMax(bound + 1, sabs);
  }
}

// ---------------------------------------------------------------------------
// main streaming kernel
// ---------------------------------------------------------------------------
struct Acc {
  double sr, si, pr, pi, ar, ai, br, bi, cr, ci;
};

// HAS_P: the p-polarised source field is present. GEN_N: normals are general
// (otherwise every normal is (0, ny, 0), the aperture / screen / source case of
// waves.py:687-689 — d.n collapses to dy*ny).
// TAB: sincos through the LDS table (|k r| < 2^42 is guaranteed by the caller).
//
// The update of one (receiving point, sample) pair comes in two halves so that the
// loop can put its scalar prefetch between them (see stream_loop): pair_head ends
// with the first use of the LDS table entry, pair_tail is pure accumulation.
struct Mid {
  double dx, dy, dz, gr, gi, h;
};

template <bool GEN_N, bool TAB>
__device__ __forceinline__ Mid pair_head(double px, double py, double pz,
                                         const double (&r)[KIRCHHOFF_REC_DOUBLES],
                                         const double2* tab, const SinCosTabRegs& kreg) {
  const double sx = r[0], sy = r[1], sz = r[2], knl = r[3];
  const double kny = r[4], k = r[5], knx = r[9], knz = r[10];
  Mid m;
  // --- bit-exact part (numpy order, no contraction) ---
  m.dx = px - sx;
  m.dy = py - sy;
  m.dz = pz - sz;
  const double s2 = (m.dx * m.dx + m.dy * m.dy) + m.dz * m.dz;
  // h = 1/(2r)
  const double rr = sqrt_rn_halfinv(s2, m.h);
  const double phase = k * rr;
  // --- the rest only needs ~1e-16 relative accuracy ---
  double dn;
  if (GEN_N) {
    dn = m.dx * knx;
    dn = fma_(m.dy, kny, dn);
    dn = fma_(m.dz, knz, dn);
  } else {
    dn = m.dy * kny;
  }
  const double cr = m.h * fma_(dn, m.h, knl);     // (k/r)(d.n/r + nl)
  double sn, cs;
  if (TAB)
    sincos_tab(phase, tab, kreg, sn, cs);
  else
    sincos_phase(phase, sn, cs);
  m.gr = cr * cs;
  m.gi = cr * sn;
  return m;
}

template <bool HAS_P>
__device__ __forceinline__ void pair_tail(const Mid& m,
                                          const double (&r)[KIRCHHOFF_REC_DOUBLES], Acc& a) {
  const double esr = r[6], esi = r[7], k2k = r[8];
  const double epr = r[11], epi = r[12], qr = r[13], qi = r[14], k2 = r[15];
  const double gr = m.gr, gi = m.gi;
  double hr, hi;
  if (HAS_P) {
    a.sr = fma_(gr, esr, a.sr);
    a.sr = fma_(-gi, esi, a.sr);
    a.si = fma_(gr, esi, a.si);
    a.si = fma_(gi, esr, a.si);
    a.pr = fma_(gr, epr, a.pr);
    a.pr = fma_(-gi, epi, a.pr);
    a.pi = fma_(gr, epi, a.pi);
    a.pi = fma_(gi, epr, a.pi);
    const double kip = k2 * m.h;                  // k/r
    const double hr0 = kip * gr;
    const double hi0 = kip * gi;
    hr = hr0 * qr;
    hr = fma_(-hi0, qi, hr);
    hi = hr0 * qi;
    hi = fma_(hi0, qr, hi);
  } else {
    // Ep == 0: k(Es+Ep) = k Es, so g*Es is shared by S and by the direction term
    double wr = gr * esr;
    wr = fma_(-gi, esi, wr);
    double wi = gr * esi;
    wi = fma_(gi, esr, wi);
    a.sr += wr;
    a.si += wi;
    const double kkip = k2k * m.h;                // k^2/r
    hr = kkip * wr;
    hi = kkip * wi;
  }
  a.ar = fma_(hr, m.dx, a.ar);
  a.ai = fma_(hi, m.dx, a.ai);
  a.br = fma_(hr, m.dy, a.br);
  a.bi = fma_(hi, m.dy, a.bi);
  a.cr = fma_(hr, m.dz, a.cr);
  a.ci = fma_(hi, m.dz, a.ci);
}

// One packed sample record in SGPRs. The record index is wave-uniform, so the
// record comes through the scalar cache (s_load) and VALU takes its fields as
// scalar operands: no VGPRs, no LDS traffic for it. The loads are inline asm
// because the compiler otherwise sinks them to their first use and waits on the
// spot. gfx950 counts SMEM and LDS returns on ONE counter (lgkmcnt) and SMEM
// returns out of order, so every wait for an LDS read is an lgkmcnt(0) that also
// drains any scalar prefetch in flight. The loop therefore issues the prefetch
// right AFTER the last LDS wait of an iteration (sched_barrier pins that place)
// and, where SGPRs allow (three register sets), two records ahead: the request
// then has a whole iteration before the next LDS wait catches it.
// The compiler does not count these loads in its own s_waitcnt bookkeeping; extra
// outstanding SMEM only makes its waits stricter, and settle() -- tied to the
// destination registers -- is the wait that guards their use.
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <bool FULL>
struct SRec;

template <>
struct SRec<false> {   // 72 bytes: Ep == 0 and every normal is (0, ny, 0)
  u32x16 lo;
  u32x2 hi;
  __device__ __forceinline__ void issue(const double* p) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x40"
                 : "=&s"(lo), "=&s"(hi)
                 : "s"(p)
                 : "memory");
  }
  __device__ __forceinline__ void settle() {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lo), "+s"(hi));
  }
  __device__ __forceinline__ void unpack(double (&r)[KIRCHHOFF_REC_DOUBLES]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = __hiloint2double((int)lo[2 * i + 1], (int)lo[2 * i]);
    r[8] = __hiloint2double((int)hi[1], (int)hi[0]);
#pragma unroll
    for (int i = 9; i < KIRCHHOFF_REC_DOUBLES; ++i) r[i] = 0.;
  }
};

template <>
struct SRec<true> {    // the whole 128-byte record
  u32x16 lo, hi;
  __device__ __forceinline__ void issue(const double* p) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40"
                 : "=&s"(lo), "=&s"(hi)
                 : "s"(p)
                 : "memory");
  }
  __device__ __forceinline__ void settle() {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lo), "+s"(hi));
  }
  __device__ __forceinline__ void unpack(double (&r)[KIRCHHOFF_REC_DOUBLES]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      r[i] = __hiloint2double((int)lo[2 * i + 1], (int)lo[2 * i]);
      r[8 + i] = __hiloint2double((int)hi[2 * i + 1], (int)hi[2 * i]);
    }
  }
};

// work on the record in `cur`; between the two halves settle `landed` (if any) and
// request the record at `pnext` into `fetch`
template <int PPT, bool HAS_P, bool GEN_N, bool TAB, class R>
__device__ __forceinline__ void stream_step(const double (&x)[PPT], const double (&y)[PPT],
                                            const double (&z)[PPT], Acc (&acc)[PPT],
                                            const double2* tab, const SinCosTabRegs& kreg,
                                            const R& cur, R* landed, R& fetch,
                                            const double* pnext) {
  double r[KIRCHHOFF_REC_DOUBLES];
  cur.unpack(r);
  Mid m[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) m[j] = pair_head<GEN_N, TAB>(x[j], y[j], z[j], r, tab, kreg);
  __builtin_amdgcn_sched_barrier(0);
  if (landed) landed->settle();
  fetch.issue(pnext);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < PPT; ++j) pair_tail<HAS_P>(m[j], r, acc[j]);
}

template <int PPT, bool HAS_P, bool GEN_N, bool TAB>
__device__ __forceinline__ void stream_loop(const double (&x)[PPT], const double (&y)[PPT],
                                            const double (&z)[PPT], Acc (&acc)[PPT],
                                            const double* __restrict__ rec,
                                            const double2* tab, int s0, int s1) {
  if (s0 >= s1) return;
  const SinCosTabRegs kreg;
  constexpr bool FULL = HAS_P || GEN_N;
  typedef SRec<FULL> R;
  const double* p = rec + (int64_t)s0 * KIRCHHOFF_REC_DOUBLES;
  // requests past the last record re-read the last one (never used): uniform loop.
  // `left` counts the records not yet worked on, the current one included.
#define KIRCHHOFF_AHEAD(q, n) ((q) + (left > (n) ? (n) : left - 1) * KIRCHHOFF_REC_DOUBLES)
  int left = s1 - s0;
  if (FULL) {
    // two register sets (2 x 32 SGPRs): request s+1 mid-iteration, settle at its end
    R A, B;
    A.issue(p);
    A.settle();
    for (;;) {
      stream_step<PPT, HAS_P, GEN_N, TAB, R>(x, y, z, acc, tab, kreg, A, nullptr, B,
                                             KIRCHHOFF_AHEAD(p, 1));
      B.settle();
      if (--left == 0) break;
      p += KIRCHHOFF_REC_DOUBLES;
      stream_step<PPT, HAS_P, GEN_N, TAB, R>(x, y, z, acc, tab, kreg, B, nullptr, A,
                                             KIRCHHOFF_AHEAD(p, 1));
      A.settle();
      if (--left == 0) break;
      p += KIRCHHOFF_REC_DOUBLES;
    }
  } else {
    // three register sets (3 x 18 SGPRs): request s+2 mid-iteration s; it is settled
    // mid-iteration s+1, right after that iteration's LDS wait has drained it anyway
    R A, B, C;
    A.issue(p);
    B.issue(KIRCHHOFF_AHEAD(p, 1));
    A.settle();   // lgkmcnt(0): B has landed as well
    for (;;) {
      stream_step<PPT, HAS_P, GEN_N, TAB, R>(x, y, z, acc, tab, kreg, A, &B, C,
                                             KIRCHHOFF_AHEAD(p, 2));
      if (--left == 0) break;
      p += KIRCHHOFF_REC_DOUBLES;
      stream_step<PPT, HAS_P, GEN_N, TAB, R>(x, y, z, acc, tab, kreg, B, &C, A,
                                             KIRCHHOFF_AHEAD(p, 2));
      if (--left == 0) break;
      p += KIRCHHOFF_REC_DOUBLES;
      stream_step<PPT, HAS_P, GEN_N, TAB, R>(x, y, z, acc, tab, kreg, C, &A, B,
                                             KIRCHHOFF_AHEAD(p, 2));
      if (--left == 0) break;
      p += KIRCHHOFF_REC_DOUBLES;
    }
    // drain the request still in flight before any of the three sets is reused
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(A.lo), "+s"(A.hi), "+s"(B.lo), "+s"(B.hi), "+s"(C.lo), "+s"(C.hi));
  }
#undef KIRCHHOFF_AHEAD
}

template <int PPT>
__global__ __launch_bounds__(KIRCHHOFF_BLOCK, KIRCHHOFF_WAVES) void kirchhoff_stream(
    int64_t np, const double* __restrict__ px, const double* __restrict__ py,
    const double* __restrict__ pz, int ns, const double* __restrict__ rec,
    const unsigned* __restrict__ flags, int nsplit, int chunk, int64_t np_pad,
    double* __restrict__ partial) {
  const int split = blockIdx.x % nsplit;
  const int64_t tile = blockIdx.x / nsplit;
  const int64_t base = tile * (int64_t)(KIRCHHOFF_BLOCK * PPT) + threadIdx.x;
  const int s0 = split * chunk;
  const int s1 = min(ns, s0 + chunk);

  __shared__ double2 tab[SINCOS_TAB_N];
  sincos_tab_fill(tab);

  double x[PPT], y[PPT], z[PPT];
  Acc acc[PPT];
  double pabs = 0.;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    int64_t p = base + (int64_t)j * KIRCHHOFF_BLOCK;
    // out-of-range lanes re-use the last pixel (their result is not stored)
    int64_t pc = p < np ? p : np - 1;
    x[j] = px[pc];
    y[j] = py[pc];
    z[j] = pz[pc];
    pabs = fmax(pabs, fabs(x[j]) + fabs(y[j]) + fabs(z[j]));
    acc[j] = Acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  }
  const unsigned f = __builtin_amdgcn_readfirstlane(*flags);   // wave-uniform dispatch
  // |k r| <= kmax (|p|_1 + |s|_1): below 2^42 the table-driven sincos is exact
  // enough (fp64_math.h); harder X-rays over longer distances take the general one
  const unsigned long long* bound = reinterpret_cast<const unsigned long long*>(flags) + 1;
  const double kmax = __longlong_as_double(bound[0]);
  const double smax = __longlong_as_double(bound[1]);
  const bool small_phase =
      wave_max_u64(__double_as_longlong(kmax * (pabs + smax))) < __double_as_longlong(0x1p42);
  if (!small_phase)
    stream_loop<PPT, true, true, false>(x, y, z, acc, rec, tab, s0, s1);
  else if (f == 0)
    stream_loop<PPT, false, false, true>(x, y, z, acc, rec, tab, s0, s1);
  else if (f == KIRCHHOFF_FLAG_EP)
    stream_loop<PPT, true, false, true>(x, y, z, acc, rec, tab, s0, s1);
  else if (f == KIRCHHOFF_FLAG_NXZ)
    stream_loop<PPT, false, true, true>(x, y, z, acc, rec, tab, s0, s1);
  else
    stream_loop<PPT, true, true, true>(x, y, z, acc, rec, tab, s0, s1);
  double* out = partial + (int64_t)split * 10 * np_pad;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    int64_t p = base + (int64_t)j * KIRCHHOFF_BLOCK;
    if (p < np) {
      out[0 * np_pad + p] = acc[j].sr;
      out[1 * np_pad + p] = acc[j].si;
      out[2 * np_pad + p] = acc[j].pr;
      out[3 * np_pad + p] = acc[j].pi;
      out[4 * np_pad + p] = acc[j].ar;
      out[5 * np_pad + p] = acc[j].ai;
      out[6 * np_pad + p] = acc[j].br;
      out[7 * np_pad + p] = acc[j].bi;
      out[8 * np_pad + p] = acc[j].cr;
      out[9 * np_pad + p] = acc[j].ci;
    }
  }
}

// ---------------------------------------------------------------------------
// finalize: add the split partials in fixed order, apply the prefactors.
//   convention 0 (numpy, waves.py:844,847): S,P *= i/(4pi); A,B,C *= i/(4pi)^2
//   convention 1 (OpenCL, diffract.cl:143-148): S,P *= -i/(4pi);
//                                              A,B,C *= (1+i)/(4pi)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kirchhoff_finalize(
    int64_t np, int nsplit, int64_t np_pad, const double* __restrict__ partial,
    int convention, double2* __restrict__ S, double2* __restrict__ P,
    double2* __restrict__ A, double2* __restrict__ B, double2* __restrict__ C) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  double v[10];
#pragma unroll
  for (int c = 0; c < 10; ++c) v[c] = 0.0;
  for (int s = 0; s < nsplit; ++s) {
    const double* in = partial + (int64_t)s * 10 * np_pad;
#pragma unroll
    for (int c = 0; c < 10; ++c) v[c] += in[c * np_pad + p];
  }
  const double inv4pi = 0.07957747154594767;  // 1/(4 pi)
  double2 o[5];
  if (convention == 0) {
    const double f2 = inv4pi * inv4pi;
    o[0] = make_double2(-v[1] * inv4pi, v[0] * inv4pi);
    o[1] = make_double2(-v[3] * inv4pi, v[2] * inv4pi);
    o[2] = make_double2(-v[5] * f2, v[4] * f2);
    o[3] = make_double2(-v[7] * f2, v[6] * f2);
    o[4] = make_double2(-v[9] * f2, v[8] * f2);
  } else {
    o[0] = make_double2(v[1] * inv4pi, -v[0] * inv4pi);
    o[1] = make_double2(v[3] * inv4pi, -v[2] * inv4pi);
    o[2] = make_double2((v[4] - v[5]) * inv4pi, (v[4] + v[5]) * inv4pi);
    o[3] = make_double2((v[6] - v[7]) * inv4pi, (v[6] + v[7]) * inv4pi);
    o[4] = make_double2((v[8] - v[9]) * inv4pi, (v[8] + v[9]) * inv4pi);
  }
  S[p] = o[0];
  P[p] = o[1];
  A[p] = o[2];
  B[p] = o[3];
  C[p] = o[4];
}

// ---------------------------------------------------------------------------
// debug kernels for the building blocks (tests/test_gpu_math.py)
// ---------------------------------------------------------------------------
__global__ void debug_sqrt_kernel(int64_t n, const double* __restrict__ x,
                                  double* __restrict__ r, double* __restrict__ ri) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // both forms of fp64_math.h: the one-correction root of the Kirchhoff loop must equal
  // the two-correction root of the reflect kernels (NaN flags a difference)
  double rinv, h;
  const double r2 = sqrt_rn_rinv(x[i], rinv);
  const double r1 = sqrt_rn_halfinv(x[i], h);
  r[i] = r1 == r2 ? r1 : __builtin_nan("");
  ri[i] = rinv;
}

__global__ void debug_divconst_kernel(int64_t n, const double* __restrict__ a, double b,
                                      double y, double* __restrict__ q) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double q0 = a[i] * y;
  const double r = fma_(-q0, b, a[i]);
  q[i] = fma_(r, y, q0);
}

__global__ void debug_sincos_kernel(int64_t n, const double* __restrict__ phi,
                                    double* __restrict__ sn, double* __restrict__ cs) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  sincos_phase(phi[i], s, c);
  sn[i] = s;
  cs[i] = c;
}

__global__ __launch_bounds__(256) void debug_sincos_tab_kernel(
    int64_t n, const double* __restrict__ phi, double* __restrict__ sn,
    double* __restrict__ cs) {
  __shared__ double2 tab[SINCOS_TAB_N];
  sincos_tab_fill(tab);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  const SinCosTabRegs kreg;
  sincos_tab(phi[i], tab, kreg, s, c);
  sn[i] = s;
  cs[i] = c;
}

}  // namespace xrt

// ---------------------------------------------------------------------------
// launch plan + launchers (called from capi.hip)
// ---------------------------------------------------------------------------
namespace xrt {

KirchhoffPlan kirchhoff_plan(int64_t np, int64_t ns, int nsplit_req, int ppt_req) {
  KirchhoffPlan pl;
  // two receiving points per lane amortise the per-sample work (record unpack, loop) and
  // measure +3 % on large problems; small ones keep one for more blocks
  pl.ppt = (ppt_req == 1 || ppt_req == 2) ? ppt_req : (np >= 65536 ? 2 : 1);
  int64_t per_block = (int64_t)KIRCHHOFF_BLOCK * pl.ppt;
  pl.tiles = (np + per_block - 1) / per_block;
  if (pl.tiles < 1) pl.tiles = 1;
  int nsplit = nsplit_req;
  if (nsplit <= 0) {
    // The kernel is VALU-bound and VGPR-limited to ~7 blocks per CU; many more
    // blocks than that (64 per CU) keep the tail short. Splits come in multiples
    // of 8 so that split == XCD under the round-robin block placement.
    nsplit = 1;
    const int64_t want_blocks = 16384;
    if (pl.tiles < want_blocks) {
      int64_t need = (want_blocks + pl.tiles - 1) / pl.tiles;
      nsplit = (int)(((need + 7) / 8) * 8);
      if (nsplit > 256) nsplit = 256;
    }
  }
  // never split finer than 64 samples per split
  while (nsplit > 1 && ns / nsplit < 64) nsplit /= 2;
  if (nsplit < 1) nsplit = 1;
  pl.nsplit = nsplit;
  pl.np_pad = ((np + 31) / 32) * 32;
  pl.chunk = (int)((ns + nsplit - 1) / nsplit);
  if (pl.chunk < 1) pl.chunk = 1;
  // 256 B in front of the records hold the sample-set flags
  pl.rec_bytes = 256 + (size_t)ns * KIRCHHOFF_REC_DOUBLES * sizeof(double);
  pl.partial_bytes = (size_t)nsplit * 10 * pl.np_pad * sizeof(double);
  return pl;
}

hipError_t kirchhoff_launch(const KirchhoffPlan& pl, int64_t np, const double* px,
                            const double* py, const double* pz, int64_t ns,
                            const double* sx, const double* sy, const double* sz,
                            int pstride, const double* nx, const double* ny,
                            const double* nz, int nstride, const double* nl,
                            const double* k, const double* Es,
                            const double* Ep, int convention, double* S, double* P,
                            double* A, double* B, double* C, void* workspace,
                            hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  unsigned* flags = reinterpret_cast<unsigned*>(workspace);
  double* rec = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 256);
  double* partial =
      reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) +
                                ((pl.rec_bytes + 255) / 256) * 256);
  hipError_t me = hipMemsetAsync(flags, 0, 256, stream);
  if (me != hipSuccess) return me;
  if (ns > 0) {
    hipLaunchKernelGGL(kirchhoff_pack, dim3((unsigned)((ns + 255) / 256)), dim3(256),
                       0, stream, ns, sx, sy, sz, pstride, nx, ny, nz, nstride, nl, k,
                       reinterpret_cast<const double2*>(Es),
                       reinterpret_cast<const double2*>(Ep), rec, flags);
  }
  if (np > 0) {
    dim3 grid((unsigned)(pl.tiles * pl.nsplit));
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (pl.ppt == 2)
      hipLaunchKernelGGL(kirchhoff_stream<2>, grid, dim3(KIRCHHOFF_BLOCK), 0, stream,
                         np, px, py, pz, (int)ns, rec, flags, pl.nsplit, pl.chunk, pl.np_pad,
                         partial);
    else
      hipLaunchKernelGGL(kirchhoff_stream<1>, grid, dim3(KIRCHHOFF_BLOCK), 0, stream,
                         np, px, py, pz, (int)ns, rec, flags, pl.nsplit, pl.chunk, pl.np_pad,
                         partial);
    if (ev1) (void)hipEventRecord(ev1, stream);
    hipLaunchKernelGGL(kirchhoff_finalize, dim3((unsigned)((np + 255) / 256)),
                       dim3(256), 0, stream, np, pl.nsplit, pl.np_pad, partial,
                       convention, reinterpret_cast<double2*>(S),
                       reinterpret_cast<double2*>(P), reinterpret_cast<double2*>(A),
                       reinterpret_cast<double2*>(B), reinterpret_cast<double2*>(C));
  }
  return hipGetLastError();
}

hipError_t debug_sqrt_launch(int64_t n, const double* x, double* r, double* ri,
                             hipStream_t stream) {
  hipLaunchKernelGGL(debug_sqrt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256),
                     0, stream, n, x, r, ri);
  return hipGetLastError();
}

hipError_t debug_divconst_launch(int64_t n, const double* a, double b, double y, double* q,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(debug_divconst_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     stream, n, a, b, y, q);
  return hipGetLastError();
}

hipError_t debug_sincos_launch(int64_t n, const double* phi, double* sn, double* cs,
                               int table, hipStream_t stream) {
  if (table)
    hipLaunchKernelGGL(debug_sincos_tab_kernel, dim3((unsigned)((n + 255) / 256)),
                       dim3(256), 0, stream, n, phi, sn, cs);
  else
    hipLaunchKernelGGL(debug_sincos_kernel, dim3((unsigned)((n + 255) / 256)),
                       dim3(256), 0, stream, n, phi, sn, cs);
  return hipGetLastError();
}

}  // namespace xrt
