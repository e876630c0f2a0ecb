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

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

__device__ __forceinline__ unsigned long long dbits(double v) {
  return (unsigned long long)__double_as_longlong(v);
}

// ---------------------------------------------------------------------------
// scan: one pass over the receiving points (blocks [0, nbp)) and one over the samples
// (the blocks after them) that classifies the launch, so that the pack kernel and the
// main kernel can take a shorter instruction stream when it is safe -- decided on the
// device, no host sync:
//   * FLAG_EP / FLAG_NXZ / FLAG_KVAR: some Ep != 0, some normal off the y axis, the
//     wavenumbers differ;
//   * FLAG_PYVAR: the receiving points do not lie on one plane y = const;
//   * extents of both point sets and the smallest |py0 - sy|, which bound how far the
//     distance of any pair is from |py0 - sy| (kirchhoff_fast below);
//   * the row length of a receiving mesh: the first p > 0 with px[p] == px[0].
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v,
                                                           unsigned long long* lds) {
  v = wave_max_u64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = lds[0];
  for (unsigned w = 1; w < (blockDim.x >> 6); ++w) r = lds[w] > r ? lds[w] : r;
  return r;
}

// (a capped grid of striding blocks, one atomic per block and quantity: the atomics all go
// to the same few addresses, where they serialise -- one per WAVE made this pass take 1 ms
// on 1e6 samples)
__global__ __launch_bounds__(256) void kirchhoff_scan(
    int64_t np, const double* __restrict__ px, const double* __restrict__ py,
    const double* __restrict__ pz, int64_t ns, const double* __restrict__ sx,
    const double* __restrict__ sy, const double* __restrict__ sz, int pstride,
    const double* __restrict__ nx, const double* __restrict__ nz, int nstride,
    const double* __restrict__ k, const double2* __restrict__ Ep, int nbp, unsigned opts,
    KirchhoffInfo* __restrict__ info) {
  __shared__ unsigned long long lds[4];
  unsigned f = 0;
  if ((int)blockIdx.x < nbp) {
    unsigned long long xm = 0, zm = 0, nr = 0;
    const double y0 = np > 0 ? py[0] : 0., x0 = np > 0 ? px[0] : 0.;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < np;
         i += (int64_t)nbp * blockDim.x) {
      if (py[i] != y0) f |= KIRCHHOFF_FLAG_PYVAR;
      const unsigned long long ax = dbits(fabs(px[i])), az = dbits(fabs(pz[i]));
      xm = ax > xm ? ax : xm;
      zm = az > zm ? az : zm;
      if (i > 0 && px[i] == x0) {
        const unsigned long long c = ~(unsigned long long)i;
        nr = c > nr ? c : nr;
      }
    }
    f = __builtin_amdgcn_readfirstlane(__reduce_or_sync(~0ull, f));
    if (f && (threadIdx.x & 63) == 0) atomicOr(&info->flags, f);
    xm = block_max_u64(xm, lds);
    zm = block_max_u64(zm, lds);
    nr = block_max_u64(nr, lds);
    if (threadIdx.x == 0) {
      atomicMax(&info->pxmax, xm);
      atomicMax(&info->pzmax, zm);
      if (nr) atomicMax(&info->not_row, nr);
      if (blockIdx.x == 0) {
        info->py0 = y0;
        info->opts = opts;
      }
    }
    return;
  }
  const int nbs = (int)gridDim.x - nbp;
  unsigned long long km = 0, s1 = 0, xm = 0, zm = 0, dm = 0;
  const double y0 = np > 0 ? py[0] : 0., k0 = ns > 0 ? k[0] : 0.;
  for (int64_t i = (int64_t)((int)blockIdx.x - nbp) * blockDim.x + threadIdx.x; i < ns;
       i += (int64_t)nbs * blockDim.x) {
    const int64_t ip = i * pstride, in = i * nstride;
    const double2 ep = Ep[i];
    const double kk = k[i], x = sx[ip], y = sy[ip], z = sz[ip];
    if (ep.x != 0. || ep.y != 0.) f |= KIRCHHOFF_FLAG_EP;
    if (nx[in] != 0. || nz[in] != 0.) f |= KIRCHHOFF_FLAG_NXZ;
    if (kk != k0) f |= KIRCHHOFF_FLAG_KVAR;
    unsigned long long t;
    t = dbits(fabs(kk));
    km = t > km ? t : km;
    t = dbits(fabs(x) + fabs(y) + fabs(z));
    s1 = t > s1 ? t : s1;
    t = dbits(fabs(x));
    xm = t > xm ? t : xm;
    t = dbits(fabs(z));
    zm = t > zm ? t : zm;
    // 1/0 = inf and NaN both sort above every finite value: no fast geometry then
    t = np > 0 ? dbits(fabs(1. / (y0 - y))) : 0;
    dm = t > dm ? t : dm;
  }
  f = __builtin_amdgcn_readfirstlane(__reduce_or_sync(~0ull, f));
  if (f && (threadIdx.x & 63) == 0) atomicOr(&info->flags, f);
  km = block_max_u64(km, lds);
  s1 = block_max_u64(s1, lds);
  xm = block_max_u64(xm, lds);
  zm = block_max_u64(zm, lds);
  dm = block_max_u64(dm, lds);
  if (threadIdx.x == 0) {
    atomicMax(&info->kmax, km);
    atomicMax(&info->s1max, s1);
    atomicMax(&info->sxmax, xm);
    atomicMax(&info->szmax, zm);
    atomicMax(&info->dyinvmax, dm);
    if ((int)blockIdx.x == nbp) info->k0 = k0;
  }
}

// The fast geometry: all normals along y, all receiving points on the plane y = py0,
// and every pair's distance r = sqrt(dy^2 + t^2), t^2 = dx^2 + dz^2, so close to |dy|
// that 1/|dy| seeds the square root as well as v_rsq_f64 does (fp64_math.h):
// |1/|dy| * r - 1| <= t^2 / (2 dy^2) <= 2^-27, with t^2 bounded by the extents of the
// two point sets. (cfg4: 0.6 mm across, 10 m apart: 1.8e-9 = 2^-29.)
__device__ __forceinline__ bool kirchhoff_fast(const KirchhoffInfo* info) {
  if (info->opts & KIRCHHOFF_OPT_NO_FAST) return false;
  if (info->flags & (KIRCHHOFF_FLAG_NXZ | KIRCHHOFF_FLAG_PYVAR)) return false;
  const double tx = __longlong_as_double(info->pxmax) + __longlong_as_double(info->sxmax);
  const double tz = __longlong_as_double(info->pzmax) + __longlong_as_double(info->szmax);
  const double di = __longlong_as_double(info->dyinvmax);
  return (tx * tx + tz * tz) * 0.5 * (di * di) <= 0x1p-27;   // false for NaN / inf
}

// the factor 2k^2 of the direction integrals is applied once per receiving point (by
// kirchhoff_finalize) instead of once per pair when there is one wavenumber and no Ep
// (fast-geometry loops only)
__device__ __forceinline__ bool kirchhoff_unik(const KirchhoffInfo* info) {
  return (info->flags & (KIRCHHOFF_FLAG_EP | KIRCHHOFF_FLAG_KVAR)) == 0 &&
         kirchhoff_fast(info);
}

// ---------------------------------------------------------------------------
// pack: sample arrays -> 16-double records. Positions / normals are read with
// an element stride so that both the SoA layout (stride 1) and the reference's
// OpenCL marshalling ns x [x,y,z,0] (stride 4, waves.py:872-879) feed it.
// General layout:
//   [0..2] x,y,z  [3] 2k nl  [4] 4k ny  [5] k  [6,7] Es  [8] 2k^2  |  [9] 4k nx
//   [10] 4k nz  [11,12] Ep  [13,14] 2k^2 (Es+Ep)
// (the first 72 bytes are all the Ep == 0, planar-normal case reads)
// Fast-geometry layout (kirchhoff_fast):
//   [0] x  [1] z  [2] dy = py0 - y  [3] dy^2  [4] 1/|dy|  [5] 1/(2|dy|)  [6] k
//   [7] 2k nl  [8] 4k ny dy  [9,10] Es  [11] 2k^2  |  [12,13] Ep  [14,15] 2k^2 (Es+Ep)
// (the first 96 bytes are all the Ep == 0 case reads)
// (1/r comes out of the sqrt iteration as h = 1/(2r): the factors 2 and the k of
// (k/r)(d.n/r + nl) = h (4k n.d h + 2k nl) are folded in here, once per sample
// instead of once per pair. numpy forms Es+Ep first too: k**2/(4pi) (Es+Ep) U / r.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kirchhoff_pack(
    int64_t ns, const double* __restrict__ sx, const double* __restrict__ sy,
    const double* __restrict__ sz, int pstride, const double* __restrict__ nx,
    const double* __restrict__ ny, const double* __restrict__ nz, int nstride,
    const double* __restrict__ nl, const double* __restrict__ k,
    const double2* __restrict__ Es, const double2* __restrict__ Ep,
    const KirchhoffInfo* __restrict__ info, double* __restrict__ rec) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const bool fast = kirchhoff_fast(info);
  const double2 es = Es[i], ep = Ep[i];
  const double kk = k[i];
  double2* o = reinterpret_cast<double2*>(rec + i * KIRCHHOFF_REC_DOUBLES);
  const int64_t ip = i * pstride, in = i * nstride;
  const double vx = nx[in], vy = ny[in], vz = nz[in];
  const double k2 = 2. * kk;
  const double k2k = k2 * kk;
  const double2 q = make_double2(k2k * (es.x + ep.x), k2k * (es.y + ep.y));
  if (fast) {
    const double dy = info->py0 - sy[ip];
    const double yi = 1. / fabs(dy);
    o[0] = make_double2(sx[ip], sz[ip]);
    o[1] = make_double2(dy, dy * dy);
    o[2] = make_double2(yi, 0.5 * yi);
    o[3] = make_double2(kk, k2 * nl[i]);
    o[4] = make_double2(k2 * (2. * vy) * dy, es.x);
    o[5] = make_double2(es.y, k2k);
    o[6] = ep;
    o[7] = q;
  } else {
    o[0] = make_double2(sx[ip], sy[ip]);
    o[1] = make_double2(sz[ip], k2 * nl[i]);
    o[2] = make_double2(k2 * (2. * vy), kk);
    o[3] = es;
    o[4] = make_double2(k2k, k2 * (2. * vx));
    o[5] = make_double2(k2 * (2. * vz), ep.x);
    o[6] = make_double2(ep.y, q.x);
    o[7] = make_double2(q.y, 0.);
    if (info->opts & KIRCHHOFF_OPT_RELAXED) {
      // relaxed loops: ks = k N / (2 pi), table steps per mm, as a double-double in the two
      // slots the loop variant does not read otherwise ([15]; [8] = 2k^2 when Ep is there,
      // [11] = Re Ep = 0 when it is not)
      const double scale = (info->opts & KIRCHHOFF_OPT_TAB4096) ? 1024. : 512.;
      const double SH = 0x1.45f306dc9c883p-1 * scale, SL = -0x1.6b01ec5417056p-55 * scale;
      const double ksh = kk * SH;
      const double ksl = fma_(kk, SL, fma_(kk, SH, -ksh));
      rec[i * KIRCHHOFF_REC_DOUBLES + 15] = ksh;
      rec[i * KIRCHHOFF_REC_DOUBLES + ((info->flags & KIRCHHOFF_FLAG_EP) ? 8 : 11)] = ksl;
    }
  }
}

// ---------------------------------------------------------------------------
// main streaming kernel
// ---------------------------------------------------------------------------
struct Acc {
  double sr, si, pr, pi, ar, ai, br, bi, cr, ci;
};

template <int PPT>
struct Pts {   // the receiving points of one lane
  double x[PPT], y[PPT], z[PPT];
};

// what the second half of a pair update needs from the first
struct Mid {
  double dx, dy, dz, gr, gi, h;
};

// accumulation half, common to both geometries. HAS_P: the p-polarised source field is
// present. UNIK (Ep == 0 only): one wavenumber, 2k^2 is applied by kirchhoff_finalize.
template <bool HAS_P, bool UNIK>
__device__ __forceinline__ void accumulate(const Mid& m, double esr, double esi, double epr,
                                           double epi, double qr, double qi, double k2k,
                                           Acc& a) {
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
    const double hr0 = m.h * gr;                  // g/(2r); q carries 2k^2 (Es+Ep)
    const double hi0 = m.h * gi;
    hr = hr0 * qr;
    hr = fma_(-hi0, qi, hr);
    hi = hr0 * qi;
    hi = fma_(hi0, qr, hi);
  } else {
    // Ep == 0: g*Es is shared by S and by the direction term
    double wr = gr * esr;
    wr = fma_(-gi, esi, wr);
    double wi = gr * esi;
    wi = fma_(gi, esr, wi);
    a.sr += wr;
    a.si += wi;
    if (UNIK) {
      hr = m.h * wr;
      hi = m.h * wi;
    } else {
      const double kkip = k2k * m.h;              // k^2/r
      hr = kkip * wr;
      hi = kkip * wi;
    }
  }
  a.ar = fma_(hr, m.dx, a.ar);
  a.ai = fma_(hi, m.dx, a.ai);
  a.br = fma_(hr, m.dy, a.br);
  a.bi = fma_(hi, m.dy, a.bi);
  a.cr = fma_(hr, m.dz, a.cr);
  a.ci = fma_(hi, m.dz, a.ci);
}

// ---- general geometry --------------------------------------------------------
// GEN_N: normals are general (otherwise every normal is (0, ny, 0), the aperture /
// screen / source case of waves.py:687-689 -- d.n collapses to dy*ny).
// TAB: sincos through the LDS table (|k r| < 2^42 is guaranteed by the caller).
// The update of one (receiving point, sample) pair comes in two halves so that the
// loop can put its scalar prefetch between them (see stream_loop): head ends with the
// first use of the LDS table entry, tail is pure accumulation.
// RELAX (opt-in, XRT_HIP_KIRCHHOFF_RELAXED; TAB only): d.d contracted into two fma, the root
// without its last correction step (r to ~1 ulp instead of correctly rounded, 1/(2r) as
// before), k r not formed -- the table steps come from r times a double-double k N / (2 pi)
// (sincos_tab_scaled). 5 issue slots of 60 less; the results are no longer numpy's doubles
// (a pair's phase moves by up to an ulp of k r ~ 6e-5 rad at cfg4's distances, as numpy's own
// rounding of k r does against the true product).
template <bool HAS_P, bool GEN_N, bool TAB, bool RELAX = false>
struct GenKern {
  static constexpr int NDW = (HAS_P || GEN_N || RELAX) ? 32 : 18;
  struct Shared {};
  template <int PPT>
  static __device__ __forceinline__ void pre(const Pts<PPT>&,
                                             const double (&)[KIRCHHOFF_REC_DOUBLES], Shared&) {}
  template <int PPT, int TN>
  static __device__ __forceinline__ Mid head(const Pts<PPT>& p, int j, const Shared&,
                                             const double (&r)[KIRCHHOFF_REC_DOUBLES],
                                             const double2* tab,
                                             const SinCosTabRegs<TN>& kreg) {
    const double sx = r[0], sy = r[1], sz = r[2], knl = r[3];
    const double kny = r[4], k = r[5], knx = r[9], knz = r[10];
    Mid m;
    // --- bit-exact part (numpy order, no contraction) ---
    m.dx = p.x[j] - sx;
    m.dy = p.y[j] - sy;
    m.dz = p.z[j] - sz;
    double rr, phase = 0.;
    if (RELAX) {
      const double s2 = fma_(m.dz, m.dz, fma_(m.dy, m.dy, m.dx * m.dx));
      const double y = __builtin_amdgcn_rsq(s2);
      const double g = s2 * y, h = 0.5 * y;
      const double r0 = fma_(-h, g, 0.5);
      rr = fma_(g, r0, g);
      m.h = fma_(h, r0, h);
    } else {
      const double s2 = (m.dx * m.dx + m.dy * m.dy) + m.dz * m.dz;
      rr = sqrt_rn_halfinv(s2, m.h);   // h = 1/(2r)
      phase = k * rr;
    }
    // --- the rest only needs ~1e-16 relative accuracy ---
    double dn;
    if (GEN_N) {
      dn = m.dx * knx;
      dn = fma_(m.dy, kny, dn);
      dn = fma_(m.dz, knz, dn);
    } else {
      dn = m.dy * kny;
    }
    const double cr = m.h * fma_(dn, m.h, knl);   // (k/r)(d.n/r + nl)
    double sn, cs;
    if (RELAX)
      sincos_tab_scaled<TN>(rr, r[15], HAS_P ? r[8] : r[11], tab, kreg, sn, cs);
    else if (TAB)
      sincos_tab<TN>(phase, tab, kreg, sn, cs);
    else
      sincos_phase(phase, sn, cs);
    m.gr = cr * cs;
    m.gi = cr * sn;
    return m;
  }
  static __device__ __forceinline__ void tail(const Mid& m, const Shared&,
                                              const double (&r)[KIRCHHOFF_REC_DOUBLES],
                                              Acc& a) {
    // (RELAX without Ep: [11] holds the low part of ks, Ep is zero)
    accumulate<HAS_P, false>(m, r[6], r[7], RELAX && !HAS_P ? 0. : r[11], r[12], r[13], r[14],
                             r[8], a);
  }
};

// ---- fast geometry (kirchhoff_fast) -------------------------------------------
// dy, dy^2, the seed of the square root and 4k ny dy are per-sample constants in
// SGPRs. SHARE: the points of the lane have one x (a column of the receiving mesh):
// dx, dx^2 + dy^2 are formed once per sample instead of once per pair.
// Per pair (Ep == 0, UNIK, no SHARE): 6 slots d.d, 6 the correctly rounded root and
// 1/(2r), 1 k r, 2 amplitude, 15 sincos (2 of them 32-bit), 2 + 14 accumulation = 46,
// against 52 + a quarter-rate v_rsq_f64 (= 4 slots) of the general loop.
template <bool HAS_P, bool UNIK, bool SHARE, bool TAB>
struct FastKern {
  static constexpr int NDW = HAS_P ? 32 : 24;
  struct Shared {
    double dx, t;
  };
  template <int PPT>
  static __device__ __forceinline__ void pre(const Pts<PPT>& p,
                                             const double (&r)[KIRCHHOFF_REC_DOUBLES],
                                             Shared& s) {
    if (SHARE) {
      s.dx = p.x[0] - r[0];
      s.t = s.dx * s.dx + r[3];
    }
  }
  template <int PPT, int TN>
  static __device__ __forceinline__ Mid head(const Pts<PPT>& p, int j, const Shared& s,
                                             const double (&r)[KIRCHHOFF_REC_DOUBLES],
                                             const double2* tab,
                                             const SinCosTabRegs<TN>& kreg) {
    const double sx = r[0], sz = r[1], dy2 = r[3], y0 = r[4], h0 = r[5];
    const double k = r[6], knl = r[7], dn = r[8];
    Mid m;
    // --- bit-exact part: (dx*dx + dy*dy) + dz*dz as numpy forms it ---
    double t;
    if (SHARE) {
      m.dx = s.dx;
      t = s.t;
    } else {
      m.dx = p.x[j] - sx;
      t = m.dx * m.dx + dy2;
    }
    m.dy = r[2];
    m.dz = p.z[j] - sz;
    const double s2 = t + m.dz * m.dz;
    const double rr = sqrt_rn_seeded(s2, y0, h0, m.h);
    const double phase = k * rr;
    // --- ~1e-16 relative from here on ---
    const double cr = m.h * fma_(dn, m.h, knl);
    double sn, cs;
    if (TAB)
      sincos_tab<TN>(phase, tab, kreg, sn, cs);
    else
      sincos_phase(phase, sn, cs);
    m.gr = cr * cs;
    m.gi = cr * sn;
    return m;
  }
  static __device__ __forceinline__ void tail(const Mid& m, const Shared&,
                                              const double (&r)[KIRCHHOFF_REC_DOUBLES],
                                              Acc& a) {
    accumulate<HAS_P, UNIK>(m, r[9], r[10], r[12], r[13], r[14], r[15], r[11], a);
  }
};

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
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int NDW>
struct SRec;

template <>
struct SRec<18> {   // 72 bytes
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
struct SRec<24> {   // 96 bytes
  u32x16 lo;
  u32x8 hi;
  __device__ __forceinline__ void issue(const double* p) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40"
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
#pragma unroll
    for (int i = 0; i < 4; ++i) r[8 + i] = __hiloint2double((int)hi[2 * i + 1], (int)hi[2 * i]);
#pragma unroll
    for (int i = 12; i < KIRCHHOFF_REC_DOUBLES; ++i) r[i] = 0.;
  }
};

template <>
struct SRec<32> {   // the whole 128-byte record
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
template <int PPT, int TN, class K, class R>
__device__ __forceinline__ void stream_step(const Pts<PPT>& pts, Acc (&acc)[PPT],
                                            const double2* tab,
                                            const SinCosTabRegs<TN>& kreg,
                                            const R& cur, R* landed, R& fetch,
                                            const double* pnext) {
  double r[KIRCHHOFF_REC_DOUBLES];
  cur.unpack(r);
  typename K::Shared sh;
  K::template pre<PPT>(pts, r, sh);
  Mid m[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) m[j] = K::template head<PPT, TN>(pts, j, sh, r, tab, kreg);
  __builtin_amdgcn_sched_barrier(0);
  if (landed) landed->settle();
  fetch.issue(pnext);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < PPT; ++j) K::tail(m[j], sh, r, acc[j]);
}

template <int PPT, int TN, class K>
__device__ __forceinline__ void stream_loop(const Pts<PPT>& pts, Acc (&acc)[PPT],
                                            const double* __restrict__ rec,
                                            const double2* tab, int s0, int s1) {
  if (s0 >= s1) return;
  const SinCosTabRegs<TN> kreg;
  typedef SRec<K::NDW> R;
  const double* p = rec + (int64_t)s0 * KIRCHHOFF_REC_DOUBLES;
  // requests past the last record re-read the last one (never used): uniform loop.
  // `left` counts the records not yet worked on, the current one included.
#define KIRCHHOFF_AHEAD(q, n) ((q) + (left > (n) ? (n) : left - 1) * KIRCHHOFF_REC_DOUBLES)
  int left = s1 - s0;
  if (K::NDW > 24) {
    // two register sets (2 x 32 SGPRs): request s+1 mid-iteration, settle at its end
    R A, B;
    A.issue(p);
    A.settle();
    for (;;) {
      stream_step<PPT, TN, K, R>(pts, acc, tab, kreg, A, nullptr, B, KIRCHHOFF_AHEAD(p, 1));
      B.settle();
      if (--left == 0) break;
      p += KIRCHHOFF_REC_DOUBLES;
      stream_step<PPT, TN, K, R>(pts, acc, tab, kreg, B, nullptr, A, KIRCHHOFF_AHEAD(p, 1));
      A.settle();
      if (--left == 0) break;
      p += KIRCHHOFF_REC_DOUBLES;
    }
  } else {
    // three register sets (3 x 18 / 24 SGPRs): request s+2 mid-iteration s; it is settled
    // mid-iteration s+1, right after that iteration's LDS wait has drained it anyway
    R A, B, C;
    A.issue(p);
    B.issue(KIRCHHOFF_AHEAD(p, 1));
    A.settle();   // lgkmcnt(0): B has landed as well
    for (;;) {
      stream_step<PPT, TN, K, R>(pts, acc, tab, kreg, A, &B, C, KIRCHHOFF_AHEAD(p, 2));
      if (--left == 0) break;
      p += KIRCHHOFF_REC_DOUBLES;
      stream_step<PPT, TN, K, R>(pts, acc, tab, kreg, B, &C, A, KIRCHHOFF_AHEAD(p, 2));
      if (--left == 0) break;
      p += KIRCHHOFF_REC_DOUBLES;
      stream_step<PPT, TN, K, R>(pts, acc, tab, kreg, C, &A, B, KIRCHHOFF_AHEAD(p, 2));
      if (--left == 0) break;
      p += KIRCHHOFF_REC_DOUBLES;
    }
    // drain the request still in flight before any of the three sets is reused
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(A.lo), "+s"(A.hi), "+s"(B.lo), "+s"(B.hi), "+s"(C.lo), "+s"(C.hi));
  }
#undef KIRCHHOFF_AHEAD
}

// Row length the lanes are laid out on. A lane owns PPT receiving points p, p + L, ...
// of PPT consecutive rows of L points: on a mesh (x fastest, L = its row length) they
// share x. Without a usable mesh L = 256 gives the plain tiling p + j*256. Bounded by
// np/16 so that the launch (sized without knowing L) carries at most 1/16 idle lanes.
__device__ __forceinline__ int64_t kirchhoff_row(const KirchhoffInfo* info, int64_t np,
                                                 int ppt, bool fast) {
  const unsigned long long nr = info->not_row;
  const int64_t L = nr ? (int64_t)~nr : 0;
  const bool ok = fast && ppt > 1 && !(info->opts & KIRCHHOFF_OPT_NO_SHARE) && L >= 64 &&
                  L <= np / 16;
  return ok ? L : KIRCHHOFF_BLOCK;
}

// loop variants, reported in KirchhoffInfo::variants (tests name them)
enum {
  KV_GEN_S_Y = 0,      // general geometry, Ep == 0, normals along y
  KV_GEN_S_N = 1,      // general normals
  KV_GEN_SP_Y = 2,     // Ep != 0, normals along y
  KV_GEN_SP_N = 3,
  KV_GEN_S_NOTAB = 4,  // |k r| >= 2^42: polynomial sincos
  KV_GEN_SP_NOTAB = 5,
  KV_FAST_S = 6,       // fast geometry, Ep == 0                         (+1: UNIK)
  KV_FAST_SP = 8,
  KV_FAST_S_SHARE = 9,    // mesh column per lane                        (+1: UNIK)
  KV_FAST_SP_SHARE = 11,
  KV_FAST_S_NOTAB = 12,   //                                             (+1: UNIK)
  KV_FAST_SP_NOTAB = 14,
  KV_GEN_S_N_RELAX = 15,  // XRT_HIP_KIRCHHOFF_RELAXED: the general-normal loops, relaxed
  KV_GEN_SP_N_RELAX = 16
};

template <int PPT>
__global__ __launch_bounds__(KIRCHHOFF_BLOCK, PPT > 2 ? 2 : KIRCHHOFF_WAVES) void kirchhoff_stream(
    int64_t np, const double* __restrict__ px, const double* __restrict__ py,
    const double* __restrict__ pz, int ns, const double* __restrict__ rec,
    KirchhoffInfo* __restrict__ info, int nsplit, int chunk, int64_t np_pad,
    double* __restrict__ partial) {
  const int split = blockIdx.x % nsplit;
  const int64_t tile = blockIdx.x / nsplit;
  const int s0 = split * chunk;
  const int s1 = min(ns, s0 + chunk);
  // launch-wide facts (wave-uniform: scalar loads)
  const unsigned f = __builtin_amdgcn_readfirstlane(info->flags);
  const bool fast = kirchhoff_fast(info);
  const bool unik = kirchhoff_unik(info);
  const int64_t L = kirchhoff_row(info, np, PPT, fast);
  const int64_t rows = (np + L - 1) / L;
  const int64_t lanes = ((rows + PPT - 1) / PPT) * L;
  if (tile * KIRCHHOFF_BLOCK >= lanes) return;   // the grid is sized for the worst L

  // four points per lane run two blocks per CU (register budget): room for the 64-KB
  // table, whose remainder needs one term less and whose offset is one instruction
  constexpr int TN = PPT >= 4 ? 4096 : SINCOS_TAB_N;
  __shared__ double2 tab[TN];
  sincos_tab_fill<TN>(tab);

  const int64_t n = tile * KIRCHHOFF_BLOCK + threadIdx.x;
  const int64_t grp = n / L;
  const int64_t col = n - grp * L;
  int64_t pidx[PPT];
  Pts<PPT> pts;
  Acc acc[PPT];
  double pabs = 0.;
  bool onex = true;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    int64_t p = (grp * PPT + j) * L + col;
    if (n >= lanes || p >= np) p = -1;
    pidx[j] = p;
    // idle slots re-use the lane's first point or the last pixel (never stored)
    const int64_t pc = p >= 0 ? p : (pidx[0] >= 0 ? pidx[0] : np - 1);
    pts.x[j] = px[pc];
    pts.y[j] = py[pc];
    pts.z[j] = pz[pc];
    onex = onex && pts.x[j] == pts.x[0];
    pabs = fmax(pabs, fabs(pts.x[j]) + fabs(pts.y[j]) + fabs(pts.z[j]));
    acc[j] = Acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  }
  // |k r| <= kmax (|p|_1 + |s|_1): below 2^42 the table-driven sincos is exact
  // enough (fp64_math.h); harder X-rays over longer distances take the general one
  const double kmax = __longlong_as_double(info->kmax);
  const double smax = __longlong_as_double(info->s1max);
  // (the reduction keeps phase * N/2pi + 1.5 * 2^52 at unit spacing: 2^42 rad for the
  // 2048-entry table, 2^41 for the 4096-entry one)
  const bool small_phase =
      wave_max_u64(dbits(kmax * (pabs + smax))) < dbits(TN > 2048 ? 0x1p41 : 0x1p42);
  const bool share = PPT > 1 && L != KIRCHHOFF_BLOCK && __all(onex);
  const bool has_p = f & KIRCHHOFF_FLAG_EP;
  const bool gen_n = f & KIRCHHOFF_FLAG_NXZ;
  const bool relaxed = __builtin_amdgcn_readfirstlane(info->opts) & KIRCHHOFF_OPT_RELAXED;
  int v;
#define KIRCHHOFF_RUN(vid, ...)                                   \
  do {                                                            \
    v = (vid);                                                    \
    stream_loop<PPT, TN, __VA_ARGS__>(pts, acc, rec, tab, s0, s1); \
  } while (0)
  if (fast) {
    if (!small_phase) {
      if (has_p)
        KIRCHHOFF_RUN(KV_FAST_SP_NOTAB, FastKern<true, false, false, false>);
      else if (unik)
        KIRCHHOFF_RUN(KV_FAST_S_NOTAB + 1, FastKern<false, true, false, false>);
      else
        KIRCHHOFF_RUN(KV_FAST_S_NOTAB, FastKern<false, false, false, false>);
    } else if (share) {
      if (has_p)
        KIRCHHOFF_RUN(KV_FAST_SP_SHARE, FastKern<true, false, true, true>);
      else if (unik)
        KIRCHHOFF_RUN(KV_FAST_S_SHARE + 1, FastKern<false, true, true, true>);
      else
        KIRCHHOFF_RUN(KV_FAST_S_SHARE, FastKern<false, false, true, true>);
    } else {
      if (has_p)
        KIRCHHOFF_RUN(KV_FAST_SP, FastKern<true, false, false, true>);
      else if (unik)
        KIRCHHOFF_RUN(KV_FAST_S + 1, FastKern<false, true, false, true>);
      else
        KIRCHHOFF_RUN(KV_FAST_S, FastKern<false, false, false, true>);
    }
  } else if (!small_phase) {
    if (has_p)
      KIRCHHOFF_RUN(KV_GEN_SP_NOTAB, GenKern<true, true, false>);
    else
      KIRCHHOFF_RUN(KV_GEN_S_NOTAB, GenKern<false, true, false>);
  } else if (!has_p) {
    if (!gen_n)
      KIRCHHOFF_RUN(KV_GEN_S_Y, GenKern<false, false, true>);
    else if (relaxed)
      KIRCHHOFF_RUN(KV_GEN_S_N_RELAX, GenKern<false, true, true, true>);
    else
      KIRCHHOFF_RUN(KV_GEN_S_N, GenKern<false, true, true>);
  } else if (!gen_n) {
    KIRCHHOFF_RUN(KV_GEN_SP_Y, GenKern<true, false, true>);
  } else if (relaxed) {
    KIRCHHOFF_RUN(KV_GEN_SP_N_RELAX, GenKern<true, true, true, true>);
  } else {
    KIRCHHOFF_RUN(KV_GEN_SP_N, GenKern<true, true, true>);
  }
#undef KIRCHHOFF_RUN
  // which variant ran (first wave to find its bit missing sets it)
  if ((threadIdx.x & 63) == 0 && s0 < s1 && !((info->variants >> v) & 1u))
    atomicOr(&info->variants, 1u << v);
  double* out = partial + (int64_t)split * 10 * np_pad;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int64_t p = pidx[j];
    if (p >= 0) {
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
    const KirchhoffInfo* __restrict__ info, int convention, double2* __restrict__ S,
    double2* __restrict__ P,
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
  if (kirchhoff_unik(info)) {
    // one wavenumber, no Ep: the loops left 2k^2 out of the direction integrals
    const double k2k = (2. * info->k0) * info->k0;
#pragma unroll
    for (int c = 4; c < 10; ++c) v[c] *= k2k;
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

__global__ void debug_sqrt_seeded_kernel(int64_t n, const double* __restrict__ x,
                                         const double* __restrict__ seed,
                                         double* __restrict__ r, double* __restrict__ h) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double hh;
  r[i] = sqrt_rn_seeded(x[i], seed[i], 0.5 * seed[i], hh);
  h[i] = hh;
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

template <int TN>
__global__ __launch_bounds__(256) void debug_sincos_tab_kernel(
    int64_t n, const double* __restrict__ phi, double* __restrict__ sn,
    double* __restrict__ cs) {
  __shared__ double2 tab[TN];
  sincos_tab_fill<TN>(tab);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  const SinCosTabRegs<TN> kreg;
  sincos_tab<TN>(phi[i], tab, kreg, s, c);
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
  pl.opts = ppt_req > 0 ? (ppt_req & ~0xff) : 0;
  const int want = ppt_req > 0 ? (ppt_req & 0xff) : 0;
  // several receiving points per lane amortise the per-sample work and, on a mesh,
  // share a column (kirchhoff_row); small problems keep one point for more blocks
  pl.ppt = (want == 1 || want == 2 || want == 4) ? want
                                                   : (np >= 131072 ? 4 : np >= 32768 ? 2 : 1);
  // lanes: rows of 256 in groups of ppt, or -- row length L <= np/16 found on the
  // device -- at most np/ppt + L; blocks beyond the actual count return at once
  const int64_t groups = ((np + KIRCHHOFF_BLOCK - 1) / KIRCHHOFF_BLOCK + pl.ppt - 1) / pl.ppt;
  int64_t lanes = groups * KIRCHHOFF_BLOCK;
  if (pl.ppt > 1 && !(pl.opts & KIRCHHOFF_OPT_NO_SHARE)) {
    const int64_t mesh = np / pl.ppt + np / 16 + 1;
    if (np / 16 >= 64 && mesh > lanes) lanes = mesh;
  }
  pl.tiles = (lanes + KIRCHHOFF_BLOCK - 1) / KIRCHHOFF_BLOCK;
  if (pl.tiles < 1) pl.tiles = 1;
  int nsplit = nsplit_req;
  if (nsplit <= 0) {
    // The kernel is VALU-bound and VGPR-limited to ~5 blocks per CU; many more
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
  // 256 B in front of the records hold what the scan kernel found
  pl.rec_bytes = sizeof(KirchhoffInfo) + (size_t)ns * KIRCHHOFF_REC_DOUBLES * sizeof(double);
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
  KirchhoffInfo* info = reinterpret_cast<KirchhoffInfo*>(workspace);
  double* rec = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) +
                                          sizeof(KirchhoffInfo));
  double* partial =
      reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) +
                                ((pl.rec_bytes + 255) / 256) * 256);
  hipError_t me = hipMemsetAsync(info, 0, sizeof(KirchhoffInfo), stream);
  if (me != hipSuccess) return me;
  if (np > 0 || ns > 0) {
    // at most 256 striding blocks per point set
    const int64_t wp = (np + 255) / 256, wsm = (ns + 255) / 256;
    const int nbp = (int)(wp < 256 ? wp : 256), nbs = (int)(wsm < 256 ? wsm : 256);
    hipLaunchKernelGGL(kirchhoff_scan, dim3((unsigned)(nbp + nbs)), dim3(256), 0, stream, np,
                       px, py, pz, ns, sx, sy, sz, pstride, nx, nz, nstride, k,
                       reinterpret_cast<const double2*>(Ep), nbp,
                       (unsigned)(pl.opts | (pl.ppt >= 4 ? KIRCHHOFF_OPT_TAB4096 : 0)), info);
  }
  if (ns > 0) {
    hipLaunchKernelGGL(kirchhoff_pack, dim3((unsigned)((ns + 255) / 256)), dim3(256),
                       0, stream, ns, sx, sy, sz, pstride, nx, ny, nz, nstride, nl, k,
                       reinterpret_cast<const double2*>(Es),
                       reinterpret_cast<const double2*>(Ep), info, rec);
  }
  if (np > 0) {
    dim3 grid((unsigned)(pl.tiles * pl.nsplit));
    if (ev0) (void)hipEventRecord(ev0, stream);
#define KIRCHHOFF_GO(N)                                                                   \
  hipLaunchKernelGGL(kirchhoff_stream<N>, grid, dim3(KIRCHHOFF_BLOCK), 0, stream, np, px, \
                     py, pz, (int)ns, rec, info, pl.nsplit, pl.chunk, pl.np_pad, partial)
    if (pl.ppt == 4)
      KIRCHHOFF_GO(4);
    else if (pl.ppt == 2)
      KIRCHHOFF_GO(2);
    else
      KIRCHHOFF_GO(1);
#undef KIRCHHOFF_GO
    if (ev1) (void)hipEventRecord(ev1, stream);
    hipLaunchKernelGGL(kirchhoff_finalize, dim3((unsigned)((np + 255) / 256)),
                       dim3(256), 0, stream, np, pl.nsplit, pl.np_pad, partial, info,
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

hipError_t debug_sqrt_seeded_launch(int64_t n, const double* x, const double* seed,
                                    double* r, double* h, hipStream_t stream) {
  hipLaunchKernelGGL(debug_sqrt_seeded_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256),
                     0, stream, n, x, seed, r, h);
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
  if (table == 2)
    hipLaunchKernelGGL(debug_sincos_tab_kernel<4096>, dim3((unsigned)((n + 255) / 256)),
                       dim3(256), 0, stream, n, phi, sn, cs);
  else if (table)
    hipLaunchKernelGGL(debug_sincos_tab_kernel<SINCOS_TAB_N>,
                       dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, phi, sn,
                       cs);
  else
    hipLaunchKernelGGL(debug_sincos_kernel, dim3((unsigned)((n + 255) / 256)),
                       dim3(256), 0, stream, n, phi, sn, cs);
  return hipGetLastError();
}

}  // namespace xrt
