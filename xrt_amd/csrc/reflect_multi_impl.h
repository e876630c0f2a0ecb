// OE.multiple_reflect (oes/reflect.py:165-264) for gfx950: ONE bounce off the surface per
// launch, the beam resident in HBM between the bounces.
//
// A bounce is _reflect_local (reflect.py:551-1139) in the form the loop calls it: `lb is
// vlb`, so the beam arrives and leaves in the element's VIRGIN local frame (bounce 0: arrives
// in the global frame), every entering ray moves to where the solver ended (lost rays too),
// rays over the edge (state 3) go back to where they were (reflect.py:225-228), nRefl counts
// the rays left in state 1 or 2. From the second bounce on the brackets are those of
// _bracketing(isMulti=True) (oes/base.py:1279-1289): the search for the next hit starts at the
// point between two bounces where the ray is farthest from the surface -- the root of
// ray . normal(x(t), y(t)) on [0, tMax], found by the same secant / Brent code with
// find_dz(derivOrder = 1) (base.py:819-821, 842-845), which has its own batch-global clamp
// range and method choice (base.py:861-878). So a bounce takes up to four dependent
// reductions over the batch:
//   directions -> axis and sign (base.py:1257-1270, :1239)
//   [isMulti] ray . normal at 0 and tMax -> clamp range and method of the tangency search
//   [isMulti] tangency of every ray; dz at it and at tMax -> range and method of the hit search
//   [first bounce] dz at the bracket ends -> the same
// They are phases of one launch whose blocks are all resident, with grid barriers in between
// (reflect_exact's scheme); the last phase solves, reflects (finish_ray: the code of the
// single-pass kernels) and writes the beam after the bounce, which is also this bounce's
// footprint in lbN.
#pragma once
#include "reflect_tu.h"

#define REFLECT_MULTI_BLOCK 256
// blocks per CU the kernel is compiled for (register budget 512 / (4 * N) VGPRs per lane)
#ifndef REFLECT_MULTI_PER_CU
#define REFLECT_MULTI_PER_CU 2
#endif

namespace xrt {

__device__ constexpr double kDs = 0.;   // raycing/__init__.py:91: margin of multiple reflections

// z - local_z (derivOrder 0) or ray . normal (derivOrder 1) at ray parameter t, base.py:801-846
template <class K, int DERIV>
__device__ __forceinline__ double multi_f(const xrt_hip_pass& P, double t, const LocalRay& r,
                                          double& x, double& y, double& z) {
  if (DERIV == 0) return find_dz<K>(P, t, r.x, r.y, r.z, r.a, r.b, r.c, x, y, z);
  x = r.x + r.a * t;
  y = r.y + r.b * t;
  z = r.z + r.c * t;
  if (surf_is_param<K>(P)) {
    double sp, phi, rr;
    ell_xyz_to_param(P, x, y, z, sp, phi, rr);
    x = sp;
    y = phi;
    z = rr;
  }
  double n[6];
  surface_normal<K>(P, x, y, x, y, n);
  return (r.a * n[3] + r.b * n[4] + r.c * n[5]) * (double)P.invert_normal;
}

struct Ends {
  double dz1, dz2;
  double x1, y1, z1, x2, y2, z2;
  bool ind1, ind2;
};
template <class K, int DERIV>
__device__ __forceinline__ Ends bracket_ends(const xrt_hip_pass& P, const LocalRay& r, double t1,
                                             double t2) {
  Ends e;
  e.dz1 = multi_f<K, DERIV>(P, t1, r, e.x1, e.y1, e.z1);
  e.dz2 = multi_f<K, DERIV>(P, t2, r, e.x2, e.y2, e.z2);
  e.ind1 = e.dz1 <= 0.;   // "lost": the solution is t1 (base.py:861)
  e.ind2 = e.dz2 >= 0.;   // "over": the solution is t2
  return e;
}

// find_intersection (base.py:848-885) between given ends, with the batch's clamp range and
// method: the iteration of solve_ray for either derivOrder. -> t, the point there as find_dz
// returns it ((s, phi, r) on a parametric surface) and ind1.
// OPT: the optimistic form (as solve_ray<K, true> of the single pass): secant, no clamp -- the
// batch's range is not known -- and an iterate that leaves the ray's own bracket is reported
// instead, with |f| at the bracket ends for the batch's secant-or-Brent decision.
template <class K, int DERIV, bool OPT = false>
__device__ __forceinline__ Hit solve_between(const xrt_hip_pass& P, const LocalRay& r, double t1,
                                             double t2, double tMinG, double tMaxG,
                                             bool use_brent, SolveAux* aux = nullptr) {
  Hit h;
  const Ends e = bracket_ends<K, DERIV>(P, r, t1, t2);
  const double t1own = t1, t2own = t2;
  if (OPT) {
    aux->adz1 = fabs(e.dz1);
    aux->adz2 = (e.ind1 || e.ind2) ? 0. : fabs(e.dz2);   // base.py:863-865
  }
  h.lost = e.ind1 ? 1 : 0;
  h.px = h.py = 0.;
  if (e.ind1) {
    h.t = t1;
    h.x = e.x1;
    h.y = e.y1;
    h.z = e.z1;
    return h;
  }
  if (e.ind2) {
    h.t = t2;
    h.x = e.x2;
    h.y = e.y2;
    h.z = e.z2;
    return h;
  }
  double dz1 = e.dz1, dz2 = e.dz2, x2 = e.x2, y2 = e.y2, z2 = e.z2;
  int numit = 2;
  if (!use_brent) {   // base.py:933-959
    bool active = true;
    while (active && numit < kMaxIteration) {
      const double t = t1, dz = dz1;
      t1 = t2;
      dz1 = dz2;
      t2 = t - (t1 - t) * dz / (dz1 - dz);
      if (OPT) {
        aux->escaped |= (t2 < t1own) || (t2 > t2own);
      } else {
        if (t2 < tMinG) t2 = tMinG;
        if (t2 > tMaxG) t2 = tMaxG;
      }
      dz2 = multi_f<K, DERIV>(P, t2, r, x2, y2, z2);
      if (same_sign(dz2, dz1)) {
        t1 = t;
        dz1 = dz;
      }
      active = fabs(dz2) > kZEps;
      ++numit;
    }
  } else {            // base.py:961-1048
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
      const double fs = multi_f<K, DERIV>(P, xs, r, x2, y2, z2);
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
  return h;
}

// a ray of the incoming beam in the virgin local frame (v: where a ray over the edge returns
// to) and in the true local frame (r), beamline.py:230-252 + reflect.py:617-635
__device__ __forceinline__ LocalRay multi_local(const xrt_hip_pass& P, const LocalRay& raw,
                                                double& vx, double& vy, double& vz) {
  LocalRay r = raw;
  if (P.in_is_global) {
    r.x = r.x - P.center[0];
    r.y = r.y - P.center[1];
    r.z = r.z - P.center[2];
    if (P.sin_az != 0.) {
      const double xn = P.cos_az * r.x - P.sin_az * r.y, yn = P.sin_az * r.x + P.cos_az * r.y;
      r.x = xn;
      r.y = yn;
    }
  }
  vx = r.x;
  vy = r.y;
  vz = r.z;
  rotate3(P.to_local, r.x, r.y, r.z);
  r.x -= P.shift[0];
  r.y -= P.shift[1];
  r.z -= P.shift[2];
  local_dir(P, r.a, r.b, r.c);
  return r;
}

__device__ __forceinline__ LocalRay multi_load(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                               int64_t i, LocalRay* raw = nullptr) {
  LocalRay q;
  q.x = in.x[i];
  q.y = in.y[i];
  q.z = in.z[i];
  q.a = in.a[i];
  q.b = in.b[i];
  q.c = in.c[i];
  if (raw) *raw = q;
  double vx, vy, vz;
  return multi_local(P, q, vx, vy, vz);
}

// one 4-double partial record per block: min t1, max t2, max |f(t1)|, max |f(t2)| over the
// entering rays (base.py:859-865), in the layout reduce_bracket_body folds
__device__ __forceinline__ void multi_write_part(double t1m, double t2m, double d1m, double d2m,
                                                 double* __restrict__ part) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
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

// statistics of the tangency search: f = ray . normal at t = 0 and at the far bracket end
template <class K>
__device__ __forceinline__ void multi_stats_tangency(const xrt_hip_pass& P,
                                                     const xrt_hip_beam& in, int axis,
                                                     int positive, double* __restrict__ part) {
  double t1m = INFINITY, t2m = -INFINITY, d1m = 0., d2m = 0.;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < in.n; i += stride) {
    const int st = in.state[i];
    const LocalRay r = multi_load(P, in, i);
    if (!entering(P, st)) continue;
    double t1, t2;
    bracket(P, axis, positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
    t1 = 0.;                                     // base.py:1280
    const Ends e = bracket_ends<K, 1>(P, r, t1, t2);
    const double dz2 = (e.ind1 || e.ind2) ? 0. : e.dz2;
    t1m = t1 < t1m ? t1 : t1m;
    t2m = t2 > t2m ? t2 : t2m;
    d1m = fmax(d1m, fabs(e.dz1));
    d2m = fmax(d2m, fabs(dz2));
  }
  multi_write_part(t1m, t2m, d1m, d2m, part);
}

// the tangency point of every entering ray (-> tang), and the statistics of the hit search
// that starts there
template <class K>
__device__ __forceinline__ void multi_tangency(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                               const GStat& g, double* __restrict__ tang,
                                               double* __restrict__ part) {
  double t1m = INFINITY, t2m = -INFINITY, d1m = 0., d2m = 0.;
  const bool brent = g.maxdz2 > g.maxdz1 * 20.;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < in.n; i += stride) {
    const int st = in.state[i];
    const LocalRay r = multi_load(P, in, i);
    if (!entering(P, st)) continue;
    double t1, t2;
    bracket(P, g.axis, g.positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
    const Hit hp = solve_between<K, 1>(P, r, 0., t2, g.t1min, g.t2max, brent);
    tang[i] = hp.t;
    t1 = hp.t + kDs;                             // base.py:1287
    const Ends e = bracket_ends<K, 0>(P, r, t1, t2);
    const double dz2 = (e.ind1 || e.ind2) ? 0. : e.dz2;
    t1m = t1 < t1m ? t1 : t1m;
    t2m = t2 > t2m ? t2 : t2m;
    d1m = fmax(d1m, fabs(e.dz1));
    d2m = fmax(d2m, fabs(dz2));
  }
  multi_write_part(t1m, t2m, d1m, d2m, part);
}

// last phase: the hit, the state, the reflection, the beam after the bounce
template <class K>
__device__ __forceinline__ void multi_finish(const xrt_hip_pass& P, const xrt_hip_material& M,
                                             const xrt_hip_beam& in, const xrt_hip_beam& out,
                                             const MultiAux& A, const GStat& g) {
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  const bool has_amp = in.Es_ri != nullptr;
  const bool brent = g.maxdz2 > g.maxdz1 * 20.;
  const bool param = surf_is_param<K>(P);
  const bool elevate = A.elev_out[0] != nullptr;
  unsigned long long kept = 0, hit = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < in.n; i += stride) {
    const int st0 = in.state[i];
    LocalRay raw;
    raw.x = in.x[i];
    raw.y = in.y[i];
    raw.z = in.z[i];
    raw.a = in.a[i];
    raw.b = in.b[i];
    raw.c = in.c[i];
    const int nr0 = A.nrefl_in ? A.nrefl_in[i] : 0;
    double el[4] = {-1., -kMaxHalfSize, -kMaxHalfSize, -kMaxHalfSize};   // reflect.py:214-218
    if (elevate && A.elev_in[0])
      for (int k = 0; k < 4; ++k) el[k] = A.elev_in[k][i];
    if (!entering(P, st0)) {
      copy_ray(out, in, i, st0, has_amp, false);
      A.nrefl_out[i] = nr0;
      A.theta[i] = 0.;
      if (elevate)
        for (int k = 0; k < 4; ++k) A.elev_out[k][i] = el[k];
      if (A.spr[0]) {   // reflect.py:1067-1069: copies of lb.x, y, z as they are
        A.spr[0][i] = raw.x;
        A.spr[1][i] = raw.y;
        A.spr[2][i] = raw.z;
      }
      continue;
    }
    double vx, vy, vz;
    const LocalRay r = multi_local(P, raw, vx, vy, vz);
    double t1, t2;
    bracket(P, g.axis, g.positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
    if (P.is_multi) {
      const double tg = A.tang[i];
      if (elevate) {   // base.py:1284-1286, reflect.py:651-659: find_dz at the tangency point
        double ex, ey, ez;
        el[0] = multi_f<K, 0>(P, tg, r, ex, ey, ez);
        if (param) {
          double cx, cy, cz;
          ell_param_to_xyz(P, ex, ey, ez, cx, cy, cz);
          ex = cx;
          ey = cy;
          ez = cz;
        }
        el[1] = ex;
        el[2] = ey;
        el[3] = ez;
      }
      t1 = tg + kDs;
    }
    Hit h = solve_between<K, 0>(P, r, t1, t2, g.t1min, g.t2max, brent);
    const double hs = h.x, hphi = h.y, hr = h.z;
    hit_done<K>(P, h);
    int st = rays_good<K>(P, h.x, h.y);
    if (h.lost) st = P.lost_num;
    RayIn q;
    q.path = in.path[i];
    q.E = in.E[i];
    double a = r.a, b = r.b, c = r.c, th = 0.;
    if (st == 1) {
      const Finished fin = finish_ray<K>(P, M, g, r, h, q, in, i, has_amp);
      a = fin.a;
      b = fin.b;
      c = fin.c;
      th = fin.theta;
      q = fin.lo;                 // (path + t, E)
      q.Jss = fin.vJss;           // lb is vlb: the matrix turned back is what stays
      q.Jpp = fin.vJpp;           // (reflect.py:1106-1110)
      q.Jsr = fin.vJsr;
      q.Jsi = fin.vJsi;
      q.Esr = fin.vEsr;
      q.Esi = fin.vEsi;
      q.Epr = fin.vEpr;
      q.Epi = fin.vEpi;
    } else {
      load_fields(in, i, has_amp, q);
    }
    // back to the virgin local frame (reflect.py:1115-1132), every entering ray
    double x = h.x + P.shift[0], y = h.y + P.shift[1], z = h.z + P.shift[2];
    rotate3(P.to_virgin, x, y, z);
    rotate3(P.to_virgin, a, b, c);
    if (st == 3) {                // reflect.py:225-228
      x = vx;
      y = vy;
      z = vz;
    }
    store_ray(out, i, x, y, z, a, b, c, q.path, q.E, q.Jss, q.Jpp, q.Jsr, q.Jsi, st, q.Esr,
              q.Esi, q.Epr, q.Epi, has_amp);
    const bool good = st == 1 || st == 2;
    A.nrefl_out[i] = nr0 + (good ? 1 : 0);
    A.theta[i] = th;
    if (elevate)
      for (int k = 0; k < 4; ++k) A.elev_out[k][i] = el[k];
    if (A.spr[0]) {
      A.spr[0][i] = hs;
      A.spr[1][i] = hphi;
      A.spr[2][i] = hr;
    }
    kept += good;
    hit += st == 1;
  }
  auto faddu = [](unsigned long long u, unsigned long long v) { return u + v; };
  kept = block_reduce(kept, faddu, lds_u);
  hit = block_reduce(hit, faddu, lds_u);
  if (threadIdx.x == 0) {
    if (kept) (void)atomicAdd(&A.counts[0], kept);
    if (hit) (void)atomicAdd(&A.counts[1], hit);
  }
}

// ---------------------------------------------------------------------------
// Round 6: the OPTIMISTIC form of a full bounce. The statistics phases of reflect_multi below are
// four passes over the beam and five grid barriers (0.7 ms of a 2.2-ms bounce at 1e7 rays) for
// decisions that come out the same for every beam that travels along the element: the axis of
// the largest direction cosine, the sign of the first entering ray, clamp ranges that never
// bite (a bracket-keeping secant iterate stays inside its own bracket, which lies inside the
// batch's range) and secant rather than Brent -- for BOTH searches of a bounce, the tangency
// point and the hit. So, as the single pass does (reflect_impl.h: decide_opt_body, fused_ray):
// multi_decide_opt assumes them from the head of the beam, reflect_multi_opt does the whole
// bounce per ray without any synchronisation -- tangency search, hit search, reflection, stores
// -- while every ray verifies the assumptions for itself and the |f| maxima at the bracket ends
// are collected (two sets of report slots, one per search); reflect_multi then opens with the
// verdict and returns at once unless something was contradicted, in which case it redoes the
// bounce exactly (same bits either way: the per-ray arithmetic is the same code). Which method
// each search takes is the caller's guess (xrt_hip_bounce.assume_*: the toroid's first two
// bounces take Brent for the hit, the later ones the secant; the verdict reports what the batch
// asks for, and the host remembers it per element and bounce).
// ---------------------------------------------------------------------------
#define MULTI_SLOTS2 (REFLECT_OPT_SLOTS + 8)      /* second set of report slots: behind the TabFast records */

template <class K>
__global__ __launch_bounds__(REFLECT_BLOCK) void multi_decide_opt(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_beam in, MultiAux A) {
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  int ub_lo[XRT_HIP_MAX_ELEM], ub_hi[XRT_HIP_MAX_ELEM];
  double dir0[3] = {0., 0., 0.};
  OptStat* slots = reinterpret_cast<OptStat*>(A.part);
  for (int k = threadIdx.x; k < REFLECT_OPT_SLOTS; k += blockDim.x) {
    slots[MULTI_SLOTS2 + k].maxdz1 = 0;
    slots[MULTI_SLOTS2 + k].maxdz2 = 0;
    slots[MULTI_SLOTS2 + k].viol = 0;
  }
  if (threadIdx.x == 0) {
    A.counts[0] = 0;
    A.counts[1] = 0;
  }
  // (P.method_hint is NULL here -- the launcher sees to it --: the single passes' hint is about
  // one search, a bounce has two)
  decide_opt_body(P, M, in, slots, A.g, lds_u, ub_lo, ub_hi, dir0);
}

// The arguments as ONE record, read phase by phase where they are used (reflect_impl.h:
// kernarg_at). Loaded in the entry block -- pass, material, two beams, the bounce's record: 570
// dwords for 100 SGPRs -- they lived in VGPR lanes: 2744 v_writelane / v_readlane among the
// kernel's 12074 VALU instructions, 600 of them inside the hit search's loop
// (profiles/r06_sgpr_late_ab.txt).
struct MultiArgs {
  xrt_hip_pass P;
  xrt_hip_material M;
  xrt_hip_beam in, out;
  MultiAux A;
};
#ifndef XRT_MULTI_EARLY_ARGS
template <class K>
__global__ __launch_bounds__(REFLECT_MULTI_BLOCK, REFLECT_MULTI_PER_CU) void reflect_multi_opt(
    MultiArgs R) {
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  constexpr unsigned oP = (unsigned)offsetof(MultiArgs, P), oM = (unsigned)offsetof(MultiArgs, M);
  constexpr unsigned oIn = (unsigned)offsetof(MultiArgs, in);
  constexpr unsigned oOut = (unsigned)offsetof(MultiArgs, out);
  constexpr unsigned oA = (unsigned)offsetof(MultiArgs, A);
  if (!R.A.g->optimistic) return;           // nothing could be assumed: reflect_multi does the bounce
  const GStat g = *R.A.g;
  const int64_t n = R.in.n;
  const bool has_amp = R.in.Es_ri != nullptr;
  const bool is_multi = R.P.is_multi != 0;
  const bool elevate = R.A.elev_out[0] != nullptr;
  const int assume = R.A.assume;
  unsigned long long kept = 0, hit = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
    const int64_t i = base + threadIdx.x;
    const bool live = i < n;
    const int64_t j = live ? i : n - 1;
    LocalRay raw;
    int st0;
    bool active;
    {
      const xrt_hip_beam& in = kernarg_at<xrt_hip_beam>(oIn);
      st0 = live ? in.state[j] : 0;
      raw.x = in.x[j];
      raw.y = in.y[j];
      raw.z = in.z[j];
      raw.a = in.a[j];
      raw.b = in.b[j];
      raw.c = in.c[j];
      active = live && entering(kernarg_at<xrt_hip_pass>(oP), st0);
    }
    SolveAux aux1, aux2;
    int viol = 0;
    if (live && !active) {
      const xrt_hip_beam& in = kernarg_at<xrt_hip_beam>(oIn);
      const xrt_hip_beam& out = kernarg_at<xrt_hip_beam>(oOut);
      const MultiAux& A = kernarg_at<MultiAux>(oA);
      const int nr0 = A.nrefl_in ? A.nrefl_in[i] : 0;
      copy_ray(out, in, i, st0, has_amp, false);
      A.nrefl_out[i] = nr0;
      A.theta[i] = 0.;
      if (elevate) {
        const double el0[4] = {-1., -kMaxHalfSize, -kMaxHalfSize, -kMaxHalfSize};
        for (int k = 0; k < 4; ++k) A.elev_out[k][i] = A.elev_in[0] ? A.elev_in[k][i] : el0[k];
      }
      if (A.spr[0]) {   // reflect.py:1067-1069: copies of lb.x, y, z as they are
        A.spr[0][i] = raw.x;
        A.spr[1][i] = raw.y;
        A.spr[2][i] = raw.z;
      }
    }
    if (active) {
      int nr0;
      double el[4] = {-1., -kMaxHalfSize, -kMaxHalfSize, -kMaxHalfSize};   // reflect.py:214-218
      {
        const MultiAux& A = kernarg_at<MultiAux>(oA);
        nr0 = A.nrefl_in ? A.nrefl_in[i] : 0;
        if (elevate && A.elev_in[0])
          for (int k = 0; k < 4; ++k) el[k] = A.elev_in[k][i];
      }
      double vx, vy, vz, t1, t2;
      LocalRay r;
      {
        const xrt_hip_pass& P = kernarg_at<xrt_hip_pass>(oP);
        r = multi_local(P, raw, vx, vy, vz);
        // the axis stands only if its cosine dominates every state-1 ray's own (fused_ray)
        viol = st0 == 1 && !dominates(g.axis, r);
        bracket(P, g.axis, g.positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
      }
      if (is_multi) {
        const xrt_hip_pass& P = kernarg_at<xrt_hip_pass>(oP);
        const Hit hp = solve_between<K, 1, true>(P, r, 0., t2, 0., 0., (assume & 2) != 0, &aux1);
        const double tg = hp.t;
        if (elevate) {   // base.py:1284-1286, reflect.py:651-659: find_dz at the tangency point
          const xrt_hip_pass& Pe = kernarg_at<xrt_hip_pass>(oP);
          double ex, ey, ez;
          el[0] = multi_f<K, 0>(Pe, tg, r, ex, ey, ez);
          if (surf_is_param<K>(Pe)) {
            double cx, cy, cz;
            ell_param_to_xyz(Pe, ex, ey, ez, cx, cy, cz);
            ex = cx;
            ey = cy;
            ez = cz;
          }
          el[1] = ex;
          el[2] = ey;
          el[3] = ez;
        }
        t1 = tg + kDs;
      }
      Hit h;
      {
        const xrt_hip_pass& P = kernarg_at<xrt_hip_pass>(oP);
        h = solve_between<K, 0, true>(P, r, t1, t2, 0., 0., (assume & 1) != 0, &aux2);
      }
      const double hs = h.x, hphi = h.y, hr = h.z;
      const xrt_hip_pass& P = kernarg_at<xrt_hip_pass>(oP);
      hit_done<K>(P, h);
      int st = rays_good<K>(P, h.x, h.y);
      if (h.lost) st = P.lost_num;
      const xrt_hip_beam& in = kernarg_at<xrt_hip_beam>(oIn);
      RayIn q;
      q.path = in.path[i];
      q.E = in.E[i];
      double a = r.a, b = r.b, c = r.c, th = 0.;
      if (st == 1) {
        const Finished fin =
            finish_ray<K>(P, kernarg_at<xrt_hip_material>(oM), g, r, h, q, in, i, has_amp);
        a = fin.a;
        b = fin.b;
        c = fin.c;
        th = fin.theta;
        q = fin.lo;                 // (path + t, E)
        q.Jss = fin.vJss;           // lb is vlb: the matrix turned back is what stays
        q.Jpp = fin.vJpp;           // (reflect.py:1106-1110)
        q.Jsr = fin.vJsr;
        q.Jsi = fin.vJsi;
        q.Esr = fin.vEsr;
        q.Esi = fin.vEsi;
        q.Epr = fin.vEpr;
        q.Epi = fin.vEpi;
      } else {
        load_fields(in, i, has_amp, q);
      }
      // back to the virgin local frame (reflect.py:1115-1132), every entering ray
      const xrt_hip_pass& Pv = kernarg_at<xrt_hip_pass>(oP);
      double x = h.x + Pv.shift[0], y = h.y + Pv.shift[1], z = h.z + Pv.shift[2];
      rotate3(Pv.to_virgin, x, y, z);
      rotate3(Pv.to_virgin, a, b, c);
      if (st == 3) {                // reflect.py:225-228
        x = vx;
        y = vy;
        z = vz;
      }
      const xrt_hip_beam& out = kernarg_at<xrt_hip_beam>(oOut);
      const MultiAux& A = kernarg_at<MultiAux>(oA);
      store_ray(out, i, x, y, z, a, b, c, q.path, q.E, q.Jss, q.Jpp, q.Jsr, q.Jsi, st, q.Esr,
                q.Esi, q.Epr, q.Epi, has_amp);
      const bool good = st == 1 || st == 2;
      A.nrefl_out[i] = nr0 + (good ? 1 : 0);
      A.theta[i] = th;
      if (elevate)
        for (int k = 0; k < 4; ++k) A.elev_out[k][i] = el[k];
      if (A.spr[0]) {
        A.spr[0][i] = hs;
        A.spr[1][i] = hphi;
        A.spr[2][i] = hr;
      }
      kept += good;
      hit += st == 1;
    }
    // all lanes of the wave together again: the reports (the tangency search's, the hit search's)
    OptStat* slots1 = reinterpret_cast<OptStat*>(kernarg_at<MultiAux>(oA).part);
    OptStat* slots2 = slots1 + MULTI_SLOTS2;
    if (is_multi) report_opt(slots1, aux1, viol);
    report_opt(is_multi ? slots2 : slots1, aux2, is_multi ? 0 : viol);
  }
  auto faddu = [](unsigned long long u, unsigned long long v) { return u + v; };
  kept = block_reduce(kept, faddu, lds_u);
  hit = block_reduce(hit, faddu, lds_u);
  if (threadIdx.x == 0) {
    unsigned long long* counts = kernarg_at<MultiAux>(oA).counts;
    if (kept) (void)atomicAdd(&counts[0], kept);
    if (hit) (void)atomicAdd(&counts[1], hit);
  }
}
#else
template <class K>
__global__ __launch_bounds__(REFLECT_MULTI_BLOCK, REFLECT_MULTI_PER_CU) void reflect_multi_opt(
    MultiArgs R) {
  // (A/B: every argument named, all loaded in the entry block)
  const xrt_hip_pass& P = R.P;
  const xrt_hip_material& M = R.M;
  const xrt_hip_beam &in = R.in, &out = R.out;
  const MultiAux& A = R.A;
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  if (!A.g->optimistic) return;             // nothing could be assumed: reflect_multi does the bounce
  const GStat g = *A.g;
  OptStat* slots1 = reinterpret_cast<OptStat*>(A.part);
  OptStat* slots2 = slots1 + MULTI_SLOTS2;
  const bool has_amp = in.Es_ri != nullptr;
  const bool param = surf_is_param<K>(P);
  const bool elevate = A.elev_out[0] != nullptr;
  unsigned long long kept = 0, hit = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < in.n; base += stride) {
    const int64_t i = base + threadIdx.x;
    const bool live = i < in.n;
    const int64_t j = live ? i : in.n - 1;
    const int st0 = live ? in.state[j] : 0;
    LocalRay raw;
    raw.x = in.x[j];
    raw.y = in.y[j];
    raw.z = in.z[j];
    raw.a = in.a[j];
    raw.b = in.b[j];
    raw.c = in.c[j];
    const bool active = live && entering(P, st0);
    SolveAux aux1, aux2;
    int viol = 0;
    if (live && !active) {
      const int nr0 = A.nrefl_in ? A.nrefl_in[i] : 0;
      copy_ray(out, in, i, st0, has_amp, false);
      A.nrefl_out[i] = nr0;
      A.theta[i] = 0.;
      if (elevate) {
        const double el0[4] = {-1., -kMaxHalfSize, -kMaxHalfSize, -kMaxHalfSize};
        for (int k = 0; k < 4; ++k) A.elev_out[k][i] = A.elev_in[0] ? A.elev_in[k][i] : el0[k];
      }
      if (A.spr[0]) {   // reflect.py:1067-1069: copies of lb.x, y, z as they are
        A.spr[0][i] = raw.x;
        A.spr[1][i] = raw.y;
        A.spr[2][i] = raw.z;
      }
    }
    if (active) {
      const int nr0 = A.nrefl_in ? A.nrefl_in[i] : 0;
      double el[4] = {-1., -kMaxHalfSize, -kMaxHalfSize, -kMaxHalfSize};   // reflect.py:214-218
      if (elevate && A.elev_in[0])
        for (int k = 0; k < 4; ++k) el[k] = A.elev_in[k][i];
      double vx, vy, vz;
      const LocalRay r = multi_local(P, raw, vx, vy, vz);
      // the axis stands only if its cosine dominates every state-1 ray's own (fused_ray)
      viol = st0 == 1 && !dominates(g.axis, r);
      double t1, t2;
      bracket(P, g.axis, g.positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
      if (P.is_multi) {
        const Hit hp = solve_between<K, 1, true>(P, r, 0., t2, 0., 0., (A.assume & 2) != 0, &aux1);
        const double tg = hp.t;
        if (elevate) {   // base.py:1284-1286, reflect.py:651-659: find_dz at the tangency point
          double ex, ey, ez;
          el[0] = multi_f<K, 0>(P, tg, r, ex, ey, ez);
          if (param) {
            double cx, cy, cz;
            ell_param_to_xyz(P, ex, ey, ez, cx, cy, cz);
            ex = cx;
            ey = cy;
            ez = cz;
          }
          el[1] = ex;
          el[2] = ey;
          el[3] = ez;
        }
        t1 = tg + kDs;
      }
      Hit h = solve_between<K, 0, true>(P, r, t1, t2, 0., 0., (A.assume & 1) != 0, &aux2);
      const double hs = h.x, hphi = h.y, hr = h.z;
      hit_done<K>(P, h);
      int st = rays_good<K>(P, h.x, h.y);
      if (h.lost) st = P.lost_num;
      RayIn q;
      q.path = in.path[i];
      q.E = in.E[i];
      double a = r.a, b = r.b, c = r.c, th = 0.;
      if (st == 1) {
        const Finished fin = finish_ray<K>(P, M, g, r, h, q, in, i, has_amp);
        a = fin.a;
        b = fin.b;
        c = fin.c;
        th = fin.theta;
        q = fin.lo;                 // (path + t, E)
        q.Jss = fin.vJss;           // lb is vlb: the matrix turned back is what stays
        q.Jpp = fin.vJpp;           // (reflect.py:1106-1110)
        q.Jsr = fin.vJsr;
        q.Jsi = fin.vJsi;
        q.Esr = fin.vEsr;
        q.Esi = fin.vEsi;
        q.Epr = fin.vEpr;
        q.Epi = fin.vEpi;
      } else {
        load_fields(in, i, has_amp, q);
      }
      // back to the virgin local frame (reflect.py:1115-1132), every entering ray
      double x = h.x + P.shift[0], y = h.y + P.shift[1], z = h.z + P.shift[2];
      rotate3(P.to_virgin, x, y, z);
      rotate3(P.to_virgin, a, b, c);
      if (st == 3) {                // reflect.py:225-228
        x = vx;
        y = vy;
        z = vz;
      }
      store_ray(out, i, x, y, z, a, b, c, q.path, q.E, q.Jss, q.Jpp, q.Jsr, q.Jsi, st, q.Esr,
                q.Esi, q.Epr, q.Epi, has_amp);
      const bool good = st == 1 || st == 2;
      A.nrefl_out[i] = nr0 + (good ? 1 : 0);
      A.theta[i] = th;
      if (elevate)
        for (int k = 0; k < 4; ++k) A.elev_out[k][i] = el[k];
      if (A.spr[0]) {
        A.spr[0][i] = hs;
        A.spr[1][i] = hphi;
        A.spr[2][i] = hr;
      }
      kept += good;
      hit += st == 1;
    }
    // all lanes of the wave together again: the reports (the tangency search's, the hit search's)
    if (P.is_multi) report_opt(slots1, aux1, viol);
    report_opt(P.is_multi ? slots2 : slots1, aux2, P.is_multi ? 0 : viol);
  }
  auto faddu = [](unsigned long long u, unsigned long long v) { return u + v; };
  kept = block_reduce(kept, faddu, lds_u);
  hit = block_reduce(hit, faddu, lds_u);
  if (threadIdx.x == 0) {
    if (kept) (void)atomicAdd(&A.counts[0], kept);
    if (hit) (void)atomicAdd(&A.counts[1], hit);
  }
}
#endif

// The launch is preceded by reflect_init(g, 1) (decisions reset, barrier counter zeroed).
template <class K>
__global__ __launch_bounds__(REFLECT_MULTI_BLOCK, REFLECT_MULTI_PER_CU) void reflect_multi(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_beam in, xrt_hip_beam out, MultiAux A) {
  GStat* g = A.g;
  unsigned phase = 0;
  if (A.gate) {
    // behind reflect_multi_opt: the verdict, folded by every block for itself (exact_gate's
    // scheme); nothing contradicted -> nothing to do
    __shared__ double lds_d[REFLECT_MAX_WAVES];
    bool full = true;
    if (g->optimistic) {
      // (slots: the tangency search's reports and, behind them, the hit search's -- of a first
      // bounce, which has no tangency search, the hit search's alone)
      const OptStat* slots = reinterpret_cast<const OptStat*>(A.part);
      double m1, m2, h1, h2;
      if (P.is_multi) {
        full = fold_opt(slots, lds_d, m1, m2, (A.assume & 2) != 0);
        full = fold_opt(slots + MULTI_SLOTS2, lds_d, h1, h2, (A.assume & 1) != 0) || full;
      } else {
        full = fold_opt(slots, lds_d, h1, h2, (A.assume & 1) != 0);
        m1 = m2 = 0.;
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.diag[9] = h2 > h1 * 20. ? 1. : 0.;       // what this batch asks for: hit search,
        A.diag[10] = m2 > m1 * 20. ? 1. : 0.;      // tangency search
      }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      g->redo = full ? 1 : 0;
      A.diag[11] = full ? 1. : 0.;
      A.diag[12] = 1.;
    }
    if (!full) return;
  } else if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.diag[11] = 0.;
    A.diag[12] = 0.;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.counts[0] = 0;
    A.counts[1] = 0;
  }
  stats_dir_body(P, in, A.part);
  grid_barrier(g, phase);
  if (blockIdx.x == 0) decide_axis_body(P, M, in, A.part, (int)gridDim.x, 8, g);
  grid_barrier(g, phase);
  GStat gl = load_gstat(g);
  if (gl.n_enter > 0) {
    if (P.is_multi) {
      multi_stats_tangency<K>(P, in, gl.axis, gl.positive, A.part);
      grid_barrier(g, phase);
      if (blockIdx.x == 0) reduce_bracket_body(A.part, (int)gridDim.x, g);
      grid_barrier(g, phase);
      gl = load_gstat(g);
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.diag[4] = gl.maxdz2 > gl.maxdz1 * 20. ? 1. : 0.;
        A.diag[10] = A.diag[4];
        A.diag[5] = gl.t1min;
        A.diag[6] = gl.t2max;
      }
      multi_tangency<K>(P, in, gl, A.tang, A.part);
    } else {
      stats_bracket_body<K>(P, in, gl.axis, gl.positive, A.part);
    }
    grid_barrier(g, phase);
    if (blockIdx.x == 0) reduce_bracket_body(A.part, (int)gridDim.x, g);
    grid_barrier(g, phase);
    gl = load_gstat(g);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.diag[0] = (double)gl.axis;
    A.diag[1] = (double)gl.positive;
    A.diag[2] = gl.maxdz2 > gl.maxdz1 * 20. ? 1. : 0.;
    A.diag[9] = A.diag[2];
    if (!P.is_multi) A.diag[10] = 0.;
    A.diag[3] = (double)gl.n_enter;
    A.diag[7] = gl.t1min;
    A.diag[8] = gl.t2max;
  }
  multi_finish<K>(P, M, in, out, A, gl);
}

// ---------------------------------------------------------------------------
// Round 6: the SPARSE form of a bounce, taken when the caller says that fewer than a quarter of
// the rays still enter (xrt_hip_bounce.entering_hint: what the previous bounce counted). The dense
// kernel above walks every lane through every phase: a bounce in which 4 % of the rays are left
// took 1.95 ms of the 2.2-3.8 ms of a full one, one with none left 1.1 ms (1e7 rays,
// profiles/r06_multiple_reflect.txt). Here a bounce is three launches, and no lane waits for a
// finished ray or for another lane's ray.
//   reflect_multi_stats  -- the batch decisions (phases between grid barriers, as before), over
//       a compacted INDEX of the entering rays that its first phase builds from the states
//       (segments of MULTI_SEG rays, each with its own list and count: no atomics, no global
//       prefix; from the third bounce on a good part of the rays is finished, and a wave that
//       holds one live ray costs as much as a full one);
//   reflect_multi_solve  -- the root search for the hit point. The reference's bracket-keeping
//       secant converges linearly on a grazing surface: 20 to 50 iterations per ray, the slowest
//       ray of 64 twice the median. So a lane whose ray is done takes the NEXT ray of its block's
//       list (a counter in LDS) instead of idling until the slowest lane of its wave is through:
//       every ray's iterates are the same numbers as before, only which lane computes them
//       changes. The hit records go to scratch;
//   reflect_multi_finish -- dense over the index: state, reflection (finish_ray), stores; the
//       rays that did not enter are copied through by a streaming sweep of the same blocks.
// The solve kernel holds no reflection code and the finish kernel no iteration: each gets the
// registers it needs instead of the maximum of both. For a FULL beam this form is slower than the
// one launch (2.7 against 2.2 ms per bounce: the search is arithmetic-bound, the finish bound by
// the bytes of a bounce, and as separate launches their times add; profiles/r06_multi_ab.txt).
// ---------------------------------------------------------------------------
#define MULTI_SEG 1024          // rays per segment of the index
#define MULTI_REFILL 16         // idle lanes of a wave that make it fetch new rays
#define MULTI_MAX_SEGS 8192     // segments one block of the solve kernel can own

// number of set lanes below this one
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// phase 0 of reflect_multi_stats: idx[seg * MULTI_SEG + k] = the k-th entering ray of segment
// seg (ascending), cnt[seg] = how many
__device__ __forceinline__ void multi_build_index(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                                  const MultiAux& A) {
  constexpr int PARTS = MULTI_SEG / REFLECT_MULTI_BLOCK;     // states a lane looks at per segment
  __shared__ int wave_tot[PARTS][REFLECT_MAX_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int seg = blockIdx.x; seg < A.nseg; seg += gridDim.x) {
    // all states of the segment are requested before any is looked at, one barrier per segment
    // (a dependent load and two barriers per 256 rays made this phase 0.3 ms of latency)
    bool ent[PARTS];
    unsigned long long m[PARTS];
#pragma unroll
    for (int u = 0; u < PARTS; ++u) {
      const int64_t i = (int64_t)seg * MULTI_SEG + u * REFLECT_MULTI_BLOCK + threadIdx.x;
      ent[u] = i < in.n && entering(P, in.state[i < in.n ? i : in.n - 1]);
    }
#pragma unroll
    for (int u = 0; u < PARTS; ++u) {
      m[u] = __ballot(ent[u]);
      if (lane == 0) wave_tot[u][wave] = __popcll(m[u]);
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int u = 0; u < PARTS; ++u) {
      int before = 0, all = 0;
      for (int w = 0; w < nw; ++w) {
        const int t = wave_tot[u][w];
        before += w < wave ? t : 0;
        all += t;
      }
      if (ent[u])
        A.idx[(int64_t)seg * MULTI_SEG + run + before + lanes_below(m[u])] =
            (int32_t)(u * REFLECT_MULTI_BLOCK + threadIdx.x);
      run += all;
    }
    if (threadIdx.x == 0) A.cnt[seg] = run;
    __syncthreads();
  }
}

// f(i) for every entering ray, this block's share (segments in turn)
template <class F>
__device__ __forceinline__ void multi_for_each(const MultiAux& A, F f) {
  for (int seg = blockIdx.x; seg < A.nseg; seg += gridDim.x) {
    const int n = A.cnt[seg];
    const int32_t* list = A.idx + (int64_t)seg * MULTI_SEG;
    for (int k = threadIdx.x; k < n; k += blockDim.x) f((int64_t)seg * MULTI_SEG + list[k]);
  }
}

// stats_dir_body (reflect_impl.h) over the index: the same partial record
__device__ __forceinline__ void multi_stats_dir(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                                const MultiAux& A) {
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  double ma = 0., mb = 0., mc = 0., emin = INFINITY, emax = -INFINITY;
  unsigned long long first = ~0ull, nent = 0, nmain = 0;
  multi_for_each(A, [&](int64_t i) {
    const int st = in.state[i];
    const double E = in.E[i];
    double a = in.a[i], b = in.b[i], c = in.c[i];
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
  });
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
  if (threadIdx.x == 0) {
    double* o = A.part + (int64_t)blockIdx.x * 8;
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

// stats_bracket_body (reflect_impl.h) over the index: first bounce, dz at the bracket ends
template <class K>
__device__ __forceinline__ void multi_stats_bracket(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                                    const MultiAux& A, int axis, int positive) {
  double t1m = INFINITY, t2m = -INFINITY, d1m = 0., d2m = 0.;
  multi_for_each(A, [&](int64_t i) {
    const LocalRay r = load_local(P, in, i);
    double t1, t2, x, y, z;
    bracket(P, axis, positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
    const double dz1 = find_dz<K>(P, t1, r.x, r.y, r.z, r.a, r.b, r.c, x, y, z);
    double dz2 = find_dz<K>(P, t2, r.x, r.y, r.z, r.a, r.b, r.c, x, y, z);
    if (dz1 <= 0. || dz2 >= 0.) dz2 = 0.;  // base.py:863-865
    t1m = t1 < t1m ? t1 : t1m;
    t2m = t2 > t2m ? t2 : t2m;
    d1m = fmax(d1m, fabs(dz1));
    d2m = fmax(d2m, fabs(dz2));
  });
  multi_write_part(t1m, t2m, d1m, d2m, A.part);
}

// statistics of the tangency search: f = ray . normal at t = 0 and at the far bracket end
template <class K>
__device__ __forceinline__ void sparse_stats_tangency(const xrt_hip_pass& P,
                                                     const xrt_hip_beam& in, const MultiAux& A,
                                                     int axis, int positive) {
  double t1m = INFINITY, t2m = -INFINITY, d1m = 0., d2m = 0.;
  multi_for_each(A, [&](int64_t i) {
    const LocalRay r = multi_load(P, in, i);
    double t1, t2;
    bracket(P, axis, positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
    t1 = 0.;                                     // base.py:1280
    const Ends e = bracket_ends<K, 1>(P, r, t1, t2);
    const double dz2 = (e.ind1 || e.ind2) ? 0. : e.dz2;
    t1m = t1 < t1m ? t1 : t1m;
    t2m = t2 > t2m ? t2 : t2m;
    d1m = fmax(d1m, fabs(e.dz1));
    d2m = fmax(d2m, fabs(dz2));
  });
  multi_write_part(t1m, t2m, d1m, d2m, A.part);
}

// the tangency point of every entering ray (-> tang), and the statistics of the hit search
// that starts there
template <class K>
__device__ __forceinline__ void sparse_tangency(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                               const MultiAux& A, const GStat& g) {
  double t1m = INFINITY, t2m = -INFINITY, d1m = 0., d2m = 0.;
  const bool brent = g.maxdz2 > g.maxdz1 * 20.;
  multi_for_each(A, [&](int64_t i) {
    const LocalRay r = multi_load(P, in, i);
    double t1, t2;
    bracket(P, g.axis, g.positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
    const Hit hp = solve_between<K, 1>(P, r, 0., t2, g.t1min, g.t2max, brent);
    A.tang[i] = hp.t;
    t1 = hp.t + kDs;                             // base.py:1287
    const Ends e = bracket_ends<K, 0>(P, r, t1, t2);
    const double dz2 = (e.ind1 || e.ind2) ? 0. : e.dz2;
    t1m = t1 < t1m ? t1 : t1m;
    t2m = t2 > t2m ? t2 : t2m;
    d1m = fmax(d1m, fabs(e.dz1));
    d2m = fmax(d2m, fabs(dz2));
  });
  multi_write_part(t1m, t2m, d1m, d2m, A.part);
}

// The launch is preceded by multi_init (decisions reset, barrier counter zeroed).
template <class K>
__global__ __launch_bounds__(REFLECT_MULTI_BLOCK) void reflect_multi_stats(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_beam in, MultiAux A) {
  GStat* g = A.g;
  unsigned phase = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.counts[0] = 0;
    A.counts[1] = 0;
  }
  multi_build_index(P, in, A);
  grid_barrier(g, phase);
  multi_stats_dir(P, in, A);
  grid_barrier(g, phase);
  if (blockIdx.x == 0) decide_axis_body(P, M, in, A.part, (int)gridDim.x, 8, g);
  grid_barrier(g, phase);
  GStat gl = load_gstat(g);
  if (gl.n_enter > 0) {
    if (P.is_multi) {
      sparse_stats_tangency<K>(P, in, A, gl.axis, gl.positive);
      grid_barrier(g, phase);
      if (blockIdx.x == 0) reduce_bracket_body(A.part, (int)gridDim.x, g);
      grid_barrier(g, phase);
      gl = load_gstat(g);
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.diag[4] = gl.maxdz2 > gl.maxdz1 * 20. ? 1. : 0.;
        A.diag[10] = A.diag[4];
        A.diag[5] = gl.t1min;
        A.diag[6] = gl.t2max;
      }
      sparse_tangency<K>(P, in, A, gl);
    } else {
      multi_stats_bracket<K>(P, in, A, gl.axis, gl.positive);
    }
    grid_barrier(g, phase);
    if (blockIdx.x == 0) reduce_bracket_body(A.part, (int)gridDim.x, g);
  }
  if (blockIdx.x == 0) {       // (this block wrote the decisions: it reads them back itself)
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      A.diag[0] = (double)ld_agent_i(&g->axis);
      A.diag[1] = (double)ld_agent_i(&g->positive);
      const double d1 = ld_agent(&g->maxdz1), d2 = ld_agent(&g->maxdz2);
      A.diag[2] = d2 > d1 * 20. ? 1. : 0.;
      A.diag[9] = A.diag[2];
      if (!P.is_multi) A.diag[10] = 0.;
      A.diag[11] = 0.;
      A.diag[12] = 2.;
      A.diag[3] = (double)ld_agent_u(&g->n_enter);
      A.diag[7] = ld_agent(&g->t1min);
      A.diag[8] = ld_agent(&g->t2max);
    }
  }
}

// ---- the hit search -----------------------------------------------------------------------
struct HitSearch {          // one ray's find_intersection (base.py:848-1048) between steps
  LocalRay r;
  double t1, t2, dz1, dz2, x2, y2, z2;
  double t3, dz3, t4;       // Brent
  bool mflag;
  int numit;
  int64_t i;
};

__device__ __forceinline__ void multi_put_hit(const MultiAux& A, int64_t i, double t, double x,
                                              double y, double z, int lost) {
  A.ht[i] = t;
  A.hx[i] = x;
  A.hy[i] = y;
  A.hz[i] = z;
  A.hlost[i] = lost;
}

// ray i enters the search: its bracket, the values at the ends. -> true if it has to iterate
// (then S holds it), false if an end of the bracket is the answer (written out)
template <class K>
__device__ __forceinline__ bool multi_hit_begin(const xrt_hip_pass& P, const xrt_hip_beam& in,
                                                const MultiAux& A, const GStat& g, bool brent,
                                                int64_t i, HitSearch& S) {
  LocalRay raw;
  raw.x = in.x[i];
  raw.y = in.y[i];
  raw.z = in.z[i];
  raw.a = in.a[i];
  raw.b = in.b[i];
  raw.c = in.c[i];
  double vx, vy, vz;
  const LocalRay r = multi_local(P, raw, vx, vy, vz);
  double t1, t2;
  bracket(P, g.axis, g.positive, r.x, r.y, r.z, r.a, r.b, r.c, t1, t2);
  if (P.is_multi) t1 = A.tang[i] + kDs;
  const Ends e = bracket_ends<K, 0>(P, r, t1, t2);
  if (e.ind1) {
    multi_put_hit(A, i, t1, e.x1, e.y1, e.z1, 1);
    return false;
  }
  if (e.ind2) {
    multi_put_hit(A, i, t2, e.x2, e.y2, e.z2, 0);
    return false;
  }
  S.r = r;
  S.i = i;
  S.t1 = t1;
  S.t2 = t2;
  S.dz1 = e.dz1;
  S.dz2 = e.dz2;
  S.x2 = e.x2;
  S.y2 = e.y2;
  S.z2 = e.z2;
  S.numit = 2;
  S.t3 = S.dz3 = S.t4 = 0.;
  S.mflag = true;
  if (brent) {            // base.py:961-976
    if (fabs(S.dz1) < fabs(S.dz2)) {
      double tmp = S.t1;
      S.t1 = S.t2;
      S.t2 = tmp;
      tmp = S.dz1;
      S.dz1 = S.dz2;
      S.dz2 = tmp;
    }
    S.t3 = S.t1;
    S.dz3 = S.dz1;
    S.t4 = 0.;
    if (!(fabs(S.dz2) > kZEps)) {       // (no iteration at all)
      multi_put_hit(A, i, S.t2, S.x2, S.y2, S.z2, 0);
      return false;
    }
  }
  return true;
}

// one step of solve_between's secant loop -> true while the ray goes on
template <class K>
__device__ __forceinline__ bool multi_secant_step(const xrt_hip_pass& P, const GStat& g,
                                                  HitSearch& S) {
  const double t = S.t1, dz = S.dz1;
  S.t1 = S.t2;
  S.dz1 = S.dz2;
  S.t2 = t - (S.t1 - t) * dz / (S.dz1 - dz);
  if (S.t2 < g.t1min) S.t2 = g.t1min;
  if (S.t2 > g.t2max) S.t2 = g.t2max;
  S.dz2 = multi_f<K, 0>(P, S.t2, S.r, S.x2, S.y2, S.z2);
  if (same_sign(S.dz2, S.dz1)) {
    S.t1 = t;
    S.dz1 = dz;
  }
  ++S.numit;
  return fabs(S.dz2) > kZEps && S.numit < kMaxIteration;
}

// ... and of its Brent loop
template <class K>
__device__ __forceinline__ bool multi_brent_step(const xrt_hip_pass& P, HitSearch& S) {
  double xa = S.t1, xb = S.t2, xc = S.t3, xd = S.t4;
  double fa = S.dz1, fb = S.dz2, fc = S.dz3;
  double xs;
  if (fa != fc && fb != fc) {
    xs = xa * fb * fc / (fa - fb) / (fa - fc) + fa * xb * fc / (fb - fa) / (fb - fc) +
         fa * fb * xc / (fc - fa) / (fc - fb);
  } else {
    xs = xb - fb * (xb - xa) / (fb - fa);
  }
  const double q = (3. * xa + xb) / 4.;
  const bool cond1 = ((xs < q) && (xs < xb)) || ((xs > q) && (xs > xb));
  const bool cond2 = S.mflag && (fabs(xs - xb) >= (fabs(xb - xc) / 2.));
  const bool cond3 = (!S.mflag) && (fabs(xs - xb) >= (fabs(xc - xd) / 2.));
  const bool cond4 = S.mflag && (fabs(xb - xc) < kZEps);
  const bool cond5 = (!S.mflag) && (fabs(xc - xd) < kZEps);
  const bool conds = cond1 || cond2 || cond3 || cond4 || cond5;
  if (conds) xs = (xa + xb) / 2.;
  S.mflag = conds;
  const double fs = multi_f<K, 0>(P, xs, S.r, S.x2, S.y2, S.z2);
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
  S.t1 = xa;
  S.t2 = xb;
  S.t3 = xc;
  S.t4 = xd;
  S.dz1 = fa;
  S.dz2 = fb;
  S.dz3 = fc;
  ++S.numit;
  return fabs(S.dz2) > kZEps && S.numit < kMaxIteration;
}

template <class K>
__global__ __launch_bounds__(REFLECT_MULTI_BLOCK) void reflect_multi_solve(
    xrt_hip_pass P, xrt_hip_beam in, MultiAux A) {
  __shared__ int pre[MULTI_MAX_SEGS + 1];      // entering rays before each of this block's segments
  __shared__ int next;
  const GStat g = *A.g;                         // (written by the launch before this one)
  if (g.n_enter == 0) return;
  const bool brent = g.maxdz2 > g.maxdz1 * 20.;
  const int s0 = (int)((int64_t)A.nseg * blockIdx.x / gridDim.x);
  const int s1 = (int)((int64_t)A.nseg * (blockIdx.x + 1) / gridDim.x);
  const int ns = s1 - s0;
  if (ns <= 0) return;
  // (a prefix sum over at most a few dozen counts: one wave, 64 at a time)
  if (threadIdx.x < 64) {
    int run = 0;
    for (int k0 = 0; k0 < ns; k0 += 64) {
      const int k = k0 + (int)threadIdx.x;
      const int c = k < ns ? A.cnt[s0 + k] : 0;
      int incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if ((int)threadIdx.x >= off) incl += v;
      }
      if (k < ns) pre[k] = run + incl - c;
      run += __shfl(incl, 63);
    }
    if (threadIdx.x == 0) {
      pre[ns] = run;
      next = 0;
    }
  }
  __syncthreads();
  const int total = pre[ns];
  const int lane = threadIdx.x & 63;
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  HitSearch S;
  S.i = 0;
  bool act = false, more = total > 0;
  for (;;) {
    const unsigned long long idle = __ballot(!act);
    const int nidle = __popcll(idle);
    if (more && (nidle >= MULTI_REFILL)) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&next, nidle);
      base = __shfl(base, 0);
      if (!act) {
        const int q = base + __popcll(idle & below);
        if (q < total) {
          // which segment: the last one whose start is <= q
          int lo = 0, hi = ns;            // pre[lo] <= q < pre[hi]
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= q) lo = mid; else hi = mid;
          }
          const int64_t seg = s0 + lo;
          const int64_t i = seg * MULTI_SEG + A.idx[seg * MULTI_SEG + (q - pre[lo])];
          act = multi_hit_begin<K>(P, in, A, g, brent, i, S);
        }
      }
      more = base + nidle < total;
      continue;
    }
    if (nidle == 64) {
      if (!more) break;
      continue;          // (never: with every lane idle the branch above refills)
    }
    if (act) {
      const bool on = brent ? multi_brent_step<K>(P, S) : multi_secant_step<K>(P, g, S);
      if (!on) {
        multi_put_hit(A, S.i, S.t2, S.x2, S.y2, S.z2, 0);
        act = false;
      }
    }
  }
}

// ---- state, reflection, the beam after the bounce ----------------------------------------
template <class K>
__global__ __launch_bounds__(REFLECT_MULTI_BLOCK, REFLECT_MULTI_PER_CU) void reflect_multi_finish(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_beam in, xrt_hip_beam out, MultiAux A) {
  __shared__ unsigned long long lds_u[REFLECT_MAX_WAVES];
  const GStat g = *A.g;
  const bool has_amp = in.Es_ri != nullptr;
  const bool param = surf_is_param<K>(P);
  const bool elevate = A.elev_out[0] != nullptr;
  unsigned long long kept = 0, hit = 0;
  for (int seg = blockIdx.x; seg < A.nseg; seg += gridDim.x) {
    // the rays that do not enter leave as they came: a streaming sweep over the segment
    for (int c0 = 0; c0 < MULTI_SEG; c0 += (int)blockDim.x) {
      const int64_t i = (int64_t)seg * MULTI_SEG + c0 + threadIdx.x;
      if (i >= in.n) break;
      const int st0 = in.state[i];
      if (entering(P, st0)) continue;
      copy_ray(out, in, i, st0, has_amp, false);
      A.nrefl_out[i] = A.nrefl_in ? A.nrefl_in[i] : 0;
      A.theta[i] = 0.;
      if (elevate) {
        const double el0[4] = {-1., -kMaxHalfSize, -kMaxHalfSize, -kMaxHalfSize};
        for (int k = 0; k < 4; ++k) A.elev_out[k][i] = A.elev_in[0] ? A.elev_in[k][i] : el0[k];
      }
      if (A.spr[0]) {   // reflect.py:1067-1069: copies of lb.x, y, z as they are
        A.spr[0][i] = in.x[i];
        A.spr[1][i] = in.y[i];
        A.spr[2][i] = in.z[i];
      }
    }
    // the entering ones: dense
    const int n = A.cnt[seg];
    const int32_t* list = A.idx + (int64_t)seg * MULTI_SEG;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
      const int64_t i = (int64_t)seg * MULTI_SEG + list[k];
      LocalRay raw;
      raw.x = in.x[i];
      raw.y = in.y[i];
      raw.z = in.z[i];
      raw.a = in.a[i];
      raw.b = in.b[i];
      raw.c = in.c[i];
      const int nr0 = A.nrefl_in ? A.nrefl_in[i] : 0;
      double el[4] = {-1., -kMaxHalfSize, -kMaxHalfSize, -kMaxHalfSize};   // reflect.py:214-218
      if (elevate && A.elev_in[0])
        for (int j = 0; j < 4; ++j) el[j] = A.elev_in[j][i];
      double vx, vy, vz;
      const LocalRay r = multi_local(P, raw, vx, vy, vz);
      if (P.is_multi && elevate) {
        // base.py:1284-1286, reflect.py:651-659: find_dz at the tangency point
        const double tg = A.tang[i];
        double ex, ey, ez;
        el[0] = multi_f<K, 0>(P, tg, r, ex, ey, ez);
        if (param) {
          double cx, cy, cz;
          ell_param_to_xyz(P, ex, ey, ez, cx, cy, cz);
          ex = cx;
          ey = cy;
          ez = cz;
        }
        el[1] = ex;
        el[2] = ey;
        el[3] = ez;
      }
      Hit h;
      h.t = A.ht[i];
      h.x = A.hx[i];
      h.y = A.hy[i];
      h.z = A.hz[i];
      h.lost = A.hlost[i];
      h.px = h.py = 0.;
      const double hs = h.x, hphi = h.y, hr = h.z;
      hit_done<K>(P, h);
      int st = rays_good<K>(P, h.x, h.y);
      if (h.lost) st = P.lost_num;
      RayIn q;
      q.path = in.path[i];
      q.E = in.E[i];
      double a = r.a, b = r.b, c = r.c, th = 0.;
      if (st == 1) {
        const Finished fin = finish_ray<K>(P, M, g, r, h, q, in, i, has_amp);
        a = fin.a;
        b = fin.b;
        c = fin.c;
        th = fin.theta;
        q = fin.lo;                 // (path + t, E)
        q.Jss = fin.vJss;           // lb is vlb: the matrix turned back is what stays
        q.Jpp = fin.vJpp;           // (reflect.py:1106-1110)
        q.Jsr = fin.vJsr;
        q.Jsi = fin.vJsi;
        q.Esr = fin.vEsr;
        q.Esi = fin.vEsi;
        q.Epr = fin.vEpr;
        q.Epi = fin.vEpi;
      } else {
        load_fields(in, i, has_amp, q);
      }
      // back to the virgin local frame (reflect.py:1115-1132), every entering ray
      double x = h.x + P.shift[0], y = h.y + P.shift[1], z = h.z + P.shift[2];
      rotate3(P.to_virgin, x, y, z);
      rotate3(P.to_virgin, a, b, c);
      if (st == 3) {                // reflect.py:225-228
        x = vx;
        y = vy;
        z = vz;
      }
      store_ray(out, i, x, y, z, a, b, c, q.path, q.E, q.Jss, q.Jpp, q.Jsr, q.Jsi, st, q.Esr,
                q.Esi, q.Epr, q.Epi, has_amp);
      const bool good = st == 1 || st == 2;
      A.nrefl_out[i] = nr0 + (good ? 1 : 0);
      A.theta[i] = th;
      if (elevate)
        for (int j = 0; j < 4; ++j) A.elev_out[j][i] = el[j];
      if (A.spr[0]) {
        A.spr[0][i] = hs;
        A.spr[1][i] = hphi;
        A.spr[2][i] = hr;
      }
      kept += good;
      hit += st == 1;
    }
  }
  auto faddu = [](unsigned long long u, unsigned long long v) { return u + v; };
  kept = block_reduce(kept, faddu, lds_u);
  hit = block_reduce(hit, faddu, lds_u);
  if (threadIdx.x == 0) {
    if (kept) (void)atomicAdd(&A.counts[0], kept);
    if (hit) (void)atomicAdd(&A.counts[1], hit);
  }
}

// gb of multiple_reflect (reflect.py:247-255): rays that were reflected at least once leave
// with state 1 in the global frame; the others are the incoming rays with the state they
// ended in.
__global__ __launch_bounds__(REFLECT_BLOCK) void multi_to_global_kernel(
    xrt_hip_pass P, xrt_hip_beam last, xrt_hip_beam orig, const int32_t* __restrict__ nrefl,
    xrt_hip_beam gb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= last.n) return;
  const bool has_amp = last.Es_ri != nullptr;
  const int st = last.state[i];
  if (nrefl[i] <= 0) {
    copy_ray(gb, orig, i, st, has_amp, false);
    return;
  }
  double x = last.x[i], y = last.y[i], z = last.z[i];
  double a = last.a[i], b = last.b[i], c = last.c[i];
  if (P.sin_az != 0.) {   // beamline.py:276-283: rotate_z(., ., cos, -sin)
    const double an = P.cos_az * a - (-P.sin_az) * b, bn = (-P.sin_az) * a + P.cos_az * b;
    a = an;
    b = bn;
    const double xn = P.cos_az * x - (-P.sin_az) * y, yn = (-P.sin_az) * x + P.cos_az * y;
    x = xn;
    y = yn;
  }
  x += P.center[0];
  y += P.center[1];
  z += P.center[2];
  const double2 js = reinterpret_cast<const double2*>(last.Jsp_ri)[i];
  double2 es = make_double2(0., 0.), ep = make_double2(0., 0.);
  if (has_amp) {
    es = reinterpret_cast<const double2*>(last.Es_ri)[i];
    ep = reinterpret_cast<const double2*>(last.Ep_ri)[i];
  }
  store_ray(gb, i, x, y, z, a, b, c, last.path[i], last.E[i], last.Jss[i], last.Jpp[i], js.x,
            js.y, 1, es.x, es.y, ep.x, ep.y, has_amp);
}

// every block of the launch has to be resident (grid barriers): as many as the occupancy of
// THIS instantiation allows, no more than the rays need
template <class K>
inline int launch_multi_dense_k(const MultiLaunch& L) {
  static int per_cu = 0;
  if (per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reflect_multi<K>, REFLECT_MULTI_BLOCK,
                                                     0) != hipSuccess || nb < 1)
      nb = 1;
    per_cu = nb > REFLECT_MULTI_PER_CU ? REFLECT_MULTI_PER_CU : nb;
  }
  int64_t blocks = (L.in->n + REFLECT_MULTI_BLOCK - 1) / REFLECT_MULTI_BLOCK;
  const int64_t cap = (int64_t)L.cus * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks > (int64_t)REFLECT_MAX_PART) blocks = REFLECT_MAX_PART;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(reflect_multi<K>, dim3((unsigned)blocks), dim3(REFLECT_MULTI_BLOCK), 0, L.st,
                     *L.P, *L.M, *L.in, *L.out, L.A);
  return (int)hipGetLastError();
}

// The statistics kernel synchronises its blocks with grid barriers: every block resident, as
// many as the occupancy of THIS instantiation allows. The other two size their grids by
// their own occupancy and walk the segments (no barriers: any number of blocks would do).
template <class K>
inline int launch_multi_sparse_k(const MultiLaunch& L) {
  static int occ[3] = {0, 0, 0};
  if (occ[0] == 0) {
    const void* fn[3] = {reinterpret_cast<const void*>(reflect_multi_stats<K>),
                         reinterpret_cast<const void*>(reflect_multi_solve<K>),
                         reinterpret_cast<const void*>(reflect_multi_finish<K>)};
    for (int k = 0; k < 3; ++k) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn[k], REFLECT_MULTI_BLOCK, 0) !=
              hipSuccess || nb < 1)
        nb = 1;
      occ[k] = nb > 8 ? 8 : nb;
    }
  }
  const int64_t nseg = L.A.nseg;
  auto grid_of = [&](int per_cu) {
    int64_t blocks = (int64_t)L.cus * per_cu;
    if (blocks > nseg) blocks = nseg;
    if (blocks < 1) blocks = 1;
    return blocks;
  };
  int64_t b0 = grid_of(occ[0] > REFLECT_MULTI_PER_CU ? REFLECT_MULTI_PER_CU : occ[0]);
  if (b0 > (int64_t)REFLECT_MAX_PART) b0 = REFLECT_MAX_PART;
  hipLaunchKernelGGL(reflect_multi_stats<K>, dim3((unsigned)b0), dim3(REFLECT_MULTI_BLOCK), 0, L.st,
                     *L.P, *L.M, *L.in, L.A);
  int64_t b1 = grid_of(occ[1]);
  const int64_t least = (nseg + MULTI_MAX_SEGS - 1) / MULTI_MAX_SEGS;
  if (b1 < least) b1 = least;
  hipLaunchKernelGGL(reflect_multi_solve<K>, dim3((unsigned)b1), dim3(REFLECT_MULTI_BLOCK), 0, L.st,
                     *L.P, *L.in, L.A);
  const int64_t b2 = grid_of(occ[2]);
  hipLaunchKernelGGL(reflect_multi_finish<K>, dim3((unsigned)b2), dim3(REFLECT_MULTI_BLOCK), 0,
                     L.st, *L.P, *L.M, *L.in, *L.out, L.A);
  return (int)hipGetLastError();
}

template <class K>
inline int launch_multi_opt_k(const MultiLaunch& L) {
  static int per_cu = 0;
  if (per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reflect_multi_opt<K>,
                                                     REFLECT_MULTI_BLOCK, 0) != hipSuccess ||
        nb < 1)
      nb = 1;
    per_cu = nb > REFLECT_MULTI_PER_CU ? REFLECT_MULTI_PER_CU : nb;
  }
  xrt_hip_pass head = *L.P;
  head.method_hint = nullptr;
  hipLaunchKernelGGL(multi_decide_opt<K>, dim3(1), dim3(REFLECT_BLOCK), 0, L.st, head, *L.M, *L.in,
                     L.A);
  int64_t blocks = (L.in->n + REFLECT_MULTI_BLOCK - 1) / REFLECT_MULTI_BLOCK;
  const int64_t cap = (int64_t)L.cus * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  MultiArgs R;
  memset(&R, 0, sizeof(R));
  R.P = *L.P;
  R.M = *L.M;
  R.in = *L.in;
  R.out = *L.out;
  R.A = L.A;
  hipLaunchKernelGGL(reflect_multi_opt<K>, dim3((unsigned)blocks), dim3(REFLECT_MULTI_BLOCK), 0,
                     L.st, R);
  return launch_multi_dense_k<K>(L);       // (A.gate set: returns at once unless contradicted)
}

// sparse: few rays still enter (exact, over an index); optimistic: a full bounce without its
// statistics phases, verified per ray; else the exact dense kernel (A.gate = 0)
template <class K>
inline int launch_multi_k(const MultiLaunch& L) {
  if (L.sparse) return launch_multi_sparse_k<K>(L);
  return L.A.gate ? launch_multi_opt_k<K>(L) : launch_multi_dense_k<K>(L);
}

}  // namespace xrt
