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
template <class K, int DERIV>
__device__ __forceinline__ Hit solve_between(const xrt_hip_pass& P, const LocalRay& r, double t1,
                                             double t2, double tMinG, double tMaxG,
                                             bool use_brent) {
  Hit h;
  const Ends e = bracket_ends<K, DERIV>(P, r, t1, t2);
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
      if (t2 < tMinG) t2 = tMinG;
      if (t2 > tMaxG) t2 = tMaxG;
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

// The launch is preceded by reflect_init(g, 1) (decisions reset, barrier counter zeroed).
template <class K>
__global__ __launch_bounds__(REFLECT_MULTI_BLOCK, REFLECT_MULTI_PER_CU) void reflect_multi(
    xrt_hip_pass P, xrt_hip_material M, xrt_hip_beam in, xrt_hip_beam out, MultiAux A) {
  GStat* g = A.g;
  unsigned phase = 0;
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
    A.diag[3] = (double)gl.n_enter;
    A.diag[7] = gl.t1min;
    A.diag[8] = gl.t2max;
  }
  multi_finish<K>(P, M, in, out, A, gl);
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
inline int launch_multi_k(const MultiLaunch& L) {
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

}  // namespace xrt
