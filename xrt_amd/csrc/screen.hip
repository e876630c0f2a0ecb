// Screen.expose (xrt/backends/raycing/screens.py:226-302) for device-resident
// beams: project every ray into the screen's (x, y, z) basis
// (beamline.py:253-264), propagate it to the plane y = 0, flag rays that never
// reach it. Pure streaming kernel: 100 B read + 100 B written per ray (132+132
// with field amplitudes), coalesced SoA accesses.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/xrt_hip.h"
#include "fp64_math.h"
#include "screen.h"
#include "screen_impl.h"

namespace xrt {

__global__ __launch_bounds__(256) void screen_expose_kernel(xrt_hip_screen S, xrt_hip_beam in,
                                                           xrt_hip_beam out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= in.n) return;
  const double gx = in.x[i] - S.center[0];
  const double gy = in.y[i] - S.center[1];
  const double gz = in.z[i] - S.center[2];
  const double ga = in.a[i], gb = in.b[i], gc = in.c[i];
  if (S.radius != 0.) {   // HemisphericScreen.expose, screens.py:526-555
    const double half_b = (ga * gx + gb * gy) + gc * gz;
    const double cq = ((gx * gx + gy * gy) + gz * gz) - S.radius * S.radius;
    double path = -half_b + sqrt(half_b * half_b - cq);
    int st = in.state[i];
    bool bad = isnan(path) || isinf(path);
    if (S.only_positive_path) bad = bad || (path < 0.);
    if (bad) {
      path = 0.;
      st = S.lost_num;
    }
    const double E = in.E[i];
    const double rx = in.x[i] + ga * path - S.center[0];
    const double ry = in.y[i] + gb * path - S.center[1];
    const double rz = in.z[i] + gc * path - S.center[2];
    const double lx = (rx * S.ex[0] + ry * S.ex[1]) + rz * S.ex[2];
    const double ly = (rx * S.ey[0] + ry * S.ey[1]) + rz * S.ey[2];
    const double lz = (rx * S.ez[0] + ry * S.ez[1]) + rz * S.ez[2];
    put(out.x, i, lx);
    put(out.y, i, ly);
    put(out.z, i, lz);
    put(out.a, i, ga);
    put(out.b, i, gb);
    put(out.c, i, gc);
    put(out.path, i, in.path[i] + path);
    put(out.E, i, E);
    put(out.Jss, i, in.Jss[i]);
    put(out.Jpp, i, in.Jpp[i]);
    reinterpret_cast<double2*>(out.Jsp_ri)[i] = reinterpret_cast<const double2*>(in.Jsp_ri)[i];
    out.state[i] = st;
    if (S.out_theta) S.out_theta[i] = asin(lz / S.radius) - S.theta_offset;
    if (S.out_phi) S.out_phi[i] = atan2(ly, lx) - S.phi_offset;
    if (in.Es_ri) {
      const double kCH = 6.626069573e-27 * 2.99792458e10 / 1.602176565e-12 * 1e8;
      const double kCHBAR = kCH / 6.283185307179586476925286766559;
      double s, co;
      sincos_phase((1e7 * (E / kCHBAR)) * path, s, co);
      const double2 es = reinterpret_cast<const double2*>(in.Es_ri)[i];
      const double2 ep = reinterpret_cast<const double2*>(in.Ep_ri)[i];
      reinterpret_cast<double2*>(out.Es_ri)[i] =
          make_double2(es.x * co - es.y * s, es.x * s + es.y * co);
      reinterpret_cast<double2*>(out.Ep_ri)[i] =
          make_double2(ep.x * co - ep.y * s, ep.x * s + ep.y * co);
    }
    return;
  }
  const bool has_amp = in.Es_ri != nullptr;
  const double2 js = reinterpret_cast<const double2*>(in.Jsp_ri)[i];
  double2 es = make_double2(0., 0.), ep = make_double2(0., 0.);
  if (has_amp) {
    es = reinterpret_cast<const double2*>(in.Es_ri)[i];
    ep = reinterpret_cast<const double2*>(in.Ep_ri)[i];
  }
  expose_flat_store(S, out, i, in.x[i], in.y[i], in.z[i], ga, gb, gc, in.path[i], in.E[i],
                    in.Jss[i], in.Jpp[i], js.x, js.y, in.state[i], es.x, es.y, ep.x, ep.y,
                    has_amp);
}

hipError_t screen_expose_launch(const xrt_hip_screen& S, const xrt_hip_beam& in,
                                const xrt_hip_beam& out, hipStream_t st) {
  if (in.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(screen_expose_kernel, dim3((unsigned)((in.n + 255) / 256)), dim3(256), 0,
                     st, S, in, out);
  return hipGetLastError();
}


// matplotlib's Path.contains_points for a closed polygon (src/_path.h, point_in_path_impl,
// radius 0), as in reflect.hip's rays_good: an edge whose ends lie on different sides of the
// horizontal through the point toggles `inside` when the crossing is to its right.
__device__ __forceinline__ bool inside_polygon(const double* v, int n, double x, double y) {
  if (n < 3 || !(isfinite(x) && isfinite(y))) return false;
  // non-finite vertices split the outline into sub-polygons, each closed on itself; inside
  // any of them = inside (matplotlib: PathNanRemover + inside_flag |= subpath_flag) -- the
  // cells of a GridAperture
  bool any = false;
  int k = 0;
  while (k < n) {
    if (!(isfinite(v[2 * k]) && isfinite(v[2 * k + 1]))) {
      ++k;
      continue;
    }
    const int start = k;
    while (k < n && isfinite(v[2 * k]) && isfinite(v[2 * k + 1])) ++k;
    bool inside = false;
    double x0 = v[2 * (k - 1)], y0 = v[2 * (k - 1) + 1];
    for (int j = start; j < k; ++j) {
      const double x1 = v[2 * j], y1 = v[2 * j + 1];
      const bool up0 = y0 >= y, up1 = y1 >= y;
      if (up0 != up1 && (((y1 - y) * (x0 - x1) >= (x1 - x) * (y0 - y1)) == up1))
        inside = !inside;
      x0 = x1;
      y0 = y1;
    }
    any = any || inside;
  }
  return any;
}

// RectangularAperture.propagate, apertures.py:334-413. Same streaming shape as
// screen_expose; additionally writes the new state back into the incoming beam.
// FULL = false: only the states are wanted (out_local NULL: nobody looks at the beam in the
// aperture's frame -- the caller can make it later from the same arrays and a copy of the states
// as they were): 52 B read and at most 4 written per ray instead of 100 + 100.
template <bool FULL>
__global__ __launch_bounds__(256) void aperture_propagate_kernel(xrt_hip_aperture A,
                                                                xrt_hip_beam in,
                                                                xrt_hip_beam lo,
                                                                xrt_hip_beam glo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= in.n) return;
  const bool has_amp = FULL && in.Es_ri != nullptr;
  const bool want_glo = FULL && glo.x != nullptr;
  int st = in.state[i];
  double x = in.x[i], y = in.y[i], z = in.z[i];
  double a = in.a[i], b = in.b[i], c = in.c[i];
  double path = FULL ? in.path[i] : 0.;
  const double E = FULL ? in.E[i] : 0.;
  double2 es = make_double2(0., 0.), ep = make_double2(0., 0.);
  if (has_amp) {
    es = reinterpret_cast<const double2*>(in.Es_ri)[i];
    ep = reinterpret_cast<const double2*>(in.Ep_ri)[i];
  }
  // (own_marks: the states are those a states-only call of this aperture left behind -- a ray it
  // stopped then was alive when it arrived, and is stopped again by the same test)
  const bool good = st > 0 || (A.own_marks && st == A.lost_num);
  if (good) {
    const ApertureRay q = aperture_ray(A, x, y, z, a, b, c);
    x = q.x;
    z = q.z;
    a = q.a;
    b = q.b;
    c = q.c;
    const double dpath = q.dpath;
    path = path + dpath;
    bool bad = q.bad;
    if (A.poly_n > 0) bad = !inside_polygon(A.poly_xz, A.poly_n, x, z);
    if (A.is_beam_stop) bad = !bad;
    if (bad) {
      st = A.lost_num;
      in.state[i] = st;   // the reference marks the incoming beam as well
    }
    y = 0.;
    if (has_amp) {  // exp(1e7j (E/CHBAR) path), apertures.py:379-382
      const double kCH = 6.626069573e-27 * 2.99792458e10 / 1.602176565e-12 * 1e8;
      const double kCHBAR = kCH / 6.283185307179586476925286766559;
      const double ph = (1e7 * (E / kCHBAR)) * dpath;
      double s, co;
      sincos_phase(ph, s, co);
      es = make_double2(es.x * co - es.y * s, es.x * s + es.y * co);
      ep = make_double2(ep.x * co - ep.y * s, ep.x * s + ep.y * co);
    }
  }
  if (!good && A.poly_n > 0 && !inside_polygon(A.poly_xz, A.poly_n, x, z))
    in.state[i] = A.lost_num;   // apertures.py:1198-1203: the incoming beam only
  if (!FULL) return;
  const double path_in = in.path[i];
  const double Jss = in.Jss[i], Jpp = in.Jpp[i];
  const double2 js = reinterpret_cast<const double2*>(in.Jsp_ri)[i];
  lo.x[i] = x;
  lo.y[i] = y;
  lo.z[i] = z;
  lo.a[i] = a;
  lo.b[i] = b;
  lo.c[i] = c;
  lo.path[i] = path;
  lo.E[i] = E;
  lo.Jss[i] = Jss;
  lo.Jpp[i] = Jpp;
  reinterpret_cast<double2*>(lo.Jsp_ri)[i] = js;
  lo.state[i] = st;
  if (has_amp) {
    reinterpret_cast<double2*>(lo.Es_ri)[i] = es;
    reinterpret_cast<double2*>(lo.Ep_ri)[i] = ep;
  }
  if (want_glo) {
    if (good) {  // virgin_local_to_global(bl, glo, center, good)
      if (A.sin_az != 0.) {
        const double an = A.cos_az * a - (-A.sin_az) * b, bn = (-A.sin_az) * a + A.cos_az * b;
        a = an;
        b = bn;
        const double xn = A.cos_az * x - (-A.sin_az) * y, yn = (-A.sin_az) * x + A.cos_az * y;
        x = xn;
        y = yn;
      }
      x += A.center[0];
      y += A.center[1];
      z += A.center[2];
      if (A.glo_adds_path) path = path + path_in;   // DoubleSlit, apertures.py:1013
    }
    put(glo.x, i, x);
    put(glo.y, i, y);
    put(glo.z, i, z);
    put(glo.a, i, a);
    put(glo.b, i, b);
    put(glo.c, i, c);
    put(glo.path, i, path);
    put(glo.E, i, E);
    put(glo.Jss, i, Jss);
    put(glo.Jpp, i, Jpp);
    reinterpret_cast<double2*>(glo.Jsp_ri)[i] = js;
    glo.state[i] = st;
    if (has_amp) {
      reinterpret_cast<double2*>(glo.Es_ri)[i] = es;
      reinterpret_cast<double2*>(glo.Ep_ri)[i] = ep;
    }
  }
}

hipError_t aperture_propagate_launch(const xrt_hip_aperture& A, const xrt_hip_beam& in,
                                     const xrt_hip_beam& lo, const xrt_hip_beam& glo,
                                     hipStream_t st) {
  if (in.n <= 0) return hipSuccess;
  if (!lo.x) {
    hipLaunchKernelGGL(aperture_propagate_kernel<false>, dim3((unsigned)((in.n + 255) / 256)),
                       dim3(256), 0, st, A, in, lo, glo);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(aperture_propagate_kernel<true>, dim3((unsigned)((in.n + 255) / 256)), dim3(256),
                     0, st, A, in, lo, glo);
  return hipGetLastError();
}

// Screen.expose of a resident beam and, right behind it, aperture.propagate of the SAME beam
// (a front-end monitor and the mask after it: screens.py:226-302, apertures.py:334-413) as ONE
// pass over the rays: the image from the states as they are, then the aperture's marks in the
// incoming beam -- 100 B read, 100 B + <= 4 B written per ray instead of 152 read in two
// launches. Flat screens, apertures without an outline of vertices; the arithmetic is that of
// screen_expose_kernel and aperture_propagate_kernel<false> (expose_flat_store, aperture_ray).
__global__ __launch_bounds__(256) void screen_expose_mark_kernel(xrt_hip_screen S,
                                                                xrt_hip_aperture A,
                                                                xrt_hip_beam in,
                                                                xrt_hip_beam out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= in.n) return;
  const double x = in.x[i], y = in.y[i], z = in.z[i];
  const double a = in.a[i], b = in.b[i], c = in.c[i];
  const int st = in.state[i];
  const bool has_amp = in.Es_ri != nullptr;
  const double2 js = reinterpret_cast<const double2*>(in.Jsp_ri)[i];
  double2 es = make_double2(0., 0.), ep = make_double2(0., 0.);
  if (has_amp) {
    es = reinterpret_cast<const double2*>(in.Es_ri)[i];
    ep = reinterpret_cast<const double2*>(in.Ep_ri)[i];
  }
  expose_flat_store(S, out, i, x, y, z, a, b, c, in.path[i], in.E[i], in.Jss[i], in.Jpp[i], js.x,
                    js.y, st, es.x, es.y, ep.x, ep.y, has_amp);
  if (st > 0 || (A.own_marks && st == A.lost_num)) {
    bool bad = aperture_ray(A, x, y, z, a, b, c).bad;
    if (A.is_beam_stop) bad = !bad;
    if (bad) in.state[i] = A.lost_num;
  }
}

hipError_t screen_expose_mark_launch(const xrt_hip_screen& S, const xrt_hip_aperture& A,
                                     const xrt_hip_beam& in, const xrt_hip_beam& out,
                                     hipStream_t st) {
  if (in.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(screen_expose_mark_kernel, dim3((unsigned)((in.n + 255) / 256)), dim3(256), 0,
                     st, S, A, in, out);
  return hipGetLastError();
}


}  // namespace xrt
