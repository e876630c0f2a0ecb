// Screen.expose (screens.py:226-302) of ONE ray held in registers, flat screens: what the
// stand-alone kernel (screen.hip) does per ray and what a ray pass does in its tail when the
// script hands the beam it makes straight on to a screen (reflect_impl.h: LateScreen) --
// one code, the same bits either way.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/xrt_hip.h"
#include "fp64_math.h"
#include "screen.h"

namespace xrt {

// streamed out once, never read again by the kernel: non-temporal (reflect_impl.h: store_ray)
template <class T>
__device__ __forceinline__ void put(T* p, int64_t i, T v) {
  __builtin_nontemporal_store(v, &p[i]);
}

// the image of one ray: position and direction in the screen's frame, the path to the plane
struct ImageRay {
  double x, z, a, b, c, path;   // (y = 0 on the plane; path = what the screen adds)
  int st;
};
// (px, py, pz), (ga, gb, gc): the ray in the global frame
__device__ __forceinline__ ImageRay expose_flat(const xrt_hip_screen& S, double px, double py,
                                                double pz, double ga, double gb, double gc,
                                                int st) {
  const double gx = px - S.center[0];
  const double gy = py - S.center[1];
  const double gz = pz - S.center[2];
  // sum(c*b for c, b in zip(basis, xyz)): ((0 + c0*x) + c1*y) + c2*z
  double x = (S.ex[0] * gx + S.ex[1] * gy) + S.ex[2] * gz;
  double y = (S.ey[0] * gx + S.ey[1] * gy) + S.ey[2] * gz;
  double z = (S.ez[0] * gx + S.ez[1] * gy) + S.ez[2] * gz;
  ImageRay r;
  r.a = (S.ex[0] * ga + S.ex[1] * gb) + S.ex[2] * gc;
  r.b = (S.ey[0] * ga + S.ey[1] * gb) + S.ey[2] * gc;
  r.c = (S.ez[0] * ga + S.ez[1] * gb) + S.ez[2] * gc;
  double path = -y / r.b;
  bool bad = isnan(path) || isinf(path);
  if (S.only_positive_path) bad = bad || (path < 0.);
  if (bad) {
    path = 0.;
    st = S.lost_num;
  }
  x = x + r.a * path;
  z = z + r.c * path;
  if (S.compress_x != 0.) x *= S.compress_x;
  if (S.compress_z != 0.) z *= S.compress_z;
  r.x = x;
  r.z = z;
  r.path = path;
  r.st = st;
  return r;
}

// ... and its record in the image beam (the rest of the ray's record as it is)
__device__ __forceinline__ void store_image(const xrt_hip_beam& out, int64_t i, const ImageRay& r,
                                            double path0, double E, double Jss, double Jpp,
                                            double Jsr, double Jsi, double Esr, double Esi,
                                            double Epr, double Epi, bool has_amp) {
  put(out.x, i, r.x);
  put(out.y, i, 0.);
  put(out.z, i, r.z);
  put(out.a, i, r.a);
  put(out.b, i, r.b);
  put(out.c, i, r.c);
  put(out.path, i, path0 + r.path);
  put(out.E, i, E);
  put(out.Jss, i, Jss);
  put(out.Jpp, i, Jpp);
  reinterpret_cast<double2*>(out.Jsp_ri)[i] = make_double2(Jsr, Jsi);
  out.state[i] = r.st;
  if (has_amp) {
    // exp(1e7j * (E/CHBAR) * path), screens.py:271-274
    const double kCH = 6.626069573e-27 * 2.99792458e10 / 1.602176565e-12 * 1e8;
    const double kCHBAR = kCH / 6.283185307179586476925286766559;
    const double ph = (1e7 * (E / kCHBAR)) * r.path;
    double s, co;
    sincos_phase(ph, s, co);
    reinterpret_cast<double2*>(out.Es_ri)[i] =
        make_double2(Esr * co - Esi * s, Esr * s + Esi * co);
    reinterpret_cast<double2*>(out.Ep_ri)[i] =
        make_double2(Epr * co - Epi * s, Epr * s + Epi * co);
  }
}

// RectangularAperture / RoundAperture / DoubleSlit / beam stops (apertures.py:334-413, 770-846,
// 931-1021) on ONE ray that is alive, given in the global frame: the ray in the aperture's
// frame on its plane, the path there, and whether the blades stop it. (No polygons: they need
// their vertex array and also relabel dead rays -- screen.hip's kernel.) One code for the
// stand-alone kernel and for an aperture in the tail of a ray pass (reflect_impl.h).
struct ApertureRay {
  double x, z, a, b, c, dpath;
  bool bad;
};
__device__ __forceinline__ ApertureRay aperture_ray(const xrt_hip_aperture& A, double px,
                                                    double py, double pz, double ga, double gb,
                                                    double gc) {
  ApertureRay r;
  const double gx = px - A.center[0], gy = py - A.center[1], gz = pz - A.center[2];
  double x = (A.ex[0] * gx + A.ex[1] * gy) + A.ex[2] * gz;
  const double y = (A.ey[0] * gx + A.ey[1] * gy) + A.ey[2] * gz;
  double z = (A.ez[0] * gx + A.ez[1] * gy) + A.ez[2] * gz;
  r.a = (A.ex[0] * ga + A.ex[1] * gb) + A.ex[2] * gc;
  r.b = (A.ey[0] * ga + A.ey[1] * gb) + A.ey[2] * gc;
  r.c = (A.ez[0] * ga + A.ez[1] * gb) + A.ez[2] * gc;
  r.dpath = -y / r.b;
  x = x + r.a * r.dpath;
  z = z + r.c * r.dpath;
  bool bad = false;
  if (A.round) bad = sqrt(x * x + z * z) > A.radius;
  if (A.blade_mask & 1) bad = bad || (x < A.blade[0]);
  if (A.blade_mask & 2) bad = bad || (x > A.blade[1]);
  if (A.blade_mask & 4) bad = bad || (z < A.blade[2]);
  if (A.blade_mask & 8) bad = bad || (z > A.blade[3]);
  if (A.has_shade) bad = bad || (z > A.shade[0] && z < A.shade[1]);
  r.x = x;
  r.z = z;
  r.bad = bad;
  return r;
}

// up to two apertures right behind an element, in the order the beam meets them: the state of
// the outgoing (global) record after them -- what aperture.propagate(gb) leaves in gb.state
__device__ __forceinline__ int apertures_mark(const TailApertures& T, double x, double y, double z,
                                              double a, double b, double c, int st) {
#pragma unroll
  for (int k = 0; k < XRT_TAIL_APERTURES; ++k) {
    if (k < T.n && st > 0) {
      bool bad = aperture_ray(T.a[k], x, y, z, a, b, c).bad;
      if (T.a[k].is_beam_stop) bad = !bad;
      if (bad) st = T.a[k].lost_num;
    }
  }
  return st;
}

__device__ __forceinline__ void expose_flat_store(
    const xrt_hip_screen& S, const xrt_hip_beam& out, int64_t i, double px, double py, double pz,
    double ga, double gb, double gc, double path0, double E, double Jss, double Jpp, double Jsr,
    double Jsi, int st, double Esr, double Esi, double Epr, double Epi, bool has_amp) {
  const ImageRay r = expose_flat(S, px, py, pz, ga, gb, gc, st);
  store_image(out, i, r, path0, E, Jss, Jpp, Jsr, Jsi, Esr, Esi, Epr, Epi, has_amp);
}

}  // namespace xrt
