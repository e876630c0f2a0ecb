// Screen.expose (screens.py:226-302) of ONE ray held in registers, flat screens: what the
// stand-alone kernel (screen.hip) does per ray and what a ray pass does in its tail when the
// script hands the beam it makes straight on to a screen (reflect_impl.h: ScreenConsumer) --
// one code, the same bits either way.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/xrt_hip.h"
#include "fp64_math.h"

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

__device__ __forceinline__ void expose_flat_store(
    const xrt_hip_screen& S, const xrt_hip_beam& out, int64_t i, double px, double py, double pz,
    double ga, double gb, double gc, double path0, double E, double Jss, double Jpp, double Jsr,
    double Jsi, int st, double Esr, double Esi, double Epr, double Epi, bool has_amp) {
  const ImageRay r = expose_flat(S, px, py, pz, ga, gb, gc, st);
  store_image(out, i, r, path0, E, Jss, Jpp, Jsr, Jsi, Esr, Esi, Epr, Epi, has_amp);
}

}  // namespace xrt
