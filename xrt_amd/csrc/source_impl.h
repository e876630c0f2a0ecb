// GeometricSource.shine (sources/geoms.py:420-535) of ONE ray, in registers: what the stand-alone
// generator kernel (source.hip) stores and what a ray pass starts from when the script hands the
// source's beam straight to an element (reflect_impl.h: reflect_fused_gen_scr) -- one code, the
// same rays either way. Counter-based Philox4x32-10 addressed by (ray, slot, call): a ray can be
// made again at any time (oracle/geosource_np.py restates the stream layout).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/xrt_hip.h"
#include "fp64_math.h"

namespace xrt {
namespace gen {



constexpr double kPI2 = 6.283185307179586476925286766559;
constexpr int SLOT_PHASE = 0, SLOT_Y = 1, SLOT_XZ = 2, SLOT_AC = 4, SLOT_E = 6;

struct U2 {
  double a, b;
};

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011): multipliers 0xD2511F53 / 0xCD9E8D57, Weyl
// key increments 0x9E3779B9 / 0xBB67AE85. counter = (ray lo, ray hi, slot, call).
__device__ __forceinline__ U2 philox_uniforms(uint64_t ray, uint32_t slot, uint32_t call,
                                              uint64_t seed) {
  uint32_t c0 = (uint32_t)ray, c1 = (uint32_t)(ray >> 32), c2 = slot, c3 = call;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // (one 32 x 32 -> 64 multiply each: v_mad_u64_u32 instead of a mul_hi and a mul_lo)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
    const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    c0 = h1 ^ c1 ^ k0;
    c1 = l1;
    c2 = h0 ^ c3 ^ k1;
    c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  // 53 bits each: the high 27 bits of one word, the high 26 of the next (numpy's recipe)
  U2 u;
  u.a = ((double)(c0 >> 5) * 67108864. + (double)(c1 >> 6)) * 0x1p-53;
  u.b = ((double)(c2 >> 5) * 67108864. + (double)(c3 >> 6)) * 0x1p-53;
  return u;
}

// sin and cos of 2 pi u, u in [0, 1): 4u quarter turns = a whole number q and a rest |r| <= 1/2
// (both exact), then the polynomials of fp64_math.h -- the library's sincos carries a
// large-argument path along that this angle never takes. Within an ulp or two of
// np.sin / np.cos (2 pi u), which round the product 2 pi u first.
__device__ __forceinline__ void sincos_turn(double u, double& sn, double& cs) {
  const double t = 4. * u;
  const double q = __builtin_rint(t);
  sincos_quarter_turns(t - q, (unsigned)(int)q, sn, cs);
}

__device__ __forceinline__ void box_muller(const U2 u, double& g1, double& g2) {
  const double radius = sqrt(-2. * log(1. - u.a));
  double sn, cs;
  sincos_turn(u.b, sn, cs);
  g1 = radius * cs;
  g2 = radius * sn;
}

// the call number of this launch: the record's, plus what the caller keeps in device memory (a
// launch captured in a HIP graph is replayed with the same record: the cell counts the replays)
__device__ __forceinline__ uint32_t call_of(const xrt_hip_geosource& G) {
  return G.call + (G.call_dev ? *G.call_dev : 0u);
}

struct Ray {
  double Jss, Jpp, Jre, Jim, Esr, Esi, Epr, Epi;
};

// exp(-v^2 / sigma^2 / 2) / PI2**0.5 / sigma * 2 * cut, geoms.py:384-385
__device__ __forceinline__ void weigh_down(Ray& r, double v, double sigma, double cut,
                                           bool amp) {
  const double w = exp(-(v * v) / (sigma * sigma) / 2.) / sqrt(kPI2) / sigma * 2. * cut;
  r.Jss *= w;
  r.Jpp *= w;
  r.Jre *= w;
  r.Jim *= w;
  if (amp) {
    const double q = sqrt(w);
    r.Esr *= q;
    r.Esi *= q;
    r.Epr *= q;
    r.Epi *= q;
  }
}

__device__ __forceinline__ double one_number(const xrt_hip_geosource& G, uint32_t call, int k,
                                             uint64_t i, uint32_t slot, Ray& r, bool amp) {
  const int law = G.law[k];
  if (law == XRT_HIP_LAW_NONE) return 0.;
  const U2 u = philox_uniforms(i, slot, call, G.seed);
  if (law == XRT_HIP_LAW_NORMAL) {
    double g1, g2;
    box_muller(u, g1, g2);
    return g1 * G.p0[k];
  }
  if (law == XRT_HIP_LAW_FLAT) return G.p0[k] + (G.p1[k] - G.p0[k]) * u.a;
  const double cut = G.p1[k];
  const double v = -cut + (cut - (-cut)) * u.a;
  weigh_down(r, v, G.p0[k], cut, amp);
  return v;
}

__device__ __forceinline__ void one_pair(const xrt_hip_geosource& G, uint32_t call, int k,
                                         bool annulus, const double* ann, uint64_t i,
                                         uint32_t slot, Ray& r, bool amp, double& first,
                                         double& second) {
  if (annulus) {          // _set_annulus, geoms.py:409-418
    const U2 u = philox_uniforms(i, slot, call, G.seed);
    double radius = ann[1];
    if (ann[1] > ann[0]) {
      const double density = 2. / (ann[1] * ann[1] - ann[0] * ann[0]);
      radius = sqrt(2. * u.a / density + ann[0] * ann[0]);
    }
    const double phi = ann[2] + (ann[3] - ann[2]) * u.b;
    double sn, cs;
    sincos(phi, &sn, &cs);
    first = radius * cs;
    second = radius * sn;
    return;
  }
  if (G.law[k] == XRT_HIP_LAW_NORMAL && G.law[k + 1] == XRT_HIP_LAW_NORMAL) {
    double g1, g2;
    box_muller(philox_uniforms(i, slot, call, G.seed), g1, g2);
    first = g1 * G.p0[k];
    second = g2 * G.p0[k + 1];
    return;
  }
  first = one_number(G, call, k, i, slot, r, amp);
  second = one_number(G, call, k + 1, i, slot + 1, r, amp);
}

__device__ __forceinline__ void turn(const xrt_hip_rotation& R, double& x, double& y,
                                     double& z) {   // _rotate.py:5-57
  for (int s = 0; s < R.n; ++s) {
    const double c = R.cosa[s], sn = R.sina[s];
    if (R.axis[s] == 0) {
      const double u = y * c - z * sn, v = y * sn + z * c;
      y = u;
      z = v;
    } else if (R.axis[s] == 1) {
      const double u = x * c - z * (-sn), v = x * (-sn) + z * c;
      x = u;
      z = v;
    } else {
      const double u = x * c - y * sn, v = x * sn + y * c;
      x = u;
      y = v;
    }
  }
}


// one ray of the source, complete
struct GenRay {
  double x, y, z, a, b, c, E;
  Ray r;
};

__device__ __forceinline__ GenRay make_ray(const xrt_hip_geosource& G, uint32_t call, int64_t i,
                                           bool amp) {
  GenRay o;
  Ray r{G.Jss, G.Jpp, G.Jsp[0], G.Jsp[1], G.Es[0], G.Es[1], G.Ep[0], G.Ep[1]};
  if (amp && G.random_ep) {     // make_polarization: Ep = uniform * 2**-0.5, geoms.py:136-137
    r.Epr = philox_uniforms((uint64_t)i, SLOT_PHASE, call, G.seed).a * 0.70710678118654757;
    r.Epi = 0.;
  }
  double x, y, z, a, c;
  y = one_number(G, call, 0, (uint64_t)i, SLOT_Y, r, amp);
  one_pair(G, call, 1, G.annulus_xz != 0, G.ann_xz, (uint64_t)i, SLOT_XZ, r, amp, x, z);
  one_pair(G, call, 3, G.annulus_ac != 0, G.ann_ac, (uint64_t)i, SLOT_AC, r, amp, a, c);
  const double ac = a * a + c * c;
  double b;
  if (G.slopes) {               // geoms.py:499-503
    b = sqrt(ac + 1.);
    a = a / b;
    c = c / b;
    b = 1.0 / b;
  } else {
    b = sqrt(1. - ac);
  }
  double E = G.e_p0;
  if (G.e_law) {
    const U2 u = philox_uniforms(G.filament ? 0ull : (uint64_t)i, SLOT_E, call, G.seed);
    if (G.e_law == 1) {
      double g1, g2;
      box_muller(u, g1, g2);
      E = G.e_p0 + G.e_p1 * g1;
    } else if (G.e_law == 2) {
      E = G.e_p0 + (G.e_p1 - G.e_p0) * u.a;
    } else {
      int k = 0;
      while (k < G.n_lines - 1 && G.e_cdf[k] <= u.a) ++k;
      E = G.e_lines[k];
    }
  }
  turn(G.rot, x, y, z);
  turn(G.rot, a, b, c);
  if (G.to_global) {            // virgin_local_to_global, beamline.py:266-287
    if (G.sin_az != 0.) {
      const double s = -G.sin_az;
      double u = a * G.cos_az - b * s, v = a * s + b * G.cos_az;
      a = u;
      b = v;
      u = x * G.cos_az - y * s;
      v = x * s + y * G.cos_az;
      x = u;
      y = v;
    }
    x += G.center[0];
    y += G.center[1];
    z += G.center[2];
  }
  o.x = x;
  o.y = y;
  o.z = z;
  o.a = a;
  o.b = b;
  o.c = c;
  o.E = E;
  o.r = r;
  return o;
}

__device__ __forceinline__ void store_gen_ray(const xrt_hip_beam& out, int64_t i, const GenRay& g,
                                              int state, bool amp) {
  __builtin_nontemporal_store(g.x, &out.x[i]);
  __builtin_nontemporal_store(g.y, &out.y[i]);
  __builtin_nontemporal_store(g.z, &out.z[i]);
  __builtin_nontemporal_store(g.a, &out.a[i]);
  __builtin_nontemporal_store(g.b, &out.b[i]);
  __builtin_nontemporal_store(g.c, &out.c[i]);
  __builtin_nontemporal_store(0., &out.path[i]);
  __builtin_nontemporal_store(g.E, &out.E[i]);
  __builtin_nontemporal_store(g.r.Jss, &out.Jss[i]);
  __builtin_nontemporal_store(g.r.Jpp, &out.Jpp[i]);
  reinterpret_cast<double2*>(out.Jsp_ri)[i] = make_double2(g.r.Jre, g.r.Jim);
  out.state[i] = state;
  if (amp) {
    reinterpret_cast<double2*>(out.Es_ri)[i] = make_double2(g.r.Esr, g.r.Esi);
    reinterpret_cast<double2*>(out.Ep_ri)[i] = make_double2(g.r.Epr, g.r.Epi);
  }
}

}  // namespace gen
}  // namespace xrt
