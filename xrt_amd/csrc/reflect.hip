// Launch logic of the reflect passes and the utility kernels of the host layer. The device
// code is in reflect_impl.h; the big kernel templates are instantiated in their own
// translation units (reflect_tu.h).
#define XRT_REFLECT_MAIN_TU
#include <string.h>

#include <mutex>

#include "reflect_impl.h"
#include "reflect_tu.h"
#include "hist.h"
#include "screen.h"
#include "source.h"

namespace xrt {

// ---- BarrierSerial (reflect.h) ---------------------------------------------------------------
namespace {
constexpr int kMaxDev = 16, kRing = 32;
struct BarrierChain {
  bool seen = false, multi = false;
  hipStream_t only = nullptr;
  hipEvent_t ring[kRing] = {};
  hipEvent_t last = nullptr;
  unsigned head = 0;
};
std::mutex g_barrier_mu;
BarrierChain g_barrier[kMaxDev];
}  // namespace

BarrierSerial::BarrierSerial(hipStream_t st) : st_(st), dev_(0), chain_(false) {
  g_barrier_mu.lock();
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) (void)hipGetLastError();
  if (cap != hipStreamCaptureStatusNone) return;       // (recorded into a graph: one worker)
  if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= kMaxDev) dev_ = 0;
  BarrierChain& C = g_barrier[dev_];
  if (!C.seen) {
    C.seen = true;
    C.only = st;
  } else if (!C.multi && st != C.only) {
    // a second stream: whatever the first one still has in flight ends before this launch,
    // and from now on the launches are chained by events
    (void)hipDeviceSynchronize();
    C.multi = true;
  }
  chain_ = C.multi;
  if (chain_ && C.last) (void)hipStreamWaitEvent(st, C.last, 0);
}

BarrierSerial::~BarrierSerial() {
  if (chain_) {
    BarrierChain& C = g_barrier[dev_];
    hipEvent_t& e = C.ring[C.head++ % kRing];
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
    if (e && hipEventRecord(e, st_) == hipSuccess) C.last = e;
  }
  g_barrier_mu.unlock();
}


// ---------------------------------------------------------------------------
// The surface functions of an element, for the host classes: OE.local_z / local_n /
// local_r / xyz_to_param / param_to_xyz evaluate HERE, with the very code the ray
// kernels use, instead of being written a second time in numpy.
//   what 0: (u, v) = (x, y) -> z of the surface above it (on a parametric surface: the z
//           of the point with these x, y; reflect.py:336-341)
//        1: (u, v) = (x, y), or (s, phi) on a parametric surface -> normals n[0..5]
//        2: (u, v) = (s, phi) -> r           3: (u, v, w) = (x, y, z) -> (s, phi, r)
//        4: (u, v, w) = (s, phi, r) -> (x, y, z)
//        5: (u, v) = (x, y) -> the state rays_good gives a hit there
// Output k of point i at o[k * n + i].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(REFLECT_BLOCK) void surface_eval_kernel(
    xrt_hip_pass P, int what, int64_t n, const double* __restrict__ u,
    const double* __restrict__ v, const double* __restrict__ w, double* __restrict__ o) {
  using K = GenericAll;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p = u[i], q = v[i], t = w ? w[i] : 0.;
  const bool param = P.surf_kind == XRT_HIP_SURF_ELLIPSE_PARAM;
  if (what == 0) {
    if (param) {
      double sp, phi, rr, x, y, z;
      ell_xyz_to_param(P, p, q, 0., sp, phi, rr);
      ell_param_to_xyz(P, sp, phi, ell_local_r(P, sp, phi), x, y, z);
      o[i] = z;
    } else {
      o[i] = surf_z<K>(P, p, q);
    }
  } else if (what == 1) {
    double nn[6];
    surface_normal<K>(P, p, q, p, q, nn);
    for (int k = 0; k < 6; ++k) o[k * n + i] = nn[k];
  } else if (what == 5) {
    o[i] = (double)rays_good<K>(P, p, q);
  } else if (what == 2) {
    o[i] = param ? ell_local_r(P, p, q) : 0.;
  } else if (what == 3) {
    double a = p, b = q, c = t;
    if (param) ell_xyz_to_param(P, p, q, t, a, b, c);
    o[i] = a;
    o[n + i] = b;
    o[2 * n + i] = c;
  } else {
    double a = p, b = q, c = t;
    if (param) ell_param_to_xyz(P, p, q, t, a, b, c);
    o[i] = a;
    o[n + i] = b;
    o[2 * n + i] = c;
  }
}

hipError_t surface_eval_launch(const xrt_hip_pass& P, int what, int64_t n, const double* u,
                               const double* v, const double* w, double* o, hipStream_t st) {
  if (P.surf_kind == XRT_HIP_SURF_USER) {   // the user's own functions, from their unit
    const UserUnit* unit = static_cast<const UserUnit*>(P.user_unit);
    if (!unit || !(what == 0 || what == 1 || what == 5)) return hipErrorInvalidValue;
    return unit->eval(&P, what, n, u, v, o, st) == 0 ? hipSuccess : hipErrorLaunchFailure;
  }
  hipLaunchKernelGGL(surface_eval_kernel,
                     dim3((unsigned)((n + REFLECT_BLOCK - 1) / REFLECT_BLOCK)),
                     dim3(REFLECT_BLOCK), 0, st, P, what, n, u, v, w, o);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// OE.local_to_global (oes/base.py:1165-1229) on a device-resident beam, in place: out
// of the element's true local frame (shift, rotations back: P.to_virgin), the coherency
// matrix and the amplitudes turned by roll + atan2(n_x, n_z) -- with the normal the
// reference takes there, at the ALREADY rotated coordinates --, then the beamline
// azimuth and the element's centre. All rays, whatever their state.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(REFLECT_BLOCK) void beam_to_global_kernel(xrt_hip_pass P,
                                                                       xrt_hip_beam b) {
  using K = GenericAll;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n) return;
  double x = b.x[i] + P.shift[0], y = b.y[i] + P.shift[1], z = b.z[i] + P.shift[2];
  double da = b.a[i], db = b.b[i], dc = b.c[i];
  rotate3(P.to_virgin, x, y, z);
  rotate3(P.to_virgin, da, db, dc);
  double px = x, py = y;
  if (P.surf_kind == XRT_HIP_SURF_ELLIPSE_PARAM) {
    double rr;
    ell_xyz_to_param(P, x, y, z, px, py, rr);
  }
  double nn[6];
  surface_normal<K>(P, x, y, px, py, nn);
  double cosY = P.cos_roll, sinY = P.sin_roll;
  if (nn[3] != 0.) {
    const double ih = frcp(fhypot(nn[3], nn[5]));
    const double cphi = nn[5] * ih, sphi = nn[3] * ih;
    cosY = P.cos_roll * cphi - P.sin_roll * sphi;
    sinY = P.sin_roll * cphi + P.cos_roll * sphi;
  } else if (nn[5] < 0.) {
    cosY = -P.cos_roll;
    sinY = -P.sin_roll;
  }
  double Jss = b.Jss[i], Jpp = b.Jpp[i];
  double2 js = reinterpret_cast<double2*>(b.Jsp_ri)[i];
  rot_coherency(cosY, sinY, Jss, Jpp, js.x, js.y);
  b.Jss[i] = Jss;
  b.Jpp[i] = Jpp;
  reinterpret_cast<double2*>(b.Jsp_ri)[i] = js;
  if (b.Es_ri) {
    const double2 es = reinterpret_cast<double2*>(b.Es_ri)[i];
    const double2 ep = reinterpret_cast<double2*>(b.Ep_ri)[i];
    const cplx Es = C(es.x, es.y), Ep = C(ep.x, ep.y);
    const cplx e1 = Es * cosY + Ep * sinY;
    const cplx e2 = Es * (-sinY) + Ep * cosY;
    reinterpret_cast<double2*>(b.Es_ri)[i] = make_double2(e1.re, e1.im);
    reinterpret_cast<double2*>(b.Ep_ri)[i] = make_double2(e2.re, e2.im);
  }
  if (P.out_to_global) {
    if (P.sin_az != 0.) {
      const double an = P.cos_az * da - (-P.sin_az) * db, bn = (-P.sin_az) * da + P.cos_az * db;
      da = an;
      db = bn;
      const double xn = P.cos_az * x - (-P.sin_az) * y, yn = (-P.sin_az) * x + P.cos_az * y;
      x = xn;
      y = yn;
    }
    x += P.center[0];
    y += P.center[1];
    z += P.center[2];
  }
  b.x[i] = x;
  b.y[i] = y;
  b.z[i] = z;
  b.a[i] = da;
  b.b[i] = db;
  b.c[i] = dc;
}

hipError_t beam_to_global_launch(const xrt_hip_pass& P, const xrt_hip_beam& b,
                                 hipStream_t st) {
  if (b.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(beam_to_global_kernel,
                     dim3((unsigned)((b.n + REFLECT_BLOCK - 1) / REFLECT_BLOCK)),
                     dim3(REFLECT_BLOCK), 0, st, P, b);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// waves.diffract around the Kirchhoff integral (waves.py:606-831), on the device. The
// O(Ns Np) integral is csrc/kirchhoff.hip; these O(N) kernels keep its inputs and
// outputs in HBM, so that a chain of diffractions does not bounce every array through
// the host between its steps.
// ---------------------------------------------------------------------------

// Before the integral: what the samples on the diffracting element contribute.
//   normal (the surface's at the sample; (0, 1, 0) for apertures, screens, sources),
//   nl = direction . normal, k = E / CHBAR 1e7 (waves.py:674-689, :841), and the field
//   with every sample that is not in state 1 switched off (Es = Ep = 0: it then adds
//   exact zeros to the sums, which spares compacting the arrays);
//   per block: sum of Jss + Jpp and of (Jss + Jpp) nl over the lit samples, their count
//   (part[3 b .. 3 b + 2]; the host adds the blocks up in order).
__global__ __launch_bounds__(REFLECT_BLOCK) void diffract_pre_kernel(
    xrt_hip_pass P, int is_oe, xrt_hip_beam s, double* __restrict__ sx,
    double* __restrict__ sy, double* __restrict__ sz, double* __restrict__ nx,
    double* __restrict__ ny, double* __restrict__ nz, double* __restrict__ nl,
    double* __restrict__ k, double2* __restrict__ Es, double2* __restrict__ Ep,
    double* __restrict__ part) {
  using K = GenericAll;
  __shared__ double lds_d[REFLECT_MAX_WAVES];
  double sumJ = 0., sumJn = 0., cnt = 0.;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s.n; i += stride) {
    const bool lit = s.state[i] == 1;
    double x = s.x[i], y = s.y[i], z = s.z[i];
    if (!lit && !(isfinite(x) && isfinite(y) && isfinite(z))) x = y = z = 0.;
    double n3[3] = {0., 1., 0.};
    if (is_oe) {
      double px = x, py = y;
      if (P.surf_kind == XRT_HIP_SURF_ELLIPSE_PARAM) {
        double rr;
        ell_xyz_to_param(P, x, y, z, px, py, rr);
      }
      double nn[6];
      surface_normal<K>(P, x, y, px, py, nn);
      n3[0] = nn[3];
      n3[1] = nn[4];
      n3[2] = nn[5];
    }
    const double cosine = s.a[i] * n3[0] + s.b[i] * n3[1] + s.c[i] * n3[2];
    sx[i] = x;
    sy[i] = y;
    sz[i] = z;
    nx[i] = n3[0];
    ny[i] = n3[1];
    nz[i] = n3[2];
    nl[i] = cosine;
    k[i] = s.E[i] / kCHBAR * 1e7;
    double2 es = make_double2(0., 0.), ep = make_double2(0., 0.);
    if (lit) {
      if (s.Es_ri) {
        es = reinterpret_cast<const double2*>(s.Es_ri)[i];
        ep = reinterpret_cast<const double2*>(s.Ep_ri)[i];
      }
      const double flux = s.Jss[i] + s.Jpp[i];
      sumJ += flux;
      sumJn += flux * cosine;
      cnt += 1.;
    }
    Es[i] = es;
    Ep[i] = ep;
  }
  auto faddd = [](double u, double v) { return u + v; };
  sumJ = block_reduce(sumJ, faddd, lds_d);
  sumJn = block_reduce(sumJn, faddd, lds_d);
  cnt = block_reduce(cnt, faddd, lds_d);
  if (threadIdx.x == 0) {
    part[3 * blockIdx.x] = sumJ;
    part[3 * blockIdx.x + 1] = sumJn;
    part[3 * blockIdx.x + 2] = cnt;
  }
}

// After the integral (waves.py:707-749): the new integrals are added to the wave's
// accumulators, and from those come the amplitudes, the coherency matrix and the
// propagation direction -- the three direction integrals share one arbitrary phase,
// removed with the dominant one's (c if the diffracting element is an optical element
// and |c| > |b| at sample 0, else b) --, everything scaled to flux.
__global__ __launch_bounds__(REFLECT_BLOCK) void wave_fields_kernel(
    int64_t n, const double2* __restrict__ nS, const double2* __restrict__ nP,
    const double2* __restrict__ nA, const double2* __restrict__ nB,
    const double2* __restrict__ nC, double2* __restrict__ aS, double2* __restrict__ aP,
    double2* __restrict__ aA, double2* __restrict__ aB, double2* __restrict__ aC,
    const double* __restrict__ energy0, double scale, int from_oe, xrt_hip_beam w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto add = [](double2 u, double2 v) { return make_double2(u.x + v.x, u.y + v.y); };
  // (every lane forms sample 0's sums itself: no ordering against the lane that stores them)
  const double2 b0 = add(aB[0], nB[0]), c0 = add(aC[0], nC[0]);
  const bool use_c = from_oe && fhypot(c0.x, c0.y) > fhypot(b0.x, b0.y);
  const double2 S = add(aS[i], nS[i]), Pp = add(aP[i], nP[i]);
  const double2 A = add(aA[i], nA[i]), B = add(aB[i], nB[i]), Cc = add(aC[i], nC[i]);
  aS[i] = S;
  aP[i] = Pp;
  aA[i] = A;
  aB[i] = B;
  aC[i] = Cc;
  const double2 car = use_c ? Cc : B;
  const double mag = fhypot(car.x, car.y);
  const double ur = mag > 0. ? car.x / mag : 1., ui = mag > 0. ? -car.y / mag : 0.;
  double a = A.x * ur - A.y * ui, b = B.x * ur - B.y * ui, c = Cc.x * ur - Cc.y * ui;
  double len = sqrt(a * a + b * b + c * c);
  if (len == 0.) len = 1.;
  w.a[i] = a / len;
  w.b[i] = b / len;
  w.c[i] = c / len;
  w.E[i] = energy0[0];
  const double rs = sqrt(scale);
  w.Jss[i] = (S.x * S.x + S.y * S.y) * scale;
  w.Jpp[i] = (Pp.x * Pp.x + Pp.y * Pp.y) * scale;
  reinterpret_cast<double2*>(w.Jsp_ri)[i] =
      make_double2((S.x * Pp.x + S.y * Pp.y) * scale, (S.y * Pp.x - S.x * Pp.y) * scale);
  reinterpret_cast<double2*>(w.Es_ri)[i] = make_double2(S.x * rs, S.y * rs);
  reinterpret_cast<double2*>(w.Ep_ri)[i] = make_double2(Pp.x * rs, Pp.y * rs);
}

// Positions (and directions) of a beam from a frame given by three basis vectors and an
// origin (screens, apertures) to the global one, in place.
__global__ __launch_bounds__(REFLECT_BLOCK) void basis_to_global_kernel(
    xrt_hip_screen F, xrt_hip_beam b, int with_directions) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n) return;
  const double x = b.x[i], y = b.y[i], z = b.z[i];
  b.x[i] = F.center[0] + x * F.ex[0] + y * F.ey[0] + z * F.ez[0];
  b.y[i] = F.center[1] + x * F.ex[1] + y * F.ey[1] + z * F.ez[1];
  b.z[i] = F.center[2] + x * F.ex[2] + y * F.ey[2] + z * F.ez[2];
  if (with_directions) {
    const double a = b.a[i], bb = b.b[i], c = b.c[i];
    b.a[i] = a * F.ex[0] + bb * F.ey[0] + c * F.ez[0];
    b.b[i] = a * F.ex[1] + bb * F.ey[1] + c * F.ez[1];
    b.c[i] = a * F.ex[2] + bb * F.ey[2] + c * F.ez[2];
  }
}

// The receiving samples live on an element (waves.py:773-824): the diffracted field,
// known in the global frame (glo), goes into the element's local s/p frame -- directions
// through the azimuth and the element's rotations, coherency matrix and amplitudes turned
// by -(roll + atan2(n_x, n_z)) -- and the flux is projected on the surface (obliquity, also
// applied to glo). A receiver that is no optical element (screen, aperture) only has the
// azimuth taken out of the directions.
__global__ __launch_bounds__(REFLECT_BLOCK) void wave_receive_kernel(xrt_hip_pass P, int is_oe,
                                                                     xrt_hip_beam w,
                                                                     xrt_hip_beam g) {
  using K = GenericAll;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w.n) return;
  double a = g.a[i], b = g.b[i], c = g.c[i];
  double Jss = g.Jss[i], Jpp = g.Jpp[i];
  double2 js = reinterpret_cast<const double2*>(g.Jsp_ri)[i];
  const double2 es0 = reinterpret_cast<const double2*>(g.Es_ri)[i];
  const double2 ep0 = reinterpret_cast<const double2*>(g.Ep_ri)[i];
  cplx Es = C(es0.x, es0.y), Ep = C(ep0.x, ep0.y);
  {
    const double an = P.cos_az * a - P.sin_az * b, bn = P.sin_az * a + P.cos_az * b;
    a = an;
    b = bn;
  }
  if (is_oe) {
    const double x = w.x[i], y = w.y[i], z = w.z[i];
    double px = x, py = y;
    if (P.surf_kind == XRT_HIP_SURF_ELLIPSE_PARAM) {
      double rr;
      ell_xyz_to_param(P, x, y, z, px, py, rr);
    }
    double nn[6];
    surface_normal<K>(P, x, y, px, py, nn);
    double cosY = P.cos_roll, sinY = P.sin_roll;
    if (nn[3] != 0.) {
      const double ih = frcp(fhypot(nn[3], nn[5]));
      const double cphi = nn[5] * ih, sphi = nn[3] * ih;
      cosY = P.cos_roll * cphi - P.sin_roll * sphi;
      sinY = P.sin_roll * cphi + P.cos_roll * sphi;
    } else if (nn[5] < 0.) {
      cosY = -P.cos_roll;
      sinY = -P.sin_roll;
    }
    rot_coherency(cosY, -sinY, Jss, Jpp, js.x, js.y);
    const cplx e1 = Es * cosY + Ep * (-sinY);
    const cplx e2 = Es * sinY + Ep * cosY;
    Es = e1;
    Ep = e2;
    rotate3(P.to_local, a, b, c);
    const double obl = fabs(-a * nn[3] - b * nn[4] - c * nn[5]);
    const double ro = sqrt(obl);
    Jss *= obl;
    Jpp *= obl;
    js.x *= obl;
    js.y *= obl;
    Es = Es * ro;
    Ep = Ep * ro;
    g.Jss[i] *= obl;
    g.Jpp[i] *= obl;
    double2 gj = reinterpret_cast<double2*>(g.Jsp_ri)[i];
    reinterpret_cast<double2*>(g.Jsp_ri)[i] = make_double2(gj.x * obl, gj.y * obl);
    reinterpret_cast<double2*>(g.Es_ri)[i] = make_double2(es0.x * ro, es0.y * ro);
    reinterpret_cast<double2*>(g.Ep_ri)[i] = make_double2(ep0.x * ro, ep0.y * ro);
  }
  w.a[i] = a;
  w.b[i] = b;
  w.c[i] = c;
  w.Jss[i] = Jss;
  w.Jpp[i] = Jpp;
  reinterpret_cast<double2*>(w.Jsp_ri)[i] = js;
  reinterpret_cast<double2*>(w.Es_ri)[i] = make_double2(Es.re, Es.im);
  reinterpret_cast<double2*>(w.Ep_ri)[i] = make_double2(Ep.re, Ep.im);
}

#define DIFFRACT_PRE_BLOCKS 256
hipError_t diffract_pre_launch(const xrt_hip_pass& P, int is_oe, const xrt_hip_beam& s,
                               double* sx, double* sy, double* sz, double* nx, double* ny,
                               double* nz, double* nl, double* k, double* Es, double* Ep,
                               double* part, int* nblocks, hipStream_t st) {
  int64_t b = (s.n + REFLECT_BLOCK - 1) / REFLECT_BLOCK;
  if (b > DIFFRACT_PRE_BLOCKS) b = DIFFRACT_PRE_BLOCKS;
  if (b < 1) b = 1;
  *nblocks = (int)b;
  hipLaunchKernelGGL(diffract_pre_kernel, dim3((unsigned)b), dim3(REFLECT_BLOCK), 0, st, P,
                     is_oe, s, sx, sy, sz, nx, ny, nz, nl, k, reinterpret_cast<double2*>(Es),
                     reinterpret_cast<double2*>(Ep), part);
  return hipGetLastError();
}

hipError_t wave_fields_launch(int64_t n, double* const* fresh, double* const* acc,
                              const double* energy0, double scale, int from_oe,
                              const xrt_hip_beam& w, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  auto c2 = [](double* p) { return reinterpret_cast<double2*>(p); };
  hipLaunchKernelGGL(wave_fields_kernel, dim3((unsigned)((n + REFLECT_BLOCK - 1) / REFLECT_BLOCK)),
                     dim3(REFLECT_BLOCK), 0, st, n, c2(fresh[0]), c2(fresh[1]), c2(fresh[2]),
                     c2(fresh[3]), c2(fresh[4]), c2(acc[0]), c2(acc[1]), c2(acc[2]), c2(acc[3]),
                     c2(acc[4]), energy0, scale, from_oe, w);
  return hipGetLastError();
}

hipError_t basis_to_global_launch(const xrt_hip_screen& F, const xrt_hip_beam& b,
                                  int with_directions, hipStream_t st) {
  if (b.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(basis_to_global_kernel,
                     dim3((unsigned)((b.n + REFLECT_BLOCK - 1) / REFLECT_BLOCK)),
                     dim3(REFLECT_BLOCK), 0, st, F, b, with_directions);
  return hipGetLastError();
}

hipError_t wave_receive_launch(const xrt_hip_pass& P, int is_oe, const xrt_hip_beam& w,
                               const xrt_hip_beam& g, hipStream_t st) {
  if (w.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(wave_receive_kernel,
                     dim3((unsigned)((w.n + REFLECT_BLOCK - 1) / REFLECT_BLOCK)),
                     dim3(REFLECT_BLOCK), 0, st, P, is_oe, w, g);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// stand-alone amplitude kernels (Material.get_amplitude / Crystal.get_amplitude)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(REFLECT_BLOCK) void material_amplitude_kernel(
    xrt_hip_material M, int64_t n, const double* __restrict__ E,
    const double* __restrict__ bdn, double2* __restrict__ rs, double2* __restrict__ rp,
    double* __restrict__ mu, double* __restrict__ nk) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  cplx n_own = C(1., 0.);
  if (M.n_fixed == 2) n_own = C(M.n_ray[2 * i], M.n_ray[2 * i + 1]);   // tabulated n(E)
  const Ampl A = material_amplitude(M, M.kind, E[i], bdn[i], full_window(),
                                    M.n_fixed == 2 ? &n_own : nullptr);
  rs[i] = make_double2(A.rs.re, A.rs.im);
  rp[i] = make_double2(A.rp.re, A.rp.im);
  if (mu) mu[i] = A.mu;
  if (nk) nk[i] = A.nk;
}

__global__ __launch_bounds__(REFLECT_BLOCK) void crystal_amplitude_kernel(
    xrt_hip_material M, int64_t n, const double* __restrict__ E,
    const double* __restrict__ g0, const double* __restrict__ gh,
    const double* __restrict__ hns, double2* __restrict__ S, double2* __restrict__ P) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Ampl A = crystal_amplitude(M, E[i], g0[i], gh[i], hns[i], full_window());
  S[i] = make_double2(A.rs.re, A.rs.im);
  P[i] = make_double2(A.rp.re, A.rp.im);
}

__global__ __launch_bounds__(REFLECT_BLOCK) void multilayer_amplitude_kernel(
    xrt_hip_material M, int64_t n, const double* __restrict__ E,
    const double* __restrict__ bdn, double2* __restrict__ rs, double2* __restrict__ rp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Ampl A = multilayer_amplitude(*M.layers, E[i], bdn[i]);
  rs[i] = make_double2(A.rs.re, A.rs.im);
  rp[i] = make_double2(A.rp.re, A.rp.im);
}

hipError_t multilayer_amplitude_launch(const xrt_hip_material& M, int64_t n, const double* E,
                                       const double* bdn, double* rs, double* rp,
                                       hipStream_t st) {
  hipLaunchKernelGGL(multilayer_amplitude_kernel,
                     dim3((unsigned)((n + REFLECT_BLOCK - 1) / REFLECT_BLOCK)),
                     dim3(REFLECT_BLOCK), 0, st, M, n, E, bdn, reinterpret_cast<double2*>(rs),
                     reinterpret_cast<double2*>(rp));
  return hipGetLastError();
}

hipError_t material_amplitude_launch(const xrt_hip_material& M, int64_t n, const double* E,
                                     const double* bdn, double* rs, double* rp, double* mu,
                                     double* nk, hipStream_t st) {
  hipLaunchKernelGGL(material_amplitude_kernel,
                     dim3((unsigned)((n + REFLECT_BLOCK - 1) / REFLECT_BLOCK)),
                     dim3(REFLECT_BLOCK), 0, st, M, n, E, bdn, reinterpret_cast<double2*>(rs),
                     reinterpret_cast<double2*>(rp), mu, nk);
  return hipGetLastError();
}

hipError_t crystal_amplitude_launch(const xrt_hip_material& M, int64_t n, const double* E,
                                    const double* g0, const double* gh, const double* hns,
                                    double* S, double* P, hipStream_t st) {
  hipLaunchKernelGGL(crystal_amplitude_kernel,
                     dim3((unsigned)((n + REFLECT_BLOCK - 1) / REFLECT_BLOCK)),
                     dim3(REFLECT_BLOCK), 0, st, M, n, E, g0, gh, hns,
                     reinterpret_cast<double2*>(S), reinterpret_cast<double2*>(P));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------
size_t reflect_workspace_bytes(int64_t n) {
  // two passes' worth (DCM) of GStat (256 B) + per-block partials, then t, x, y, z
  // (4 x 8n) + state (4n) of the crystal tail
  const size_t a = ((size_t)n * 8 + 255) / 256 * 256;
  const size_t s = ((size_t)n * 4 + 255) / 256 * 256;
  return 2 * (256 + REFLECT_PART_BYTES) + 4 * a + s;
}

// any array of one beam is an array of the other
static bool beams_overlap(const xrt_hip_beam& a, const xrt_hip_beam& b) {
  const void* pa[] = {a.x, a.y, a.z, a.a, a.b, a.c, a.path, a.E, a.Jss, a.Jpp, a.Jsp_ri,
                      a.state, a.Es_ri, a.Ep_ri};
  const void* pb[] = {b.x, b.y, b.z, b.a, b.b, b.c, b.path, b.E, b.Jss, b.Jpp, b.Jsp_ri,
                      b.state, b.Es_ri, b.Ep_ri};
  for (const void* u : pa)
    for (const void* v : pb)
      if (u && u == v) return true;
  return false;
}

__global__ void reflect_init(GStat* g, int redo) {
  gstat_reset(g, redo);
  g->bar = 0;
  g->hang = 0;
}

// blocks of reflect_exact: all of them resident at once (one 256-lane block per CU:
// 1 wave per SIMD, up to 512 VGPRs -- the generic code of every phase without spilling)
static unsigned exact_blocks(int64_t n) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 64;
  }
  int64_t b = (n + REFLECT_EXACT_BLOCK - 1) / REFLECT_EXACT_BLOCK;
  if (b > cus) b = cus;
  if (b > (int64_t)REFLECT_MAX_PART) b = REFLECT_MAX_PART;
  return (unsigned)(b < 1 ? 1 : b);
}

struct WsLayout {
  GStat *g1, *g2;
  double *part1, *part2;
  double *ht, *hx, *hy, *hz;
  int32_t* hst;
};

static WsLayout ws_layout(void* workspace, int64_t n) {
  const size_t a = ((size_t)n * 8 + 255) / 256 * 256;
  char* w = reinterpret_cast<char*>(workspace);
  WsLayout L;
  L.g1 = reinterpret_cast<GStat*>(w);
  L.part1 = reinterpret_cast<double*>(w + 256);
  L.g2 = reinterpret_cast<GStat*>(w + 256 + REFLECT_PART_BYTES);
  L.part2 = reinterpret_cast<double*>(w + 2 * 256 + REFLECT_PART_BYTES);
  char* base = w + 2 * (256 + REFLECT_PART_BYTES);
  L.ht = reinterpret_cast<double*>(base);
  L.hx = reinterpret_cast<double*>(base + a);
  L.hy = reinterpret_cast<double*>(base + 2 * a);
  L.hz = reinterpret_cast<double*>(base + 3 * a);
  L.hst = reinterpret_cast<int32_t*>(base + 4 * a);
  return L;
}

int user_unit_abi() { return XRT_USER_UNIT_ABI; }

// (the conditions under which reflect_pass_launch below picks a lean kernel and the optimistic
// single pass, for a caller that has to know beforehand)
bool reflect_pass_carries_screen(const xrt_hip_pass& P, const xrt_hip_material& M,
                                 const xrt_hip_screen& S) {
  const bool nis = P.no_intersection_search != 0;
  const bool need_mean = deflects_as_crystal(M) && !M.geom_transmitted;
  const bool wide = P.surf_kind == XRT_HIP_SURF_BENT_BRAGG || P.surf_kind == XRT_HIP_SURF_VFM ||
                    P.surf_kind == XRT_HIP_SURF_DUALVFM || P.surf_kind == XRT_HIP_SURF_DICED;
  if (nis || need_mean || wide || P.fe_c || P.g_ray_x || M.kind == XRT_HIP_MAT_MULTILAYER ||
      P.surf_kind >= XRT_HIP_SURF_BLAZED || P.grating || P.asymmetric || M.n_fixed == 2 ||
      S.radius != 0.)
    return false;
  if (M.kind == XRT_HIP_MAT_MIRROR)
    return P.surf_kind == XRT_HIP_SURF_TOROID || P.surf_kind == XRT_HIP_SURF_FLAT ||
           P.surf_kind == XRT_HIP_SURF_BENTFLAT;
  return M.kind == XRT_HIP_MAT_PLATE && P.surf_kind == XRT_HIP_SURF_FLAT;
}

hipError_t reflect_pass_launch(const xrt_hip_pass& P, const xrt_hip_material& M,
                               const xrt_hip_beam& in, const xrt_hip_beam& restore,
                               const xrt_hip_beam& lb, const xrt_hip_beam& vb, double* theta,
                               void* workspace, hipStream_t st, hipEvent_t ev0,
                               hipEvent_t ev1, hipEvent_t evk0, hipEvent_t evk1,
                               bool force_exact, const xrt_hip_screen* scr,
                               const xrt_hip_beam* sb, bool keep_virgin, int* fused,
                               const xrt_hip_geosource* src, const PlotTailPlan* plot,
                               bool keep_screen, const TailApertures* ap) {
  static_assert(sizeof(GStat) <= 256, "workspace head slot");
  // apertures right behind the element (their marks in the outgoing beam's states) with or
  // without a screen behind them: a screen that images nothing stands in
  xrt_hip_screen no_screen;
  xrt_hip_beam no_image;
  if (ap && ap->n == 0) ap = nullptr;
  if (ap && !scr) {
    memset(&no_screen, 0, sizeof(no_screen));
    memset(&no_image, 0, sizeof(no_image));
    no_image.n = in.n;
    scr = &no_screen;
    sb = &no_image;
    keep_virgin = true;          // (the marked global beam is what the caller wants)
  }
  const bool real_screen = scr && scr != &no_screen;
  if (fused) *fused = 0;
  static_assert(REFLECT_OPT_SLOTS * sizeof(OptStat) <= REFLECT_PART_BYTES, "report slots");
  const int64_t n = in.n;
  if (n <= 0) return hipSuccess;
  const WsLayout L = ws_layout(workspace, n);
  GStat* g = L.g1;
  double* part = L.part1;
  OptStat* opt = reinterpret_cast<OptStat*>(part);
  const dim3 grid((unsigned)((n + REFLECT_FUSED_BLOCK - 1) / REFLECT_FUSED_BLOCK)),
      block(REFLECT_BLOCK), fblock(REFLECT_FUSED_BLOCK);
  // The optimistic single pass (see decide_opt_body) needs the input intact for a
  // possible redo, and surfaces that bracket at all.
  const bool aliased = beams_overlap(in, lb) || beams_overlap(in, vb) ||
                       beams_overlap(restore, lb) || beams_overlap(restore, vb);
  const bool nis = P.no_intersection_search != 0;
  const bool searches = !nis && P.surf_kind != XRT_HIP_SURF_BLAZED;
  const bool need_mean = deflects_as_crystal(M) && !M.geom_transmitted;
  // Bragg crystals on surfaces outside the crystal kernels' families (conics, cone, lens
  // paraboloid, VFM / DualVFM -- rare: a paraboloidal analyser): the generic exact sequence of
  // the surface's family knows both, the fused crystal kernels (family 0 + bent shapes) do not
  const bool xtal_elsewhere =
      need_mean && M.kind == XRT_HIP_MAT_CRYSTAL &&
      (P.surf_kind == XRT_HIP_SURF_ELLIPSE_PARAM || P.surf_kind == XRT_HIP_SURF_PARABOLOID ||
       P.surf_kind == XRT_HIP_SURF_CONE || P.surf_kind == XRT_HIP_SURF_VFM ||
       P.surf_kind == XRT_HIP_SURF_DUALVFM || P.surf_kind == XRT_HIP_SURF_USER);
  // OE(figureError = ...): the Figured kernels hold the height-map spline; a Bragg crystal with
  // a figure error takes their exact sequence (its fused crystal kernels do not)
  const bool figured = P.fe_c != nullptr;
  const bool xtal_figured = figured && need_mean && M.kind == XRT_HIP_MAT_CRYSTAL;
  const bool optimistic = searches && !force_exact && !aliased && !xtal_elsewhere && !xtal_figured;
  const bool flat_xtal = P.surf_kind == XRT_HIP_SURF_FLAT;
  const bool layers = M.kind == XRT_HIP_MAT_MULTILAYER;
  const bool wide = P.surf_kind == XRT_HIP_SURF_BENT_BRAGG || P.surf_kind == XRT_HIP_SURF_VFM ||
                    P.surf_kind == XRT_HIP_SURF_DUALVFM || P.surf_kind == XRT_HIP_SURF_DICED;
  // which instantiation serves this (surface, material): see reflect_tu.h
  const bool plain = !P.grating && !P.asymmetric && !P.no_intersection_search;
  int family_spec;   // what reflect_exact is compiled for: surface family x layered / zones
  if (P.g_ray_x)
    family_spec = SP_PER_RAY_ZONES;
  else if (layers)
    family_spec = wide ? SP_LAYERED2 : (P.surf_kind >= XRT_HIP_SURF_BLAZED ? SP_LAYERED1 : SP_LAYERED0);
  else
    family_spec = wide ? SP_GENERIC2 : (P.surf_kind >= XRT_HIP_SURF_BLAZED ? SP_GENERIC1 : SP_GENERIC0);
  if (figured) {
    if (layers || P.g_ray_x || (need_mean && (!xtal_figured || nis)) ||
        P.surf_kind == XRT_HIP_SURF_USER || P.surf_kind == XRT_HIP_SURF_ELLIPSE_PARAM)
      return hipErrorInvalidValue;      // (capi.hip says why before it gets here)
    family_spec = wide ? SP_FIGURED2 : (P.surf_kind >= XRT_HIP_SURF_BLAZED ? SP_FIGURED1 : SP_FIGURED0);
  }
  int spec = family_spec;
  if (figured) {
    // (no lean kernel, no crystal kernel)
  } else if (need_mean && layers) {
    spec = wide ? SP_LAYERED2 : (P.surf_kind >= XRT_HIP_SURF_BLAZED ? SP_LAYERED1 : SP_LAYERED0);
  } else if (need_mean) {
    // Bragg-reflecting crystals sit on flat surfaces in practice (DCM): that case is
    // compiled with the kinds fixed. Each ray takes its own sign of beamInDotNormal
    // and raises any_neg / any_pos (see reflect_fused_xtal).
    if (M.thick && M.structure != 2)
      spec = flat_xtal ? SP_THICK_FLAT : SP_THICK_ANY;
    else
      spec = flat_xtal ? SP_FLAT_XTAL : SP_ANY_XTAL;
  } else if (!need_mean && family_spec == SP_GENERIC0 && plain && M.n_fixed != 2) {
    // (a per-ray refractive index -- Material(refractiveIndex = table) -- stays with the
    // generic kernels: the lean ones do not carry its test)
    if (M.kind == XRT_HIP_MAT_MIRROR && P.surf_kind == XRT_HIP_SURF_TOROID)
      spec = SP_TOROID_MIRROR;
    else if (M.kind == XRT_HIP_MAT_MIRROR && P.surf_kind == XRT_HIP_SURF_FLAT)
      spec = SP_FLAT_MIRROR;
    else if (M.kind == XRT_HIP_MAT_MIRROR && P.surf_kind == XRT_HIP_SURF_BENTFLAT)
      spec = SP_BENT_MIRROR;
    else if (M.kind == XRT_HIP_MAT_PLATE && P.surf_kind == XRT_HIP_SURF_FLAT)
      spec = SP_FLAT_PLATE;
  }
  PassAux A;
  A.theta = theta;
  A.g = g;
  A.part = part;
  A.ht = L.ht;
  A.hx = L.hx;
  A.hy = L.hy;
  A.hz = L.hz;
  A.hst = L.hst;
  A.aliased = aliased ? 1 : 0;
  // A screen in the tail of the pass (Screen.expose of the global beam, flat screens): the
  // lean kernels carry it; every other pass is followed by the screen's own launch. In the
  // fused form `vb` is written only if the caller keeps it (or if the exact sequence has to
  // redo the pass: then the image is made from it afterwards).
  const bool lean = spec == SP_TOROID_MIRROR || spec == SP_FLAT_MIRROR ||
                    spec == SP_BENT_MIRROR || spec == SP_FLAT_PLATE;
  // ... and the fused kernels of a single flat Bragg crystal (apertures and screen; no plot, no source)
  const bool xtal_tail = need_mean && !layers && !figured && M.kind == XRT_HIP_MAT_CRYSTAL &&
                         (spec == SP_THICK_FLAT || spec == SP_FLAT_XTAL) &&
                         family_spec == SP_GENERIC0 && !plot && !src &&
                         P.surf_kind != XRT_HIP_SURF_USER;
  const bool fuse_screen = scr && sb && optimistic && (lean || xtal_tail) && scr->radius == 0.;
  const TailApertures none{};
  // ... and the plot of the screen's image behind it (plot_tail.h): only in a tail
  if (plot && !fuse_screen) return hipErrorInvalidValue;   // (capi.hip asks ..._fusable first)
  xrt_hip_beam sb_fused;
  if (plot) {
    sb_fused = *sb;
    if (!keep_screen) {       // the image itself is not wanted: its records are all that leaves
      sb_fused.x = sb_fused.y = sb_fused.z = sb_fused.a = sb_fused.b = sb_fused.c = nullptr;
      sb_fused.path = sb_fused.E = sb_fused.Jss = sb_fused.Jpp = sb_fused.Jsp_ri = nullptr;
      sb_fused.state = nullptr;
      sb_fused.Es_ri = sb_fused.Ep_ri = nullptr;
    }
    sb = &sb_fused;
  }
  xrt_hip_beam vb_fused = vb;
  if (fuse_screen && !keep_virgin) {
    vb_fused.x = vb_fused.y = vb_fused.z = vb_fused.a = vb_fused.b = vb_fused.c = nullptr;
    vb_fused.path = vb_fused.E = vb_fused.Jss = vb_fused.Jpp = vb_fused.Jsp_ri = nullptr;
    vb_fused.state = nullptr;
    vb_fused.Es_ri = vb_fused.Ep_ri = nullptr;
  }
  // The source in the head of the pass (GeometricSource.shine on the device): `in` is then a
  // scratch beam that is filled only if the exact sequence has to redo the pass. Mirrors only
  // (the lean kernels), with the screen in the tail; otherwise the generator's own launch
  // fills `in` first and everything is as usual.
  const bool fuse_source = src && fuse_screen && spec != SP_FLAT_PLATE && src->state > 0 &&
                           P.good_mode == 0 && restore.x == in.x;
  if (src && !fuse_source) {
    const hipError_t ge = geosource_shine_launch(*src, in, st);
    if (ge != hipSuccess) return ge;
  }
  const FusedLaunch FL{grid, fblock, st, &P, &M, &in, &restore, &lb,
                       fuse_screen ? &vb_fused : &vb, theta, g, opt,
                       fuse_screen ? scr : nullptr, fuse_screen ? sb : nullptr,
                       fuse_source ? src : nullptr, plot ? &plot->Q : nullptr,
                       fuse_screen && ap ? ap : &none};
  const ExactLaunch XL{dim3(exact_blocks(n)), dim3(REFLECT_EXACT_BLOCK), st, &P, &M, &in,
                       &restore, &lb, &vb, A};
  bool launched = true;
  const UserUnit* unit = P.surf_kind == XRT_HIP_SURF_USER
                             ? static_cast<const UserUnit*>(P.user_unit) : nullptr;
  if (P.surf_kind == XRT_HIP_SURF_USER &&
      (!unit || layers != (unit->layered != 0) ||
       (need_mean && !layers && (!xtal_elsewhere || nis))))
    return hipErrorInvalidValue;        // (capi.hip says why before it gets here)
  // the solve + finish kernel: mode 0 (optimistic) or 2 (no statistics needed)
  auto launch_fused = [&](int mode) {
    if (plot)
      launched &= tu_hot_fused_scr_plot(spec, mode, FL);
    else if (fuse_screen && xtal_tail)
      launched &= tu_hot_xtal_scr(spec, mode, FL);
    else if (fuse_screen)
      launched &= tu_hot_fused_scr(spec, mode, FL);
    else if (unit)      // (need_mean here: a multilayer deflecting as a crystal -- layered flavour)
      launched &= (need_mean ? unit->xtal(mode, &FL) : unit->fused(mode, &FL)) == 0;
    else if (figured)
      launched &= tu_figured_fused(spec, mode, FL);
    else if (need_mean)
      launched &= tu_hot_xtal(spec, mode, FL) || tu_xtal_xtal(spec, mode, FL) ||
                  tu_layered_xtal(spec, mode, FL);
    else
      launched &= tu_hot_fused(spec, mode, FL) || tu_generic_fused(spec, mode, FL) ||
                  tu_layered_fused(spec, mode, FL);
  };
  auto launch_exact = [&]() {
    BarrierSerial one_at_a_time(st);      // (grid barriers: reflect.h)
    if (unit)
      launched &= unit->exact(&XL) == 0;
    else if (figured)
      launched &= tu_figured_exact0(family_spec, XL) || tu_figured_exact1(family_spec, XL);
    else
      launched &= tu_exact0(family_spec, XL) || tu_exact1(family_spec, XL) ||
                  tu_exact2(family_spec, XL) || tu_exact3(family_spec, XL);
  };
  if (ev0) (void)hipEventRecord(ev0, st);
  if (fuse_source) {
    // ray 0 made by the decide kernel stands for the head of the beam; the pass makes its rays
    // itself; the beam is written out only for a redo
    xrt_hip_beam head;
    memset(&head, 0, sizeof(head));
    head.n = 1;
    head.state = reinterpret_cast<int32_t*>(L.ht);
    head.a = L.ht + 1;
    head.b = L.ht + 2;
    head.c = L.ht + 3;
    head.E = L.ht + 4;
    head.x = head.y = head.z = head.path = head.Jss = head.Jpp = L.ht + 5;
    head.Jsp_ri = L.ht + 6;
    hipLaunchKernelGGL(reflect_decide_opt_gen, dim3(1), block, 0, st, P, M, *src, head, part, g);
    if (evk0) (void)hipEventRecord(evk0, st);
    launched &= plot ? tu_hot_fused_gen_scr_plot(spec, FL) : tu_hot_fused_gen_scr(spec, FL);
    if (evk1) (void)hipEventRecord(evk1, st);
    // verdict, and only if it was contradicted: the source's beam, the exact sequence, the image
    {
      BarrierSerial one_at_a_time(st);
      tu_exact0_redo_scr(XL, *scr, *sb, src, plot ? &plot->Q : nullptr, ap ? ap : &none);
    }
  } else if (fuse_screen) {
    hipLaunchKernelGGL(reflect_decide_opt, dim3(1), block, 0, st, P, M, in, part, g);
    if (evk0) (void)hipEventRecord(evk0, st);
    launch_fused(0);
    if (evk1) (void)hipEventRecord(evk1, st);
    {
      BarrierSerial one_at_a_time(st);
      tu_exact0_redo_scr(XL, *scr, *sb, nullptr, plot ? &plot->Q : nullptr, ap ? ap : &none);
    }
  } else if (optimistic) {
    // assumptions from the head of the beam -> the pass on them, every ray checking ->
    // reflect_exact: folds the reports, returns at once unless one was contradicted
    hipLaunchKernelGGL(reflect_decide_opt, dim3(1), block, 0, st, P, M, in, part, g);
    if (evk0) (void)hipEventRecord(evk0, st);
    launch_fused(0);
    if (evk1) (void)hipEventRecord(evk1, st);
    launch_exact();
  } else if (nis) {
    // no intersection search: no batch statistics; only a crystal batch with both signs
    // of beamInDotNormal needs the tail
    const bool tail_only = need_mean && aliased;   // the tail reads every ray before writing
    hipLaunchKernelGGL(reflect_init, dim3(1), dim3(1), 0, st, g, tail_only ? 1 : 0);
    if (evk0) (void)hipEventRecord(evk0, st);
    if (tail_only) {
      launch_exact();
    } else {
      launch_fused(2);
      if (need_mean) launch_exact();
    }
    if (evk1) (void)hipEventRecord(evk1, st);
  } else {
    hipLaunchKernelGGL(reflect_init, dim3(1), dim3(1), 0, st, g, 1);
    if (evk0) (void)hipEventRecord(evk0, st);
    launch_exact();
    if (evk1) (void)hipEventRecord(evk1, st);
  }
  if (scr && sb) {
    // (fused: the redo, if any, has marked the real vb and made the image from it --
    // reflect_redo_scr); otherwise the apertures' and the screen's own launches
    if (!fuse_screen) {
      if (ap) {
        xrt_hip_beam nothing;
        memset(&nothing, 0, sizeof(nothing));
        nothing.n = in.n;
        for (int k = 0; k < ap->n; ++k) {
          const hipError_t ae = aperture_propagate_launch(ap->a[k], vb, nothing, nothing, st);
          if (ae != hipSuccess) return ae;
        }
      }
      if (real_screen) {
        const hipError_t se = screen_expose_launch(*scr, vb, *sb, st);
        if (se != hipSuccess) return se;
      }
    }
    if (fused)
      *fused = (fuse_screen && real_screen ? 1 : 0) | (fuse_source ? 2 : 0) | (plot ? 4 : 0) |
               (fuse_screen && ap ? 8 : 0);
  }
  if (plot) {
    // the records of the pass (or of its redo) into the plot's accumulators: two small kernels
    const hipError_t pe = plot_tail_finish(*plot, st);
    if (pe != hipSuccess) return pe;
  }
  if (ev1) (void)hipEventRecord(ev1, st);
  if (!launched) return hipErrorInvalidDeviceFunction;   // no unit holds this spec
  return hipGetLastError();
}

// both crystals flat, both Bragg-reflecting (the reference's DCM), the same thickness
// class (one kernel instantiation serves both), beam enters in the global frame and
// leaves in it
bool reflect_dcm_fusable(const xrt_hip_pass& P1, const xrt_hip_material& M1,
                         const xrt_hip_pass& P2, const xrt_hip_material& M2) {
  auto bragg = [](const xrt_hip_material& M) {   // (one-element lattices: the fused kernel
    // looks the anomalous factor up once for both crystals)
    return M.kind == XRT_HIP_MAT_CRYSTAL && !M.geom_transmitted && M.structure != 2;
  };
  auto flat = [](const xrt_hip_pass& P) {
    return P.surf_kind == XRT_HIP_SURF_FLAT && !P.no_intersection_search && !P.grating;
  };
  if (P1.fe_c || P2.fe_c) return false;     // a figure error: two passes of the Figured kernels
  // the two faces of a flat plate (Plate.double_refract): what the lean plate kernel takes
  auto face = [](const xrt_hip_pass& P, const xrt_hip_material& M) {
    return M.kind == XRT_HIP_MAT_PLATE && M.n_fixed != 2 && !P.asymmetric && !P.g_ray_x;
  };
  const bool pair = (bragg(M1) && bragg(M2) && (M1.thick != 0) == (M2.thick != 0)) ||
                    (face(P1, M1) && face(P2, M2));
  return pair && flat(P1) && flat(P2) &&
         !P1.out_to_global && !P2.in_is_global && !P1.only_state1_out && !P2.only_state1_out;
}

hipError_t reflect_dcm_launch(const xrt_hip_pass& P1, const xrt_hip_material& M1,
                              const xrt_hip_pass& P2, const xrt_hip_material& M2,
                              const xrt_hip_beam& in, const xrt_hip_beam& lo1,
                              const xrt_hip_beam& lo2, const xrt_hip_beam& gb2,
                              double* theta1, double* theta2, void* workspace,
                              hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                              hipEvent_t evk0, hipEvent_t evk1, bool force_exact,
                              const xrt_hip_screen* scr, const xrt_hip_beam* sb, bool keep_global,
                              const TailApertures* ap, int* fused) {
  const int64_t n = in.n;
  if (fused) *fused = 0;
  if (n <= 0) return hipSuccess;
  // apertures / a flat screen right behind the monochromator: in the tail of the fused kernel of
  // the crystal pairs (the plate pair and a forced exact sequence: their own launches below)
  if (ap && ap->n == 0) ap = nullptr;
  const bool tail = (scr && sb) || ap;
  const bool fuse_tail = tail && !force_exact && M1.kind != XRT_HIP_MAT_PLATE &&
                         (!scr || scr->radius == 0.);
  xrt_hip_screen no_screen;
  xrt_hip_beam no_image;
  memset(&no_screen, 0, sizeof(no_screen));
  memset(&no_image, 0, sizeof(no_image));
  no_image.n = n;
  const TailApertures none{};
  if (!scr || !sb) keep_global = true;      // (the marked global beam is what the caller wants)
  const WsLayout L = ws_layout(workspace, n);
  const dim3 grid((unsigned)((n + REFLECT_DCM_BLOCK - 1) / REFLECT_DCM_BLOCK)),
      block(REFLECT_BLOCK), fblock(REFLECT_DCM_BLOCK);
  // (redo: the beam between the crystals lives in gb2's arrays)
  PassAux a1, a2;
  a1.theta = theta1;
  a1.g = L.g1;
  a1.part = L.part1;
  a1.ht = L.ht;
  a1.hx = L.hx;
  a1.hy = L.hy;
  a1.hz = L.hz;
  a1.hst = L.hst;
  a1.aliased = 0;
  a2 = a1;
  a2.theta = theta2;
  a2.g = L.g2;
  a2.part = L.part2;
  a2.aliased = 1;
  const DcmLaunch DL{grid, fblock, st, &P1, &P2, &M1, &M2, &in, &lo1, &lo2, &gb2, theta1, theta2,
                     L.g1, L.g2, reinterpret_cast<OptStat*>(L.part1),
                     reinterpret_cast<OptStat*>(L.part2), a1, a2};
  if (ev0) (void)hipEventRecord(ev0, st);
  if (force_exact) {
    hipLaunchKernelGGL(reflect_init, dim3(1), dim3(1), 0, st, L.g1, 1);
    hipLaunchKernelGGL(reflect_init, dim3(1), dim3(1), 0, st, L.g2, 1);
    if (evk0) (void)hipEventRecord(evk0, st);
  } else {
    hipLaunchKernelGGL(reflect_decide_dcm, dim3(1), block, 0, st, P1, M1, P2, M2, in, L.part1,
                       L.part2, L.g1, L.g2);
    if (evk0) (void)hipEventRecord(evk0, st);
    const int spec = M1.kind == XRT_HIP_MAT_PLATE ? SP_FLAT_PLATE
                                                  : (M1.thick ? SP_THICK_FLAT : SP_FLAT_XTAL);
    if (fuse_tail) {
      xrt_hip_beam gb_fused = gb2;
      if (!keep_global) {
        gb_fused.x = gb_fused.y = gb_fused.z = gb_fused.a = gb_fused.b = gb_fused.c = nullptr;
        gb_fused.path = gb_fused.E = gb_fused.Jss = gb_fused.Jpp = gb_fused.Jsp_ri = nullptr;
        gb_fused.state = nullptr;
        gb_fused.Es_ri = gb_fused.Ep_ri = nullptr;
      }
      if (!tu_hot_dcm_scr(spec, DL, gb_fused, scr ? *scr : no_screen, scr && sb ? *sb : no_image,
                          ap ? *ap : none))
        return hipErrorInvalidDeviceFunction;
    } else if (!(tu_hot_dcm(spec, DL) || tu_xtal_dcm(spec, DL) || tu_hot_plate2(spec, DL))) {
      return hipErrorInvalidDeviceFunction;
    }
    if (evk1) (void)hipEventRecord(evk1, st);
  }
  DcmLaunch XD = DL;
  XD.grid = dim3(exact_blocks(n));
  XD.block = dim3(REFLECT_EXACT_BLOCK);
  {
    BarrierSerial one_at_a_time(st);
    if (fuse_tail)      // (the redo, if any, marks the real gb2 and makes the image from it)
      tu_exact0_dcm_redo_scr(XD, scr ? *scr : no_screen, scr && sb ? *sb : no_image,
                             ap ? *ap : none);
    else
      tu_exact0_dcm(XD);
  }
  if (tail && !fuse_tail) {
    if (ap) {
      for (int k = 0; k < ap->n; ++k) {
        const hipError_t ae = aperture_propagate_launch(ap->a[k], gb2, no_image, no_image, st);
        if (ae != hipSuccess) return ae;
      }
    }
    if (scr && sb) {
      const hipError_t se = screen_expose_launch(*scr, gb2, *sb, st);
      if (se != hipSuccess) return se;
    }
  }
  if (fused) *fused = (fuse_tail && scr && sb ? 1 : 0) | (fuse_tail && ap ? 8 : 0);
  if (force_exact && evk1) (void)hipEventRecord(evk1, st);
  if (ev1) (void)hipEventRecord(ev1, st);
  return hipGetLastError();
}

}  // namespace xrt
