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

namespace xrt {

__global__ __launch_bounds__(256) void screen_expose_kernel(xrt_hip_screen S, xrt_hip_beam in,
                                                           xrt_hip_beam out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= in.n) return;
  const double gx = in.x[i] - S.center[0];
  const double gy = in.y[i] - S.center[1];
  const double gz = in.z[i] - S.center[2];
  const double ga = in.a[i], gb = in.b[i], gc = in.c[i];
  // sum(c*b for c, b in zip(basis, xyz)): ((0 + c0*x) + c1*y) + c2*z
  double x = (S.ex[0] * gx + S.ex[1] * gy) + S.ex[2] * gz;
  double y = (S.ey[0] * gx + S.ey[1] * gy) + S.ey[2] * gz;
  double z = (S.ez[0] * gx + S.ez[1] * gy) + S.ez[2] * gz;
  const double a = (S.ex[0] * ga + S.ex[1] * gb) + S.ex[2] * gc;
  const double b = (S.ey[0] * ga + S.ey[1] * gb) + S.ey[2] * gc;
  const double c = (S.ez[0] * ga + S.ez[1] * gb) + S.ez[2] * gc;
  double path = -y / b;
  int st = in.state[i];
  bool bad = isnan(path) || isinf(path);
  if (S.only_positive_path) bad = bad || (path < 0.);
  if (bad) {
    path = 0.;
    st = S.lost_num;
  }
  x = x + a * path;
  z = z + c * path;
  y = 0.;
  if (S.compress_x != 0.) x *= S.compress_x;
  if (S.compress_z != 0.) z *= S.compress_z;
  out.x[i] = x;
  out.y[i] = y;
  out.z[i] = z;
  out.a[i] = a;
  out.b[i] = b;
  out.c[i] = c;
  const double E = in.E[i];
  out.path[i] = in.path[i] + path;
  out.E[i] = E;
  out.Jss[i] = in.Jss[i];
  out.Jpp[i] = in.Jpp[i];
  reinterpret_cast<double2*>(out.Jsp_ri)[i] = reinterpret_cast<const double2*>(in.Jsp_ri)[i];
  out.state[i] = st;
  if (in.Es_ri) {
    // exp(1e7j * (E/CHBAR) * path), screens.py:271-274
    const double kCH = 6.626069573e-27 * 2.99792458e10 / 1.602176565e-12 * 1e8;
    const double kCHBAR = kCH / 6.283185307179586476925286766559;
    const double ph = (1e7 * (E / kCHBAR)) * path;
    double s, co;
    sincos_phase(ph, s, co);
    const double2 es = reinterpret_cast<const double2*>(in.Es_ri)[i];
    const double2 ep = reinterpret_cast<const double2*>(in.Ep_ri)[i];
    reinterpret_cast<double2*>(out.Es_ri)[i] =
        make_double2(es.x * co - es.y * s, es.x * s + es.y * co);
    reinterpret_cast<double2*>(out.Ep_ri)[i] =
        make_double2(ep.x * co - ep.y * s, ep.x * s + ep.y * co);
  }
}

hipError_t screen_expose_launch(const xrt_hip_screen& S, const xrt_hip_beam& in,
                                const xrt_hip_beam& out, hipStream_t st) {
  if (in.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(screen_expose_kernel, dim3((unsigned)((in.n + 255) / 256)), dim3(256), 0,
                     st, S, in, out);
  return hipGetLastError();
}

}  // namespace xrt
