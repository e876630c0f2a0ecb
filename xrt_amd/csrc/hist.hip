// Weighted 2-D histogram of a device-resident beam (the reduce step after the hot
// path in every run_ray_tracing iteration; xrt/multipro.py:111-177,
// raycing/__init__.py:170-300). One lane = one ray: two coalesced 8-B loads for
// the coordinates, the state and the J components for the weight, one fp64
// atomic add into the bin. Bin search = numpy's: estimate by scaling, then fix
// against the linspace edges so that rays on an edge land where np.histogram2d
// puts them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/xrt_hip.h"
#include "hist.h"

namespace xrt {

__device__ __forceinline__ int find_bin(double v, double lo, double hi, int bins) {
  if (!(v >= lo && v <= hi)) return -1;
  const double step = (hi - lo) / (double)bins;   // np.linspace: arange*step + start
  int b = (int)(((v - lo) / (hi - lo)) * (double)bins);
  if (b >= bins) b = bins - 1;
  if (b < 0) b = 0;
  // edges[j] = j*step + lo, edges[bins] = hi exactly
  auto edge = [&](int j) { return j == bins ? hi : (double)j * step + lo; };
  while (b > 0 && v < edge(b)) --b;
  while (b < bins - 1 && v >= edge(b + 1)) ++b;
  return b;
}

__global__ __launch_bounds__(256) void hist2d_kernel(
    xrt_hip_beam beam, const double* __restrict__ x, const double* __restrict__ y,
    double xf, double yf, int ray_flags, int flux_kind, double srcw, int bx, double xlo,
    double xhi, int by, double ylo, double yhi, double* __restrict__ hist,
    double* __restrict__ counters) {
  __shared__ double lds[8][4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i < beam.n) {
    const int st = beam.state[i];
    if (st > 0) c[3] = 1.;
    if (st == 1) c[4] = 1.;
    if (st == 2) c[5] = 1.;
    if (st == 3) c[6] = 1.;
    if (st < 0) c[7] = 1.;
    bool sel = false;
    if ((ray_flags & 1) && st == 1) sel = true;
    if ((ray_flags & 2) && st == 2) sel = true;
    if ((ray_flags & 4) && st == 3) sel = true;
    if ((ray_flags & 8) && st < 0) sel = true;
    if ((ray_flags & 16) && st > 0) sel = true;
    if (sel) {
      double w;
      if (flux_kind == 1)
        w = beam.Jss[i];
      else if (flux_kind == 2)
        w = beam.Jpp[i];
      else if (flux_kind == 3)
        w = 2. * beam.Jsp_ri[2 * i];
      else if (flux_kind == 4)
        w = 2. * beam.Jsp_ri[2 * i + 1];
      else if (flux_kind == 5)
        w = (beam.Jss[i] + beam.Jpp[i]) * beam.E[i] * 1.602176565e-19;
      else
        w = beam.Jss[i] + beam.Jpp[i];
      w *= srcw;
      c[0] = 1.;
      c[1] = w;
      const int ix = find_bin(x[i] * xf, xlo, xhi, bx);
      const int iy = find_bin(y[i] * yf, ylo, yhi, by);
      if (ix >= 0 && iy >= 0) {
        c[2] = w;
        atomicAdd(&hist[(int64_t)iy * bx + ix], w);
      }
    }
  }
  if (counters) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double v = c[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if ((threadIdx.x & 63) == 0) lds[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      const double v = lds[threadIdx.x][0] + lds[threadIdx.x][1] + lds[threadIdx.x][2] +
                       lds[threadIdx.x][3];
      if (v != 0.) atomicAdd(&counters[threadIdx.x], v);
    }
  }
}

// ---------------------------------------------------------------------------
// All histograms of one XYCPlot in one pass (multipro.py:316-361): the 2-D
// intensity histogram, its RGB twin colourised by the colour axis (hue = the
// normalised colour datum, saturation, value = flux; matplotlib's hsv_to_rgb),
// and the 1-D histograms of x, y and the colour datum, each with flux and RGB
// weights. The 1-D histograms are independent of the 2-D range, like the three
// separate np.histogram calls of the reference.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void hsv_to_rgb(double h, double s, double v, double& r, double& g,
                                           double& b) {
  const int i = (int)(h * 6.0);
  const double f = h * 6.0 - (double)i;
  const double p = v * (1.0 - s);
  const double q = v * (1.0 - s * f);
  const double t = v * (1.0 - s * (1.0 - f));
  switch (i % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
  if (s == 0.) r = g = b = v;
}

__global__ __launch_bounds__(256) void plot_hist_kernel(
    xrt_hip_beam beam, const double* __restrict__ x, const double* __restrict__ y,
    const double* __restrict__ cd, xrt_hip_plot P, double* __restrict__ h2,
    double* __restrict__ h2rgb, double* __restrict__ hx, double* __restrict__ hy,
    double* __restrict__ hc, double* __restrict__ counters) {
  __shared__ double lds[8][4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i < beam.n) {
    const int st = beam.state[i];
    if (st > 0) c[3] = 1.;
    if (st == 1) c[4] = 1.;
    if (st == 2) c[5] = 1.;
    if (st == 3) c[6] = 1.;
    if (st < 0) c[7] = 1.;
    bool sel = false;
    if ((P.ray_flags & 1) && st == 1) sel = true;
    if ((P.ray_flags & 2) && st == 2) sel = true;
    if ((P.ray_flags & 4) && st == 3) sel = true;
    if ((P.ray_flags & 8) && st < 0) sel = true;
    if ((P.ray_flags & 16) && st > 0) sel = true;
    if (sel) {
      double w;
      if (P.flux_kind == 1)
        w = beam.Jss[i];
      else if (P.flux_kind == 2)
        w = beam.Jpp[i];
      else if (P.flux_kind == 3)
        w = 2. * beam.Jsp_ri[2 * i];
      else if (P.flux_kind == 4)
        w = 2. * beam.Jsp_ri[2 * i + 1];
      else if (P.flux_kind == 5)
        w = (beam.Jss[i] + beam.Jpp[i]) * beam.E[i] * 1.602176565e-19;
      else
        w = beam.Jss[i] + beam.Jpp[i];
      w *= P.source_weight;
      c[0] = 1.;
      c[1] = w;
      const double cv = cd[i] * P.c_factor;
      double h01 = ((cv - P.c_lim[0]) * P.color_factor) / (P.c_lim[1] - P.c_lim[0]);
      if (h01 < 0.) h01 = 0.;
      if (h01 > 1.) h01 = 1.;
      double rgb[3];
      hsv_to_rgb(h01, P.color_saturation, w, rgb[0], rgb[1], rgb[2]);
      const int ix = find_bin(x[i] * P.x_factor, P.x_lim[0], P.x_lim[1], P.bins_x);
      const int iy = find_bin(y[i] * P.y_factor, P.y_lim[0], P.y_lim[1], P.bins_y);
      if (ix >= 0 && iy >= 0) {
        c[2] = w;
        const int64_t b = (int64_t)iy * P.bins_x + ix;
        atomicAdd(&h2[b], w);
        if (h2rgb)
          for (int k = 0; k < 3; ++k) atomicAdd(&h2rgb[3 * b + k], rgb[k]);
      }
      if (hx && ix >= 0) {
        atomicAdd(&hx[4 * ix], w);
        for (int k = 0; k < 3; ++k) atomicAdd(&hx[4 * ix + 1 + k], rgb[k]);
      }
      if (hy && iy >= 0) {
        atomicAdd(&hy[4 * iy], w);
        for (int k = 0; k < 3; ++k) atomicAdd(&hy[4 * iy + 1 + k], rgb[k]);
      }
      if (hc) {
        const int ic = find_bin(cv, P.c_lim[0], P.c_lim[1], P.bins_c);
        if (ic >= 0) {
          atomicAdd(&hc[4 * ic], w);
          for (int k = 0; k < 3; ++k) atomicAdd(&hc[4 * ic + 1 + k], rgb[k]);
        }
      }
    }
  }
  if (counters) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double v = c[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if ((threadIdx.x & 63) == 0) lds[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      const double v = lds[threadIdx.x][0] + lds[threadIdx.x][1] + lds[threadIdx.x][2] +
                       lds[threadIdx.x][3];
      if (v != 0.) atomicAdd(&counters[threadIdx.x], v);
    }
  }
}

hipError_t plot_hist_launch(const xrt_hip_beam& beam, const double* x, const double* y,
                            const double* c, const xrt_hip_plot& P, double* h2, double* h2rgb,
                            double* hx, double* hy, double* hc, double* counters,
                            hipStream_t st) {
  if (beam.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(plot_hist_kernel, dim3((unsigned)((beam.n + 255) / 256)), dim3(256), 0, st,
                     beam, x, y, c, P, h2, h2rgb, hx, hy, hc, counters);
  return hipGetLastError();
}

hipError_t hist2d_launch(const xrt_hip_beam& beam, const double* x, const double* y, double xf,
                         double yf, int ray_flags, int flux_kind, double srcw, int bx,
                         double xlo, double xhi, int by, double ylo, double yhi, double* hist,
                         double* counters, hipStream_t st) {
  if (beam.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(hist2d_kernel, dim3((unsigned)((beam.n + 255) / 256)), dim3(256), 0, st,
                     beam, x, y, xf, yf, ray_flags, flux_kind, srcw, bx, xlo, xhi, by, ylo, yhi,
                     hist, counters);
  return hipGetLastError();
}

}  // namespace xrt
