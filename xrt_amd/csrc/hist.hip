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

// what one ray contributes to the histograms of a plot
struct PlotRay {
  int sel;             // selected by ray_flags
  int ix, iy, ic;      // bins (-1: outside the axis range)
  double w, rgb[3];
};

__device__ __forceinline__ void count_state(int st, double (&c)[8]) {
  if (st > 0) c[3] += 1.;
  if (st == 1) c[4] += 1.;
  if (st == 2) c[5] += 1.;
  if (st == 3) c[6] += 1.;
  if (st < 0) c[7] += 1.;
}

__device__ __forceinline__ PlotRay plot_ray(const xrt_hip_beam& beam, const double* x,
                                            const double* y, const double* cd,
                                            const xrt_hip_plot& P, int64_t i, int st,
                                            bool want_c) {
  PlotRay r;
  r.ix = r.iy = r.ic = -1;
  r.w = 0.;
  r.rgb[0] = r.rgb[1] = r.rgb[2] = 0.;
  bool sel = false;
  if ((P.ray_flags & 1) && st == 1) sel = true;
  if ((P.ray_flags & 2) && st == 2) sel = true;
  if ((P.ray_flags & 4) && st == 3) sel = true;
  if ((P.ray_flags & 8) && st < 0) sel = true;
  if ((P.ray_flags & 16) && st > 0) sel = true;
  r.sel = sel;
  if (!sel) return r;
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
  r.w = w;
  const double cv = cd[i] * P.c_factor;
  double h01 = ((cv - P.c_lim[0]) * P.color_factor) / (P.c_lim[1] - P.c_lim[0]);
  if (h01 < 0.) h01 = 0.;
  if (h01 > 1.) h01 = 1.;
  hsv_to_rgb(h01, P.color_saturation, w, r.rgb[0], r.rgb[1], r.rgb[2]);
  r.ix = find_bin(x[i] * P.x_factor, P.x_lim[0], P.x_lim[1], P.bins_x);
  r.iy = find_bin(y[i] * P.y_factor, P.y_lim[0], P.y_lim[1], P.bins_y);
  if (want_c) r.ic = find_bin(cv, P.c_lim[0], P.c_lim[1], P.bins_c);
  return r;
}

__device__ __forceinline__ void flush_counters(double (&c)[8], double* counters,
                                               double (*lds)[16]) {
  const int nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double v = c[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) lds[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    double v = 0.;
    for (int w = 0; w < nw; ++w) v += lds[threadIdx.x][w];
    if (v != 0.) atomicAdd(&counters[threadIdx.x], v);
  }
}

// General form: one fp64 global atomic per ray and histogram cell. `parts` selects
// the 2-D histograms (1) and/or the 1-D histograms with the counters (2): it serves
// whatever does not fit the LDS-privatised kernels below.
__global__ __launch_bounds__(256) void plot_hist_kernel(
    xrt_hip_beam beam, const double* __restrict__ x, const double* __restrict__ y,
    const double* __restrict__ cd, xrt_hip_plot P, double* __restrict__ h2,
    double* __restrict__ h2rgb, double* __restrict__ hx, double* __restrict__ hy,
    double* __restrict__ hc, double* __restrict__ counters, int parts) {
  __shared__ double lds[8][16];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i < beam.n) {
    const int st = beam.state[i];
    count_state(st, c);
    const PlotRay r = plot_ray(beam, x, y, cd, P, i, st, hc != nullptr);
    if (r.sel) {
      c[0] = 1.;
      c[1] = r.w;
      if (r.ix >= 0 && r.iy >= 0) {
        c[2] = r.w;
        if (parts & 1) {
          const int64_t b = (int64_t)r.iy * P.bins_x + r.ix;
          atomicAdd(&h2[b], r.w);
          if (h2rgb)
            for (int k = 0; k < 3; ++k) atomicAdd(&h2rgb[3 * b + k], r.rgb[k]);
        }
      }
      if (parts & 2) {
        if (hx && r.ix >= 0) {
          atomicAdd(&hx[4 * r.ix], r.w);
          for (int k = 0; k < 3; ++k) atomicAdd(&hx[4 * r.ix + 1 + k], r.rgb[k]);
        }
        if (hy && r.iy >= 0) {
          atomicAdd(&hy[4 * r.iy], r.w);
          for (int k = 0; k < 3; ++k) atomicAdd(&hy[4 * r.iy + 1 + k], r.rgb[k]);
        }
        if (hc && r.ic >= 0) {
          atomicAdd(&hc[4 * r.ic], r.w);
          for (int k = 0; k < 3; ++k) atomicAdd(&hc[4 * r.ic + 1 + k], r.rgb[k]);
        }
      }
    }
  }
  if (counters && (parts & 2)) flush_counters(c, counters, lds);
}

// ---------------------------------------------------------------------------
// LDS-privatised forms. With global atomics only, 1e7 rays issue 1.6e8 fp64 atomics,
// most of them onto the few hundred cells of the 1-D histograms: 27 ms measured, 35x
// the reflect pass that produced the beam. Here every block keeps its own copy of
// the cells in LDS (ds_add_f64), strides over the beam and adds its non-zero cells to
// the global arrays once.
//   plot_hist1d_lds: the three 1-D histograms (4 values per bin) + the counters;
//   plot_hist_lds: the fused form further down (2-D planes + the 1-D work of the first pass).
// ---------------------------------------------------------------------------
#define HIST_LDS_BUDGET (156 * 1024)

__global__ __launch_bounds__(256) void plot_hist1d_lds(
    xrt_hip_beam beam, const double* __restrict__ x, const double* __restrict__ y,
    const double* __restrict__ cd, xrt_hip_plot P, double* __restrict__ hx,
    double* __restrict__ hy, double* __restrict__ hc, double* __restrict__ counters) {
  extern __shared__ double cells[];     // [bx*4 | by*4 | bc*4]
  __shared__ double lds[8][16];
  const int nx = hx ? 4 * P.bins_x : 0, ny = hy ? 4 * P.bins_y : 0;
  const int nc = hc ? 4 * P.bins_c : 0;
  double* lx = cells;
  double* ly = cells + nx;
  double* lc = ly + ny;
  for (int k = threadIdx.x; k < nx + ny + nc; k += blockDim.x) cells[k] = 0.;
  __syncthreads();
  double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < beam.n; i += stride) {
    const int st = beam.state[i];
    count_state(st, c);
    const PlotRay r = plot_ray(beam, x, y, cd, P, i, st, hc != nullptr);
    if (!r.sel) continue;
    c[0] += 1.;
    c[1] += r.w;
    if (r.ix >= 0 && r.iy >= 0) c[2] += r.w;
    if (hx && r.ix >= 0) {
      atomicAdd(&lx[4 * r.ix], r.w);
      for (int k = 0; k < 3; ++k) atomicAdd(&lx[4 * r.ix + 1 + k], r.rgb[k]);
    }
    if (hy && r.iy >= 0) {
      atomicAdd(&ly[4 * r.iy], r.w);
      for (int k = 0; k < 3; ++k) atomicAdd(&ly[4 * r.iy + 1 + k], r.rgb[k]);
    }
    if (hc && r.ic >= 0) {
      atomicAdd(&lc[4 * r.ic], r.w);
      for (int k = 0; k < 3; ++k) atomicAdd(&lc[4 * r.ic + 1 + k], r.rgb[k]);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nx; k += blockDim.x)
    if (lx[k] != 0.) atomicAdd(&hx[k], lx[k]);
  for (int k = threadIdx.x; k < ny; k += blockDim.x)
    if (ly[k] != 0.) atomicAdd(&hy[k], ly[k]);
  for (int k = threadIdx.x; k < nc; k += blockDim.x)
    if (lc[k] != 0.) atomicAdd(&hc[k], lc[k]);
  if (counters) flush_counters(c, counters, lds);
}

// ---- one fused, LDS-privatised pass ----------------------------------------------------------
// A block (1024 lanes, one per CU) keeps `nch` of the four 2-D planes (flux, R, G, B; from
// `ch0` on) and, in the pass that has ch0 == 0 (if `counters` is given), the three 1-D
// histograms and the ray counters in LDS (ds_add_f64), strides over the beam and adds its
// non-zero cells to the global arrays once. A 128 x 128 plot is four passes (one 128-KB plane
// each; the first carries the 1-D work), a 64 x 64 plot one. Global fp64 atomics, the
// alternative, run at 2.4e10 per second on this chip whatever the table size, scope or
// distribution (tools/probes/probe_atomics.hip): the 4e7 updates of the 2-D planes alone
// would take 1.7 ms.
// The loop is unrolled four rays deep with every load issued before the first use: the version
// that fetched a ray's fields after looking at its state ran at 1.6 TB/s (two dependent trips
// per ray, 16 waves per CU).
struct RayData {
  int st;
  double x, y, c, jss, jpp, extra;
};
__device__ __forceinline__ RayData fetch_ray(const xrt_hip_beam& beam, const double* x,
                                             const double* y, const double* cd,
                                             const xrt_hip_plot& P, int64_t i) {
  RayData d;
  d.st = beam.state[i];
  d.x = x[i];
  d.y = y[i];
  d.c = cd[i];
  d.jss = beam.Jss[i];
  d.jpp = beam.Jpp[i];
  d.extra = 0.;
  if (P.flux_kind == 3)
    d.extra = beam.Jsp_ri[2 * i];
  else if (P.flux_kind == 4)
    d.extra = beam.Jsp_ri[2 * i + 1];
  else if (P.flux_kind == 5)
    d.extra = beam.E[i];
  return d;
}
// plot_ray on fetched data (same arithmetic)
__device__ __forceinline__ PlotRay eval_ray(const RayData& d, const xrt_hip_plot& P,
                                            bool want_c) {
  PlotRay r;
  r.ix = r.iy = r.ic = -1;
  r.w = 0.;
  r.rgb[0] = r.rgb[1] = r.rgb[2] = 0.;
  const int st = d.st;
  bool sel = false;
  if ((P.ray_flags & 1) && st == 1) sel = true;
  if ((P.ray_flags & 2) && st == 2) sel = true;
  if ((P.ray_flags & 4) && st == 3) sel = true;
  if ((P.ray_flags & 8) && st < 0) sel = true;
  if ((P.ray_flags & 16) && st > 0) sel = true;
  r.sel = sel;
  if (!sel) return r;
  double w;
  if (P.flux_kind == 1)
    w = d.jss;
  else if (P.flux_kind == 2)
    w = d.jpp;
  else if (P.flux_kind == 3 || P.flux_kind == 4)
    w = 2. * d.extra;
  else if (P.flux_kind == 5)
    w = (d.jss + d.jpp) * d.extra * 1.602176565e-19;
  else
    w = d.jss + d.jpp;
  w *= P.source_weight;
  r.w = w;
  const double cv = d.c * P.c_factor;
  double h01 = ((cv - P.c_lim[0]) * P.color_factor) / (P.c_lim[1] - P.c_lim[0]);
  if (h01 < 0.) h01 = 0.;
  if (h01 > 1.) h01 = 1.;
  hsv_to_rgb(h01, P.color_saturation, w, r.rgb[0], r.rgb[1], r.rgb[2]);
  r.ix = find_bin(d.x * P.x_factor, P.x_lim[0], P.x_lim[1], P.bins_x);
  r.iy = find_bin(d.y * P.y_factor, P.y_lim[0], P.y_lim[1], P.bins_y);
  if (want_c) r.ic = find_bin(cv, P.c_lim[0], P.c_lim[1], P.bins_c);
  return r;
}

#define HIST_UNROLL 4
__global__ __launch_bounds__(1024) void plot_hist_lds(
    xrt_hip_beam beam, const double* __restrict__ x, const double* __restrict__ y,
    const double* __restrict__ cd, xrt_hip_plot P, double* __restrict__ h2,
    double* __restrict__ h2rgb, int ch0, int nch, double* __restrict__ hx,
    double* __restrict__ hy, double* __restrict__ hc, double* __restrict__ counters,
    double* __restrict__ scratch) {
  extern __shared__ double cells[];     // [nch][by][bx] | bx*4 | by*4 | bc*4
  __shared__ double lds[8][16];
  const bool lines = ch0 == 0 && counters != nullptr;   // this pass carries the 1-D work
  const int plane = P.bins_x * P.bins_y;
  const int nx = lines && hx ? 4 * P.bins_x : 0, ny = lines && hy ? 4 * P.bins_y : 0;
  const int nc = lines && hc ? 4 * P.bins_c : 0;
  double* lx = cells + (int64_t)nch * plane;
  double* ly = lx + nx;
  double* lc = ly + ny;
  for (int k = threadIdx.x; k < nch * plane + nx + ny + nc; k += blockDim.x) cells[k] = 0.;
  __syncthreads();
  double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < beam.n;
       i0 += HIST_UNROLL * stride) {
    RayData d[HIST_UNROLL];
#pragma unroll
    for (int u = 0; u < HIST_UNROLL; ++u) {
      const int64_t i = i0 + u * stride;
      d[u] = fetch_ray(beam, x, y, cd, P, i < beam.n ? i : i0);
      if (i >= beam.n) d[u].st = 0;      // (state 0: counted nowhere, selected by no flag)
    }
#pragma unroll
    for (int u = 0; u < HIST_UNROLL; ++u) {
      if (lines) count_state(d[u].st, c);
      const PlotRay r = eval_ray(d[u], P, nc != 0);
      if (!r.sel) continue;
      if (lines) {
        c[0] += 1.;
        c[1] += r.w;
        if (r.ix >= 0 && r.iy >= 0) c[2] += r.w;
        if (nx && r.ix >= 0) {
          atomicAdd(&lx[4 * r.ix], r.w);
          for (int k = 0; k < 3; ++k) atomicAdd(&lx[4 * r.ix + 1 + k], r.rgb[k]);
        }
        if (ny && r.iy >= 0) {
          atomicAdd(&ly[4 * r.iy], r.w);
          for (int k = 0; k < 3; ++k) atomicAdd(&ly[4 * r.iy + 1 + k], r.rgb[k]);
        }
        if (nc && r.ic >= 0) {
          atomicAdd(&lc[4 * r.ic], r.w);
          for (int k = 0; k < 3; ++k) atomicAdd(&lc[4 * r.ic + 1 + k], r.rgb[k]);
        }
      }
      if (r.ix < 0 || r.iy < 0) continue;
      const int b = r.iy * P.bins_x + r.ix;
      for (int k = 0; k < nch; ++k) {
        const int ch = ch0 + k;
        const double v = ch == 0 ? r.w : r.rgb[ch - 1];
        if (v != 0.) atomicAdd(&cells[k * plane + b], v);
      }
    }
  }
  __syncthreads();
  if (scratch) {
    // the block's planes go to its slot of the scratch area as they are (coalesced stores);
    // plot_hist_reduce adds the slots up. 256 blocks flushing 16 384 cells each through
    // global atomics were 4e6 atomics = 0.17 ms of a 0.24-ms pass.
    double* slot = scratch + (int64_t)blockIdx.x * nch * plane;
    for (int b = threadIdx.x; b < nch * plane; b += blockDim.x) slot[b] = cells[b];
  } else {
    for (int k = 0; k < nch; ++k) {
      const int ch = ch0 + k;
      for (int b = threadIdx.x; b < plane; b += blockDim.x) {
        const double v = cells[k * plane + b];
        if (v == 0.) continue;
        if (ch == 0)
          atomicAdd(&h2[b], v);
        else
          atomicAdd(&h2rgb[3 * (int64_t)b + ch - 1], v);
      }
    }
  }
  for (int k = threadIdx.x; k < nx; k += blockDim.x)
    if (lx[k] != 0.) atomicAdd(&hx[k], lx[k]);
  for (int k = threadIdx.x; k < ny; k += blockDim.x)
    if (ly[k] != 0.) atomicAdd(&hy[k], ly[k]);
  for (int k = threadIdx.x; k < nc; k += blockDim.x)
    if (lc[k] != 0.) atomicAdd(&hc[k], lc[k]);
  if (lines) flush_counters(c, counters, lds);
}

// sums the per-block planes of one plot_hist_lds pass into the histograms: blockIdx.y takes
// one of HIST_REDUCE_PARTS groups of slots, one atomic per cell and group
#define HIST_REDUCE_PARTS 8
__global__ __launch_bounds__(256) void plot_hist_reduce(const double* __restrict__ scratch,
                                                        int nblocks, int plane, int ch0, int nch,
                                                        double* __restrict__ h2,
                                                        double* __restrict__ h2rgb) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nch * plane) return;
  const int per = (nblocks + HIST_REDUCE_PARTS - 1) / HIST_REDUCE_PARTS;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  double v = 0.;
  for (int blk = b0; blk < b1; ++blk) v += scratch[(int64_t)blk * nch * plane + j];
  if (v == 0.) return;
  const int k = j / plane, b = j - k * plane, ch = ch0 + k;
  if (ch == 0)
    atomicAdd(&h2[b], v);
  else
    atomicAdd(&h2rgb[3 * (int64_t)b + ch - 1], v);
}

hipError_t plot_hist_launch(const xrt_hip_beam& beam, const double* x, const double* y,
                            const double* c, const xrt_hip_plot& P, double* h2, double* h2rgb,
                            double* hx, double* hy, double* hc, double* counters,
                            hipStream_t st) {
  if (beam.n <= 0) return hipSuccess;
  const dim3 full((unsigned)((beam.n + 255) / 256));
  int general = 0;   // parts left to the global-atomics kernel
  const size_t b1 = sizeof(double) * 4 *
                    ((hx ? P.bins_x : 0) + (hy ? P.bins_y : 0) + (hc ? P.bins_c : 0));
  const size_t plane = sizeof(double) * (size_t)P.bins_x * (size_t)P.bins_y;
  const int nchan = h2rgb ? 4 : 1;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess)
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  unsigned lds_blocks = (unsigned)((beam.n + 1023) / 1024);
  if (lds_blocks > (unsigned)cus) lds_blocks = (unsigned)cus;   // one block per CU
  if (plane > 0 && plane <= HIST_LDS_BUDGET) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(plot_hist_lds),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       HIST_LDS_BUDGET);
    if (e != hipSuccess) return e;
  }
  // scratch for the per-block planes of a pass (stream-ordered allocation: the pool keeps it
  // between calls); without it the blocks flush through atomics
  double* scratch = nullptr;
  const size_t cells_pp = plane / sizeof(double);
  auto pass = [&](int ch0, int nch, size_t extra, bool lines) {
    hipLaunchKernelGGL(plot_hist_lds, dim3(lds_blocks), dim3(1024), nch * plane + extra, st, beam,
                       x, y, c, P, h2, h2rgb, ch0, nch, lines ? hx : nullptr,
                       lines ? hy : nullptr, lines ? hc : nullptr, lines ? counters : nullptr,
                       scratch);
    if (scratch) {
      const int total = (int)(nch * cells_pp);
      hipLaunchKernelGGL(plot_hist_reduce, dim3((total + 255) / 256, HIST_REDUCE_PARTS), dim3(256), 0, st, scratch,
                         (int)lds_blocks, (int)cells_pp, ch0, nch, h2, h2rgb);
    }
  };
  if (plane > 0 && plane <= HIST_LDS_BUDGET && lds_blocks > 8) {
    int most = (int)(HIST_LDS_BUDGET / plane);
    if (most > nchan) most = nchan;
    if (hipMallocAsync(reinterpret_cast<void**>(&scratch), (size_t)lds_blocks * most * plane,
                       st) != hipSuccess) {
      (void)hipGetLastError();
      scratch = nullptr;
    }
  }
  if (plane > 0 && plane + b1 <= HIST_LDS_BUDGET && counters) {
    // the fused passes: as many of the (flux, R, G, B) planes per pass as fit the LDS, the
    // 1-D histograms and the counters riding with the first
    for (int ch0 = 0; ch0 < nchan;) {
      const size_t extra = ch0 == 0 ? b1 : 0;
      int nch = (int)((HIST_LDS_BUDGET - extra) / plane);
      if (nch > nchan - ch0) nch = nchan - ch0;
      pass(ch0, nch, extra, true);
      ch0 += nch;
    }
    if (scratch) (void)hipFreeAsync(scratch, st);
    return hipGetLastError();
  }
  // 1-D histograms + counters
  if (b1 <= 48 * 1024) {
    unsigned blocks = full.x < 2048u ? full.x : 2048u;
    hipLaunchKernelGGL(plot_hist1d_lds, dim3(blocks), dim3(256), b1, st, beam, x, y, c, P, hx, hy,
                       hc, counters);
  } else {
    general |= 2;
  }
  // 2-D histograms: as many of the (flux, R, G, B) planes per pass as fit the LDS
  if (plane > 0 && plane <= HIST_LDS_BUDGET) {
    int per_pass = (int)(HIST_LDS_BUDGET / plane);
    if (per_pass > nchan) per_pass = nchan;
    for (int ch0 = 0; ch0 < nchan; ch0 += per_pass) {
      const int nch = ch0 + per_pass <= nchan ? per_pass : nchan - ch0;
      pass(ch0, nch, 0, false);
    }
  } else if (plane > 0) {
    general |= 1;
  }
  if (scratch) (void)hipFreeAsync(scratch, st);
  if (general)
    hipLaunchKernelGGL(plot_hist_kernel, full, dim3(256), 0, st, beam, x, y, c, P, h2, h2rgb, hx,
                       hy, hc, counters, general);
  return hipGetLastError();
}

hipError_t hist2d_launch(const xrt_hip_beam& beam, const double* x, const double* y, double xf,
                         double yf, int ray_flags, int flux_kind, double srcw, int bx,
                         double xlo, double xhi, int by, double ylo, double yhi, double* hist,
                         double* counters, hipStream_t st) {
  if (beam.n <= 0) return hipSuccess;
  const size_t plane = sizeof(double) * (size_t)bx * (size_t)by;
  if (plane > 0 && plane <= HIST_LDS_BUDGET) {
    // the flux plane of a plot without colour axis: same LDS-privatised kernels
    xrt_hip_plot P;
    P.x_factor = xf;
    P.y_factor = yf;
    P.c_factor = 0.;
    P.source_weight = srcw;
    P.x_lim[0] = xlo;
    P.x_lim[1] = xhi;
    P.y_lim[0] = ylo;
    P.y_lim[1] = yhi;
    P.c_lim[0] = 0.;
    P.c_lim[1] = 1.;
    P.color_factor = 0.;
    P.color_saturation = 0.;
    P.bins_x = bx;
    P.bins_y = by;
    P.bins_c = 1;
    P.ray_flags = ray_flags;
    P.flux_kind = flux_kind;
    return plot_hist_launch(beam, x, y, x, P, hist, nullptr, nullptr, nullptr, nullptr, counters,
                            st);
  }
  hipLaunchKernelGGL(hist2d_kernel, dim3((unsigned)((beam.n + 255) / 256)), dim3(256), 0, st,
                     beam, x, y, xf, yf, ray_flags, flux_kind, srcw, bx, xlo, xhi, by, ylo, yhi,
                     hist, counters);
  return hipGetLastError();
}

}  // namespace xrt
