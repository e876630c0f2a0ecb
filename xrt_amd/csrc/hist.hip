// Weighted histograms of a device-resident beam (the reduce step after the hot
// path in every run_ray_tracing iteration; xrt/multipro.py:111-177,
// raycing/__init__.py:170-300). Bin search = numpy's: estimate by scaling, then fix
// against the linspace edges so that rays on an edge land where np.histogram2d
// puts them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/xrt_hip.h"
#include "fp64_math.h"
#include "hist.h"
#include "kernarg.h"
#include "plot_tail.h"

namespace xrt {

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
                                            const xrt_hip_plot& P, const PlotAxes& A,
                                            int64_t i, int st, bool want_c) {
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
  double h01 = div_rn((cv - A.c.lo) * P.color_factor, A.c.hi - A.c.lo);
  if (h01 < 0.) h01 = 0.;
  if (h01 > 1.) h01 = 1.;
  hsv_to_rgb(h01, P.color_saturation, w, r.rgb[0], r.rgb[1], r.rgb[2]);
  r.ix = find_bin(x[i] * P.x_factor, A.x);
  r.iy = find_bin(y[i] * P.y_factor, A.y);
  if (want_c) r.ic = find_bin(cv, A.c);
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
    const double* __restrict__ cd, xrt_hip_plot P, PlotAxes A, double* __restrict__ h2,
    double* __restrict__ h2rgb, double* __restrict__ hx, double* __restrict__ hy,
    double* __restrict__ hc, double* __restrict__ counters, int parts) {
  __shared__ double lds[8][16];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i < beam.n) {
    const int st = beam.state[i];
    count_state(st, c);
    const PlotRay r = plot_ray(beam, x, y, cd, P, A, i, st, hc != nullptr);
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

// Small beams: ONE launch. The three-kernel route below costs 45-50 us whatever the beam
// (zeroing, writing and adding up 32 MB of plane copies; three dependent launches). Here the
// planes take global atomics (4 per ray inside the plot), the 1-D histograms are kept per block
// in LDS and flushed once. Same per-ray arithmetic (plot_ray), sums in another order. Measured
// (tools/probe_e2e_sizes.py, graph replays, same box): 2e3 rays 0.110 -> 0.091 ms per iteration;
// at 1e5 rays the atomics of a focused beam (many rays per cell) make it 5 us SLOWER than the
// three kernels (0.111 against 0.106): the limit is set below that.
#define HIST_SMALL_RAYS 32768
__global__ __launch_bounds__(256) void plot_hist_small(
    xrt_hip_beam beam, const double* __restrict__ x, const double* __restrict__ y,
    const double* __restrict__ cd, xrt_hip_plot P, PlotAxes A, double* __restrict__ h2,
    double* __restrict__ h2rgb, double* __restrict__ hx, double* __restrict__ hy,
    double* __restrict__ hc, double* __restrict__ counters) {
  extern __shared__ double cells[];      // [bx][4] | [by][4] | [bc][4]
  __shared__ double lds[8][16];
  const int nx = hx ? A.x.bins : 0, ny = hy ? A.y.bins : 0, nc = hc ? A.c.bins : 0;
  double* lx = cells;
  double* ly = lx + 4 * nx;
  double* lc = ly + 4 * ny;
  const int nl = 4 * (nx + ny + nc);
  for (int k = threadIdx.x; k < nl; k += blockDim.x) cells[k] = 0.;
  __syncthreads();
  double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < beam.n; i += stride) {
    const int st = beam.state[i];
    count_state(st, c);
    const PlotRay r = plot_ray(beam, x, y, cd, P, A, i, st, hc != nullptr);
    if (!r.sel) continue;
    c[0] += 1.;
    c[1] += r.w;
    if (r.ix >= 0 && r.iy >= 0) {
      c[2] += r.w;
      if (h2) {
        const int64_t b = (int64_t)r.iy * P.bins_x + r.ix;
        atomicAdd(&h2[b], r.w);
        if (h2rgb)
        {
          atomicAdd(&h2rgb[3 * b], r.rgb[0]);
          atomicAdd(&h2rgb[3 * b + 1], r.rgb[1]);
          atomicAdd(&h2rgb[3 * b + 2], r.rgb[2]);
        }
      }
    }
    if (nx && r.ix >= 0) {
      atomicAdd(&lx[4 * r.ix], r.w);
      atomicAdd(&lx[4 * r.ix + 1], r.rgb[0]);
      atomicAdd(&lx[4 * r.ix + 2], r.rgb[1]);
      atomicAdd(&lx[4 * r.ix + 3], r.rgb[2]);
    }
    if (ny && r.iy >= 0) {
      atomicAdd(&ly[4 * r.iy], r.w);
      atomicAdd(&ly[4 * r.iy + 1], r.rgb[0]);
      atomicAdd(&ly[4 * r.iy + 2], r.rgb[1]);
      atomicAdd(&ly[4 * r.iy + 3], r.rgb[2]);
    }
    if (nc && r.ic >= 0) {
      atomicAdd(&lc[4 * r.ic], r.w);
      atomicAdd(&lc[4 * r.ic + 1], r.rgb[0]);
      atomicAdd(&lc[4 * r.ic + 2], r.rgb[1]);
      atomicAdd(&lc[4 * r.ic + 3], r.rgb[2]);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nl; k += blockDim.x) {
    const double v = cells[k];
    if (v == 0.) continue;
    double* dst = k < 4 * nx ? hx + k : (k < 4 * (nx + ny) ? hy + (k - 4 * nx)
                                                           : hc + (k - 4 * (nx + ny)));
    atomicAdd(dst, v);
  }
  if (counters) flush_counters(c, counters, lds);
}

// ---------------------------------------------------------------------------
// LDS-privatised forms. Global fp64 atomics run at 2.4e10 per second on this chip whatever the
// table size, the scope of the atomic or the spread of the addresses (tools/probes/
// probe_atomics.hip): the 1.6e8 updates of a plot of 1e7 rays take 27 ms that way (round 2's
// first version), the 4e7 updates of the four 2-D planes (flux, R, G, B) alone 1.7 ms. So every
// cell is accumulated in LDS (ds_add_f64), per-block copies go to a stream-ordered scratch area
// as plain coalesced stores, and a small reduce kernel adds the copies up.
//
//   plot_hist_rays -- every ray read ONCE, in chunks of 1024 consecutive rays per block step,
//     loads four rays deep: weight, hue and the three bins of every ray; the ray counters; the
//     1-D histograms in LDS, laid out [weight][bin] so that the lanes of one ds_add_f64 spread
//     over the banks ([bin][weight] put them on 8 of 64); and
//       DIRECT  -- the 2-D planes fit the LDS beside them (64 x 64 bins): accumulated here, one
//                  1024-lane block per CU;
//       RECORDS -- they do not (the four planes of a 128 x 128 plot are 512 KB, of a 256 x 256
//                  plot 2 MB; a CU has 160 KB). The plot is cut into T rectangular TILES of bins
//                  whose planes do fit (64 x 64 bins: T = 4 / 16), and the chunk is SORTED BY
//                  TILE in LDS (count, prefix, scatter) and written out as (w, hue, cell within
//                  the tile) in that order, with the T + 1 run starts of the chunk -- 20 B per
//                  ray, coalesced, no atomics between blocks.
//   plot_hist_tiles -- block (tile, slice), 1024 lanes, one per CU: its waves walk through the
//     chunks of the slice, read the tile's run of each (contiguous: the runs of four chunks are
//     taken together so that all lanes have a ray), hue -> RGB, ds_add_f64 into the tile's planes.
//     Each ray is read by exactly one block; scanning every ray in every tile's block instead
//     cost T times the issue slots (1.25 ms at 256 x 256), a per-wave queue in front of the
//     work 0.4 ms.
//   plot_hist_reduce -- adds the copies up.
// ---------------------------------------------------------------------------
#define HIST_LDS_BUDGET (159 * 1024)
#define HIST_BLOCK 1024          // lanes of a block that owns a CU's LDS
#define HIST_CHUNK 1024          // rays per block step
#define HIST_MAX_TILES 64
#define HIST_MAX_RAYS 0x7fffffffll
#define HIST_REDUCE_PARTS 16

struct HistPlan {
  int ntx, nty;      // tiles along x, y
  int tx, ty;        // bins per tile
  int slices;        // chunk slices of plot_hist_tiles
  int nchan;         // 1 (flux) or 4 (flux, R, G, B)
  int lines;         // the 1-D histograms and counters are wanted
  int derive;        // the x and y histograms of the rays INSIDE the 2-D range are the column and
                     // row sums of the planes (same linspace edges, same find_bin): the reduce
                     // adds those; plot_hist_rays keeps the 1-D updates of the other rays only
};

// the sorted chunks: ray records in tile order and, per chunk, where each tile's run starts
struct HistRecords {
  double* w;
  double* hue;
  unsigned* cell;
  unsigned* start;   // [chunk][T + 2]: start[t] .. start[t + 1]; bucket T = rays outside the plot
  unsigned* counts;  // [block of plot_hist_rays][T]: its rays per tile
  int* share;        // [T + 1]: which blocks of plot_hist_tiles take which tile
  int nsrc;          // blocks of plot_hist_rays
};

// How the blocks of plot_hist_tiles are shared out among the tiles: every tile that has rays
// gets one block, the rest go in proportion to the rays (a focused beam puts ALL its rays into
// one or four of the 16 tiles of a 256 x 256 plot: with a fixed number of blocks per tile
// 1/16 or 1/4 of the chip did all the work). share[t] .. share[t + 1] are tile t's blocks.
// plot_hist_rays leaves its per-block ray counts per tile ([block][T], plain stores: atomics on
// a handful of shared addresses cost 8 ns EACH on this chip -- 13 000 of them made that kernel
// 100 us slower); every block of plot_hist_tiles adds them up for itself (48 KB out of L2) and
// derives the same table; block 0 leaves it for plot_hist_reduce.
__device__ __forceinline__ void make_tile_shares(const unsigned* __restrict__ counts, int nsrc,
                                                 int T, int nblocks, int64_t nchunks,
                                                 unsigned* tot, int* share, int* table_out) {
  if (threadIdx.x < T) tot[threadIdx.x] = 0;
  __syncthreads();
  {
    // thread i takes the counts i, i + blockDim, ...: consecutive threads, consecutive words
    unsigned acc = 0;
    const int n = nsrc * T;
    int t = threadIdx.x % T;               // (blockDim is a multiple of T only if T is a power
    for (int i = threadIdx.x; i < n; i += blockDim.x) {       // of two: keep it general)
      t = i % T;
      acc = counts[i];
      if (acc) atomicAdd(&tot[t], acc);
    }
  }
  __syncthreads();
  // what a tile costs: its rays, and the walk over the run table of EVERY chunk, which a wave
  // does four chunks at a time -- one round of loads, as many as 256 rays take. (Shared out by
  // rays alone, the tiles at the rim of a Gaussian footprint got one block each, which then
  // walked the 9766 chunks of 1e7 rays on its own: 260 us.)
  const double walk = 64. * (double)nchunks;
  double sum = 0.;
  int nonempty = 0;
  for (int t = 0; t < T; ++t) {
    if (tot[t]) {
      sum += (double)tot[t] + walk;
      ++nonempty;
    }
  }
  long long n = 0;
  if (threadIdx.x < T && tot[threadIdx.x]) {
    n = 1 + (long long)((double)(nblocks - nonempty) *
                        (((double)tot[threadIdx.x] + walk) / sum));
    if (n > nchunks) n = nchunks;                    // a slice takes whole chunks
  }
  __syncthreads();
  if (threadIdx.x < T) tot[threadIdx.x] = (unsigned)n;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < T; ++t) {
      share[t] = run;
      run += (int)tot[t];
    }
    // (each quotient is rounded down and the ones add up to `nonempty`: never above nblocks)
    share[T] = run < nblocks ? run : nblocks;
  }
  __syncthreads();
  if (table_out && (int)threadIdx.x <= T) table_out[threadIdx.x] = share[threadIdx.x];
}
__device__ __forceinline__ void load_tile_shares(const int* __restrict__ table, int T,
                                                 int* share) {
  if ((int)threadIdx.x <= T) share[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
}

enum { HIST_LINES_ONLY = 0, HIST_DIRECT = 1, HIST_RECORDS = 2 };

struct RayData {
  int st;
  double x, y, c, jss, jpp, extra;
};

struct HistRaysArgs {       // plot_hist_rays' arguments as one record (kernarg.h)
  xrt_hip_beam beam;
  const double *x, *y, *cd;
  xrt_hip_plot P;
  PlotAxes A;
  HistPlan H;
  double *counters, *plane_copies, *line_copies;
  HistRecords R;
};
// (only DIRECT needs the LDS of a whole CU: the others run as 256-lane blocks, several per CU)
template <int MODE>
#ifndef HIST_RAYS_PER_CU
#define HIST_RAYS_PER_CU 3      // blocks of plot_hist_rays per CU (RECORDS / LINES_ONLY)
#endif
#ifndef HIST_RAYS_MIN_BLOCKS
#define HIST_RAYS_MIN_BLOCKS 1  // second argument of its launch bounds
#endif
__global__ __launch_bounds__(MODE == HIST_DIRECT ? HIST_BLOCK : 256,
                             MODE == HIST_DIRECT ? 1 : HIST_RAYS_MIN_BLOCKS) void plot_hist_rays(
    HistRaysArgs G) {
  // One record of arguments. Reading them chunk by chunk where they are used (kernarg.h; -DHIST_LATE_ARGS)
  // takes 68 -> 3 SGPRs out of the VGPR lanes and 263 -> 6 lane moves out of the code, and LOSES: 156-157
  // against 148-150 us per 1e7 rays, same box (profiles/r06_sgpr_late_ab.txt) -- this kernel is bound by
  // its block barriers and the LDS sort, not by VALU issue, and the scalar loads of every chunk wait.
#ifndef HIST_LATE_ARGS
#define HIST_ARG(T, m) G.m
#else
#define HIST_ARG(T, m) kernarg_at<T>((unsigned)offsetof(HistRaysArgs, m))
#endif
  const PlotAxes& A = G.A;          // (here: the bin counts, the LDS layout)
  const HistPlan& H = G.H;
  const int64_t nrays = G.beam.n;
  const int flux_kind = G.P.flux_kind;
  constexpr int LANES = MODE == HIST_DIRECT ? HIST_BLOCK : 256;
  constexpr int U = HIST_CHUNK / LANES;     // rays per lane and step: 1 (DIRECT: depth comes from
                                            // the prefetch below) or 4
  extern __shared__ double cells[];   // DIRECT: [nchan][by][bx] | then [4][bx] [4][by] [4][bc] |
                                      // RECORDS: staging w, hue [1024], cell [1024]
  __shared__ double lds[8][16];
  __shared__ unsigned bucket[HIST_MAX_TILES + 2], first[HIST_MAX_TILES + 2];
  __shared__ unsigned tile_sum[HIST_MAX_TILES];
  if (MODE == HIST_RECORDS && threadIdx.x < HIST_MAX_TILES) tile_sum[threadIdx.x] = 0;
  const bool lines = H.lines != 0;
  const int plane = A.x.bins * A.y.bins;
  const int n2 = MODE == HIST_DIRECT ? H.nchan * plane : 0;
  const int nx = lines ? A.x.bins : 0, ny = lines ? A.y.bins : 0, nc = lines ? A.c.bins : 0;
  double* lx = cells + n2;
  double* ly = lx + 4 * nx;
  double* lc = ly + 4 * ny;
  const int ncells = n2 + 4 * (nx + ny + nc);
  double* stage_w = cells + ncells;
  double* stage_h = stage_w + HIST_CHUNK;
  unsigned* stage_c = reinterpret_cast<unsigned*>(stage_h + HIST_CHUNK);
  const int T = H.ntx * H.nty;
  for (int k = threadIdx.x; k < ncells; k += LANES) cells[k] = 0.;
  __syncthreads();
  int cn[6] = {0, 0, 0, 0, 0, 0};      // selected, alive, good, out, over, dead
  double cw = 0., cw_in = 0.;          // flux of the selected rays, of those inside the 2-D range
  const double crange = A.c.hi - A.c.lo;
  const int64_t nchunks = (nrays + HIST_CHUNK - 1) / HIST_CHUNK;

  auto fetch = [&](int64_t chunk, RayData (&d)[U]) {
    const xrt_hip_beam& beam = HIST_ARG(xrt_hip_beam, beam);
    const double* x = HIST_ARG(const double*, x);
    const double* y = HIST_ARG(const double*, y);
    const double* cd = HIST_ARG(const double*, cd);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = chunk * HIST_CHUNK + u * LANES + threadIdx.x;
      const int64_t j = i < nrays ? i : nrays - 1;
      d[u].st = __builtin_nontemporal_load(beam.state + j);
      d[u].x = __builtin_nontemporal_load(x + j);
      d[u].y = __builtin_nontemporal_load(y + j);
      d[u].c = __builtin_nontemporal_load(cd + j);
      d[u].jss = __builtin_nontemporal_load(beam.Jss + j);
      d[u].jpp = __builtin_nontemporal_load(beam.Jpp + j);
      d[u].extra = 0.;
      if (flux_kind == 3)
        d[u].extra = beam.Jsp_ri[2 * j];
      else if (flux_kind == 4)
        d[u].extra = beam.Jsp_ri[2 * j + 1];
      else if (flux_kind == 5)
        d[u].extra = beam.E[j];
      if (i >= nrays) d[u].st = 0;      // (state 0: counted nowhere, selected by no flag)
    }
  };

  RayData nxt[U];
  if ((int64_t)blockIdx.x < nchunks) fetch(blockIdx.x, nxt);
  for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    RayData d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) d[u] = nxt[u];
    // the next chunk is requested before this one is worked on
    fetch(chunk + gridDim.x < nchunks ? chunk + gridDim.x : chunk, nxt);
    unsigned tl[U], cell[U];
    double rw[U], rh[U];
    if (MODE == HIST_RECORDS) {
      if (threadIdx.x < T + 1) bucket[threadIdx.x] = 0;
      __syncthreads();
    }
    const xrt_hip_plot& P = HIST_ARG(xrt_hip_plot, P);
    const PlotAxes& A = HIST_ARG(PlotAxes, A);
    const HistPlan& H = HIST_ARG(HistPlan, H);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int st = d[u].st;
      if (lines) {
        cn[1] += st > 0;
        cn[2] += st == 1;
        cn[3] += st == 2;
        cn[4] += st == 3;
        cn[5] += st < 0;
      }
      tl[u] = T;
      cell[u] = 0;
      rw[u] = rh[u] = 0.;
      if (ray_selected(st, P.ray_flags)) {
        double w;
        if (P.flux_kind == 1)
          w = d[u].jss;
        else if (P.flux_kind == 2)
          w = d[u].jpp;
        else if (P.flux_kind == 3 || P.flux_kind == 4)
          w = 2. * d[u].extra;
        else if (P.flux_kind == 5)
          w = (d[u].jss + d[u].jpp) * d[u].extra * 1.602176565e-19;
        else
          w = d[u].jss + d[u].jpp;
        w *= P.source_weight;
        const double cv = d[u].c * P.c_factor;
        double h01 = div_rn((cv - A.c.lo) * P.color_factor, crange);
        if (h01 < 0.) h01 = 0.;
        if (h01 > 1.) h01 = 1.;
        const int ix = find_bin(d[u].x * P.x_factor, A.x);
        const int iy = find_bin(d[u].y * P.y_factor, A.y);
        const bool inside = ix >= 0 && iy >= 0;
        if (lines || MODE == HIST_DIRECT) {
          double r, g, b;
          hsv_to_rgb(h01, P.color_saturation, w, r, g, b);
          if (lines) {
            cn[0] += 1;
            cw += w;
            if (inside) cw_in += w;
            // (8 of the 16 ds_add_f64 per ray, 40-50 us per 1e7 rays, for sums the planes hold)
            const bool own_xy = !(H.derive && inside);
            if (own_xy && ix >= 0) {
              atomicAdd(&lx[ix], w);
              atomicAdd(&lx[nx + ix], r);
              atomicAdd(&lx[2 * nx + ix], g);
              atomicAdd(&lx[3 * nx + ix], b);
            }
            if (own_xy && iy >= 0) {
              atomicAdd(&ly[iy], w);
              atomicAdd(&ly[ny + iy], r);
              atomicAdd(&ly[2 * ny + iy], g);
              atomicAdd(&ly[3 * ny + iy], b);
            }
            const int ic = find_bin(cv, A.c);
            if (ic >= 0) {
              atomicAdd(&lc[ic], w);
              atomicAdd(&lc[nc + ic], r);
              atomicAdd(&lc[2 * nc + ic], g);
              atomicAdd(&lc[3 * nc + ic], b);
            }
          }
          if (MODE == HIST_DIRECT && inside) {
            const int bb = iy * A.x.bins + ix;
            if (w != 0.) atomicAdd(&cells[bb], w);
            if (H.nchan > 1) {
              if (r != 0.) atomicAdd(&cells[plane + bb], r);
              if (g != 0.) atomicAdd(&cells[2 * plane + bb], g);
              if (b != 0.) atomicAdd(&cells[3 * plane + bb], b);
            }
          }
        }
        if (MODE == HIST_RECORDS && inside) {
          const int tjx = ix / H.tx, tjy = iy / H.ty;
          tl[u] = (unsigned)(tjy * H.ntx + tjx);
          cell[u] = (unsigned)((iy - tjy * H.ty) * H.tx + (ix - tjx * H.tx));
          rw[u] = w;
          rh[u] = h01;
        }
      }
    }
    if (MODE == HIST_RECORDS) {
      // counting sort of the chunk by tile: rank within the bucket, bucket starts, scatter into
      // the staging arrays, coalesced copy out
      unsigned rank[U];
#pragma unroll
      for (int u = 0; u < U; ++u) rank[u] = atomicAdd(&bucket[tl[u]], 1u);
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int t = 0; t <= T; ++t) {
          first[t] = run;
          run += bucket[t];
        }
        first[T + 1] = run;
      }
      __syncthreads();
      const HistRecords& R = HIST_ARG(HistRecords, R);
      if (threadIdx.x < T + 2) R.start[chunk * (T + 2) + threadIdx.x] = first[threadIdx.x];
      if (threadIdx.x < T) tile_sum[threadIdx.x] += bucket[threadIdx.x];   // (its own thread's)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned pos = first[tl[u]] + rank[u];
        stage_w[pos] = rw[u];
        stage_h[pos] = rh[u];
        stage_c[pos] = cell[u];
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = u * LANES + threadIdx.x;
        const int64_t o = chunk * HIST_CHUNK + k;
        __builtin_nontemporal_store(stage_w[k], R.w + o);
        __builtin_nontemporal_store(stage_h[k], R.hue + o);
        __builtin_nontemporal_store(stage_c[k], R.cell + o);
      }
    }
  }
  __syncthreads();
  if (MODE == HIST_RECORDS && threadIdx.x < T)
    HIST_ARG(HistRecords, R).counts[(int64_t)blockIdx.x * T + threadIdx.x] = tile_sum[threadIdx.x];
  // this block's copies, as they are (coalesced stores): the reduce kernel adds the blocks up
  if (MODE == HIST_DIRECT) {
    double* out = G.plane_copies + (int64_t)blockIdx.x * n2;
    for (int k = threadIdx.x; k < n2; k += LANES) out[k] = cells[k];
  }
  if (lines) {
    const int nl = 4 * (nx + ny + nc);
    double* out = G.line_copies + (int64_t)blockIdx.x * nl;
    for (int k = threadIdx.x; k < nl; k += LANES) out[k] = lx[k];
    double* counters = G.counters;
    if (counters) {
      double c[8] = {(double)cn[0], cw,           cw_in,         (double)cn[1],
                     (double)cn[2], (double)cn[3], (double)cn[4], (double)cn[5]};
      flush_counters(c, counters, lds);
    }
  }
}

#define HIST_GROUP 4     // chunks whose runs a wave takes together
template <int NCH>
__global__ __launch_bounds__(HIST_BLOCK) void plot_hist_tiles(
    int64_t nchunks, HistRecords R, double saturation, PlotAxes A, HistPlan H,
    double* __restrict__ plane_copies) {
  extern __shared__ double cells[];     // [NCH][ty][tx]
  __shared__ int share[HIST_MAX_TILES + 1];
  __shared__ unsigned tot[HIST_MAX_TILES];
  const int T = H.ntx * H.nty;
#ifdef HIST_STATIC_SHARES
  if (threadIdx.x <= T) share[threadIdx.x] = threadIdx.x * ((int)gridDim.x / T);
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x <= T) R.share[threadIdx.x] = share[threadIdx.x];
#else
  make_tile_shares(R.counts, R.nsrc, T, (int)gridDim.x, nchunks, tot, share,
                   blockIdx.x == 0 ? R.share : nullptr);
#endif
  int tile = 0;
  while (tile < T && (int)blockIdx.x >= share[tile + 1]) ++tile;
  if (tile >= T) return;                // (more blocks than the tiles take)
  const int slice = (int)blockIdx.x - share[tile], slices = share[tile + 1] - share[tile];
  const int copy_slot = (int)blockIdx.x;
  const int tcells = H.tx * H.ty;
  for (int k = threadIdx.x; k < NCH * tcells; k += blockDim.x) cells[k] = 0.;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  // the chunks of the slice: slice, slice + S, ...; wave w takes them HIST_GROUP at a time
#ifdef HIST_SKIP_LOOP
  const int64_t mine = 0;
#else
  const int64_t mine = slice < nchunks ? (nchunks - slice + slices - 1) / slices : 0;
#endif
  // lanes 0..3 fetch the run of one chunk each (everybody gets all four below); the runs of the
  // NEXT group are requested before this group's records are
  auto runs_of = [&](int64_t j0, int64_t& c_l, unsigned& s_l, unsigned& e_l) {
    c_l = -1;
    s_l = e_l = 0;
    if (lane < HIST_GROUP && j0 + lane < mine) {
      c_l = slice + (j0 + lane) * slices;
      s_l = R.start[c_l * (T + 2) + tile];
      e_l = R.start[c_l * (T + 2) + tile + 1];
    }
  };
  int64_t c_l, c_n;
  unsigned s_l, e_l, s_n, e_n;
  runs_of((int64_t)wave * HIST_GROUP, c_l, s_l, e_l);
  for (int64_t j0 = (int64_t)wave * HIST_GROUP; j0 < mine; j0 += (int64_t)nwaves * HIST_GROUP) {
    runs_of(j0 + (int64_t)nwaves * HIST_GROUP, c_n, s_n, e_n);
    int64_t base[HIST_GROUP];
    int upto[HIST_GROUP];
    int total = 0;
    const int64_t safe = __shfl(c_l, 0) * HIST_CHUNK;   // what lanes without a ray read
#pragma unroll
    for (int g = 0; g < HIST_GROUP; ++g) {
      const int64_t c = __shfl(c_l, g);
      const unsigned s0 = __shfl(s_l, g), e0 = __shfl(e_l, g);
      base[g] = c * HIST_CHUNK + s0 - total;     // record index = base[g] + position in the group
      total += (int)(e0 - s0);
      upto[g] = total;
    }
    for (int i0 = 0; i0 < total; i0 += 64 * 4) {
      double w[4], hue[4];
      unsigned cell[4];
      bool on[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 64 + lane;
        on[u] = i < total;
        const int g = i < upto[0] ? 0 : i < upto[1] ? 1 : i < upto[2] ? 2 : 3;
        const int64_t k = on[u] ? base[g] + i : safe;
        w[u] = __builtin_nontemporal_load(R.w + k);
        hue[u] = __builtin_nontemporal_load(R.hue + k);
        cell[u] = __builtin_nontemporal_load(R.cell + k);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!on[u] || cell[u] >= (unsigned)tcells) continue;
        if (NCH > 1) {
          double r, g, b;
          hsv_to_rgb(hue[u], saturation, w[u], r, g, b);
          if (r != 0.) atomicAdd(&cells[tcells + cell[u]], r);
          if (g != 0.) atomicAdd(&cells[2 * tcells + cell[u]], g);
          if (b != 0.) atomicAdd(&cells[3 * tcells + cell[u]], b);
        }
        if (w[u] != 0.) atomicAdd(&cells[cell[u]], w[u]);
      }
    }
    c_l = c_n;
    s_l = s_n;
    e_l = e_n;
  }
  __syncthreads();
  // the tile as it is into this block's copy, [block][chan][ty][tx] (coalesced)
  double* out = plane_copies + (int64_t)copy_slot * NCH * tcells;
  for (int k = threadIdx.x; k < NCH * tcells; k += blockDim.x) out[k] = cells[k];
}

// ---------------------------------------------------------------------------
// plot_tail_tiles: the records a ray kernel wrote in its tail (plot_tail.h: every WAVE's 64
// rays sorted by tile, one row of a byte table per wave) added up per tile in LDS -- what
// plot_hist_tiles does for the 1024-ray chunks of plot_hist_rays. Buckets 0 .. T - 1 are the
// tiles of the 2-D histogram, bucket T holds the rays outside it (they still count for the
// 1-D histograms of the axis they are inside of, and for the colour histogram) and the rays the
// plot does not select. Block = (bucket, slice of the chunks); a wave takes 64 consecutive
// chunks at a time: lane l the run of chunk l, a prefix sum over the lanes gives every ray of
// the 64 runs a number, and lane j of a round finds its ray's chunk by a binary search over the
// 64 sums (LDS). The colour histogram is kept here as well (the pass has no LDS to keep it in);
// the x and y histograms of the rays inside the plot are the row and column sums of the planes
// (plot_hist_reduce, `derive`). Bucket T's blocks also add up the five state counts per wave.
// ---------------------------------------------------------------------------
#define TAIL_SAMPLE 2048
__device__ __forceinline__ void make_tail_shares(const PlotTail& Q, int64_t nchunks, int nblocks,
                                                 unsigned* tot, int* share) {
  const int T1 = Q.T + 1;
  if ((int)threadIdx.x < T1) tot[threadIdx.x] = 0;
  __syncthreads();
  // how the rays are spread over the buckets: from a sample of the waves (the shares only
  // balance the work; every bucket gets a block whether the sample saw a ray of it or not)
  const int64_t S = nchunks < TAIL_SAMPLE ? nchunks : TAIL_SAMPLE;
  const int64_t step = nchunks / S;
  {
    // thread = (bucket, stripe of the sampled waves): its own sum in a register, ONE addition
    // to the bucket's total at the end (an atomic per run and thread put every lane of a wave
    // on the same few words: 190 ns per instruction, 0.1 ms for a beam that fills all tiles)
    const int t = (int)threadIdx.x % T1, stripe = (int)threadIdx.x / T1;
    const int stripes = (int)blockDim.x / T1;
    unsigned mine = 0;
    if (stripe < stripes)
      for (int64_t k = stripe; k < S; k += stripes) {
        const unsigned char* row = Q.tab + k * step * Q.pitch;
        mine += (unsigned)row[t + 1] - (unsigned)row[t];
      }
    if (mine) atomicAdd(&tot[t], mine);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double walk = 0.5 * (double)S;      // a bucket's walk over the table, in rays
    double sum = 0.;
    for (int t = 0; t < T1; ++t) sum += (double)tot[t] + walk;
    const int rest = nblocks - T1;
    const int64_t most = (nchunks + 63) / 64;          // a slice takes 64 chunks at least
    int run = 0, biggest = 0;
    for (int t = 0; t < T1; ++t) {
      if (tot[t] > tot[biggest]) biggest = t;
      long long nb = 1 + (long long)((double)rest * (((double)tot[t] + walk) / sum));
      if (nb > most) nb = most;
      share[t] = run;
      run += (int)nb;
      tot[t] = (unsigned)nb;
    }
    share[T1] = run < nblocks ? run : nblocks;
  }
  __syncthreads();
}

template <int NCH>
__global__ __launch_bounds__(HIST_BLOCK) void plot_tail_tiles(
    int64_t nchunks, PlotTail Q, double* __restrict__ plane_copies,
    double* __restrict__ line_copies, int* __restrict__ share_out,
    double* __restrict__ counters) {
  extern __shared__ double cells[];     // [NCH][ty][tx] | [4][nx] [4][ny] [4][nc] | search tables
  __shared__ int share[HIST_MAX_TILES + 2];
  __shared__ unsigned tot[HIST_MAX_TILES + 1];
  __shared__ double lds[8][16];
  const int T = Q.T;
  make_tail_shares(Q, nchunks, (int)gridDim.x, tot, share);
  if (blockIdx.x == 0 && (int)threadIdx.x <= T) share_out[threadIdx.x] = share[threadIdx.x];
  int tile = 0;
  while (tile <= T && (int)blockIdx.x >= share[tile + 1]) ++tile;
  const bool idle = tile > T;           // (more blocks than the buckets take)
  const bool rest = tile == T;          // the rays outside the 2-D histogram
  const int tcells = Q.tx * Q.ty;
  const int nx = Q.A.x.bins, ny = Q.A.y.bins, nc = Q.A.c.bins;
  const int nl = 4 * (nx + ny + nc);
  const int nplane = rest || idle ? 0 : NCH * tcells;
  // a tile's block keeps the colour histogram only ([4][nc] behind its planes), the others the
  // three of them
  double* lx = cells + nplane;
  double* ly = lx + (nplane ? 0 : 4 * nx);
  double* lc = ly + (nplane ? 0 : 4 * ny);
  const int nlines = nplane ? 4 * nc : nl;
  for (int k = threadIdx.x; k < nplane + nlines; k += blockDim.x) cells[k] = 0.;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  int* pre = reinterpret_cast<int*>(cells + (nplane + nlines)) + wave * 128;   // [64] sums
  int* bas = pre + 64;                                                        // [64] bases
  __syncthreads();
  int n_sel = 0, cnt[5] = {0, 0, 0, 0, 0};
  double w_all = 0., w_in = 0.;
  const double crange = Q.A.c.hi - Q.A.c.lo;
  if (!idle) {
    const int slice = (int)blockIdx.x - share[tile], slices = share[tile + 1] - share[tile];
    int64_t per = (nchunks + slices - 1) / slices;
    per = (per + 63) / 64 * 64;
    const int64_t c0 = slice * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
    // rows a wave takes at a time: 64 (a lane each), or fewer when the slice is short -- a beam of
    // 1e5 rays is 1563 rows, 64 of them 4096 rays that ONE wave would add up while the other
    // fifteen of its block watch (70 us instead of 20); all 64 lanes work on the rays either way
    int G = 64;
    while (G > 4 && (c1 - c0) < (int64_t)nwaves * 2 * G) G >>= 1;
    // lane l takes the run of chunk g0 + l; the bytes of the NEXT 64 chunks are requested before
    // this group is worked on (a bucket with few rays is all walk: one dependent load per step
    // made a cold tile's block the last to finish, 0.3 ms at 1e7 rays)
    auto row_of = [&](int64_t c, unsigned& s_, unsigned& e_, unsigned (&k_)[5]) {
      s_ = e_ = 0;
#pragma unroll
      for (int k = 0; k < 5; ++k) k_[k] = 0;
      if (c < c1 && lane < G) {
        const unsigned char* row = Q.tab + c * Q.pitch;
        s_ = row[tile];
        e_ = row[tile + 1];
        if (rest) {
#pragma unroll
          for (int k = 0; k < 5; ++k) k_[k] = row[T + 2 + k];
        }
      }
    };
    // (four groups of 64 waves' rows are requested at a time: a bucket with few rays is all walk,
    // and one dependent byte load per 64 rows made a cold tile's blocks the last to finish)
    constexpr int AHEAD = 4;
    const int64_t hop = (int64_t)nwaves * G;
    unsigned sa[AHEAD], ea[AHEAD], ka[AHEAD][5], sb[AHEAD], eb[AHEAD], kb[AHEAD][5];
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) row_of(c0 + (int64_t)wave * G + q * hop + lane, sa[q], ea[q], ka[q]);
    for (int64_t gq = c0 + (int64_t)wave * G; gq < c1; gq += AHEAD * hop) {
#pragma unroll
      for (int q = 0; q < AHEAD; ++q) row_of(gq + (AHEAD + q) * hop + lane, sb[q], eb[q], kb[q]);
#pragma unroll
      for (int q = 0; q < AHEAD; ++q) {
      const int64_t g0 = gq + q * hop;
      if (g0 >= c1) break;
      const unsigned s0 = sa[q], e0 = ea[q];
      const unsigned (&k0)[5] = ka[q];
      const int64_t c = g0 + lane;
#pragma unroll
      for (int k = 0; k < 5; ++k) cnt[k] += (int)k0[k];
      const int len = (int)(e0 - s0);
      if (__ballot(len != 0) != 0ull) {
        int incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int v = __shfl_up(incl, off);
          if (lane >= off) incl += v;
        }
        const int total = __shfl(incl, 63);
        // every run whole (a focused beam: all rays of these waves in this tile): ray j is
        // record g0 * 64 + j, nothing to search
        const bool whole = total == 64 * G;
        if (!whole) {
          pre[lane] = incl;
          bas[lane] = (int)(c * 64 + s0) - (incl - len);   // record = bas[chunk] + number
        }
        const int safe = (int)(g0 * 64);                   // what lanes without a ray read
        for (int j0 = 0; j0 < total; j0 += 256) {
          double w[4], hue[4];
          unsigned word[4];
          bool on[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * 64 + lane;
            on[u] = j < total;
            int64_t k = safe + j;
            if (!whole) {
              int g = 0;
#pragma unroll
              for (int stp = 32; stp > 0; stp >>= 1)
                if (pre[g + stp - 1] <= j) g += stp;
              k = on[u] ? (int64_t)(bas[g] + j) : (int64_t)safe;
            }
            w[u] = Q.w[k];
            hue[u] = Q.hue[k];
            word[u] = Q.word[k];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (!on[u]) continue;
            if (rest && !(word[u] & PLOT_TAIL_SELECTED)) continue;
            // hue and colour bin from the colour datum (plot_hist_rays' arithmetic)
            const double cv = hue[u];
            double h01 = div_rn((cv - Q.A.c.lo) * Q.P.color_factor, crange);
            if (h01 < 0.) h01 = 0.;
            if (h01 > 1.) h01 = 1.;
            const int ic = Q.want_c ? find_bin(cv, Q.A.c) : -1;
            double r, g, b;
            hsv_to_rgb(h01, Q.P.color_saturation, w[u], r, g, b);
            n_sel += 1;
            w_all += w[u];
            if (!rest) {
              w_in += w[u];
              const unsigned cell = word[u];
#ifndef TAIL_AB_NO_PLANES
              if (cell < (unsigned)tcells) {
                if (w[u] != 0.) atomicAdd(&cells[cell], w[u]);
                if (NCH > 1) {
                  if (r != 0.) atomicAdd(&cells[tcells + cell], r);
                  if (g != 0.) atomicAdd(&cells[2 * tcells + cell], g);
                  if (b != 0.) atomicAdd(&cells[3 * tcells + cell], b);
                }
              }
#endif
            } else {
              const int ix = (int)(word[u] & 0x7ffu) - 1, iy = (int)((word[u] >> 11) & 0x7ffu) - 1;
              if (ix >= 0) {
                atomicAdd(&lx[ix], w[u]);
                atomicAdd(&lx[nx + ix], r);
                atomicAdd(&lx[2 * nx + ix], g);
                atomicAdd(&lx[3 * nx + ix], b);
              }
              if (iy >= 0) {
                atomicAdd(&ly[iy], w[u]);
                atomicAdd(&ly[ny + iy], r);
                atomicAdd(&ly[2 * ny + iy], g);
                atomicAdd(&ly[3 * ny + iy], b);
              }
            }
#ifndef TAIL_AB_NO_CLINES
            if (ic >= 0) {
              atomicAdd(&lc[ic], w[u]);
              atomicAdd(&lc[nc + ic], r);
              atomicAdd(&lc[2 * nc + ic], g);
              atomicAdd(&lc[3 * nc + ic], b);
            }
#endif
          }
        }
      }
      }
#pragma unroll
      for (int q = 0; q < AHEAD; ++q) {
        sa[q] = sb[q];
        ea[q] = eb[q];
#pragma unroll
        for (int k = 0; k < 5; ++k) ka[q][k] = kb[q][k];
      }
    }
  }
  __syncthreads();
  // this block's copies as they are: the tile [chan][ty][tx], the lines [4][nx] [4][ny] [4][nc]
  if (nplane) {
    double* out = plane_copies + (int64_t)blockIdx.x * NCH * tcells;
    for (int k = threadIdx.x; k < nplane; k += blockDim.x) out[k] = cells[k];
  }
  {
    double* out = line_copies + (int64_t)blockIdx.x * nl;
    const int head = nplane ? 4 * (nx + ny) : 0;       // (a tile's block: zeros for x and y)
    for (int k = threadIdx.x; k < nl; k += blockDim.x)
      out[k] = k < head ? 0. : lx[k - head];
  }
  if (counters) {
    double c[8] = {(double)n_sel, w_all, w_in, (double)cnt[0], (double)cnt[1], (double)cnt[2],
                   (double)cnt[3], (double)cnt[4]};
    flush_counters(c, counters, lds);
  }
}

// Adds the copies up into the histograms, one launch: blocks [0, nb2) take the 2-D planes
// ([copy][chan][by][bx] -> h2, h2rgb), the rest the 1-D histograms ([copy][ [4][bx] | [4][by] |
// [4][bc] ] -> h[bin][4]); blockIdx.y takes one of gridDim.y groups of copies (one atomic per
// cell and group), four independent sums per thread so that the loads overlap.
__device__ __forceinline__ double sum_copies(const double* __restrict__ p, int64_t pitch, int s0,
                                             int s1) {
  double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
  int sl = s0;
  for (; sl + 3 < s1; sl += 4) {
    a0 += p[(int64_t)sl * pitch];
    a1 += p[(int64_t)(sl + 1) * pitch];
    a2 += p[(int64_t)(sl + 2) * pitch];
    a3 += p[(int64_t)(sl + 3) * pitch];
  }
  for (; sl < s1; ++sl) a0 += p[(int64_t)sl * pitch];
  // (sixteen sums instead of four: no difference, 24.9 against 23.6 us at 1e7 rays)
  return (a0 + a1) + (a2 + a3);
}
__global__ __launch_bounds__(256) void plot_hist_reduce(
    const double* __restrict__ planes, int ncopies, int plane, int nchan, int nb2,
    const double* __restrict__ lines, int nline_copies, int nx, int ny, int nc,
    double* __restrict__ h2, double* __restrict__ h2rgb, double* __restrict__ hx,
    double* __restrict__ hy, double* __restrict__ hc, HistPlan H, int bins_x, int bins_y,
    const int* __restrict__ tile_share, int plane_parts, int nbl, int line_parts) {
  __shared__ int share[HIST_MAX_TILES + 1];
  // a 1-D grid of the blocks that have work: nb2 * plane_parts for the planes, then nbl *
  // line_parts for the lines (as (nb2 + nbl) x 16 blocks, 15 of 16 plane blocks of a 256 x 256
  // plot -- plane_parts = 1 -- came and went: 16576 blocks for the work of 1216, ~10 us of every plot)
  const int plane_blocks = nb2 * plane_parts;
  const bool for_planes = (int)blockIdx.x < plane_blocks;
  const int rest = (int)blockIdx.x - plane_blocks;
  const int block_x = for_planes ? (int)blockIdx.x % (nb2 > 0 ? nb2 : 1)
                                 : nb2 + rest % (nbl > 0 ? nbl : 1);
  const int part = for_planes ? (int)blockIdx.x / (nb2 > 0 ? nb2 : 1) : rest / (nbl > 0 ? nbl : 1);
  const bool tiled = tile_share != nullptr;     // copies of plot_hist_tiles: [block][chan][ty][tx]
  if (tiled && for_planes) load_tile_shares(tile_share, H.ntx * H.nty, share);
  if (for_planes) {
    // block = (channel, 8 rows, 32 columns) of the plot: 256-B row segments of every copy, and
    // with H.derive the patch's column and row sums for the 1-D histograms of x and y
    __shared__ double scol[32], srow[8];
    const int nbx = (bins_x + 31) / 32, nby = (bins_y + 7) / 8;
    int q = block_x;
    const int ibx = q % nbx;
    q /= nbx;
    const int iby = q % nby, ch = q / nby;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int bx = ibx * 32 + cx, by = iby * 8 + ry;
    if (H.derive) {
      if (threadIdx.x < 32) scol[threadIdx.x] = 0.;
      if (threadIdx.x < 8) srow[threadIdx.x] = 0.;
      __syncthreads();
    }
    double v = 0.;
    if (bx < bins_x && by < bins_y && ch < nchan) {
      const int b = by * bins_x + bx;
      int c0 = 0, cn = ncopies;
      int64_t pitch = (int64_t)nchan * plane;
      const double* src = planes + (int64_t)ch * plane + b;
      if (tiled) {
        const int tjx = bx / H.tx, tjy = by / H.ty;
        const int t = tjy * H.ntx + tjx, tcells = H.tx * H.ty;
        c0 = share[t];
        cn = share[t + 1];
        pitch = (int64_t)nchan * tcells;
        src = planes + (int64_t)ch * tcells + (by - tjy * H.ty) * H.tx + (bx - tjx * H.tx);
      }
      // (plane_parts groups of copies per cell; ONE group for the larger plots: every cell then
      // has one thread and adds without an atomic -- at 2.4e10 global atomics per second the 16
      // groups of a 256 x 256 plot cost 44 us)
      const int per = (cn - c0 + plane_parts - 1) / plane_parts;
      const int s0 = c0 + part * per, s1 = min(cn, s0 + per);
      if (s0 < s1) v = sum_copies(src, pitch, s0, s1);
      if (v != 0.) {
        double* dst = ch == 0 ? &h2[b] : &h2rgb[3 * (int64_t)b + ch - 1];
        if (plane_parts == 1)
          *dst += v;
        else
          atomicAdd(dst, v);
      }
    }
    if (H.derive) {
      if (v != 0.) {
        atomicAdd(&scol[cx], v);
        atomicAdd(&srow[ry], v);
      }
      __syncthreads();
      if (threadIdx.x < 32) {
        if (hx && bx < bins_x && scol[cx] != 0.) atomicAdd(&hx[4 * bx + ch], scol[cx]);
      } else if (threadIdx.x < 40) {
        const int r = threadIdx.x - 32, yy = iby * 8 + r;
        if (hy && yy < bins_y && srow[r] != 0.) atomicAdd(&hy[4 * yy + ch], srow[r]);
      }
    }
    return;
  }
  const int nl = 4 * (nx + ny + nc);
  const int j = (block_x - nb2) * blockDim.x + threadIdx.x;
  if (j >= nl) return;
  const int per = (nline_copies + line_parts - 1) / line_parts;
  const int s0 = part * per, s1 = min(nline_copies, s0 + per);
  const double v = sum_copies(lines + j, nl, s0, s1);
  if (v == 0.) return;
  if (j < 4 * nx) {
    if (hx) atomicAdd(&hx[4 * (j % nx) + j / nx], v);
  } else if (j < 4 * (nx + ny)) {
    const int k = j - 4 * nx;
    if (hy) atomicAdd(&hy[4 * (k % ny) + k / ny], v);
  } else {
    const int k = j - 4 * (nx + ny);
    if (hc) atomicAdd(&hc[4 * (k % nc) + k / nc], v);
  }
}

// The scratch of a call comes from the device's stream-ordered pool, which by default hands its
// memory back at every synchronisation: a plot per iteration would allocate a quarter of a
// gigabyte from the driver each time (0.15 ms). Once per device: keep it.
static void keep_pool_memory(int dev) {
  static bool done[64] = {};
  if (dev < 0 || dev >= 64 || done[dev]) return;
  done[dev] = true;
  hipMemPool_t pool = nullptr;
  if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
    uint64_t keep = UINT64_MAX;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
  }
  (void)hipGetLastError();
}

// the smallest number of tiles whose planes fit `budget` bytes of LDS
static bool plan_tiles(int bx, int by, int nchan, size_t budget, int max_tiles, HistPlan& H) {
  int best = 0;
  for (int a = 1; a <= max_tiles; ++a)
    for (int b = 1; a * b <= max_tiles; ++b) {
      const int tx = (bx + a - 1) / a, ty = (by + b - 1) / b;
      if ((size_t)tx * ty * nchan * sizeof(double) > budget || tx * ty > 0xfffff) continue;
      // fewer tiles first; then rows as long as possible (coalesced flush)
      if (!best || a * b < best || (a * b == best && a < H.ntx)) {
        best = a * b;
        H.ntx = a;
        H.nty = b;
        H.tx = tx;
        H.ty = ty;
      }
    }
  return best != 0;
}

// *ws* / *ws_bytes*: scratch of the caller (size from plot_hist_scratch_bytes), used in stream
// order; NULL or too small: taken from the device's stream-ordered pool for this call. With
// *need* set nothing is launched: the size a call with these arguments would use is returned.
hipError_t plot_hist_launch(const xrt_hip_beam& beam, const double* x, const double* y,
                            const double* c, const xrt_hip_plot& P, double* h2, double* h2rgb,
                            double* hx, double* hy, double* hc, double* counters,
                            hipStream_t st, void* ws, size_t ws_bytes, size_t* need) {
  if (need) *need = 0;
  if (beam.n <= 0) return hipSuccess;
  PlotAxes A;
  A.x = axis_bins(P.x_lim[0], P.x_lim[1], P.bins_x);
  A.y = axis_bins(P.y_lim[0], P.y_lim[1], P.bins_y);
  A.c = axis_bins(P.c_lim[0], P.c_lim[1], P.bins_c > 0 ? P.bins_c : 1);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess)
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (!need && !ws) keep_pool_memory(dev);
  const bool want_lines = hx || hy || hc || counters;
  // (the kernel keeps all three 1-D histograms; absent ones are dropped by the reduce)
  const size_t b1 = sizeof(double) * 4 * ((size_t)A.x.bins + A.y.bins + A.c.bins);
  const size_t plane = sizeof(double) * (size_t)P.bins_x * (size_t)P.bins_y;
  const size_t stage = HIST_CHUNK * (8 + 8 + 4);
  HistPlan H = {};
  H.nchan = h2rgb ? 4 : 1;
  H.ntx = H.nty = 1;
  H.tx = P.bins_x;
  H.ty = P.bins_y;
  const bool fits = beam.n < HIST_MAX_RAYS;
  const bool lines = fits && want_lines && b1 <= 64 * 1024;
  H.lines = lines;
  H.derive = 0;
  int mode = HIST_LINES_ONLY;
  if (h2 && fits) {
    if (H.nchan * plane + (lines ? b1 : 0) <= HIST_LDS_BUDGET - 2048)
      mode = HIST_DIRECT;
    else if (plan_tiles(P.bins_x, P.bins_y, H.nchan, HIST_LDS_BUDGET,
                        cus < HIST_MAX_TILES ? cus : HIST_MAX_TILES, H))
      mode = HIST_RECORDS;
  }
  // (the 1-D weights are flux, R, G, B: derivable when all four planes are made)
  H.derive = lines && mode != HIST_LINES_ONLY && H.nchan == 4;
  int general = (h2 && mode == HIST_LINES_ONLY ? 1 : 0) | (want_lines && !lines ? 2 : 0);
  // small beams: the one-launch form (the caller's scratch is not needed: *need stays 0)
  static const bool no_small = getenv("XRT_HIP_HIST_NO_SMALL") != nullptr;
  if (beam.n <= HIST_SMALL_RAYS && b1 <= 64 * 1024 && !no_small) {
    if (need) return hipSuccess;
    const int64_t want = (beam.n + 1023) / 1024;
    const unsigned nblk = (unsigned)(want < cus ? want : cus);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(plot_hist_small),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)b1);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(plot_hist_small, dim3(nblk), dim3(256), b1, st, beam, x, y, c, P, A, h2,
                       h2rgb, hx, hy, hc, counters);
    return hipGetLastError();
  }
  if (mode != HIST_LINES_ONLY || lines) {
    const int64_t chunks = (beam.n + HIST_CHUNK - 1) / HIST_CHUNK;
    const int64_t most = (int64_t)cus * (mode == HIST_DIRECT ? 1 : HIST_RAYS_PER_CU);
    const int nblk = (int)(chunks < most ? chunks : most);        // plot_hist_rays: fills the CUs
    const int T = H.ntx * H.nty;
    int ncopies = nblk;                                           // copies of the planes
    if (mode == HIST_RECORDS) {
      // plot_hist_tiles: one block per CU, shared out among the tiles by their ray counts
      // (tile_shares); every block leaves a copy of ITS tile
      // (fewer blocks for small beams -- less to zero, write and reduce -- was tried: at 1e5
      // rays 55 blocks took 29.8 us where 256 take 18.4; the launch is bound by the latency of
      // a block's phases, not by the 32 MB of copies)
      ncopies = cus > T ? cus : T;
      H.slices = ncopies;
    }
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t nl = b1 / sizeof(double);
    const size_t planes_b = pad(mode == HIST_LINES_ONLY ? 0
                                : mode == HIST_RECORDS
                                    ? (size_t)ncopies * H.nchan * H.tx * H.ty * sizeof(double)
                                    : (size_t)ncopies * H.nchan * plane);
    const size_t lines_b = pad(lines ? (size_t)nblk * b1 : 0);
    const size_t recs = mode == HIST_RECORDS ? (size_t)chunks * HIST_CHUNK : 0;
    const size_t start_b = pad(mode == HIST_RECORDS ? (size_t)chunks * (T + 2) * 4 : 0);
    const size_t counts_b = pad(mode == HIST_RECORDS ? (size_t)nblk * T * 4 : 0);
    const size_t scratch_b =
        planes_b + lines_b + pad(recs * 8) * 2 + pad(recs * 4) + start_b + counts_b + 1024;
    if (need) {
      *need = scratch_b + 256;       // (+ alignment of the caller's pointer)
      return hipSuccess;
    }
    char* scratch = nullptr;
    bool own = false;
    if (ws) {
      char* aligned = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 255) / 256 * 256);
      if (aligned + scratch_b <= static_cast<char*>(ws) + ws_bytes) scratch = aligned;
    }
    if (!scratch) {
      own = true;
      if (hipMallocAsync(reinterpret_cast<void**>(&scratch), scratch_b, st) != hipSuccess) {
        (void)hipGetLastError();
        scratch = nullptr;
      }
    }
    if (scratch) {
      char* q = scratch;
      double* plane_copies = reinterpret_cast<double*>(q);
      q += planes_b;
      double* line_copies = reinterpret_cast<double*>(q);
      q += lines_b;
      HistRecords R;
      R.w = reinterpret_cast<double*>(q);
      q += pad(recs * 8);
      R.hue = reinterpret_cast<double*>(q);
      q += pad(recs * 8);
      R.cell = reinterpret_cast<unsigned*>(q);
      q += pad(recs * 4);
      R.start = reinterpret_cast<unsigned*>(q);
      q += start_b;
      R.counts = reinterpret_cast<unsigned*>(q);
      q += counts_b;
      R.share = reinterpret_cast<int*>(q);
      R.nsrc = nblk;
      hipError_t e = hipSuccess;
      const size_t lds1 = (mode == HIST_DIRECT ? H.nchan * plane : 0) + (lines ? b1 : 0) +
                          (mode == HIST_RECORDS ? stage : 0);
      auto rays = mode == HIST_DIRECT    ? plot_hist_rays<HIST_DIRECT>
                  : mode == HIST_RECORDS ? plot_hist_rays<HIST_RECORDS>
                                         : plot_hist_rays<HIST_LINES_ONLY>;
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(rays),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
      if (e != hipSuccess) {
        if (own) (void)hipFreeAsync(scratch, st);
        return e;
      }
      HistRaysArgs G;
      G.beam = beam;
      G.x = x;
      G.y = y;
      G.cd = c;
      G.P = P;
      G.A = A;
      G.H = H;
      G.counters = counters;
      G.plane_copies = plane_copies;
      G.line_copies = line_copies;
      G.R = R;
      hipLaunchKernelGGL(rays, dim3((unsigned)nblk), dim3(mode == HIST_DIRECT ? HIST_BLOCK : 256),
                         lds1, st, G);
      if (mode == HIST_RECORDS) {
        auto tiles = H.nchan > 1 ? plot_hist_tiles<4> : plot_hist_tiles<1>;
        const size_t lds2 = sizeof(double) * (size_t)H.nchan * H.tx * H.ty;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(tiles),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) {
          if (own) (void)hipFreeAsync(scratch, st);
          return e;
        }
        hipLaunchKernelGGL(tiles, dim3((unsigned)ncopies), dim3(HIST_BLOCK), lds2, st, chunks, R,
                           P.color_saturation, A, H, plane_copies);
      }
      {
        const int total = mode != HIST_LINES_ONLY ? H.nchan * P.bins_x * P.bins_y : 0;
        const int nb2 =
            total ? H.nchan * ((P.bins_x + 31) / 32) * ((P.bins_y + 7) / 8) : 0;   // 8 x 32 patches
        const int nbl = lines && (hx || hy || hc) ? (int)((nl + 255) / 256) : 0;
        const int plane_parts = total >= 32768 ? 1 : HIST_REDUCE_PARTS;
        if (nb2 + nbl > 0)
          hipLaunchKernelGGL(plot_hist_reduce,
                             dim3((unsigned)(nb2 * plane_parts + nbl * HIST_REDUCE_PARTS)),
                             dim3(256), 0, st, plane_copies, ncopies, P.bins_x * P.bins_y, H.nchan,
                             nb2, line_copies, nblk, lines ? A.x.bins : 0, lines ? A.y.bins : 0,
                             lines ? A.c.bins : 0, h2, h2rgb, hx, hy, hc, H, P.bins_x, P.bins_y,
                             mode == HIST_RECORDS ? R.share : nullptr, plane_parts, nbl,
                             HIST_REDUCE_PARTS);
      }
      if (own) (void)hipFreeAsync(scratch, st);
    } else {
      general = (h2 ? 1 : 0) | (want_lines ? 2 : 0);
    }
  }
  if (need) return hipSuccess;
  if (general) {
    const dim3 full((unsigned)((beam.n + 255) / 256));
    hipLaunchKernelGGL(plot_hist_kernel, full, dim3(256), 0, st, beam, x, y, c, P, A, h2, h2rgb,
                       hx, hy, hc, counters, general);
  }
  return hipGetLastError();
}

// ---- a plot in the tail of a pass: the host side (hist.h) ---------------------------------------
hipError_t plot_tail_plan(int64_t n, const xrt_hip_plot_tail& t, PlotTailPlan* plan, size_t* need) {
  if (need) *need = 0;
  const xrt_hip_plot& P = t.plot;
  const bool fields_ok = t.x_field >= 0 && t.x_field <= XRT_HIP_FIELD_ZPRIME && t.y_field >= 0 &&
                         t.y_field <= XRT_HIP_FIELD_ZPRIME && t.c_field >= 0 &&
                         t.c_field <= XRT_HIP_FIELD_ZPRIME;
  const bool shape_ok = n > 0 && n < HIST_MAX_RAYS && P.bins_x >= 1 && P.bins_y >= 1 &&
                        P.bins_x <= PLOT_TAIL_MAX_BINS_XY && P.bins_y <= PLOT_TAIL_MAX_BINS_XY &&
                        P.bins_c >= 1 && P.bins_c <= PLOT_TAIL_MAX_BINS_C && fields_ok;
  if (!shape_ok) return plan ? hipErrorInvalidValue : hipSuccess;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess)
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  PlotTail Q = {};
  Q.P = P;
  Q.A.x = axis_bins(P.x_lim[0], P.x_lim[1], P.bins_x);
  Q.A.y = axis_bins(P.y_lim[0], P.y_lim[1], P.bins_y);
  Q.A.c = axis_bins(P.c_lim[0], P.c_lim[1], P.bins_c);
  Q.fx = t.x_field;
  Q.fy = t.y_field;
  Q.fc = t.c_field;
  // tiles whose four planes fit the LDS beside the colour histogram and the search tables
  const size_t b1 = sizeof(double) * 4 * ((size_t)P.bins_x + P.bins_y + P.bins_c);
  const size_t search = (HIST_BLOCK / 64) * 128 * sizeof(int);
  const size_t beside = sizeof(double) * 4 * (size_t)P.bins_c + search + 256;
  if (b1 + search + 256 > HIST_LDS_BUDGET || beside >= HIST_LDS_BUDGET)
    return plan ? hipErrorInvalidValue : hipSuccess;
  HistPlan H = {};
  H.nchan = 4;
  const int most = cus < PLOT_TAIL_MAX_TILES ? cus : PLOT_TAIL_MAX_TILES;
  if (!plan_tiles(P.bins_x, P.bins_y, 4, HIST_LDS_BUDGET - beside, most, H) || H.tx * H.ty > 0xffff)
    return plan ? hipErrorInvalidValue : hipSuccess;
  Q.T = H.ntx * H.nty;
  Q.ntx = H.ntx;
  Q.tx = H.tx;
  Q.ty = H.ty;
  Q.mtx = H.tx == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)H.tx) + 1u;
  Q.mty = H.ty == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)H.ty) + 1u;
  Q.pitch = Q.T + 7 <= 32 ? 32 : 64;
  const int64_t chunks = (n + 63) / 64;
  const int ncopies = cus > Q.T + 1 ? cus : Q.T + 1;
  auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t recs = (size_t)chunks * 64;
  const size_t planes_b = pad((size_t)ncopies * 4 * H.tx * H.ty * sizeof(double));
  const size_t lines_b = pad((size_t)ncopies * b1);
  const size_t tab_b = pad((size_t)chunks * Q.pitch);
  const size_t total = planes_b + lines_b + 2 * pad(recs * 8) + pad(recs * 4) + tab_b + 1024 + 256;
  if (need) *need = total;
  if (!plan) return hipSuccess;
  if (!t.hist2d || !t.hist2d_rgb || !t.hist_x || !t.hist_y || !t.counters || !t.workspace)
    return hipErrorInvalidValue;
  char* q = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(t.workspace) + 255) / 256 * 256);
  if (q + total - 256 > static_cast<char*>(t.workspace) + t.workspace_bytes)
    return hipErrorInvalidValue;
  plan->plane_copies = reinterpret_cast<double*>(q);
  q += planes_b;
  plan->line_copies = reinterpret_cast<double*>(q);
  q += lines_b;
  Q.w = reinterpret_cast<double*>(q);
  q += pad(recs * 8);
  Q.hue = reinterpret_cast<double*>(q);
  q += pad(recs * 8);
  Q.word = reinterpret_cast<unsigned*>(q);
  q += pad(recs * 4);
  Q.tab = reinterpret_cast<unsigned char*>(q);
  q += tab_b;
  plan->share = reinterpret_cast<int*>(q);
  Q.want_c = t.hist_c != nullptr;
  Q.chunks = chunks;
  plan->Q = Q;
  plan->n = n;
  plan->chunks = chunks;
  plan->ncopies = ncopies;
  plan->cus = cus;
  plan->tiles_x = H.ntx;
  plan->tiles_y = H.nty;
  plan->h2 = t.hist2d;
  plan->h2rgb = t.hist2d_rgb;
  plan->hx = t.hist_x;
  plan->hy = t.hist_y;
  plan->hc = t.hist_c;
  plan->counters = t.counters;
  return hipSuccess;
}

hipError_t plot_tail_finish(const PlotTailPlan& L, hipStream_t st) {
  const PlotTail& Q = L.Q;
  const int nx = Q.A.x.bins, ny = Q.A.y.bins, nc = Q.A.c.bins;
  const size_t nl = 4 * ((size_t)nx + ny + nc);
  const size_t tile_b = sizeof(double) * (4 * (size_t)Q.tx * Q.ty + 4 * (size_t)nc);
  const size_t lds = (tile_b > nl * sizeof(double) ? tile_b : nl * sizeof(double)) +
                     (HIST_BLOCK / 64) * 128 * sizeof(int);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(plot_tail_tiles<4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(plot_tail_tiles<4>, dim3((unsigned)L.ncopies), dim3(HIST_BLOCK), lds, st,
                     L.chunks, Q, L.plane_copies, L.line_copies, L.share, L.counters);
  HistPlan H = {};
  H.ntx = L.tiles_x;
  H.nty = L.tiles_y;
  H.tx = Q.tx;
  H.ty = Q.ty;
  H.nchan = 4;
  H.lines = 1;
  H.derive = 1;
  H.slices = L.ncopies;
  const int total = 4 * nx * ny;
  const int nb2 = 4 * ((nx + 31) / 32) * ((ny + 7) / 8);
  const int nbl = (int)((nl + 255) / 256);
  const int plane_parts = total >= 32768 ? 1 : HIST_REDUCE_PARTS;
  hipLaunchKernelGGL(plot_hist_reduce,
                     dim3((unsigned)(nb2 * plane_parts + nbl * HIST_REDUCE_PARTS)), dim3(256),
                     0, st, L.plane_copies, L.ncopies, nx * ny, 4, nb2, L.line_copies, L.ncopies,
                     nx, ny, nc, L.h2, L.h2rgb, L.hx, L.hy, L.hc, H, nx, ny, L.share, plane_parts,
                     nbl, HIST_REDUCE_PARTS);
  return hipGetLastError();
}

hipError_t hist2d_launch(const xrt_hip_beam& beam, const double* x, const double* y, double xf,
                         double yf, int ray_flags, int flux_kind, double srcw, int bx,
                         double xlo, double xhi, int by, double ylo, double yhi, double* hist,
                         double* counters, hipStream_t st) {
  if (beam.n <= 0) return hipSuccess;
  // the flux plane of a plot without colour axis: the same kernels
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
                          st, nullptr, 0, nullptr);
}

}  // namespace xrt
