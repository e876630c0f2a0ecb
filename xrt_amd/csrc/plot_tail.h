// The binning of an XYCPlot (multipro.py:316-361, raycing/__init__.py:170-300) per ray, shared by
// the histogram kernels (hist.hip) and by the ray kernels that carry a PLOT in their tail
// (reflect_impl.h: LateScreenPlot) -- one code, the same bins either way.
//
// A plot in the tail of a pass (round 6): run_ray_tracing's accumulate_plot of a screen image
// whose pass has not been launched joins that pass the way Screen.expose does. The tail takes
// the image record of its ray from the registers, forms weight, hue and the three bins, and the
// WAVE sorts its 64 rays by tile of the 2-D histogram with ballots (no LDS, no block barrier)
// and writes them out as 20-B records in that order, with one row of a byte table per wave:
// where each tile's run starts, and how many of its rays are alive / good / out / over / dead.
// plot_tail_tiles (hist.hip) then adds every tile's runs up in LDS as plot_hist_tiles does for
// the records of plot_hist_rays, which disappears from the iteration together with the image
// nobody else reads (100 B written + 44 B read per ray -> 20 B written).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/xrt_hip.h"
#include "fp64_math.h"

namespace xrt {

// One histogram axis: np.linspace(lo, hi, bins + 1) has edges[j] = j*step + lo with
// step = (hi - lo)/bins (one division, one multiplication, one addition: formed here
// with the same three roundings) and edges[bins] = hi exactly. `scale` only feeds the
// first guess of the bin, which the edge tests then correct: any rounding of it will do.
struct AxisBins {
  double lo, hi, step, scale;
  int bins;
};
inline AxisBins axis_bins(double lo, double hi, int bins) {
  AxisBins a;
  a.lo = lo;
  a.hi = hi;
  a.step = (hi - lo) / (double)bins;
  a.scale = (double)bins / (hi - lo);
  a.bins = bins;
  return a;
}
struct PlotAxes {
  AxisBins x, y, c;
};
__device__ __forceinline__ double bin_edge(const AxisBins& a, int j) {
  return j == a.bins ? a.hi : (double)j * a.step + a.lo;
}
// numpy's own search on uniform bins (lib/histograms.py: scale, truncate, one step down and one
// step up against the real edges). The guess is within one bin of the answer for any axis
// whose step is not lost in the rounding of its limits, and for monotone edges the result is
// then searchsorted's (np.histogram2d), the last edge belonging to the last bin.
__device__ __forceinline__ int find_bin(double v, const AxisBins& a) {
  if (!(v >= a.lo && v <= a.hi)) return -1;
  int b = (int)((v - a.lo) * a.scale);
  b = min(max(b, 0), a.bins - 1);
  b -= (b > 0 && v < (double)b * a.step + a.lo) ? 1 : 0;
  b += (b < a.bins - 1 && v >= (double)(b + 1) * a.step + a.lo) ? 1 : 0;
  return b;
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


__device__ __forceinline__ bool ray_selected(int st, int ray_flags) {
  bool sel = false;
  if ((ray_flags & 1) && st == 1) sel = true;
  if ((ray_flags & 2) && st == 2) sel = true;
  if ((ray_flags & 4) && st == 3) sel = true;
  if ((ray_flags & 8) && st < 0) sel = true;
  if ((ray_flags & 16) && st > 0) sel = true;
  return sel;
}


// ---- the plot in the tail of a pass ------------------------------------------------------------
#define PLOT_TAIL_MAX_TILES 56          // T + 2 run starts + 5 state counts in <= 64 bytes
#define PLOT_TAIL_MAX_BINS_XY 2046      // 11 bits each for ix + 1, iy + 1 of a ray outside the tiles
#define PLOT_TAIL_MAX_BINS_C 4094       // (the colour histogram lives in LDS beside the planes)
#define PLOT_TAIL_SELECTED 0x80000000u

struct PlotTail {
  xrt_hip_plot P;
  PlotAxes A;
  int fx, fy, fc;          // XRT_HIP_FIELD_*: what the axes show of the screen's image
  int T, ntx, tx, ty;      // tiles of the 2-D histogram (hist.hip: plan_tiles)
  unsigned mtx, mty;       // floor(2^32 / tx) + 1 (0 for tx = 1): ix / tx = umulhi(ix, mtx)
  int pitch;               // bytes per wave (64 rays) of the run table: 32 or 64
  int want_c;              // the colour histogram is wanted (ePos)
  int64_t chunks;          // waves of 64 rays the beam has
  double* w;               // [chunks * 64] records, each wave's rays sorted by tile
  double* hue;             // the colour DATUM (x c_factor): hue and colour bin are formed by
                           // plot_tail_tiles, whose arithmetic units idle behind the LDS
  unsigned* word;          // a tile's ray: cell in the tile;
                           // others (bucket T): (ix + 1) | (iy + 1) << 11
                           //                    | PLOT_TAIL_SELECTED if the plot's ray flags take it
  unsigned char* tab;      // [chunk][pitch]: run starts of buckets 0 .. T + 1, then the five counts
};

// what the ray of this lane gives the plot (kept in registers between take() and emit())
struct PlotStash {
  double w, hue;
  unsigned word;
  int tile;                // 0 .. T - 1, or T
  int st;                  // the state of the image's ray (0: no ray in this lane)
};

__device__ __forceinline__ double plot_field(int f, double x, double y, double z, double a,
                                             double b, double c, double path, double E) {
  // (selects, not a switch: a jump table would index the candidates through scratch memory)
  double v = E;
  v = f == XRT_HIP_FIELD_X ? x : v;
  v = f == XRT_HIP_FIELD_Y ? y : v;
  v = f == XRT_HIP_FIELD_Z ? z : v;
  v = f == XRT_HIP_FIELD_A ? a : v;
  v = f == XRT_HIP_FIELD_B ? b : v;
  v = f == XRT_HIP_FIELD_C ? c : v;
  v = f == XRT_HIP_FIELD_PATH ? path : v;
  if (f == XRT_HIP_FIELD_XPRIME || f == XRT_HIP_FIELD_ZPRIME)
    v = div_rn(f == XRT_HIP_FIELD_XPRIME ? a : c, b);     // (torch: a / b)
  return v;
}

// the image's ray -> its record (the arithmetic of hist.hip: plot_hist_rays, the same bits)
__device__ __forceinline__ PlotStash plot_tail_take(const PlotTail& Q, double x, double y,
                                                    double z, double a, double b, double c,
                                                    double path, double E, double Jss, double Jpp,
                                                    double Jsr, double Jsi, int st) {
  PlotStash s;
  s.w = s.hue = 0.;
  s.word = 0;
  s.tile = Q.T;
  s.st = st;
#ifdef TAIL_AB_NO_TAKE                 /* A/B: no weight, no bins (wrong plots) */
  return s;
#endif
  if (!ray_selected(st, Q.P.ray_flags)) return s;
  double w;
  if (Q.P.flux_kind == 1)
    w = Jss;
  else if (Q.P.flux_kind == 2)
    w = Jpp;
  else if (Q.P.flux_kind == 3)
    w = 2. * Jsr;
  else if (Q.P.flux_kind == 4)
    w = 2. * Jsi;
  else if (Q.P.flux_kind == 5)
    w = (Jss + Jpp) * E * 1.602176565e-19;
  else
    w = Jss + Jpp;
  w *= Q.P.source_weight;
  // (x, z, energy -- the plot of a screen image -- without the selects: a branch on uniform values)
  const bool usual = Q.fx == XRT_HIP_FIELD_X && Q.fy == XRT_HIP_FIELD_Z && Q.fc == XRT_HIP_FIELD_E;
  double vx = x, vy = z, vc = E;
  if (!usual) {
    vx = plot_field(Q.fx, x, y, z, a, b, c, path, E);
    vy = plot_field(Q.fy, x, y, z, a, b, c, path, E);
    vc = plot_field(Q.fc, x, y, z, a, b, c, path, E);
  }
  const double cv = vc * Q.P.c_factor;
  const int ix = find_bin(vx * Q.P.x_factor, Q.A.x);
  const int iy = find_bin(vy * Q.P.y_factor, Q.A.y);
  s.w = w;
  s.hue = cv;
  if (ix >= 0 && iy >= 0) {
    // (exact for ix < 2^16: the bins of a plot)
    const int tjx = Q.mtx ? (int)__umulhi((unsigned)ix, Q.mtx) : ix;
    const int tjy = Q.mty ? (int)__umulhi((unsigned)iy, Q.mty) : iy;
    s.tile = tjy * Q.ntx + tjx;
    s.word = (unsigned)((iy - tjy * Q.ty) * Q.tx + (ix - tjx * Q.tx));
  } else {
    s.word = (unsigned)(ix + 1) | (unsigned)(iy + 1) << 11 | PLOT_TAIL_SELECTED;
  }
  return s;
}

// inclusive prefix sum over the 64 lanes in data-parallel-primitive moves (no trip through the LDS
// crossbar: six dependent ds_bpermute at the very end of a wave cost more than the arithmetic)
__device__ __forceinline__ unsigned wave_prefix_sum(unsigned v) {
  // row_shr:1, 2, 4, 8 within rows of 16 lanes (zeros shifted in), then lane 15 of row 0 / 2 into
  // rows 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31)
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

// All 64 lanes of the wave, converged; lane l holds the record of ray 64 * chunk + l (st 0: none).
__device__ __forceinline__ void plot_tail_emit(const PlotTail& Q, int64_t chunk, const PlotStash& s) {
  if (chunk >= Q.chunks) return;       // (the last block's waves beyond the end of the beam)
#ifdef TAIL_AB_NO_EMIT                 /* A/B (tools/ab_tail_pass.sh): what the wave's sort and stores cost */
  return;
#endif
  const int lane = (int)__lane_id();
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  // rank of every ray within its bucket; lane t: how many rays bucket t has
  int rank = 0;
  unsigned here = 0;
  unsigned long long rem = ~0ull;
#ifdef TAIL_AB_NO_SORT                 /* A/B: the records unsorted (wrong plots, the stores' cost alone) */
  rem = 0;
  rank = lane;
#endif
  while (rem) {
    const int first = __builtin_ctzll(rem);
    const int t = __builtin_amdgcn_readlane(s.tile, first);
    const unsigned long long m = __ballot(s.tile == t);
    if (s.tile == t) rank = __popcll(m & below);
    if (lane == t) here = (unsigned)__popcll(m);
    rem &= ~m;
  }
  // bucket starts: exclusive prefix over the lanes (lane T + 1 and up: the total, 64)
  const unsigned incl = wave_prefix_sum(here);
  const unsigned start = incl - here;
  const int pos = (int)__shfl(start, s.tile) + rank;
  // the row of the byte table
  const unsigned alive = (unsigned)__popcll(__ballot(s.st > 0));
  const unsigned good = (unsigned)__popcll(__ballot(s.st == 1));
  const unsigned out = (unsigned)__popcll(__ballot(s.st == 2));
  const unsigned over = (unsigned)__popcll(__ballot(s.st == 3));
  const unsigned dead = (unsigned)__popcll(__ballot(s.st < 0));
  const int k = lane - (Q.T + 2);
  const unsigned byte = k < 0 ? start : k == 0 ? alive : k == 1 ? good : k == 2 ? out
                        : k == 3 ? over : k == 4 ? dead : 0u;
  if (lane < Q.pitch) Q.tab[chunk * Q.pitch + lane] = (unsigned char)byte;
  // the records in bucket order: every lane stores at its place within the wave's 512-B (256-B)
  // window -- the same lines as an ordered store, no permutation through the LDS crossbar.
  // (plain stores: plot_tail_tiles reads the records right behind this kernel, several tiles'
  // blocks each line of a wide beam -- out of the memory-side cache if they are still there)
  const int64_t o = chunk * 64 + pos;
  Q.w[o] = s.w;
  Q.hue[o] = s.hue;
  Q.word[o] = s.word;
}

}  // namespace xrt
