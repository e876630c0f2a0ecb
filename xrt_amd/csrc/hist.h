#pragma once
#include <hip/hip_runtime.h>
#include "../../include/xrt_hip.h"
#include "plot_tail.h"
namespace xrt {
hipError_t plot_hist_launch(const xrt_hip_beam& beam, const double* x, const double* y,
                            const double* c, const xrt_hip_plot& P, double* h2, double* h2rgb,
                            double* hx, double* hy, double* hc, double* counters,
                            hipStream_t st, void* ws, size_t ws_bytes, size_t* need);
hipError_t hist2d_launch(const xrt_hip_beam& beam, const double* x, const double* y, double xf,
                         double yf, int ray_flags, int flux_kind, double srcw, int bx,
                         double xlo, double xhi, int by, double ylo, double yhi, double* hist,
                         double* counters, hipStream_t st);

// A plot in the tail of a pass (plot_tail.h): the host side. plot_tail_plan lays the caller's
// scratch out and fills the record the ray kernel takes (Q); plot_tail_finish launches the two
// kernels that turn the records the pass wrote into the plot's accumulators.
struct PlotTailPlan {
  PlotTail Q;
  int64_t n, chunks;
  int ncopies, cus;
  int tiles_x, tiles_y;
  double *plane_copies, *line_copies;
  int* share;
  double *h2, *h2rgb, *hx, *hy, *hc, *counters;
};
// *need (optional): the scratch a plot of this shape takes, 0 = it cannot ride a pass. With
// *plan* the scratch of *t* is laid out (hipErrorInvalidValue if it cannot or is too small).
hipError_t plot_tail_plan(int64_t n, const xrt_hip_plot_tail& t, PlotTailPlan* plan, size_t* need);
hipError_t plot_tail_finish(const PlotTailPlan& plan, hipStream_t st);
}
