#pragma once
#include <hip/hip_runtime.h>
#include "../../include/xrt_hip.h"
namespace xrt {
hipError_t plot_hist_launch(const xrt_hip_beam& beam, const double* x, const double* y,
                            const double* c, const xrt_hip_plot& P, double* h2, double* h2rgb,
                            double* hx, double* hy, double* hc, double* counters,
                            hipStream_t st, void* ws, size_t ws_bytes, size_t* need);
hipError_t hist2d_launch(const xrt_hip_beam& beam, const double* x, const double* y, double xf,
                         double yf, int ray_flags, int flux_kind, double srcw, int bx,
                         double xlo, double xhi, int by, double ylo, double yhi, double* hist,
                         double* counters, hipStream_t st);
}
