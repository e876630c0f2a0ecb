#pragma once
#include <hip/hip_runtime.h>
#include "../../include/xrt_hip.h"
namespace xrt {
// up to two apertures right behind an element, in the order the beam meets them (the tail of a
// pass, reflect_impl.h; apertures_mark in screen_impl.h)
#define XRT_TAIL_APERTURES 2
struct TailApertures {
  int n;
  xrt_hip_aperture a[XRT_TAIL_APERTURES];
};
hipError_t screen_expose_launch(const xrt_hip_screen& S, const xrt_hip_beam& in,
                                const xrt_hip_beam& out, hipStream_t st);
hipError_t screen_expose_mark_launch(const xrt_hip_screen& S, const xrt_hip_aperture& A,
                                     const xrt_hip_beam& in, const xrt_hip_beam& out,
                                     hipStream_t st);
hipError_t aperture_propagate_launch(const xrt_hip_aperture& A, const xrt_hip_beam& in,
                                     const xrt_hip_beam& lo, const xrt_hip_beam& glo,
                                     hipStream_t st);
}
