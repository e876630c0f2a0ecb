#pragma once
#include <hip/hip_runtime.h>
#include "../../include/xrt_hip.h"
namespace xrt {
using UndulatorArgs = xrt_hip_undulator;
using UndulatorMap = xrt_hip_undulator_map;
enum { UND_FAR = XRT_HIP_UND_FAR, UND_TAPER = XRT_HIP_UND_TAPER, UND_NF = XRT_HIP_UND_NF };
constexpr int UND_NODE_DOUBLES = 16;
// workspace: jend * UND_NODE_DOUBLES doubles (packed node records), then the (cos, sin) table of
// the in-loop sincos (2048 double2): the pack kernels fill it once per call, every block of the
// sum kernels copies it into LDS -- computing it per block was 300 of a ray's 4500 instructions
constexpr int UND_TAB_DOUBLES = 2 * 2048;
hipError_t undulator_pack_launch(const UndulatorArgs& a, void* workspace, hipStream_t st);
hipError_t undulator_sum_launch(const UndulatorArgs& a, int64_t n, const double* gamma,
                                const double* wu, const double* w, const double* ww1,
                                const double* ddphi, const double* ddpsi, double* Is_ri,
                                double* Ip_ri, const void* workspace, hipStream_t st);
hipError_t undulator_imap_launch(const UndulatorArgs& a, const UndulatorMap& m, int64_t n,
                                 const double* w, const double* theta, const double* psi,
                                 const double* gamma, double* I, double* Es_ri, double* Ep_ri,
                                 const void* workspace, hipStream_t st);
// workspace: as above
hipError_t custom_field_launch(const xrt_hip_custom_field& a, int64_t n, const double* emcg,
                               const double* gamma, const double* w, const double* ddphi,
                               const double* ddpsi, double* Is_ri, double* Ip_ri,
                               void* workspace, hipStream_t st, hipEvent_t e0,
                               hipEvent_t e1);
// n grid points, field on the 2n-1 half-step points; outputs n values each, betam one
hipError_t trajectory_launch(int filament, int64_t n, const double* wt, const double* Bx,
                             const double* By, const double* Bz, double gamma, double emcg,
                             double* betax, double* betay, double* trajx, double* trajy,
                             double* trajz, double* betam, hipStream_t st);
hipError_t bend_imap_launch(const xrt_hip_bend& m, int64_t n, const double* E,
                            const double* theta, const double* psi, const double* gamma,
                            double* I, double* Es_ri, double* Ep_ri, hipStream_t st);
hipError_t gauss_beam_launch(const xrt_hip_gauss& G, int64_t n, const double* x,
                             const double* y, const double* z, const double* E,
                             const double* dS, double dS_scalar, double* amp_ri, double* a,
                             double* b, double* c, hipStream_t st);
hipError_t bessel_k_probe_launch(int64_t n, const double* x, double* k13, double* k23,
                                 hipStream_t st);
}
