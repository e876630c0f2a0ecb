/* xrt_hip.h — C ABI of libxrt_hip.so, the MI355X (gfx950) compute backend for the
 * hot path of xrt's raycing engine.
 *
 * Plain C, ctypes/cffi loadable: only pointers, sizes and scalars cross the
 * boundary. Every entry point returns 0 on success and a negative code on
 * failure; xrt_hip_last_error() then returns a thread-local message.
 * "_dev" entry points take DEVICE pointers (data already resident in HBM) and a
 * hipStream_t passed as void* (NULL = default stream); they are asynchronous
 * unless a timing output is requested. Entry points without "_dev" take HOST
 * pointers, are blocking, and do their own staging.
 *
 * Reference interfaces replaced (paths relative to the xrt source tree):
 *   xrt_hip_kirchhoff_f64      <- XRT_CL.run_parallel('integrate_kirchhoff', ...)
 *                                 as marshalled by _diffraction_integral_CL
 *                                 (xrt/backends/raycing/waves.py:854-896,
 *                                  myopencl.py:414-583, cl/diffract.cl:80-151)
 *   xrt_hip_kirchhoff_f64_dev  <- the same integral, numpy form
 *                                 _diffraction_integral_conv (waves.py:834-851),
 *                                 on device-resident SoA arrays
 *   xrt_hip_reflect_f64_dev    <- OE._reflect_local + the global<->local
 *                                 transforms around it in OE.reflect
 *                                 (oes/reflect.py:18-163, 551-1139;
 *                                  oes/base.py:801-1048, 1094-1163, 1231-1295)
 *   xrt_hip_screen_expose_f64_dev <- Screen.expose (screens.py:226-302)
 */
#ifndef XRT_HIP_H
#define XRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define XRT_HIP_API __attribute__((visibility("default")))
#else
#define XRT_HIP_API
#endif

#define XRT_HIP_OK 0
#define XRT_HIP_ERR_ARG (-1)     /* bad argument */
#define XRT_HIP_ERR_HIP (-2)     /* HIP runtime error (message has hipGetErrorString) */
#define XRT_HIP_ERR_NODEV (-3)   /* no usable GPU */
#define XRT_HIP_ERR_NOMEM (-4)   /* workspace too small / allocation failed */

/* ---- library ---------------------------------------------------------- */
XRT_HIP_API int xrt_hip_version(void);              /* 100*major + minor */
XRT_HIP_API int xrt_hip_device_count(void);         /* >=0, or negative error */
XRT_HIP_API const char* xrt_hip_last_error(void);   /* thread-local, never NULL */

/* ---- P2: Fresnel-Kirchhoff diffraction integral ------------------------
 * convention 0 = numpy path (+i k/4pi; waves.py:844,847)
 * convention 1 = OpenCL kernel (-i/4pi and the (1+i) factor on the direction
 *                integrals; cl/diffract.cl:136-148)                          */
#define XRT_HIP_KIRCHHOFF_NUMPY 0
#define XRT_HIP_KIRCHHOFF_OPENCL 1

/* Launch plan. nsplit_req<=0 / ppt_req<=0 = choose automatically.
 * Outputs may be NULL. */
XRT_HIP_API int xrt_hip_kirchhoff_plan(int64_t np, int64_t ns, int nsplit_req, int ppt_req,
                           size_t* workspace_bytes, int* nsplit, int* ppt);

/* Device-resident form. All arrays fp64. px,py,pz[np]: receiving points in the
 * diffracting element's local frame. Samples [ns]: position sx,sy,sz; surface
 * normal nx,ny,nz; nl = (ray direction).(normal) ("cosGamma"); wavenumber k
 * [1/mm]; Es, Ep complex interleaved (re,im) [2*ns]. Outputs S,P,A,B,C complex
 * interleaved [2*np] = (Es, Ep, aE, bE, cE) of the reference.
 * workspace: >= workspace_bytes from xrt_hip_kirchhoff_plan with the same
 * (np, ns, nsplit_req, ppt_req). kernel_ms: if not NULL the call synchronises
 * and returns the duration of the main kernel (HIP events on `stream`). */
XRT_HIP_API int xrt_hip_kirchhoff_f64_dev(
    int64_t np, const double* px, const double* py, const double* pz,
    int64_t ns, const double* sx, const double* sy, const double* sz,
    const double* nx, const double* ny, const double* nz, const double* nl,
    const double* k, const double* Es_ri, const double* Ep_ri, int convention,
    double* S_ri, double* P_ri, double* A_ri, double* B_ri, double* C_ri,
    void* workspace, size_t workspace_bytes, int nsplit_req, int ppt_req,
    void* stream, float* kernel_ms);

/* Host form with exactly the reference's OpenCL marshalling
 * (waves.py:860-894): pos_xyzw / nrm_xyzw are ns*4 doubles = ns x [x,y,z,0]
 * (numpy (4,ns) order='F'). The receiving points are split evenly over
 * dev_ids[0..ndev) like XRT_CL.run_parallel_max does over OpenCL devices
 * (myopencl.py:455-533); samples are replicated. Blocking. kernel_ms (optional)
 * receives the slowest device's main-kernel time. */
XRT_HIP_API int xrt_hip_kirchhoff_f64(
    int ndev, const int* dev_ids, int64_t np, const double* px, const double* py,
    const double* pz, int64_t ns, const double* cos_gamma, const double* Es_ri,
    const double* Ep_ri, const double* k, const double* pos_xyzw,
    const double* nrm_xyzw, int convention, double* S_ri, double* P_ri,
    double* A_ri, double* B_ri, double* C_ri, float* kernel_ms);

/* ---- building-block checks (used by the GPU tests only) ---------------- */
XRT_HIP_API int xrt_hip_debug_sqrt_f64_dev(int64_t n, const double* x, double* r, double* rinv,
                               void* stream);
XRT_HIP_API int xrt_hip_debug_sincos_f64_dev(int64_t n, const double* phi, double* sn, double* cs,
                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XRT_HIP_H */
