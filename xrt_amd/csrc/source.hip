// GeometricSource.shine on the device (xrt/backends/raycing/sources/geoms.py:420-535): one
// thread per ray draws its coordinates from counter-based Philox4x32-10 blocks and writes the
// beam's SoA arrays straight into HBM -- 100 B per ray (132 with amplitudes), all stores,
// coalesced and non-temporal; nothing is read. The laws and the order of the arithmetic are
// the reference's; the stream layout is documented in oracle/geosource_np.py (the CPU
// restatement this kernel is tested against).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/xrt_hip.h"
#include "fp64_math.h"
#include "source.h"
#include "source_impl.h"

namespace xrt {

using namespace gen;

__global__ __launch_bounds__(256) void geosource_shine_kernel(xrt_hip_geosource G,
                                                             xrt_hip_beam out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= out.n) return;
  const bool amp = out.Es_ri != nullptr;
  store_gen_ray(out, i, make_ray(G, call_of(G), i, amp), G.state, amp);
}

__global__ __launch_bounds__(256) void geosource_probe_kernel(xrt_hip_geosource G, int64_t n,
                                                             int32_t* flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ray r{};
  double a, c;
  one_pair(G, call_of(G), 3, G.annulus_ac != 0, G.ann_ac, (uint64_t)i, SLOT_AC, r, false, a,
           c);
  if (a * a + c * c > 1.) atomicOr(flag, 1);
}

hipError_t geosource_shine_launch(const xrt_hip_geosource& G, const xrt_hip_beam& out,
                                  hipStream_t st) {
  if (out.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(geosource_shine_kernel, dim3((unsigned)((out.n + 255) / 256)), dim3(256), 0,
                     st, G, out);
  return hipGetLastError();
}

hipError_t geosource_probe_launch(const xrt_hip_geosource& G, int64_t n, int32_t* flag,
                                  hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(geosource_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     G, n, flag);
  return hipGetLastError();
}

}  // namespace xrt
