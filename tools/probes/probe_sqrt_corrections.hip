// How often does the second Goldschmidt correction of the f64 sqrt sequence
// (fp64_math.h, sqrt_rn_halfinv) change the result? Counts, over N pseudo-random
// arguments of the magnitude the Kirchhoff kernel sees (r^2 ~ 1e8 mm^2) and over
// [1, 4), the arguments for which one correction and two corrections differ.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/probe_sqrt tools/probes/probe_sqrt_corrections.hip && /tmp/probe_sqrt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint64_t splitmix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__global__ void probe(uint64_t per_thread, int wide, unsigned long long* diff,
                      unsigned long long* diff_ref) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long d = 0, dr = 0;
  for (uint64_t j = 0; j < per_thread; ++j) {
    const uint64_t bits = splitmix(tid * per_thread + j);
    // mantissa random; exponent: [1,4) or 2^26..2^28 (r^2 of 8..16 m in mm^2)
    const uint64_t e = wide ? (1023ull + 26 + (bits >> 63)) : (1023ull + (bits >> 63));
    const double x = __longlong_as_double((long long)((e << 52) | (bits & 0xfffffffffffffull)));
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r0 = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r0, g);
    h = __builtin_fma(h, r0, h);
    double d0 = __builtin_fma(-g, g, x);
    const double g1 = __builtin_fma(d0, h, g);
    double d1 = __builtin_fma(-g1, g1, x);
    const double g2 = __builtin_fma(d1, h, g1);
    d += g1 != g2;
    dr += g2 != __builtin_sqrt(x);
  }
  if (d) atomicAdd(diff, d);
  if (dr) atomicAdd(diff_ref, dr);
}

int main() {
  unsigned long long *dd, h[2];
  hipMalloc(&dd, 16);
  for (int wide = 0; wide < 2; ++wide) {
    hipMemset(dd, 0, 16);
    const uint64_t per_thread = 1 << 16;
    const unsigned blocks = 1 << 14;   // x 256 threads x 65536 = 2.7e11 arguments
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, per_thread, wide, dd, dd + 1);
    hipDeviceSynchronize();
    hipMemcpy(h, dd, 16, hipMemcpyDeviceToHost);
    printf("%s: %.3e arguments, one vs two corrections differ for %llu, two corrections vs "
           "the compiler's sqrt for %llu\n", wide ? "2^26..2^28" : "[1,4)",
           (double)blocks * 256 * per_thread, h[0], h[1]);
  }
  return 0;
}
