// Probe: (1) lane layout of v_mfma_f64_4x4x4_4b_f64, (2) its issue/throughput cost
// next to a stream of fp64 VALU work in the same wave. Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void layout_kernel(double* out) {
  // A = 1 at lane la only, B = 1 at lane lb only -> which D lane becomes 1?
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = lane == la ? 1.0 : 0.0;
      double b = lane == lb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(la * 64 + lb) * 64 + lane] = d;
    }
}

template <int NV, int NM>
__global__ __launch_bounds__(256) void mix_kernel(int iters, double* out, double seed) {
  double x0 = seed + threadIdx.x, x1 = seed * 2 + threadIdx.x, x2 = x0 * 0.5, x3 = x1 * 0.25;
  double d0 = 0, d1 = 0, d2 = 0;
  const double c = 1.0000001, e = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) {
      x0 = __builtin_fma(x0, c, e);
      x1 = __builtin_fma(x1, c, e);
      x2 = __builtin_fma(x2, c, e);
      x3 = __builtin_fma(x3, c, e);
    }
    if (NM >= 2) {
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, x1, d0, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, x3, d0, 0, 0, 0);
    }
    if (NM >= 4) {
      d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, x3, d1, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, x1, d1, 0, 0, 0);
    }
    if (NM >= 6) {
      d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, x3, d2, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, x0, d2, 0, 0, 0);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + d0 + d1 + d2;
}

template <int NV, int NM>
float run(int blocks, int iters, double* out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  mix_kernel<NV, NM><<<blocks, 256>>>(iters, out, 1.0);
  hipEventRecord(a);
  mix_kernel<NV, NM><<<blocks, 256>>>(iters, out, 1.0);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 64 * 64 * 64 + 256 * 8192 * 8);
  layout_kernel<<<1, 64>>>(out);
  std::vector<double> h(64 * 64 * 64);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  // for each D lane, find (la, lb) pairs feeding it
  printf("D lane <- list of (A lane, B lane)\n");
  for (int ld = 0; ld < 64; ld += 1) {
    printf("D%2d:", ld);
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb)
        if (h[(la * 64 + lb) * 64 + ld] != 0.0) printf(" (%d,%d)", la, lb);
    printf("\n");
    if (ld == 7) ld = 15;
    if (ld == 19) ld = 59;
  }
  const int blocks = 256 * 4 * 2, iters = 20000;   // 8 waves/SIMD
  float t;
  t = run<48, 0>(blocks, iters, out); printf("VALU 48 fma, 0 mfma: %8.3f ms  -> %.2f cyc/iter/wave-slot\n", t, t * 1e-3 * 2.4e9 / iters / 8);
  t = run<48, 2>(blocks, iters, out); printf("VALU 48 fma, 2 mfma: %8.3f ms\n", t);
  t = run<48, 4>(blocks, iters, out); printf("VALU 48 fma, 4 mfma: %8.3f ms\n", t);
  t = run<48, 6>(blocks, iters, out); printf("VALU 48 fma, 6 mfma: %8.3f ms\n", t);
  t = run<4, 6>(blocks, iters, out);  printf("VALU  4 fma, 6 mfma: %8.3f ms\n", t);
  t = run<60, 0>(blocks, iters, out); printf("VALU 60 fma, 0 mfma: %8.3f ms\n", t);
  return 0;
}
