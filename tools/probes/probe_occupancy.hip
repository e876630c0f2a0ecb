// One-shot streaming (13 arrays in, 26 out, 8 B per lane) with ~NI fp64 instructions per ray in
// between: how does the time depend on the waves per SIMD (forced through the VGPR allocation,
// not LDS), and does a static LDS allocation cost anything by itself?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_occupancy.hip -o /tmp/po && /tmp/po
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define NIN 13
#ifndef BLOCK
#define BLOCK 128
#endif
#ifndef NOUT
#define NOUT 26
#endif
struct Arrays { double* p[48]; };

template <int NI>
__device__ __forceinline__ void work(const double (&t)[NIN], double (&o)[NOUT]) {
  double a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = t[k] + t[(k + 5) % NIN];
#pragma unroll 4
  for (int it = 0; it < NI / 8; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], 0.999999, a[(k + 1) & 7]);
  }
#pragma unroll
  for (int w = 0; w < NOUT; ++w) o[w] = a[w & 7] + w;
}

// VREG: highest VGPR touched -> allocation -> waves per SIMD = 512 / (VREG + 1) rounded down
template <int NI, int VREG, int LDSB>
__global__ __launch_bounds__(BLOCK) void kern(Arrays in, Arrays out, long n) {
  __shared__ double lds[LDSB ? LDSB / 8 : 1];
  if (VREG == 63) asm volatile("" ::: "v63");
  if (VREG == 79) asm volatile("" ::: "v79");
  if (VREG == 95) asm volatile("" ::: "v95");
  if (VREG == 127) asm volatile("" ::: "v127");
  if (VREG == 167) asm volatile("" ::: "v167");
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double t[NIN], o[NOUT];
#pragma unroll
  for (int r = 0; r < NIN; ++r) t[r] = in.p[r][i];
  if (LDSB) {   // park two of the values in LDS and take them back (keeps the allocation alive)
    lds[threadIdx.x] = t[0];
    t[0] = lds[threadIdx.x] * 1.0;
  }
  work<NI>(t, o);
#pragma unroll
  for (int w = 0; w < NOUT; ++w) out.p[w][i] = o[w];
}

template <int NI, int VREG, int LDSB>
void run(const Arrays& in, const Arrays& out, long n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
  for (int k = 0; k < 3; ++k) kern<NI, VREG, LDSB><<<grid, BLOCK>>>(in, out, n);
  hipEventRecord(e0);
  for (int k = 0; k < 10; ++k) kern<NI, VREG, LDSB><<<grid, BLOCK>>>(in, out, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  printf("%5d fp64 instr/ray, %d waves/SIMD, LDS %5d B/block: %.3f ms  %.2f TB/s\n", NI, 512 / (VREG + 1),
         LDSB, ms, (NIN + NOUT) * 8.0 * n / 1e9 / ms);
}

template <int NI>
void sweep(const Arrays& in, const Arrays& out, long n) {
  run<NI, 63, 0>(in, out, n);
  run<NI, 79, 0>(in, out, n);
  run<NI, 95, 0>(in, out, n);
  run<NI, 127, 0>(in, out, n);
  run<NI, 167, 0>(in, out, n);
  run<NI, 95, 1024>(in, out, n);
  run<NI, 95, 12288>(in, out, n);
  run<NI, 127, 12288>(in, out, n);
}

int main(int argc, char** argv) {
  const long n = 10000000;
  Arrays in, out;
  for (int k = 0; k < NIN; ++k) { hipMalloc(&in.p[k], n * 8); hipMemset(in.p[k], 0, n * 8); }
  for (int k = 0; k < NOUT; ++k) hipMalloc(&out.p[k], n * 8);
  sweep<16>(in, out, n);
  sweep<800>(in, out, n);
  sweep<1600>(in, out, n);
  sweep<2000>(in, out, n);
  return 0;
}
