// Issue rate of the fp64 VALU instructions the node loops are made of (gfx950):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_fp64_rates tools/probes/probe_fp64_rates.hip && /tmp/probe_fp64_rates
// Each kernel runs ITER x 48 instructions of one kind in 4 independent dependency chains per lane,
// 8 waves per SIMD; cycles per instruction = time x clock / (ITER x 48 x 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 20000
#define OPS(STMT) \
  for (int it = 0; it < ITER; ++it) { _Pragma("unroll") for (int k = 0; k < 12; ++k) { STMT } }

template <int KIND>
__global__ __launch_bounds__(256) void probe(double* out, double seed, double sc) {
  double a = seed + threadIdx.x, b = a + 1., c = a + 2., d = a + 3.;
  const double m = 1.0000001, q = 1e-9;
  if (KIND == 0) OPS(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));)
  if (KIND == 1) OPS(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
  if (KIND == 2) OPS(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(q));)
  if (KIND == 3) OPS(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(sc));)
  if (KIND == 4) OPS(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(sc), "v"(q));)
  if (KIND == 5) OPS(asm volatile("v_fma_f64 %0, -%0, %4, 1.0\n v_fma_f64 %1, -%1, %4, 1.0\n v_fma_f64 %2, -%2, %4, 1.0\n v_fma_f64 %3, -%3, %4, 1.0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
  if (KIND == 6) OPS(asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));)
  if (KIND == 7) OPS(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
  if (KIND == 8) OPS(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_and_b32 %1, %1, %2\n v_lshlrev_b32 %2, 1, %3\n v_add_u32 %3, %3, %0" : "+v"(((int*)&a)[0]), "+v"(((int*)&b)[0]), "+v"(((int*)&c)[0]), "+v"(((int*)&d)[0]) : : "vcc");)
  if (KIND == 9) OPS(asm volatile("v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %5\n v_fma_f64 %2, %2, %4, %5\n v_mul_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));)
  // one dependency chain per lane: what a wave can issue on its own
  if (KIND == 10) OPS(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(q));)
  if (KIND == 11) OPS(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(m), "v"(q));)
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int KIND>
static void run(const char* name, double* out, int blocks = 2048) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<KIND><<<blocks, 256>>>(out, 1.5, 1.0000001);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<KIND><<<blocks, 256>>>(out, 1.5, 1.0000001);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // 2048 blocks x 4 waves over 1024 SIMDs = 8 waves per SIMD
  const int waves = blocks * 4 / 1024;     // per SIMD
  printf("%-34s %d waves/SIMD %8.3f ms  %.2f cycles / instruction at 2.15 GHz\n", name, waves, ms,
         ms * 1e-3 * 2.15e9 / (double(ITER) * 48 * waves));
}
int main() {
  double* out;
  hipMalloc(&out, 2048 * 256 * 8);
  run<0>("v_fma_f64 v,v,v", out);
  run<1>("v_mul_f64 v,v", out);
  run<2>("v_add_f64 v,v", out);
  run<3>("v_mul_f64 v,s", out);
  run<4>("v_fma_f64 v,s,v", out);
  run<5>("v_fma_f64 -v,v,1.0", out);
  run<6>("v_fmac_f64 v,v", out);
  run<7>("v_rcp_f64", out);
  run<8>("32-bit mix", out);
  run<9>("mul/add/fma/mul mix", out);
  for (int blocks : {256, 512, 1024, 2048}) {
    run<10>("v_fma_f64, ONE chain per lane", out, blocks);
    run<11>("v_fma_f64, two chains per lane", out, blocks);
    run<0>("v_fma_f64, four chains per lane", out, blocks);
  }
  return 0;
}
