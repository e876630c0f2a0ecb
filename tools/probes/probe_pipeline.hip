// Does a software pipeline through LDS-direct loads lift a compute-heavy streaming kernel to the
// memory floor?  The ray kernels read 13 arrays (100 B per ray), spend ~1500 VALU instructions per
// ray at 4 waves per SIMD and write 26 arrays (208 B). One-shot waves (load -> compute -> store)
// leave the memory system idle while they compute. Here:
//   mode 0: one wave = 64 rays, one-shot (the present structure)
//   mode 1: resident waves loop over batches; the NEXT batch's record is on its way into LDS
//           (global_load_lds_dwordx4, no VGPRs) while the present one is being computed
//   mode 2: resident waves, plain loads at the top of every iteration (no prefetch)
// All at 4 waves per SIMD (LDS-limited on purpose).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_pipeline.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define NIN 13
#define NOUT 26
struct Arrays {
  double* p[32];
};

// ~NI dependent-ish fp64 instructions on 8 chains
template <int NI>
__device__ __forceinline__ void work(const double (&t)[NIN], double (&o)[NOUT]) {
  double a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = t[k] + t[(k + 5) % NIN];
#pragma unroll 4
  for (int it = 0; it < NI / 8; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], 0.999999, a[(k + 1) & 7] * 1e-9);
  }
#pragma unroll
  for (int w = 0; w < NOUT; ++w) o[w] = a[w & 7] + w;
}

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

template <int MODE, int NI>
__global__ __launch_bounds__(128) void kern(Arrays in, Arrays out, long n) {
  extern __shared__ double lds[];   // per wave: 7 KB (14 slots of 512 B), padded to force occupancy
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nbatch = n / 64;
  if (MODE == 0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double t[NIN], o[NOUT];
#pragma unroll
    for (int r = 0; r < NIN; ++r) t[r] = in.p[r][i];
    work<NI>(t, o);
#pragma unroll
    for (int w = 0; w < NOUT; ++w) out.p[w][i] = o[w];
    return;
  }
  const long wid = (long)blockIdx.x * (blockDim.x >> 6) + wave;
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  if (MODE == 2) {
    for (long b = wid; b < nbatch; b += nw) {
      const long i = b * 64 + lane;
      double t[NIN], o[NOUT];
#pragma unroll
      for (int r = 0; r < NIN; ++r) t[r] = in.p[r][i];
      work<NI>(t, o);
#pragma unroll
      for (int w = 0; w < NOUT; ++w) out.p[w][i] = o[w];
    }
    return;
  }
  // MODE 1
  double* mine = lds + wave * (14 * 64);
  const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)mine);   // LDS byte address (low 32 bits of the shared pointer)
  const int half = lane >> 5, l2 = (lane & 31) * 2;
  auto prefetch = [&](long b) {
#pragma unroll
    for (int pr = 0; pr < 7; ++pr) {
      const int r = pr * 2 + half < NIN ? pr * 2 + half : NIN - 1;
      glds16(in.p[r] + b * 64 + l2, lbase + pr * 1024);
    }
  };
  long b = wid;
  if (b < nbatch) prefetch(b);
  for (; b < nbatch; b += nw) {
    // everything but the stores of the previous batch (issued after the prefetch) has landed
    asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
    double t[NIN], o[NOUT];
#pragma unroll
    for (int r = 0; r < NIN; ++r) t[r] = mine[r * 64 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (b + nw < nbatch) prefetch(b + nw);
    work<NI>(t, o);
    const long i = b * 64 + lane;
#pragma unroll
    for (int w = 0; w < NOUT; ++w) out.p[w][i] = o[w];
  }
}

template <int MODE, int NI>
float run(const Arrays& in, const Arrays& out, long n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t shmem = 20 * 1024;            // 8 blocks of 2 waves per CU = 4 waves per SIMD
  unsigned grid = MODE == 0 ? (unsigned)((n + 127) / 128) : 256 * 8;
  hipFuncSetAttribute((const void*)kern<MODE, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  for (int k = 0; k < 3; ++k) kern<MODE, NI><<<grid, 128, shmem>>>(in, out, n);
  hipEventRecord(e0);
  const int reps = 10;
  for (int k = 0; k < reps; ++k) kern<MODE, NI><<<grid, 128, shmem>>>(in, out, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int NI>
void sweep(const Arrays& in, const Arrays& out, long n) {
  const double gb = (double)(NIN + NOUT) * 8 * n / 1e9;
  const float a = run<0, NI>(in, out, n), b = run<1, NI>(in, out, n), c = run<2, NI>(in, out, n);
  printf("%5d VALU/ray: one-shot %.3f ms (%.2f TB/s) | resident + LDS prefetch %.3f ms (%.2f TB/s) | resident, plain loads %.3f ms (%.2f TB/s)\n",
         NI, a, gb / a, b, gb / b, c, gb / c);
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 10000000 / 64 * 64;
  Arrays in, out;
  for (int k = 0; k < NIN; ++k) {
    hipMalloc(&in.p[k], n * 8);
    hipMemset(in.p[k], 0, n * 8);
  }
  for (int k = 0; k < NOUT; ++k) hipMalloc(&out.p[k], n * 8);
  sweep<16>(in, out, n);
  sweep<400>(in, out, n);
  sweep<800>(in, out, n);
  sweep<1200>(in, out, n);
  sweep<1600>(in, out, n);
  sweep<2400>(in, out, n);
  // correctness of the LDS path: out[0] of mode 1 == mode 0
  return 0;
}
