// fp64 atomic adds into a histogram of `cells` doubles from 1e7 x 4 random updates: how fast,
// and does the memory scope of the atomic matter (agent = what atomicAdd gives; workgroup =
// executed in the XCD's own L2 -- only valid for a table that one XCD alone touches)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_atomics.hip -o /tmp/pa && /tmp/pa
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <int SCOPE, bool PER_XCD>
__global__ void upd(double* tab, long cells, const unsigned* idx, long n, unsigned* seen) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned x = PER_XCD ? xcc_id() : 0;
  if (PER_XCD && threadIdx.x == 0) atomicOr(seen, 1u << x);
  double* t = tab + (PER_XCD ? (long)x * cells * 4 : 0);
  const unsigned b = idx[i] % (unsigned)cells;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    (void)__hip_atomic_fetch_add(&t[(long)k * cells + b], 1.0 + k, __ATOMIC_RELAXED, SCOPE);
}

__global__ void fill(unsigned* idx, long n, int gaussian) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long s = 0x9E3779B97F4A7C15ull * (i + 1);
  s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
  unsigned a = (unsigned)s, b = (unsigned)(s >> 32);
  if (gaussian) {   // sum of four uniforms ~ a bell: a beam's footprint on the plot
    unsigned g = ((a & 0xffff) + (a >> 16) + (b & 0xffff) + (b >> 16)) >> 2;   // 0..65535
    idx[i] = g * 65537u;
  } else {
    idx[i] = a;
  }
}

template <int SCOPE, bool PER_XCD>
void run(const char* what, double* tab, long cells, const unsigned* idx, long n, unsigned* seen) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipMemset(tab, 0, cells * 4 * 8 * 8);
  upd<SCOPE, PER_XCD><<<(n + 255) / 256, 256>>>(tab, cells, idx, n, seen);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) upd<SCOPE, PER_XCD><<<(n + 255) / 256, 256>>>(tab, cells, idx, n, seen);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  // total of channel 0 over all copies must be 6 n
  double* h = (double*)malloc(cells * 4 * 8 * 8);
  hipMemcpy(h, tab, cells * 4 * 8 * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int x = 0; x < 8; ++x)
    for (long c = 0; c < cells; ++c) sum += h[(long)x * cells * 4 + c];
  free(h);
  printf("%-44s cells %7ld: %.3f ms, %.2e atomics/s, channel-0 total %.0f (want %.0f)\n", what, cells,
         ms, 4.0 * n / ms * 1e3, sum, 6.0 * n);
}

int main() {
  const long n = 10000000;
  unsigned *idx, *seen;
  double* tab;
  hipMalloc(&idx, n * 4);
  hipMalloc(&seen, 4);
  hipMemset(seen, 0, 4);
  hipMalloc(&tab, (size_t)262144 * 4 * 8 * 8);
  for (int gaussian = 0; gaussian < 2; ++gaussian) {
    fill<<<(n + 255) / 256, 256>>>(idx, n, gaussian);
    printf("--- %s cell indices\n", gaussian ? "bell-shaped" : "uniform");
    for (long cells : {16384L, 65536L, 262144L}) {
      run<__HIP_MEMORY_SCOPE_AGENT, false>("agent scope, one table", tab, cells, idx, n, seen);
      run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope, one table per XCD", tab, cells, idx, n, seen);
      run<__HIP_MEMORY_SCOPE_AGENT, true>("agent scope, one table per XCD", tab, cells, idx, n, seen);
    }
  }
  unsigned s;
  hipMemcpy(&s, seen, 4, hipMemcpyDeviceToHost);
  printf("XCC ids seen: 0x%x\n", s);
  return 0;
}
