// Would an interleaved Beam (one 104-B record of 13 doubles per ray instead of 13 arrays) let a
// ray kernel stream faster than the 39 8-B streams of the SoA layout (probe_stream.hip: 0.565-
// 0.615 ms for 13 in / 26 out on 1e7 rays)? Same bytes (104 B in, 208 B out per ray), three
// access patterns:
//   wide    -- 1 input and 2 output arrays, 16 B per lane, lanes consecutive (a plain copy: the
//              ceiling of any layout)
//   record  -- lane i reads ITS record (6 x 16 B + 8 B at i * 104) and writes two records: what a
//              ray kernel would do with no transposition
//   lds     -- the wave reads its 64 records as 6656 consecutive bytes (16 B per lane), turns them
//              through the LDS into one ray per lane, and writes the same way
//   soa     -- 13 in / 26 out arrays, 8 B per lane (the present layout, for the same box)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_stream_wide.hip -o /tmp/psw && /tmp/psw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define NF 13
typedef double v2d __attribute__((ext_vector_type(2)));

struct Soa {
  double* p[40];
};

__global__ void k_wide(const v2d* __restrict__ in, v2d* __restrict__ o1, v2d* __restrict__ o2,
                       long nv) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  v2d t = __builtin_nontemporal_load(in + i);
  __builtin_nontemporal_store(t + 1., o1 + i);
  __builtin_nontemporal_store(t + 2., o2 + i);
}

__global__ void k_record(const double* __restrict__ in, double* __restrict__ o1,
                         double* __restrict__ o2, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* r = in + i * NF;
  double t[NF];
  // (records are 8-B aligned only: 104 = 6.5 x 16; even rays start on 16 B)
#pragma unroll
  for (int k = 0; k < NF; ++k) t[k] = r[k];
  double s = 0.;
#pragma unroll
  for (int k = 0; k < NF; ++k) s += t[k];
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    o1[i * NF + k] = t[k] + s;
    o2[i * NF + k] = t[k] - s;
  }
}

// block = 256 lanes = 4 waves; every wave turns its own 64 records through its own LDS slab
__global__ __launch_bounds__(256) void k_lds(const double* __restrict__ in, double* __restrict__ o1,
                                             double* __restrict__ o2, long n) {
  __shared__ double slab[4][64 * NF];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long w0 = ((long)blockIdx.x * 4 + wave) * 64;     // first ray of the wave
  if (w0 >= n) return;
  double* s = slab[wave];
  const v2d* src = reinterpret_cast<const v2d*>(in + w0 * NF);   // 64 * 104 B = 416 x 16 B
  const int nrec = (int)((n - w0) < 64 ? (n - w0) : 64);
  const int nvec = nrec * NF / 2;
  v2d q[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int j = k * 64 + lane;
    q[k] = j < nvec ? __builtin_nontemporal_load(src + j) : v2d{0., 0.};
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int j = k * 64 + lane;
    if (j < 416) reinterpret_cast<v2d*>(s)[j] = q[k];
  }
  __builtin_amdgcn_wave_barrier();
  double t[NF];
#pragma unroll
  for (int k = 0; k < NF; ++k) t[k] = s[lane * NF + k];
  double sum = 0.;
#pragma unroll
  for (int k = 0; k < NF; ++k) sum += t[k];
  for (int pass = 0; pass < 2; ++pass) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < NF; ++k) s[lane * NF + k] = pass ? t[k] - sum : t[k] + sum;
    __builtin_amdgcn_wave_barrier();
    v2d* dst = reinterpret_cast<v2d*>((pass ? o2 : o1) + w0 * NF);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int j = k * 64 + lane;
      if (j < nvec) __builtin_nontemporal_store(reinterpret_cast<v2d*>(s)[j], dst + j);
    }
  }
}

__global__ void k_soa(Soa in, Soa out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double t[NF];
#pragma unroll
  for (int k = 0; k < NF; ++k) t[k] = in.p[k][i];
  double s = 0.;
#pragma unroll
  for (int k = 0; k < NF; ++k) s += t[k];
#pragma unroll
  for (int k = 0; k < 2 * NF; ++k) __builtin_nontemporal_store(t[k % NF] + s, &out.p[k][i]);
}

template <class F>
static float time_it(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms / 10 < best) best = ms / 10;
  }
  return best;
}

int main() {
  const long n = 10000000;
  const size_t rec = (size_t)n * NF * sizeof(double);
  double *in, *o1, *o2;
  hipMalloc(&in, rec + 4096);
  hipMalloc(&o1, rec + 4096);
  hipMalloc(&o2, rec + 4096);
  hipMemset(in, 0, rec);
  const double bytes = 3. * rec;
  auto report = [&](const char* name, float ms) {
    printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms * 1e-9);
  };
  for (int block : {128, 256, 512}) {
    const long nv = n * NF / 2;
    char nm[64];
    snprintf(nm, sizeof nm, "wide, 16 B/lane, block %d", block);
    report(nm, time_it([&] {
             hipLaunchKernelGGL(k_wide, dim3((unsigned)((nv + block - 1) / block)), dim3(block), 0, 0,
                                (const v2d*)in, (v2d*)o1, (v2d*)o2, nv);
           }));
  }
  for (int block : {64, 128, 256}) {
    char nm[64];
    snprintf(nm, sizeof nm, "record per lane, block %d", block);
    report(nm, time_it([&] {
             hipLaunchKernelGGL(k_record, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, 0,
                                in, o1, o2, n);
           }));
  }
  report("records through LDS, block 256",
         time_it([&] { hipLaunchKernelGGL(k_lds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, in, o1, o2, n); }));
  hipFree(in);
  hipFree(o1);
  hipFree(o2);
  Soa a, b;
  const size_t pitch = ((size_t)n * 8 + 511) / 512 * 512;
  double* blk;
  hipMalloc(&blk, pitch * 39);
  hipMemset(blk, 0, pitch * 13);
  for (int k = 0; k < 13; ++k) a.p[k] = blk + pitch / 8 * k;
  for (int k = 0; k < 26; ++k) b.p[k] = blk + pitch / 8 * (13 + k);
  for (int block : {128, 256})  {
    char nm[64];
    snprintf(nm, sizeof nm, "soa 13 in / 26 out, 8 B/lane, block %d", block);
    report(nm, time_it([&] {
             hipLaunchKernelGGL(k_soa, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, 0, a, b, n);
           }));
  }
  return 0;
}
