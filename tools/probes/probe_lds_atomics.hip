// LDS atomic throughput per CU: what does one ds_add_* wave-instruction cost, by data type,
// by how the 64 lanes' addresses are spread, and against the non-atomic alternative
// (each lane owns its cell: ds_read, add, ds_write)? Decides the design of csrc/hist.hip:
// its kernels issue 16 ds_add_f64 per ray.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_lds_atomics.hip -o /tmp/pla && /tmp/pla
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int CELLS = 16384;   // doubles in LDS (128 KB)
constexpr int ITERS = 2048;

__device__ __forceinline__ unsigned hashu(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// SPREAD: 0 = random over all cells, 1 = all lanes one cell, 2 = lane l -> cell l (no conflict,
// consecutive), 3 = random over 40 cells, 4 = random over 1024 cells, 5 = pairs of lanes share
enum { T_F64 = 0, T_U64 = 1, T_F32 = 2, T_U32 = 3, T_RMW64 = 4, T_F64_RTN = 5, T_NONE = 6 };

template <int TYPE, int SPREAD>
__global__ __launch_bounds__(1024) void probe(double* out, int blocks_per_cu) {
  extern __shared__ double cells[];
  for (int k = threadIdx.x; k < CELLS; k += blockDim.x) cells[k] = 0.;
  __syncthreads();
  unsigned s = hashu(blockIdx.x * 1024u + threadIdx.x + 1u);
  double acc = 0.;
  for (int it = 0; it < ITERS; ++it) {
    s = s * 1664525u + 1013904223u;
    unsigned idx;
    if (SPREAD == 0) idx = (s >> 8) % CELLS;
    else if (SPREAD == 1) idx = (it * 7) % CELLS;
    else if (SPREAD == 2) idx = (threadIdx.x + it * 64) % CELLS;
    else if (SPREAD == 3) idx = (s >> 8) % 40u;
    else if (SPREAD == 4) idx = (s >> 8) % 1024u;
    else idx = ((threadIdx.x >> 1) + it * 64) % CELLS;
    const double v = 1.0 + (double)(s & 7);
    if (TYPE == T_F64) {
      atomicAdd(&cells[idx], v);
    } else if (TYPE == T_F64_RTN) {
      acc += atomicAdd(&cells[idx], v);
    } else if (TYPE == T_U64) {
      atomicAdd(reinterpret_cast<unsigned long long*>(cells) + idx, (unsigned long long)(s & 7) + 1);
    } else if (TYPE == T_F32) {
      atomicAdd(reinterpret_cast<float*>(cells) + idx, (float)v);
    } else if (TYPE == T_U32) {
      atomicAdd(reinterpret_cast<unsigned*>(cells) + idx, (s & 7) + 1);
    } else if (TYPE == T_RMW64) {
      cells[idx] = cells[idx] + v;          // (racy unless SPREAD == 2: the cost of the plain form)
    } else {
      acc += v * (double)idx;
    }
  }
  __syncthreads();
  double t = acc;
  for (int k = threadIdx.x; k < CELLS; k += blockDim.x) t += cells[k];
  if (t == 12345.678) out[0] = t;
}

template <int TYPE, int SPREAD>
void run(const char* what, double* out) {
  const int blocks = 256;   // one 1024-lane block per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<TYPE, SPREAD>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, CELLS * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<TYPE, SPREAD><<<blocks, 1024, CELLS * 8>>>(out, 1);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) probe<TYPE, SPREAD><<<blocks, 1024, CELLS * 8>>>(out, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  const double lane_ops = 1024.0 * ITERS;           // per CU
  printf("%-52s %.3f ms  %.2f lane-updates / ns / CU  (%.1f ns per wave-instruction per CU)\n", what,
         ms, lane_ops / (ms * 1e6), ms * 1e6 / (16.0 * ITERS));
}

int main() {
  double* out;
  hipMalloc(&out, 64);
  run<T_NONE, 0>("no LDS op (index + value arithmetic only)", out);
  run<T_F64, 0>("ds_add_f64, random over 16384 cells", out);
  run<T_F64, 4>("ds_add_f64, random over 1024 cells", out);
  run<T_F64, 3>("ds_add_f64, random over 40 cells", out);
  run<T_F64, 1>("ds_add_f64, all lanes ONE cell", out);
  run<T_F64, 2>("ds_add_f64, lane l -> cell l (conflict-free)", out);
  run<T_F64, 5>("ds_add_f64, lane pairs share a cell", out);
  run<T_F64_RTN, 0>("ds_add_rtn_f64, random over 16384 cells", out);
  run<T_U64, 0>("ds_add_u64, random over 16384 cells", out);
  run<T_U64, 3>("ds_add_u64, random over 40 cells", out);
  run<T_U64, 2>("ds_add_u64, conflict-free", out);
  run<T_F32, 0>("ds_add_f32, random over 16384 cells", out);
  run<T_F32, 3>("ds_add_f32, random over 40 cells", out);
  run<T_F32, 2>("ds_add_f32, conflict-free", out);
  run<T_U32, 0>("ds_add_u32, random over 16384 cells", out);
  run<T_U32, 2>("ds_add_u32, conflict-free", out);
  run<T_RMW64, 2>("read + add + write f64, lane-owned cells", out);
  run<T_RMW64, 0>("read + add + write f64, random (racy: cost only)", out);
  return 0;
}
