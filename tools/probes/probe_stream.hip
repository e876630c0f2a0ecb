// What does the memory system give a kernel with the access pattern of the ray kernels --
// R input arrays and W output arrays of doubles, one ray per lane (8 B per lane and access) --
// and what changes it: rays per lane (16-B accesses), block size, persistent waves, the
// block -> chunk mapping over the XCDs, non-temporal stores?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_stream.hip -o /tmp/probe_stream && /tmp/probe_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define MAXA 48
struct Arrays {
  double* p[MAXA];
};

__device__ __forceinline__ unsigned chunk_block(bool xcd) {
  if (!xcd) return blockIdx.x;
  const unsigned nb = gridDim.x, b = blockIdx.x;
  const unsigned q = nb >> 3, rem = nb & 7u, x = b & 7u, j = b >> 3;
  return x * q + (x < rem ? x : rem) + j;
}

template <int R, int W, int VEC, bool NT>
__global__ void stream(Arrays in, Arrays out, long n, int xcd, int persist) {
  const long nv = n / VEC;
  long i = (long)chunk_block(xcd != 0) * blockDim.x + threadIdx.x;
  const long stride = persist ? (long)gridDim.x * blockDim.x : nv;
  for (; i < nv; i += stride) {
    double acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.;
    if constexpr (VEC == 1) {
      double t[R];
#pragma unroll
      for (int r = 0; r < R; ++r) t[r] = in.p[r][i];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[0] += t[r];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        if (NT)
          __builtin_nontemporal_store(acc[0] + w, &out.p[w][i]);
        else
          out.p[w][i] = acc[0] + w;
      }
    } else {
      double2 t[R];
#pragma unroll
      for (int r = 0; r < R; ++r) t[r] = reinterpret_cast<const double2*>(in.p[r])[i];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        acc[0] += t[r].x;
        acc[1] += t[r].y;
      }
#pragma unroll
      for (int w = 0; w < W; ++w) {
        double2 o = make_double2(acc[0] + w, acc[1] + w);
        typedef double v2d __attribute__((ext_vector_type(2)));
        if (NT)
          __builtin_nontemporal_store(v2d{o.x, o.y}, &reinterpret_cast<v2d*>(out.p[w])[i]);
        else
          reinterpret_cast<double2*>(out.p[w])[i] = o;
      }
    }
  }
}

template <int R, int W, int VEC, bool NT>
float run(const Arrays& in, const Arrays& out, long n, int block, int xcd, int persist, int waves_per_simd,
          int occ = 0) {   // occ: waves per SIMD enforced through the LDS allocation (0 = no limit)
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  long nv = n / VEC;
  unsigned grid = (unsigned)((nv + block - 1) / block);
  if (persist) grid = 256 * 4 * waves_per_simd * 64 / block;
  size_t shmem = 0;
  if (occ) {   // blocks per CU = occ * 4 * 64 / block, each gets an equal share of 160 KB
    shmem = (size_t)(160 * 1024) / (occ * 256 / block) - 256;
    hipFuncSetAttribute((const void*)stream<R, W, VEC, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  }
  for (int k = 0; k < 3; ++k) stream<R, W, VEC, NT><<<grid, block, shmem>>>(in, out, n, xcd, persist);
  hipEventRecord(e0);
  const int reps = 10;
  for (int k = 0; k < reps; ++k) stream<R, W, VEC, NT><<<grid, block, shmem>>>(in, out, n, xcd, persist);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int R, int W>
void sweep(const Arrays& in, const Arrays& out, long n) {
  const double gb = (double)(R + W) * 8 * n / 1e9;
  auto line = [&](const char* what, float ms) {
    printf("R=%2d W=%2d  %-44s %7.3f ms  %6.2f TB/s\n", R, W, what, ms, gb / ms);
  };
  line("8 B/lane, block 128", run<R, W, 1, false>(in, out, n, 128, 0, 0, 0));
  line("8 B/lane, block 256", run<R, W, 1, false>(in, out, n, 256, 0, 0, 0));
  line("8 B/lane, block 512", run<R, W, 1, false>(in, out, n, 512, 0, 0, 0));
  line("8 B/lane, block 64", run<R, W, 1, false>(in, out, n, 64, 0, 0, 0));
  line("8 B/lane, block 128, XCD chunks", run<R, W, 1, false>(in, out, n, 128, 1, 0, 0));
  line("8 B/lane, block 128, nt stores", run<R, W, 1, true>(in, out, n, 128, 0, 0, 0));
  line("8 B/lane, block 256, persistent x4", run<R, W, 1, false>(in, out, n, 256, 0, 1, 4));
  line("8 B/lane, block 256, persistent x8", run<R, W, 1, false>(in, out, n, 256, 0, 1, 8));
  line("16 B/lane, block 128", run<R, W, 2, false>(in, out, n, 128, 0, 0, 0));
  line("16 B/lane, block 256", run<R, W, 2, false>(in, out, n, 256, 0, 0, 0));
  line("16 B/lane, block 64", run<R, W, 2, false>(in, out, n, 64, 0, 0, 0));
  line("16 B/lane, block 128, XCD chunks", run<R, W, 2, false>(in, out, n, 128, 1, 0, 0));
  line("16 B/lane, block 128, nt stores", run<R, W, 2, true>(in, out, n, 128, 0, 0, 0));
  line("16 B/lane, block 256, persistent x4", run<R, W, 2, false>(in, out, n, 256, 0, 1, 4));
  for (int occ = 2; occ <= 8; ++occ) {
    char what[64];
    snprintf(what, sizeof what, "8 B/lane, block 128, %d waves per SIMD", occ);
    line(what, run<R, W, 1, false>(in, out, n, 128, 0, 0, 0, occ));
  }
  line("8 B/lane, block 512, 4 waves per SIMD", run<R, W, 1, false>(in, out, n, 512, 0, 0, 0, 4));
  line("8 B/lane, block 512, 6 waves per SIMD", run<R, W, 1, false>(in, out, n, 512, 0, 0, 0, 6));
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 10000000;
  Arrays in, out;
  for (int k = 0; k < MAXA; ++k) {
    in.p[k] = out.p[k] = nullptr;
  }
  for (int k = 0; k < 13; ++k) {
    hipMalloc(&in.p[k], n * 8);
    hipMemset(in.p[k], 0, n * 8);
  }
  for (int k = 0; k < 40; ++k) hipMalloc(&out.p[k], n * 8);
  if (argc > 2) {               // the occupancy question only
    sweep<13, 26>(in, out, n);
    sweep<13, 40>(in, out, n);
    return 0;
  }
  sweep<13, 13>(in, out, n);    // Screen.expose: 100 B in, 100 B out
  sweep<13, 26>(in, out, n);    // OE.reflect: 100 B in, 208 B out
  sweep<13, 40>(in, out, n);    // DCM.double_reflect: 100 B in, 316 B out
  sweep<13, 1>(in, out, n);     // read only, nearly
  sweep<1, 26>(in, out, n);     // write only, nearly
  return 0;
}
