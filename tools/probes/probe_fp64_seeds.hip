// How good are v_rcp_f64 / v_rsq_f64 as seeds, and are the cheap forms bit-identical to the
// compiler's IEEE division / sqrt / within tolerance of libm's acos?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/probe_fp64_seeds.hip -o /tmp/probe_seeds && /tmp/probe_seeds
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../xrt_amd/csrc/fp64_math.h"

using namespace xrt;

__device__ unsigned long long rng(unsigned long long& s) {
  s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s;
}

struct Out {
  double rcp_err, rsq_err, rcp1_err, acos_err;
  unsigned long long div_mismatch, div_checked_mismatch, sqrt_mismatch, n;
};

__global__ void probe(Out* out, int iters) {
  unsigned long long s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  double e_rcp = 0, e_rsq = 0, e_rcp1 = 0, e_acos = 0;
  unsigned long long dm = 0, dcm = 0, sm = 0;
  for (int it = 0; it < iters; ++it) {
    // a, b with exponents spread over 2^-60 .. 2^60, random mantissas and signs
    const unsigned long long r1 = rng(s), r2 = rng(s);
    const int ea = (int)(rng(s) % 121) - 60, eb = (int)(rng(s) % 121) - 60;
    double a = ldexp(1.0 + (double)(r1 >> 12) * 0x1p-52, ea);
    double b = ldexp(1.0 + (double)(r2 >> 12) * 0x1p-52, eb);
    if (r1 & 1) a = -a;
    if (r2 & 1) b = -b;
    const double y = __builtin_amdgcn_rcp(b);
    e_rcp = fmax(e_rcp, fabs(fma(-b, y, 1.0)));
    const double y1 = fma(fma(-b, y, 1.0), y, y);
    e_rcp1 = fmax(e_rcp1, fabs(fma(-b, y1, 1.0)));
    const double ab = fabs(b);
    const double z = __builtin_amdgcn_rsq(ab);
    e_rsq = fmax(e_rsq, fabs(fma(-ab * z, z, 1.0)) * 0.5);
    const double q = a / b;
    if (__double_as_longlong(q) != __double_as_longlong(div_rn(a, b))) ++dm;
    double dummy;
    if (__double_as_longlong(sqrt(ab)) != __double_as_longlong(sqrt_rn_rinv(ab, dummy))) ++sm;
    const double x = (double)(long long)(r1 >> 11) * 0x1p-52 - 1.0;   // [-1, 1)
    e_acos = fmax(e_acos, fabs(acos_np(x) - acos(x)));
  }
  atomicAdd(&out->div_mismatch, dm);
  atomicAdd(&out->div_checked_mismatch, dcm);
  atomicAdd(&out->sqrt_mismatch, sm);
  atomicAdd(&out->n, (unsigned long long)iters);
  // max via bit patterns of non-negative doubles
  atomicMax((unsigned long long*)&out->rcp_err, (unsigned long long)__double_as_longlong(e_rcp));
  atomicMax((unsigned long long*)&out->rsq_err, (unsigned long long)__double_as_longlong(e_rsq));
  atomicMax((unsigned long long*)&out->rcp1_err, (unsigned long long)__double_as_longlong(e_rcp1));
  atomicMax((unsigned long long*)&out->acos_err, (unsigned long long)__double_as_longlong(e_acos));
}

int main() {
  Out* d;
  hipMalloc(&d, sizeof(Out));
  hipMemset(d, 0, sizeof(Out));
  probe<<<1024, 256>>>(d, 4096);
  Out h;
  hipMemcpy(&h, d, sizeof(Out), hipMemcpyDeviceToHost);
  printf("samples %llu\n", h.n);
  printf("v_rcp_f64 seed: max |1 - b y| = %.3e (2^%.1f)\n", h.rcp_err, log2(h.rcp_err));
  printf("after one Newton step:        %.3e (2^%.1f)\n", h.rcp1_err, log2(h.rcp1_err));
  printf("v_rsq_f64 seed: max rel err   = %.3e (2^%.1f)\n", h.rsq_err, log2(h.rsq_err));
  printf("div_rn   != a / b : %llu\n", h.div_mismatch);
  printf("sqrt_rn_rinv != sqrt : %llu\n", h.sqrt_mismatch);
  printf("max |acos_np - acos| = %.3e\n", h.acos_err);
  return 0;
}
