// OE.multiple_reflect: the instantiations of reflect_multi for the built-in surface families
// (user-defined surfaces bring theirs in their unit, user_unit.hip.in) and the launch logic
// of one bounce.
#include <stdlib.h>

#include "reflect_multi_impl.h"

namespace xrt {

bool tu_multi(int spec, const MultiLaunch& L) {
  switch (spec) {
    case SP_GENERIC0: return launch_multi_k<Generic0>(L) == 0;
    case SP_GENERIC1: return launch_multi_k<Generic1>(L) == 0;
    case SP_GENERIC2: return launch_multi_k<Generic2>(L) == 0;
    default: return false;
  }
}

__global__ void multi_init(GStat* g) {
  gstat_reset(g, 1);
  g->bar = 0;
  g->hang = 0;
}

struct MultiWs {
  unsigned long long* counts;
  double* diag;
  GStat* g;
  double* part;
  double* tang;
  double *ht, *hx, *hy, *hz;        // (the sparse form's hit records and index)
  int32_t *hlost, *idx, *cnt;
  int nseg;
};
static size_t pad256(size_t b) { return (b + 255) / 256 * 256; }
static MultiWs multi_ws(void* workspace, int64_t n) {
  char* w = reinterpret_cast<char*>(workspace);
  MultiWs L;
  L.counts = reinterpret_cast<unsigned long long*>(w);
  L.diag = reinterpret_cast<double*>(w + 128);
  L.g = reinterpret_cast<GStat*>(w + 256);
  L.part = reinterpret_cast<double*>(w + 512);
  char* q = w + 512 + REFLECT_PART_BYTES;
  const size_t d = pad256((size_t)n * 8), i4 = pad256((size_t)n * 4);
  L.nseg = (int)((n + MULTI_SEG - 1) / MULTI_SEG);
  L.tang = reinterpret_cast<double*>(q);
  q += d;
  L.ht = reinterpret_cast<double*>(q);
  q += d;
  L.hx = reinterpret_cast<double*>(q);
  q += d;
  L.hy = reinterpret_cast<double*>(q);
  q += d;
  L.hz = reinterpret_cast<double*>(q);
  q += d;
  L.hlost = reinterpret_cast<int32_t*>(q);
  q += i4;
  L.idx = reinterpret_cast<int32_t*>(q);
  q += pad256((size_t)L.nseg * MULTI_SEG * 4);
  L.cnt = reinterpret_cast<int32_t*>(q);
  return L;
}

size_t bounce_workspace_bytes(int64_t n) {
  const size_t nseg = (size_t)((n + MULTI_SEG - 1) / MULTI_SEG);
  return 512 + REFLECT_PART_BYTES + 5 * pad256((size_t)n * 8) + pad256((size_t)n * 4) +
         pad256(nseg * MULTI_SEG * 4) + pad256(nseg * 4);
}

static int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 64;
  }
  return cus;
}

hipError_t reflect_bounce_launch(const xrt_hip_pass& P, const xrt_hip_material& M,
                                 const xrt_hip_beam& in, const xrt_hip_beam& out,
                                 const xrt_hip_bounce& B, void* workspace, hipStream_t st,
                                 bool want_info) {
  static_assert(sizeof(GStat) <= 256, "workspace slot");
  if (in.n <= 0) return hipSuccess;
  const MultiWs W = multi_ws(workspace, in.n);
  MultiLaunch L;
  L.st = st;
  L.P = &P;
  L.M = &M;
  L.in = &in;
  L.out = &out;
  L.cus = device_cus();
  L.A.g = W.g;
  L.A.part = W.part;
  L.A.tang = W.tang;
  L.A.diag = W.diag;
  L.A.counts = W.counts;
  L.A.nrefl_in = B.nrefl_in;
  L.A.nrefl_out = B.nrefl_out;
  L.A.theta = B.theta;
  for (int k = 0; k < 4; ++k) {
    L.A.elev_in[k] = B.elev_in[k];
    L.A.elev_out[k] = B.elev_out[k];
  }
  for (int k = 0; k < 3; ++k) L.A.spr[k] = B.spr_out[k];
  L.A.idx = W.idx;
  L.A.cnt = W.cnt;
  L.A.nseg = W.nseg;
  L.A.ht = W.ht;
  L.A.hx = W.hx;
  L.A.hy = W.hy;
  L.A.hz = W.hz;
  L.A.hlost = W.hlost;
  // (XRT_HIP_MULTI_FORM = dense / sparse / exact: one form for every bounce -- full bounces
  // optimistic / all bounces over an index / the exact dense kernel --, for A/B runs and tests)
  const char* form = getenv("XRT_HIP_MULTI_FORM");
  L.sparse = B.entering_hint > 0 && B.entering_hint * 4 < in.n;
  if (form && (form[0] == 'd' || form[0] == 'e')) L.sparse = 0;
  if (form && form[0] == 's') L.sparse = 1;
  // the optimistic form of a full bounce, unless the caller wants the batch statistics (only the
  // exact phases collect them) or XRT_HIP_REFLECT_EXACT=1 / XRT_HIP_MULTI_FORM=exact says no
  const char* ex = getenv("XRT_HIP_REFLECT_EXACT");
  L.A.gate = !L.sparse && !want_info && B.assume_hit_brent >= 0 && !(ex && ex[0] == '1') &&
             !(form && form[0] == 'e');
  L.A.assume = (B.assume_hit_brent > 0 ? 1 : 0) | (B.assume_tangency_brent > 0 ? 2 : 0);
  hipLaunchKernelGGL(multi_init, dim3(1), dim3(1), 0, st, W.g);
  bool launched;
  BarrierSerial one_at_a_time(st);        // (grid barriers between the phases: reflect.h)
  if (P.surf_kind == XRT_HIP_SURF_USER) {
    const UserUnit* unit = static_cast<const UserUnit*>(P.user_unit);
    if (!unit || !unit->multi) return hipErrorInvalidValue;
    launched = unit->multi(&L) == 0;
  } else {
    const bool wide = P.surf_kind == XRT_HIP_SURF_BENT_BRAGG || P.surf_kind == XRT_HIP_SURF_VFM ||
                      P.surf_kind == XRT_HIP_SURF_DUALVFM || P.surf_kind == XRT_HIP_SURF_DICED;
    const int spec =
        wide ? SP_GENERIC2 : (P.surf_kind >= XRT_HIP_SURF_BLAZED ? SP_GENERIC1 : SP_GENERIC0);
    launched = tu_multi(spec, L);
  }
  if (!launched) return hipErrorInvalidDeviceFunction;
  return hipGetLastError();
}

hipError_t multi_to_global_launch(const xrt_hip_pass& P, const xrt_hip_beam& last,
                                  const xrt_hip_beam& orig, const int32_t* nrefl,
                                  const xrt_hip_beam& gb, hipStream_t st) {
  if (last.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(multi_to_global_kernel,
                     dim3((unsigned)((last.n + REFLECT_BLOCK - 1) / REFLECT_BLOCK)),
                     dim3(REFLECT_BLOCK), 0, st, P, last, orig, nrefl, gb);
  return hipGetLastError();
}

}  // namespace xrt
