// The kernels the headline workloads run: mirror / plate passes with the kinds fixed at
// compile time (cfg2: ToroidMirror), thick flat Bragg crystals and the fused DCM (cfg3).
#include "reflect_tu.h"

namespace xrt {

bool tu_hot_fused(int spec, int mode, const FusedLaunch& L) {
  // (launch_fused_k: the record of arguments; -DXRT_FUSED_EARLY_ARGS for the A/B: one by one)
  switch (spec) {
    case SP_TOROID_MIRROR: launch_fused_k<ToroidMirror>(mode, L); return true;
    case SP_FLAT_MIRROR: launch_fused_k<FlatMirror>(mode, L); return true;
    case SP_BENT_MIRROR: launch_fused_k<BentMirror>(mode, L); return true;
    case SP_FLAT_PLATE: launch_fused_k<FlatPlate>(mode, L); return true;
  }
  return false;
}

bool tu_hot_xtal(int spec, int mode, const FusedLaunch& L) {
  if (spec != SP_THICK_FLAT) return false;
  launch_xtal_k<ThickFlat>(mode, L);
  return true;
}

bool tu_hot_dcm(int spec, const DcmLaunch& L) {
  if (spec != SP_THICK_FLAT) return false;
  launch_dcm_k<ThickFlat>(L);
  return true;
}

}  // namespace xrt

#ifdef XRT_PROBE_TIMING
// the per-wave records of the last launch: [wave][8] = ticks of sections 0..4, [7] = start stamp
extern "C" __attribute__((visibility("default"))) int xrt_probe_ticks(unsigned long long* out,
                                                                      long long waves) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(xrt::g_probe_ticks), (size_t)waves * 64) != hipSuccess)
    return -1;
  return 0;
}
#endif
