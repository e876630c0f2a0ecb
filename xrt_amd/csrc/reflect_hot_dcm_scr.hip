// DCM.double_reflect with apertures / a screen in its tail (reflect_fused_dcm_scr): thick and thin
// flat Bragg crystals.
#include "reflect_tu.h"

namespace xrt {

bool tu_hot_dcm_scr(int spec, const DcmLaunch& L, const xrt_hip_beam& gb2, const xrt_hip_screen& S,
                    const xrt_hip_beam& sb, const TailApertures& ap) {
  switch (spec) {
    case SP_THICK_FLAT: launch_dcm_scr_k<ThickFlat>(L, gb2, S, sb, ap); return true;
    case SP_FLAT_XTAL: launch_dcm_scr_k<FlatXtal>(L, gb2, S, sb, ap); return true;
  }
  return false;
}

}  // namespace xrt
