// A single flat Bragg crystal with apertures / a screen in the tail of its pass
// (reflect_fused_xtal_scr): thick and thin crystals.
#include "reflect_tu.h"

namespace xrt {

bool tu_hot_xtal_scr(int spec, int mode, const FusedLaunch& L) {
  switch (spec) {
    case SP_THICK_FLAT: launch_xtal_scr_k<ThickFlat>(mode, L); return true;
    case SP_FLAT_XTAL: launch_xtal_scr_k<FlatXtal>(mode, L); return true;
  }
  return false;
}

}  // namespace xrt
