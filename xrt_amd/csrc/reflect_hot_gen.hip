// The lean mirror passes with the source in their head and a screen in their tail
// (reflect_fused_gen_scr): GeometricSource.shine -> OE.reflect -> Screen.expose as one pass.
#include "reflect_tu.h"

namespace xrt {

bool tu_hot_fused_gen_scr(int spec, const FusedLaunch& L) {
  switch (spec) {
    case SP_TOROID_MIRROR: launch_fused_gen_scr_k<ToroidMirror>(L); return true;
    case SP_FLAT_MIRROR: launch_fused_gen_scr_k<FlatMirror>(L); return true;
    case SP_BENT_MIRROR: launch_fused_gen_scr_k<BentMirror>(L); return true;
  }
  return false;
}

}  // namespace xrt
