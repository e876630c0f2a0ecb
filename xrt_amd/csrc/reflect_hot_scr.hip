// The lean mirror / plate passes with a screen fused into their tail (reflect_fused_scr):
// OE.reflect -> Screen.expose as one pass over the beam.
#include "reflect_tu.h"

namespace xrt {

bool tu_hot_fused_scr(int spec, int mode, const FusedLaunch& L) {
  switch (spec) {
    case SP_TOROID_MIRROR: launch_fused_scr_k<ToroidMirror>(mode, L); return true;
    case SP_FLAT_MIRROR: launch_fused_scr_k<FlatMirror>(mode, L); return true;
    case SP_BENT_MIRROR: launch_fused_scr_k<BentMirror>(mode, L); return true;
    case SP_FLAT_PLATE: launch_fused_scr_k<FlatPlate>(mode, L); return true;
  }
  return false;
}

}  // namespace xrt
