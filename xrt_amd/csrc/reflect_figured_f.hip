// OE(figureError = ...): the fused pass with the height-map spline in the intersection search and
// in the normal (reflect_impl.h: figure_height, figure_turn_normal), one kernel per surface
// family, surface and material kinds read at run time.
#include "reflect_tu.h"

namespace xrt {

bool tu_figured_fused(int spec, int mode, const FusedLaunch& L) {
  switch (spec) {
    case SP_FIGURED0: launch_fused_k<Figured<0>>(mode, L); return true;
    case SP_FIGURED1: launch_fused_k<Figured<1>>(mode, L); return true;
    case SP_FIGURED2: launch_fused_k<Figured<2>>(mode, L); return true;
  }
  return false;
}

}  // namespace xrt
