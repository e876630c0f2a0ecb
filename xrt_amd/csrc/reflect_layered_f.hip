// Layered materials (Multilayer / Coated), specular deflection: Parratt's recursion per ray.
#include "reflect_tu.h"

namespace xrt {

bool tu_layered_fused(int spec, int mode, const FusedLaunch& L) {
  switch (spec) {
    case SP_LAYERED0: launch_fused_k<Layered0>(mode, L); return true;
    case SP_LAYERED1: launch_fused_k<Layered1>(mode, L); return true;
    case SP_LAYERED2: launch_fused_k<Layered2>(mode, L); return true;
  }
  return false;
}

}  // namespace xrt
