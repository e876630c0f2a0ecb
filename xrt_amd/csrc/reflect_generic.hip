// The fused pass with surface and material read at run time, one kernel per surface family,
// and the kernels of the general zone plate.
#include "reflect_tu.h"

namespace xrt {

bool tu_generic_fused(int spec, int mode, const FusedLaunch& L) {
  switch (spec) {
    case SP_GENERIC0: launch_fused_k<Generic0>(mode, L); return true;
    case SP_GENERIC1: launch_fused_k<Generic1>(mode, L); return true;
    case SP_GENERIC2: launch_fused_k<Generic2>(mode, L); return true;
    case SP_PER_RAY_ZONES: launch_fused_k<PerRayZones>(mode, L); return true;
  }
  return false;
}

}  // namespace xrt
