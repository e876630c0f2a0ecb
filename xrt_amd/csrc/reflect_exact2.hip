// The exact sequence of layered materials on family-0 surfaces and of the general zone plate.
#include "reflect_tu.h"

namespace xrt {

bool tu_exact2(int spec, const ExactLaunch& L) {
  switch (spec) {
    case SP_LAYERED0: launch_exact_k<Layered0>(L); return true;
    case SP_PER_RAY_ZONES: launch_exact_k<PerRayZones>(L); return true;
  }
  return false;
}

}  // namespace xrt
