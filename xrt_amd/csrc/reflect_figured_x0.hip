// OE(figureError = ...): the exact sequence of family-0 surfaces (also what Bragg crystals with a
// figure error take).
#include "reflect_tu.h"

namespace xrt {

bool tu_figured_exact0(int spec, const ExactLaunch& L) {
  if (spec != SP_FIGURED0) return false;
  launch_exact_k<Figured<0>>(L);
  return true;
}

}  // namespace xrt
