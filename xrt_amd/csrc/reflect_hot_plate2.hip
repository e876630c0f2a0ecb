// Both faces of a flat plate in one pass (reflect_fused_plate2): Plate.double_refract.
#include "reflect_tu.h"

namespace xrt {

bool tu_hot_plate2(int spec, const DcmLaunch& L) {
  if (spec != SP_FLAT_PLATE) return false;
  launch_plate2_k<FlatPlate>(L);
  return true;
}

}  // namespace xrt
