// Crystal passes other than the thick flat one: thick crystals on any family-0 surface, thin
// crystals / Laue geometries / crystals from their unit cell, flat or not; the fused DCM
// of thin crystals.
#include "reflect_tu.h"

namespace xrt {

bool tu_xtal_xtal(int spec, int mode, const FusedLaunch& L) {
  switch (spec) {
    case SP_THICK_ANY: launch_xtal_k<ThickAny>(mode, L); return true;
    case SP_FLAT_XTAL: launch_xtal_k<FlatXtal>(mode, L); return true;
    case SP_ANY_XTAL: launch_xtal_k<AnyXtal>(mode, L); return true;
  }
  return false;
}

bool tu_xtal_dcm(int spec, const DcmLaunch& L) {
  if (spec != SP_FLAT_XTAL) return false;
  launch_dcm_k<FlatXtal>(L);
  return true;
}

}  // namespace xrt
