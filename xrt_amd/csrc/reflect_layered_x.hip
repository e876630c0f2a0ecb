// Layered materials deflecting like a Bragg crystal of their period (Multilayer).
#include "reflect_tu.h"

namespace xrt {

bool tu_layered_xtal(int spec, int mode, const FusedLaunch& L) {
  switch (spec) {
    case SP_LAYERED0: launch_xtal_k<Layered0>(mode, L); return true;
    case SP_LAYERED1: launch_xtal_k<Layered1>(mode, L); return true;
    case SP_LAYERED2: launch_xtal_k<Layered2>(mode, L); return true;
  }
  return false;
}

}  // namespace xrt
