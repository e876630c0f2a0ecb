// The exact sequence of surface families 1 and 2.
#include "reflect_tu.h"

namespace xrt {

bool tu_exact1(int spec, const ExactLaunch& L) {
  switch (spec) {
    case SP_GENERIC1: launch_exact_k<Generic1>(L); return true;
    case SP_GENERIC2: launch_exact_k<Generic2>(L); return true;
  }
  return false;
}

}  // namespace xrt
