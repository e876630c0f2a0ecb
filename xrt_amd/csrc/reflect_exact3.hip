// The exact sequence of layered materials on surface families 1 and 2.
#include "reflect_tu.h"

namespace xrt {

bool tu_exact3(int spec, const ExactLaunch& L) {
  switch (spec) {
    case SP_LAYERED1: launch_exact_k<Layered1>(L); return true;
    case SP_LAYERED2: launch_exact_k<Layered2>(L); return true;
  }
  return false;
}

}  // namespace xrt
