// OE(figureError = ...): the exact sequence of surface families 1 and 2.
#include "reflect_tu.h"

namespace xrt {

bool tu_figured_exact1(int spec, const ExactLaunch& L) {
  switch (spec) {
    case SP_FIGURED1: launch_exact_k<Figured<1>>(L); return true;
    case SP_FIGURED2: launch_exact_k<Figured<2>>(L); return true;
  }
  return false;
}

}  // namespace xrt
