// The lean mirror / plate passes with a screen AND the plot of its image in their tail
// (reflect_fused_scr_plot, reflect_fused_gen_scr_plot): OE.reflect -> Screen.expose ->
// accumulate_plot of run_ray_tracing as one pass over the beam (plot_tail.h).
#include "reflect_tu.h"

namespace xrt {

bool tu_hot_fused_scr_plot(int spec, int mode, const FusedLaunch& L) {
  switch (spec) {
    case SP_TOROID_MIRROR: launch_fused_scr_plot_k<ToroidMirror>(mode, L); return true;
    case SP_FLAT_MIRROR: launch_fused_scr_plot_k<FlatMirror>(mode, L); return true;
    case SP_BENT_MIRROR: launch_fused_scr_plot_k<BentMirror>(mode, L); return true;
    case SP_FLAT_PLATE: launch_fused_scr_plot_k<FlatPlate>(mode, L); return true;
  }
  return false;
}

bool tu_hot_fused_gen_scr_plot(int spec, const FusedLaunch& L) {
  switch (spec) {
    case SP_TOROID_MIRROR: launch_fused_gen_scr_plot_k<ToroidMirror>(L); return true;
    case SP_FLAT_MIRROR: launch_fused_gen_scr_plot_k<FlatMirror>(L); return true;
    case SP_BENT_MIRROR: launch_fused_gen_scr_plot_k<BentMirror>(L); return true;
  }
  return false;
}

}  // namespace xrt
