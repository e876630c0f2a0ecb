// The exact sequence (verdict kernel + redo) of family-0 passes and of the fused DCM.
#include "reflect_tu.h"

namespace xrt {

bool tu_exact0(int spec, const ExactLaunch& L) {
  if (spec != SP_GENERIC0) return false;
  launch_exact_k<Generic0>(L);
  return true;
}

// (the lean kernels are family 0: Generic0 redoes their passes)
void tu_exact0_redo_scr(const ExactLaunch& L, const xrt_hip_screen& S, const xrt_hip_beam& sb,
                        const xrt_hip_geosource* src, const PlotTail* plot,
                        const TailApertures* ap) {
  const xrt_hip_geosource none{};
  const PlotTail no_plot{};           // (w null: no plot behind the screen)
  const PlotTail& Q = plot ? *plot : no_plot;
  if (src)
    hipLaunchKernelGGL((reflect_redo_scr<Generic0, true>), L.grid, L.block, 0, L.st, *L.P, *L.M,
                       *src, *L.in, *L.restore, *L.lb, *L.vb, L.A, S, sb, Q, *ap);
  else
    hipLaunchKernelGGL((reflect_redo_scr<Generic0, false>), L.grid, L.block, 0, L.st, *L.P, *L.M,
                       none, *L.in, *L.restore, *L.lb, *L.vb, L.A, S, sb, Q, *ap);
}

void tu_exact0_dcm(const DcmLaunch& L) {
  hipLaunchKernelGGL(reflect_dcm_exact<Generic0>, L.grid, L.block, 0, L.st, *L.P1, *L.M1, *L.P2,
                     *L.M2, *L.in, *L.lo1, *L.lo2, *L.gb2, L.A1, L.A2);
}

void tu_exact0_dcm_redo_scr(const DcmLaunch& L, const xrt_hip_screen& S, const xrt_hip_beam& sb,
                            const TailApertures& ap) {
  hipLaunchKernelGGL(reflect_dcm_redo_scr<Generic0>, L.grid, L.block, 0, L.st, *L.P1, *L.M1, *L.P2,
                     *L.M2, *L.in, *L.lo1, *L.lo2, *L.gb2, L.A1, L.A2, S, sb, ap);
}

}  // namespace xrt
