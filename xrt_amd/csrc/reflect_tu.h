// The kernel templates of reflect_impl.h are instantiated in several translation units so
// that hipcc compiles them in parallel (the whole set in one file took three minutes) and a
// change to the hot kernels recompiles one small file. Each unit exports plain launchers;
// reflect.hip picks the unit by the spec id.
#pragma once
#include <string.h>

#include "reflect_impl.h"

namespace xrt {

using FlatXtal = Spec<0, XRT_HIP_SURF_FLAT, XRT_HIP_MAT_CRYSTAL, false>;
using AnyXtal = Spec<0, -1, XRT_HIP_MAT_CRYSTAL, false>;
using ToroidMirror = Spec<0, XRT_HIP_SURF_TOROID, XRT_HIP_MAT_MIRROR, true>;
using FlatMirror = Spec<0, XRT_HIP_SURF_FLAT, XRT_HIP_MAT_MIRROR, true>;
using BentMirror = Spec<0, XRT_HIP_SURF_BENTFLAT, XRT_HIP_MAT_MIRROR, true>;
using FlatPlate = Spec<0, XRT_HIP_SURF_FLAT, XRT_HIP_MAT_PLATE, true>;   // filters, windows
using ThickFlat = ThickXtal<XRT_HIP_SURF_FLAT>;
using ThickAny = ThickXtal<-1>;

enum SpecId {
  SP_GENERIC0, SP_GENERIC1, SP_GENERIC2, SP_PER_RAY_ZONES,
  SP_LAYERED0, SP_LAYERED1, SP_LAYERED2,
  SP_TOROID_MIRROR, SP_FLAT_MIRROR, SP_BENT_MIRROR, SP_FLAT_PLATE,
  SP_THICK_FLAT, SP_THICK_ANY, SP_FLAT_XTAL, SP_ANY_XTAL,
  SP_FIGURED0, SP_FIGURED1, SP_FIGURED2
};

struct FusedLaunch {   // one launch of reflect_fused / reflect_fused_xtal
  dim3 grid, block;
  hipStream_t st;
  const xrt_hip_pass* P;
  const xrt_hip_material* M;
  const xrt_hip_beam *in, *restore, *lb, *vb;
  double* theta;
  GStat* g;
  OptStat* opt;
  const xrt_hip_screen* scr;   // a screen in the tail of the pass (reflect_fused_scr), or null
  const xrt_hip_beam* sb;      //   its image
  const xrt_hip_geosource* src;   // the source in its head (reflect_fused_gen_scr), or null
  const PlotTail* plot;        // a plot behind the screen (reflect_fused_scr_plot), or null
  const TailApertures* ap;     // apertures right behind the element (n = 0: none)
};
struct ExactLaunch {   // reflect_exact
  dim3 grid, block;
  hipStream_t st;
  const xrt_hip_pass* P;
  const xrt_hip_material* M;
  const xrt_hip_beam *in, *restore, *lb, *vb;
  PassAux A;
};
struct DcmLaunch {     // reflect_fused_dcm / reflect_dcm_exact
  dim3 grid, block;
  hipStream_t st;
  const xrt_hip_pass *P1, *P2;
  const xrt_hip_material *M1, *M2;
  const xrt_hip_beam *in, *lo1, *lo2, *gb2;
  double *theta1, *theta2;
  GStat *g1, *g2;
  OptStat *opt1, *opt2;
  PassAux A1, A2;      // (the exact redo)
};

// one bounce of OE.multiple_reflect (reflect_multi_impl.h)
struct MultiAux {
  GStat* g;
  double* part;                 // partial records, one per block
  double* tang;                 // [n] ray parameter of the tangency point of each entering ray
  double* diag;                 // [16] what the host may ask about the batch decisions
  unsigned long long* counts;   // [0] rays in state 1 or 2 after the bounce, [1] in state 1
  const int32_t* nrefl_in;      // lb.nRefl before the bounce; NULL (first bounce) = zeros
  int32_t* nrefl_out;
  double* theta;                // lb.theta of this bounce (0 where the ray did not hit)
  const double* elev_in[4];     // elevationD / X / Y / Z before the bounce; NULL = the
  double* elev_out[4];          //   initial values (reflect.py:214-218); out NULL = not wanted
  double* spr[3];               // lb.s / phi / r of a parametric surface, or NULL
  // the sparse form of a bounce (reflect_multi_impl.h): the index of the entering rays in
  // segments of MULTI_SEG rays (offsets within the segment, ascending; cnt[seg] of them), the hit
  // records between its solve and its finish kernel
  int32_t* idx;
  int32_t* cnt;
  int nseg;
  double *ht, *hx, *hy, *hz;
  int32_t* hlost;
  // the optimistic form of a full bounce: reflect_multi is launched behind reflect_multi_opt
  // and opens with its verdict (gate); assume: bit 0 the hit search takes Brent, bit 1 the
  // tangency search does
  int gate;
  int assume;
};
struct MultiLaunch {   // one launch of reflect_multi
  hipStream_t st;
  const xrt_hip_pass* P;
  const xrt_hip_material* M;
  const xrt_hip_beam *in, *out;
  MultiAux A;
  int cus;             // compute units of the device
  int sparse;          // few rays still enter: index + three launches instead of the one
};

// what a user-surface unit and the library that loads it must agree on
#define XRT_USER_UNIT_ABI                                                                     \
  ((int)(sizeof(xrt_hip_pass) * 31 + sizeof(xrt_hip_material) * 17 + sizeof(xrt_hip_beam) * 5 + \
         sizeof(xrt::FusedLaunch) * 7 + sizeof(xrt::ExactLaunch) * 3 + sizeof(xrt::GStat) +    \
         sizeof(xrt::MultiLaunch) * 11))

template <class K>
inline void launch_fused_late_k(int mode, const FusedLaunch& L);
template <class K>
inline void launch_fused_k(int mode, const FusedLaunch& L) {
#ifndef XRT_FUSED_EARLY_ARGS
  launch_fused_late_k<K>(mode, L);     // (every family: the record of arguments, kernarg.h)
#else
  if (mode == 0)
    hipLaunchKernelGGL((reflect_fused<K, 0>), L.grid, L.block, 0, L.st, *L.P, *L.M, *L.in,
                       *L.restore, *L.lb, *L.vb, L.theta, L.g, L.opt);
  else
    hipLaunchKernelGGL((reflect_fused<K, 2>), L.grid, L.block, 0, L.st, *L.P, *L.M, *L.in,
                       *L.restore, *L.lb, *L.vb, L.theta, L.g, L.opt);
#endif
}
// the argument record of the kernels with a tail (reflect_impl.h: FusedTailArgs)
inline FusedTailArgs fused_tail_args(const FusedLaunch& L) {
  FusedTailArgs A;
  memset(&A, 0, sizeof(A));
  A.P = *L.P;
  A.M = *L.M;
  if (L.src) A.G = *L.src;
  A.in = *L.in;
  A.restore = *L.restore;
  A.lb = *L.lb;
  A.vb = *L.vb;
  A.theta = L.theta;
  A.gp = L.g;
  A.opt = L.opt;
  A.scr.S = *L.scr;
  A.scr.out = *L.sb;
  A.scr.ap = *L.ap;
  if (L.plot) A.Q = *L.plot;
  return A;
}
// (the plain pass of the lean kernels on the record: scr, G, Q left zero)
template <class K>
inline void launch_fused_late_k(int mode, const FusedLaunch& L) {
  FusedTailArgs A;
  memset(&A, 0, sizeof(A));
  A.P = *L.P;
  A.M = *L.M;
  A.in = *L.in;
  A.restore = *L.restore;
  A.lb = *L.lb;
  A.vb = *L.vb;
  A.theta = L.theta;
  A.gp = L.g;
  A.opt = L.opt;
  if (mode == 0)
    hipLaunchKernelGGL((reflect_fused<K, 0>), L.grid, L.block, 0, L.st, A);
  else
    hipLaunchKernelGGL((reflect_fused<K, 2>), L.grid, L.block, 0, L.st, A);
}
template <class K>
inline void launch_fused_scr_k(int mode, const FusedLaunch& L) {
  const FusedTailArgs A = fused_tail_args(L);
  if (mode == 0)
    hipLaunchKernelGGL((reflect_fused_scr<K, 0>), L.grid, L.block, 0, L.st, A);
  else
    hipLaunchKernelGGL((reflect_fused_scr<K, 2>), L.grid, L.block, 0, L.st, A);
}
template <class K>
inline void launch_fused_gen_scr_k(const FusedLaunch& L) {
  const FusedTailArgs A = fused_tail_args(L);
  hipLaunchKernelGGL(reflect_fused_gen_scr<K>, L.grid, L.block, 0, L.st, A);
}
template <class K>
inline void launch_fused_scr_plot_k(int mode, const FusedLaunch& L) {
  const FusedTailArgs A = fused_tail_args(L);
  if (mode == 0)
    hipLaunchKernelGGL((reflect_fused_scr_plot<K, 0>), L.grid, L.block, 0, L.st, A);
  else
    hipLaunchKernelGGL((reflect_fused_scr_plot<K, 2>), L.grid, L.block, 0, L.st, A);
}
template <class K>
inline void launch_fused_gen_scr_plot_k(const FusedLaunch& L) {
  const FusedTailArgs A = fused_tail_args(L);
  hipLaunchKernelGGL(reflect_fused_gen_scr_plot<K>, L.grid, L.block, 0, L.st, A);
}
template <class K>
inline void launch_xtal_k(int mode, const FusedLaunch& L) {
#ifndef XRT_FUSED_EARLY_ARGS
  FusedTailArgs A;
  memset(&A, 0, sizeof(A));
  A.P = *L.P;
  A.M = *L.M;
  A.in = *L.in;
  A.restore = *L.restore;
  A.lb = *L.lb;
  A.vb = *L.vb;
  A.theta = L.theta;
  A.gp = L.g;
  A.opt = L.opt;
  if (mode == 0)
    hipLaunchKernelGGL((reflect_fused_xtal<K, 0>), L.grid, L.block, 0, L.st, A);
  else
    hipLaunchKernelGGL((reflect_fused_xtal<K, 2>), L.grid, L.block, 0, L.st, A);
  return;
#else
  if (mode == 0)
    hipLaunchKernelGGL((reflect_fused_xtal<K, 0>), L.grid, L.block, 0, L.st, *L.P, *L.M, *L.in,
                       *L.restore, *L.lb, *L.vb, L.theta, L.g, &L.g->any_neg, L.opt);
  else
    hipLaunchKernelGGL((reflect_fused_xtal<K, 2>), L.grid, L.block, 0, L.st, *L.P, *L.M, *L.in,
                       *L.restore, *L.lb, *L.vb, L.theta, L.g, &L.g->any_neg, L.opt);
#endif
}
template <class K>
inline void launch_xtal_scr_k(int mode, const FusedLaunch& L) {
  const FusedTailArgs A = fused_tail_args(L);
  if (mode == 0)
    hipLaunchKernelGGL((reflect_fused_xtal_scr<K, 0>), L.grid, L.block, 0, L.st, A);
  else
    hipLaunchKernelGGL((reflect_fused_xtal_scr<K, 2>), L.grid, L.block, 0, L.st, A);
}
template <class K>
inline void launch_exact_k(const ExactLaunch& L) {
  hipLaunchKernelGGL(reflect_exact<K>, L.grid, L.block, 0, L.st, *L.P, *L.M, *L.in, *L.restore,
                     *L.lb, *L.vb, L.A);
}
// the argument record of the pair's kernels (reflect_impl.h: DcmTailArgs); scr left zero
inline DcmTailArgs dcm_tail_args(const DcmLaunch& L, const xrt_hip_beam& gb2) {
  DcmTailArgs A;
  memset(&A, 0, sizeof(A));
  A.P1 = *L.P1;
  A.M1 = *L.M1;
  A.P2 = *L.P2;
  A.M2 = *L.M2;
  A.in = *L.in;
  A.lo1 = *L.lo1;
  A.lo2 = *L.lo2;
  A.gb2 = gb2;
  A.theta1 = L.theta1;
  A.theta2 = L.theta2;
  A.g1p = L.g1;
  A.g2p = L.g2;
  A.flags1 = &L.g1->any_neg;
  A.flags2 = &L.g2->any_neg;
  A.opt1 = L.opt1;
  A.opt2 = L.opt2;
  return A;
}
template <class K>
inline void launch_dcm_k(const DcmLaunch& L) {
#ifdef XRT_DCM_EARLY_ARGS
  hipLaunchKernelGGL(reflect_fused_dcm<K>, L.grid, L.block, 0, L.st, *L.P1, *L.M1, *L.P2, *L.M2,
                     *L.in, *L.lo1, *L.lo2, *L.gb2, L.theta1, L.theta2, L.g1, L.g2,
                     &L.g1->any_neg, &L.g2->any_neg, L.opt1, L.opt2);
#else
  const DcmTailArgs A = dcm_tail_args(L, *L.gb2);
  hipLaunchKernelGGL(reflect_fused_dcm<K>, L.grid, L.block, 0, L.st, A);
#endif
}

template <class K>
inline void launch_dcm_scr_k(const DcmLaunch& L, const xrt_hip_beam& gb2, const xrt_hip_screen& S,
                             const xrt_hip_beam& sb, const TailApertures& ap) {
  DcmTailArgs A = dcm_tail_args(L, gb2);
  A.scr.S = S;
  A.scr.out = sb;
  A.scr.ap = ap;
  if (!sb.x)          // nothing exposes the beam: apertures alone
    hipLaunchKernelGGL(reflect_fused_dcm_marks<K>, L.grid, L.block, 0, L.st, A);
  else
    hipLaunchKernelGGL(reflect_fused_dcm_scr<K>, L.grid, L.block, 0, L.st, A);
}
template <class K>
inline void launch_plate2_k(const DcmLaunch& L) {
#ifdef XRT_DCM_EARLY_ARGS
  hipLaunchKernelGGL(reflect_fused_plate2<K>, L.grid, L.block, 0, L.st, *L.P1, *L.M1, *L.P2, *L.M2,
                     *L.in, *L.lo1, *L.lo2, *L.gb2, L.theta1, L.theta2, L.g1, L.g2, L.opt1, L.opt2);
#else
  const DcmTailArgs A = dcm_tail_args(L, *L.gb2);
  hipLaunchKernelGGL(reflect_fused_plate2<K>, L.grid, L.block, 0, L.st, A);
#endif
}

// the units (each returns false for a spec it does not hold)
bool tu_hot_fused(int spec, int mode, const FusedLaunch& L);        // reflect_hot.hip
bool tu_hot_fused_scr(int spec, int mode, const FusedLaunch& L);    // reflect_hot_scr.hip
bool tu_hot_fused_gen_scr(int spec, const FusedLaunch& L);          // reflect_hot_gen.hip
bool tu_hot_fused_scr_plot(int spec, int mode, const FusedLaunch& L);      // reflect_hot_plot.hip
bool tu_hot_fused_gen_scr_plot(int spec, const FusedLaunch& L);
bool tu_hot_xtal(int spec, int mode, const FusedLaunch& L);
bool tu_hot_xtal_scr(int spec, int mode, const FusedLaunch& L);    // reflect_hot_xtal_scr.hip
bool tu_hot_dcm(int spec, const DcmLaunch& L);
bool tu_hot_plate2(int spec, const DcmLaunch& L);                   // reflect_hot_plate2.hip
// (gb2: the global beam as the fused kernel sees it -- null arrays if nobody keeps it)
bool tu_hot_dcm_scr(int spec, const DcmLaunch& L, const xrt_hip_beam& gb2, const xrt_hip_screen& S,
                    const xrt_hip_beam& sb, const TailApertures& ap);   // reflect_hot_dcm_scr.hip
bool tu_xtal_xtal(int spec, int mode, const FusedLaunch& L);        // reflect_xtal.hip
bool tu_xtal_dcm(int spec, const DcmLaunch& L);
bool tu_generic_fused(int spec, int mode, const FusedLaunch& L);    // reflect_generic.hip
bool tu_layered_fused(int spec, int mode, const FusedLaunch& L);    // reflect_layered_f.hip
bool tu_layered_xtal(int spec, int mode, const FusedLaunch& L);     // reflect_layered_x.hip
bool tu_figured_fused(int spec, int mode, const FusedLaunch& L);    // reflect_figured_f.hip
bool tu_figured_exact0(int spec, const ExactLaunch& L);             // reflect_figured_x0.hip
bool tu_figured_exact1(int spec, const ExactLaunch& L);             // reflect_figured_x1.hip
bool tu_exact0(int spec, const ExactLaunch& L);                     // reflect_exact0.hip
void tu_exact0_redo_scr(const ExactLaunch& L, const xrt_hip_screen& S, const xrt_hip_beam& sb,
                        const xrt_hip_geosource* src, const PlotTail* plot,
                        const TailApertures* ap);
void tu_exact0_dcm(const DcmLaunch& L);
void tu_exact0_dcm_redo_scr(const DcmLaunch& L, const xrt_hip_screen& S, const xrt_hip_beam& sb,
                            const TailApertures& ap);
bool tu_exact1(int spec, const ExactLaunch& L);                     // reflect_exact1.hip
bool tu_exact2(int spec, const ExactLaunch& L);                     // reflect_exact2.hip
bool tu_exact3(int spec, const ExactLaunch& L);                     // reflect_exact3.hip
bool tu_multi(int spec, const MultiLaunch& L);                      // reflect_multi.hip

}  // namespace xrt
