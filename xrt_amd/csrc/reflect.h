// Internal (C++) interface between reflect.hip and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/xrt_hip.h"
#include "screen.h"

#define REFLECT_BLOCK 256
// threads per block of the two big kernels (one ray per lane). Measured, three alternating
// runs each on one box: reflect_fused 64 / 128 / 256 / 512 threads -> 0.672 / 0.695 / 0.712 /
// 0.760 ms (64 costs the pass more than it gains the kernel). reflect_fused_dcm (round 3, since
// it ends without a block barrier): 128 / 256 / 512 threads -> 0.962 / 0.948 / 0.987 ms.
#ifndef REFLECT_FUSED_BLOCK
#define REFLECT_FUSED_BLOCK 128
#endif
#ifndef REFLECT_DCM_BLOCK
#define REFLECT_DCM_BLOCK 256
#endif
#define REFLECT_MAX_WAVES 16                      /* waves of the largest block (1024 lanes) */
// waves per SIMD the fused kernel is compiled for (register budget 512/N VGPRs)
#ifndef REFLECT_FUSED_WAVES
#define REFLECT_FUSED_WAVES 4
#endif
#define REFLECT_MAX_PART 8192u                    /* partial records (blocks) per reduction */
#define REFLECT_PART_DOUBLES 16                   /* widest partial record (stats_dir_y) */
#define REFLECT_PART_BYTES (REFLECT_MAX_PART * REFLECT_PART_DOUBLES * 8)

namespace xrt {

// Kernels that synchronise their blocks with grid barriers (reflect_exact, reflect_dcm_exact,
// reflect_redo_scr, reflect_multi: every block must be resident) are safe alone on the device
// and in stream order; two of them launched on DIFFERENT streams (run_ray_tracing(threads=N):
// one stream per worker, the reference's xrt/runner.py:311-320) could be co-scheduled half
// resident each and spin on one another. While a process has used ONE stream for them this
// guard costs a mutex and nothing on the device; from the moment a second stream shows up
// (one hipDeviceSynchronize, once) every such launch waits for the event recorded behind the
// previous one -- on the device, the host does not block -- so that they run one after another
// whatever their streams ("re-entrant per device or serialise"). Launches recorded
// into a HIP graph are left alone (one worker records and replays, runner.py).
class BarrierSerial {
 public:
  explicit BarrierSerial(hipStream_t st);
  ~BarrierSerial();
  BarrierSerial(const BarrierSerial&) = delete;
  BarrierSerial& operator=(const BarrierSerial&) = delete;

 private:
  hipStream_t st_;
  int dev_;
  bool chain_;
};

// batch-global decisions of one pass, kept in device memory (workspace head)
struct GStat {
  double maxa, maxb, maxc;           // max |a|,|b|,|c| over entering rays with state 1
  unsigned long long first_good;     // index of the first entering ray
  unsigned long long n_enter, n_main;
  int axis, positive;                // bracketing axis; sign of the first ray's component
  double t1min, t2max, maxdz1, maxdz2;
  int bracket_valid;                 // the bracket statistics above are already final
  int optimistic;                    // the single-pass geometry assumptions were set up
  int redo;                          // 1: the exact statistics + fused sequence must run
  int any_neg, any_pos;              // optimistic crystal pass saw beamInDotNormal <0 / >=0;
                                     // both set -> mixed batch, the exact two passes redo it
  unsigned long long n_good1;        // rays that ended in state 1
  double sum_bdn;                    // sum of beamInDotNormal over them
  double emin, emax;                 // energy range of the entering rays
  int tab_lo[XRT_HIP_MAX_ELEM];      // per element: upper_bound(E table, emin) ...
  int tab_hi[XRT_HIP_MAX_ELEM];      // ... and upper_bound(E table, emax): the f1/f2
                                     // binary search of every ray stays inside
  double win_lo, win_hi;             // energies the windows (of all elements) are valid
                                     // for: [win_lo, win_hi); -inf, +inf = the whole batch
  unsigned bar;                      // arrival counter of reflect_exact's grid barrier
  int hang;                          // a grid barrier gave up waiting (never expected)
  const struct TabFast* tab_fast;    // this pass's TabFast records (one per element), or
                                     // null: not prepared, or overwritten since (exact redo)
};

// np.interp on an element's (E, f1, f2) table for the energies of a beamline's beam, without
// touching the table: the four knots around the head ray's energy and slope / offset of the
// three intervals between them (slope = (f[j+1] - f[j]) / (E[j+1] - E[j]), the reference's own
// quotient, element.py:252-263), written once per pass by the decide kernel. They are the same
// for every ray: the fused kernels fetch them through the scalar cache and select the interval
// with two compares -- no binary search of dependent loads, no divisions per ray. A ray whose
// energy is outside [x[0], x[3]) searches the table as before.
struct TabFast {
  double x[4];
  double f1[3], s1[3], f2[3], s2[3];
};
static_assert(sizeof(TabFast) == 128, "one record per 128-byte line");

// What the optimistic fused pass reports back. Same-address atomics from 150 000
// waves serialise at the memory side (3 ms for 1e7 rays, measured), so the reports
// are spread over REFLECT_OPT_SLOTS cache lines (slot = block % slots; fire-and-forget
// unsigned max on the bit patterns of non-negative doubles) and a one-block kernel
// folds the slots. They live at the head of the partial-record area, which is idle
// while the fused kernel runs.
#define REFLECT_OPT_SLOTS 256
struct OptStat {
  unsigned long long maxdz1, maxdz2;   // max |dz| at the bracket ends (bit patterns)
  int viol;                            // a ray contradicted an assumption
  int pad[27];
};
static_assert(sizeof(OptStat) == 128, "one slot per 128-byte line");

// The kernels of a user-defined surface (XRT_HIP_SURF_USER), compiled at run time into a unit
// of their own (csrc/user_unit.hip.in) and opened by xrt_hip_user_surface_load: what
// xrt_hip_pass.user_unit points to. The launch records (reflect_tu.h) cross the boundary as
// they are; user_unit_abi() of both sides must agree.
struct UserUnit {
  void* dl;
  int (*fused)(int mode, const void* fused_launch);
  int (*exact)(const void* exact_launch);
  int (*xtal)(int mode, const void* fused_launch);   // layered flavour only (else NULL)
  int layered;                                       // the unit holds the layered kernels
  int (*eval)(const xrt_hip_pass* P, int what, int64_t n, const double* u, const double* v,
              double* o, void* stream);
  int (*multi)(const void* multi_launch);            // general flavour only (else NULL)
};
int user_unit_abi();

size_t reflect_workspace_bytes(int64_t n);

hipError_t reflect_pass_launch(const xrt_hip_pass& P, const xrt_hip_material& M,
                               const xrt_hip_beam& in, const xrt_hip_beam& restore,
                               const xrt_hip_beam& lb, const xrt_hip_beam& vb, double* theta,
                               void* workspace, hipStream_t st, hipEvent_t ev0,
                               hipEvent_t ev1, hipEvent_t evk0, hipEvent_t evk1,
                               bool force_exact, const xrt_hip_screen* scr = nullptr,
                               const xrt_hip_beam* sb = nullptr, bool keep_virgin = true,
                               int* fused = nullptr, const xrt_hip_geosource* src = nullptr,
                               const struct PlotTailPlan* plot = nullptr,
                               bool keep_screen = true,
                               const TailApertures* ap = nullptr);
// would this pass carry a screen (and a plot) in its tail: one of the lean kernels, optimistic
bool reflect_pass_carries_screen(const xrt_hip_pass& P, const xrt_hip_material& M,
                                 const xrt_hip_screen& S);

// DCM.double_reflect in one kernel: both crystals per ray, the beam between them stays
// in registers. lo1 / lo2: local beams of the two crystals, gb2: global beam after the
// second one (also the scratch of the exact two-pass redo).
bool reflect_dcm_fusable(const xrt_hip_pass& P1, const xrt_hip_material& M1,
                         const xrt_hip_pass& P2, const xrt_hip_material& M2);
hipError_t reflect_dcm_launch(const xrt_hip_pass& P1, const xrt_hip_material& M1,
                              const xrt_hip_pass& P2, const xrt_hip_material& M2,
                              const xrt_hip_beam& in, const xrt_hip_beam& lo1,
                              const xrt_hip_beam& lo2, const xrt_hip_beam& gb2,
                              double* theta1, double* theta2, void* workspace,
                              hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                              hipEvent_t evk0, hipEvent_t evk1, bool force_exact,
                              const xrt_hip_screen* scr = nullptr, const xrt_hip_beam* sb = nullptr,
                              bool keep_global = true, const TailApertures* ap = nullptr,
                              int* fused = nullptr);

// One bounce of OE.multiple_reflect (reflect_multi_impl.h); workspace: counts (16 B) | diag
// (128 B) | GStat (256 B) | partial records | tang [n].
size_t bounce_workspace_bytes(int64_t n);
hipError_t reflect_bounce_launch(const xrt_hip_pass& P, const xrt_hip_material& M,
                                 const xrt_hip_beam& in, const xrt_hip_beam& out,
                                 const xrt_hip_bounce& B, void* workspace, hipStream_t st,
                                 bool want_info = false);
hipError_t multi_to_global_launch(const xrt_hip_pass& P, const xrt_hip_beam& last,
                                  const xrt_hip_beam& orig, const int32_t* nrefl,
                                  const xrt_hip_beam& gb, hipStream_t st);

hipError_t surface_eval_launch(const xrt_hip_pass& P, int what, int64_t n, const double* u,
                               const double* v, const double* w, double* o, hipStream_t st);
hipError_t beam_to_global_launch(const xrt_hip_pass& P, const xrt_hip_beam& b, hipStream_t st);

#define DIFFRACT_PRE_MAX_BLOCKS 256
hipError_t diffract_pre_launch(const xrt_hip_pass& P, int is_oe, const xrt_hip_beam& s,
                               double* sx, double* sy, double* sz, double* nx, double* ny,
                               double* nz, double* nl, double* k, double* Es, double* Ep,
                               double* part, int* nblocks, hipStream_t st);
hipError_t wave_fields_launch(int64_t n, double* const* fresh, double* const* acc,
                              const double* energy0, double scale, int from_oe,
                              const xrt_hip_beam& w, hipStream_t st);
hipError_t basis_to_global_launch(const xrt_hip_screen& F, const xrt_hip_beam& b,
                                  int with_directions, hipStream_t st);
hipError_t wave_receive_launch(const xrt_hip_pass& P, int is_oe, const xrt_hip_beam& w,
                               const xrt_hip_beam& g, hipStream_t st);

hipError_t material_amplitude_launch(const xrt_hip_material& M, int64_t n, const double* E,
                                     const double* bdn, double* rs, double* rp, double* mu,
                                     double* nk, hipStream_t st);
hipError_t multilayer_amplitude_launch(const xrt_hip_material& M, int64_t n, const double* E,
                                       const double* bdn, double* rs, double* rp,
                                       hipStream_t st);
hipError_t crystal_amplitude_launch(const xrt_hip_material& M, int64_t n, const double* E,
                                    const double* g0, const double* gh, const double* hns,
                                    double* S, double* P, hipStream_t st);

}  // namespace xrt
