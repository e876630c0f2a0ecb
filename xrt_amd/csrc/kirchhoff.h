// Internal (C++) interface between kirchhoff.hip and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define KIRCHHOFF_BLOCK 256
#define KIRCHHOFF_REC_DOUBLES 16
#ifndef KIRCHHOFF_WAVES
#define KIRCHHOFF_WAVES 4   /* waves per SIMD the stream kernel is register-budgeted for */
#endif
// sample-set / receiving-point classification, written by kirchhoff_scan
#define KIRCHHOFF_FLAG_EP 1u        /* some Ep != 0 */
#define KIRCHHOFF_FLAG_NXZ 2u       /* some normal has an x or z component */
#define KIRCHHOFF_FLAG_KVAR 4u      /* the wavenumbers are not all equal */
#define KIRCHHOFF_FLAG_PYVAR 8u     /* the receiving points do not share one y */
// bits of ppt_req above the point count (testing knobs, see include/xrt_hip.h)
#define KIRCHHOFF_OPT_NO_FAST 0x100   /* keep the general-geometry loops */
#define KIRCHHOFF_OPT_NO_SHARE 0x200  /* no per-lane sharing of the mesh column */
#define KIRCHHOFF_OPT_RELAXED 0x400   /* the general-normal loops in their relaxed form */
#define KIRCHHOFF_OPT_TAB4096 0x800   /* (internal) the launch uses the 4096-step table */

namespace xrt {

// 256 bytes in front of the sample records: what kirchhoff_scan found out about the
// launch. Everything is accumulated with atomicOr / atomicMax on zeroed memory
// (non-negative doubles order like their bit patterns).
struct KirchhoffInfo {
  unsigned flags;
  unsigned variants;                  // bit v = some wave ran loop variant v (tests)
  unsigned long long kmax, s1max;     // max |k|, max |s|_1          (table-sincos bound)
  unsigned long long sxmax, szmax;    // max |sx|, max |sz|          (paraxial bound)
  unsigned long long dyinvmax;        // max 1/|py0 - sy|
  unsigned long long pxmax, pzmax;    // max |px|, max |pz|
  unsigned long long not_row;         // ~(smallest p > 0 with px[p] == px[0]); 0: none
  double py0, k0;                     // py[0], k[0]
  unsigned opts;                      // KIRCHHOFF_OPT_* of this launch
  unsigned pad[41];
};
static_assert(sizeof(KirchhoffInfo) == 256, "info block is 256 bytes");

struct KirchhoffPlan {
  int ppt;               // receiving points per lane (1, 2 or 4)
  int opts;              // KIRCHHOFF_OPT_*
  int nsplit;            // sample splits (grid = tiles * nsplit)
  int chunk;             // samples per split
  int64_t tiles;         // blocks per split (upper bound over the possible row lengths)
  int64_t np_pad;        // row pitch of the partial-sum workspace
  size_t rec_bytes;      // info block (256 B) + packed sample records
  size_t partial_bytes;  // nsplit * 10 * np_pad doubles
  size_t workspace_bytes() const { return ((rec_bytes + 255) / 256) * 256 + partial_bytes; }
};

KirchhoffPlan kirchhoff_plan(int64_t np, int64_t ns, int nsplit_req, int ppt_req);

hipError_t kirchhoff_launch(const KirchhoffPlan& pl, int64_t np, const double* px,
                            const double* py, const double* pz, int64_t ns,
                            const double* sx, const double* sy, const double* sz,
                            int pstride, const double* nx, const double* ny,
                            const double* nz, int nstride, const double* nl,
                            const double* k, const double* Es,
                            const double* Ep, int convention, double* S, double* P,
                            double* A, double* B, double* C, void* workspace,
                            hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);

hipError_t debug_sqrt_launch(int64_t n, const double* x, double* r, double* ri,
                             hipStream_t stream);
hipError_t debug_sqrt_seeded_launch(int64_t n, const double* x, const double* seed,
                                    double* r, double* h, hipStream_t stream);
hipError_t debug_divconst_launch(int64_t n, const double* a, double b, double y, double* q,
                                 hipStream_t stream);
hipError_t debug_sincos_launch(int64_t n, const double* phi, double* sn, double* cs,
                               int table, hipStream_t stream);

}  // namespace xrt
