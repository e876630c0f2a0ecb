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
#define KIRCHHOFF_FLAG_EP 1u
#define KIRCHHOFF_FLAG_NXZ 2u

namespace xrt {

struct KirchhoffPlan {
  int ppt;               // receiving points per lane (1 or 2)
  int nsplit;            // sample splits (grid = tiles * nsplit)
  int chunk;             // samples per split
  int64_t tiles;         // pixel tiles of KIRCHHOFF_BLOCK*ppt
  int64_t np_pad;        // row pitch of the partial-sum workspace
  size_t rec_bytes;      // flags (256 B) + packed sample records
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
hipError_t debug_divconst_launch(int64_t n, const double* a, double b, double y, double* q,
                                 hipStream_t stream);
hipError_t debug_sincos_launch(int64_t n, const double* phi, double* sn, double* cs,
                               int table, hipStream_t stream);

}  // namespace xrt
