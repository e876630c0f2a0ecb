// C ABI of libxrt_hip.so (see include/xrt_hip.h for the contract).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "../../include/xrt_hip.h"
#include "kirchhoff.h"
#include "reflect.h"
#include "screen.h"
#include "source.h"
#include "hist.h"
#include "undulator.h"

namespace {

thread_local char g_err[512] = "";
thread_local hipEvent_t g_next_pass_events[4] = {nullptr, nullptr, nullptr, nullptr};

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                          \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess)                                                      \
      return fail(XRT_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,             \
                  hipGetErrorString(e_), __FILE__, __LINE__);                  \
  } while (0)

// RAII device buffer for the host-pointer entry points
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace

extern "C" {

int xrt_hip_version(void) { return 100; }

const char* xrt_hip_last_error(void) { return g_err; }

int xrt_hip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(XRT_HIP_ERR_NODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  return n;
}

int xrt_hip_kirchhoff_plan(int64_t np, int64_t ns, int nsplit_req, int ppt_req,
                           size_t* workspace_bytes, int* nsplit, int* ppt) {
  if (np < 0 || ns < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  xrt::KirchhoffPlan pl = xrt::kirchhoff_plan(np, ns, nsplit_req, ppt_req);
  if (workspace_bytes) *workspace_bytes = pl.workspace_bytes();
  if (nsplit) *nsplit = pl.nsplit;
  if (ppt) *ppt = pl.ppt;
  return XRT_HIP_OK;
}

int xrt_hip_kirchhoff_f64_dev(
    int64_t np, const double* px, const double* py, const double* pz, int64_t ns,
    const double* sx, const double* sy, const double* sz, const double* nx,
    const double* ny, const double* nz, const double* nl, const double* k,
    const double* Es_ri, const double* Ep_ri, int convention, double* S_ri,
    double* P_ri, double* A_ri, double* B_ri, double* C_ri, void* workspace,
    size_t workspace_bytes, int nsplit_req, int ppt_req, void* stream,
    float* kernel_ms) {
  if (np < 0 || ns < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (ns > 2147483647LL) return fail(XRT_HIP_ERR_ARG, "more than 2^31-1 samples");
  if (convention != 0 && convention != 1)
    return fail(XRT_HIP_ERR_ARG, "convention must be 0 (numpy) or 1 (OpenCL)");
  if (np > 0 && (!px || !py || !pz || !S_ri || !P_ri || !A_ri || !B_ri || !C_ri))
    return fail(XRT_HIP_ERR_ARG, "NULL receiving-point / output pointer");
  if (ns > 0 && (!sx || !sy || !sz || !nx || !ny || !nz || !nl || !k || !Es_ri || !Ep_ri))
    return fail(XRT_HIP_ERR_ARG, "NULL sample pointer");
  xrt::KirchhoffPlan pl = xrt::kirchhoff_plan(np, ns, nsplit_req, ppt_req);
  if (!workspace || workspace_bytes < pl.workspace_bytes())
    return fail(XRT_HIP_ERR_NOMEM, "workspace %zu B < required %zu B", workspace_bytes,
                pl.workspace_bytes());
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernel_ms) {
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
  }
  hipError_t e = xrt::kirchhoff_launch(pl, np, px, py, pz, ns, sx, sy, sz, 1, nx, ny, nz,
                                       1, nl, k, Es_ri, Ep_ri, convention, S_ri, P_ri,
                                       A_ri, B_ri, C_ri, workspace, st, e0, e1);
  if (e != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return fail(XRT_HIP_ERR_HIP, "kirchhoff launch: %s", hipGetErrorString(e));
  }
  if (kernel_ms) {
    *kernel_ms = 0.f;
    if (np > 0) {
      HIP_TRY(hipEventSynchronize(e1));
      HIP_TRY(hipEventElapsedTime(kernel_ms, e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  return XRT_HIP_OK;
}

int xrt_hip_kirchhoff_f64(int ndev, const int* dev_ids, int64_t np, const double* px,
                          const double* py, const double* pz, int64_t ns,
                          const double* cos_gamma, const double* Es_ri,
                          const double* Ep_ri, const double* k, const double* pos_xyzw,
                          const double* nrm_xyzw, int convention, double* S_ri,
                          double* P_ri, double* A_ri, double* B_ri, double* C_ri,
                          float* kernel_ms) {
  if (ndev < 1 || !dev_ids) return fail(XRT_HIP_ERR_ARG, "need >=1 device id");
  if (np < 0 || ns < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (ns > 2147483647LL) return fail(XRT_HIP_ERR_ARG, "more than 2^31-1 samples");
  if (convention != 0 && convention != 1)
    return fail(XRT_HIP_ERR_ARG, "convention must be 0 (numpy) or 1 (OpenCL)");
  if (np > 0 && (!px || !py || !pz || !S_ri || !P_ri || !A_ri || !B_ri || !C_ri))
    return fail(XRT_HIP_ERR_ARG, "NULL receiving-point / output pointer");
  if (ns > 0 && (!cos_gamma || !Es_ri || !Ep_ri || !k || !pos_xyzw || !nrm_xyzw))
    return fail(XRT_HIP_ERR_ARG, "NULL sample pointer");
  int prev_dev = 0;
  HIP_TRY(hipGetDevice(&prev_dev));

  struct Slice {
    int dev;
    int64_t p0, n;
    DevBuf px, py, pz, nl, es, ep, k, pos, nrm, out[5], ws;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    xrt::KirchhoffPlan pl;
  };
  std::vector<Slice> sl(ndev);
  int rc = XRT_HIP_OK;
  double* outs[5] = {S_ri, P_ri, A_ri, B_ri, C_ri};

  auto run = [&]() -> int {
    for (int d = 0; d < ndev; ++d) {
      Slice& s = sl[d];
      s.dev = dev_ids[d];
      s.p0 = np * d / ndev;
      s.n = np * (d + 1) / ndev - s.p0;
      HIP_TRY(hipSetDevice(s.dev));
      HIP_TRY(hipStreamCreate(&s.st));
      HIP_TRY(hipEventCreate(&s.e0));
      HIP_TRY(hipEventCreate(&s.e1));
      s.pl = xrt::kirchhoff_plan(s.n, ns, 0, 0);
      const size_t pb = (size_t)s.n * sizeof(double), sb = (size_t)ns * sizeof(double);
      HIP_TRY(s.px.alloc(pb));
      HIP_TRY(s.py.alloc(pb));
      HIP_TRY(s.pz.alloc(pb));
      HIP_TRY(s.nl.alloc(sb));
      HIP_TRY(s.k.alloc(sb));
      HIP_TRY(s.es.alloc(2 * sb));
      HIP_TRY(s.ep.alloc(2 * sb));
      HIP_TRY(s.pos.alloc(4 * sb));
      HIP_TRY(s.nrm.alloc(4 * sb));
      HIP_TRY(s.ws.alloc(s.pl.workspace_bytes()));
      for (int c = 0; c < 5; ++c) HIP_TRY(s.out[c].alloc(2 * pb));
      if (s.n > 0) {
        HIP_TRY(hipMemcpyAsync(s.px.p, px + s.p0, pb, hipMemcpyHostToDevice, s.st));
        HIP_TRY(hipMemcpyAsync(s.py.p, py + s.p0, pb, hipMemcpyHostToDevice, s.st));
        HIP_TRY(hipMemcpyAsync(s.pz.p, pz + s.p0, pb, hipMemcpyHostToDevice, s.st));
      }
      if (ns > 0) {
        HIP_TRY(hipMemcpyAsync(s.nl.p, cos_gamma, sb, hipMemcpyHostToDevice, s.st));
        HIP_TRY(hipMemcpyAsync(s.k.p, k, sb, hipMemcpyHostToDevice, s.st));
        HIP_TRY(hipMemcpyAsync(s.es.p, Es_ri, 2 * sb, hipMemcpyHostToDevice, s.st));
        HIP_TRY(hipMemcpyAsync(s.ep.p, Ep_ri, 2 * sb, hipMemcpyHostToDevice, s.st));
        HIP_TRY(hipMemcpyAsync(s.pos.p, pos_xyzw, 4 * sb, hipMemcpyHostToDevice, s.st));
        HIP_TRY(hipMemcpyAsync(s.nrm.p, nrm_xyzw, 4 * sb, hipMemcpyHostToDevice, s.st));
      }
      const double* pos = s.pos.as<double>();
      const double* nrm = s.nrm.as<double>();
      hipError_t e = xrt::kirchhoff_launch(
          s.pl, s.n, s.px.as<double>(), s.py.as<double>(), s.pz.as<double>(), ns, pos,
          pos + 1, pos + 2, 4, nrm, nrm + 1, nrm + 2, 4, s.nl.as<double>(),
          s.k.as<double>(), s.es.as<double>(), s.ep.as<double>(), convention,
          s.out[0].as<double>(), s.out[1].as<double>(), s.out[2].as<double>(),
          s.out[3].as<double>(), s.out[4].as<double>(), s.ws.p, s.st, s.e0, s.e1);
      if (e != hipSuccess)
        return fail(XRT_HIP_ERR_HIP, "kirchhoff launch on device %d: %s", s.dev,
                    hipGetErrorString(e));
      if (s.n > 0)
        for (int c = 0; c < 5; ++c)
          HIP_TRY(hipMemcpyAsync(outs[c] + 2 * s.p0, s.out[c].p, 2 * pb,
                                 hipMemcpyDeviceToHost, s.st));
    }
    float worst = 0.f;
    for (int d = 0; d < ndev; ++d) {
      Slice& s = sl[d];
      HIP_TRY(hipSetDevice(s.dev));
      HIP_TRY(hipStreamSynchronize(s.st));
      if (s.n > 0) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s.e0, s.e1));
        if (ms > worst) worst = ms;
      }
    }
    if (kernel_ms) *kernel_ms = worst;
    return XRT_HIP_OK;
  };
  rc = run();
  for (int d = 0; d < ndev; ++d) {
    Slice& s = sl[d];
    if (s.st || s.e0 || s.e1) (void)hipSetDevice(s.dev);
    if (s.st) {
      (void)hipStreamSynchronize(s.st);
      (void)hipStreamDestroy(s.st);
    }
    if (s.e0) (void)hipEventDestroy(s.e0);
    if (s.e1) (void)hipEventDestroy(s.e1);
  }
  // DevBuf destructors free on whatever device is current; hipFree works across devices
  (void)hipSetDevice(prev_dev);
  return rc;
}


size_t xrt_hip_reflect_workspace_bytes(int64_t n) {
  return xrt::reflect_workspace_bytes(n < 0 ? 0 : n);
}

int xrt_hip_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(xrt_hip_beam);
    case 1: return (int)sizeof(xrt_hip_rotation);
    case 2: return (int)sizeof(xrt_hip_pass);
    case 3: return (int)sizeof(xrt_hip_material);
    case 4: return (int)sizeof(xrt_hip_screen);
    case 5: return (int)sizeof(xrt_hip_aperture);
    case 6: return (int)sizeof(xrt_hip_undulator);
    case 7: return (int)sizeof(xrt_hip_undulator_map);
    case 8: return (int)sizeof(xrt_hip_plot);
    case 9: return (int)sizeof(xrt_hip_custom_field);
    case 10: return (int)sizeof(xrt_hip_bend);
    case 11: return (int)sizeof(xrt_hip_multilayer);
    case 12: return (int)sizeof(xrt_hip_gauss);
    case 13: return (int)sizeof(xrt_hip_geosource);
    case 14: return (int)sizeof(xrt_hip_bounce);
    case 15: return (int)sizeof(xrt_hip_plot_tail);
    case 16: return (int)sizeof(xrt_hip_tail);
    default: return fail(XRT_HIP_ERR_ARG, "xrt_hip_sizeof: unknown struct %d", which);
  }
}

static int check_beam(const xrt_hip_beam* b, const char* name, int64_t n, bool need_amp) {
  if (!b) return fail(XRT_HIP_ERR_ARG, "%s: NULL beam", name);
  if (b->n != n) return fail(XRT_HIP_ERR_ARG, "%s: %lld rays, expected %lld", name,
                             (long long)b->n, (long long)n);
  if (n > 0 && (!b->x || !b->y || !b->z || !b->a || !b->b || !b->c || !b->path || !b->E ||
                !b->Jss || !b->Jpp || !b->Jsp_ri || !b->state))
    return fail(XRT_HIP_ERR_ARG, "%s: NULL field pointer", name);
  if (n > 0 && need_amp && (!b->Es_ri || !b->Ep_ri))
    return fail(XRT_HIP_ERR_ARG, "%s: Es/Ep missing although the input beam has them", name);
  return XRT_HIP_OK;
}

static int check_pass(const xrt_hip_pass* pass, const xrt_hip_material* material) {
  if (!pass || !material) return fail(XRT_HIP_ERR_ARG, "NULL pass / material");
  if (pass->to_local.n < 0 || pass->to_local.n > XRT_HIP_MAX_ROT || pass->to_virgin.n < 0 ||
      pass->to_virgin.n > XRT_HIP_MAX_ROT)
    return fail(XRT_HIP_ERR_ARG, "rotation sequence longer than %d", XRT_HIP_MAX_ROT);
  if (pass->surf_kind < XRT_HIP_SURF_FLAT || pass->surf_kind > XRT_HIP_SURF_USER)
    return fail(XRT_HIP_ERR_ARG, "unknown surface kind %d", pass->surf_kind);
  if (pass->surf_kind == XRT_HIP_SURF_USER) {
    if (!pass->user_unit)
      return fail(XRT_HIP_ERR_ARG, "user-defined surface without its compiled unit "
                                   "(xrt_hip_user_surface_load)");
    // (a crystal's atomic planes follow the user's surface: local_n gives one normal, which
    // serves as both -- the reference does the same with a three-component local_n)
    // Multilayer / Coated: Parratt's recursion lives in the LAYERED flavour of a unit only
    const xrt::UserUnit* unit = static_cast<const xrt::UserUnit*>(pass->user_unit);
    if ((material->kind == XRT_HIP_MAT_MULTILAYER) != (unit->layered != 0))
      return fail(XRT_HIP_ERR_ARG, "multilayers on user-defined surfaces need the layered "
                                   "flavour of the surface's unit, every other material the "
                                   "general one (this unit: %s)",
                  unit->layered ? "layered" : "general");
    if (pass->no_intersection_search && material->kind == XRT_HIP_MAT_CRYSTAL)
      return fail(XRT_HIP_ERR_ARG, "crystals on user-defined surfaces: not without the "
                                   "intersection search");
    if (pass->asymmetric || (pass->grating && (pass->grating != 1 || pass->g_ray_x)))
      return fail(XRT_HIP_ERR_ARG, "user-defined surfaces take plain gratings only (no zone "
                                   "plates, no asymmetric cut)");
  }
  if (pass->fe_c) {       // OE(figureError = ...)
    if (!pass->fe_tx || !pass->fe_ty || !pass->fe_cx || !pass->fe_cy || pass->fe_k < 1 ||
        pass->fe_k > 3 || pass->fe_ntx < 2 * pass->fe_k + 2 || pass->fe_nty < 2 * pass->fe_k + 2)
      return fail(XRT_HIP_ERR_ARG, "figure error: knots, coefficients and both derivative "
                                   "arrays of a spline of degree 1..3");
    for (int a = 0; a < 2; ++a)
      if (pass->fe_grid[a] && (pass->fe_k != 3 || !(pass->fe_step[a] > 0.) ||
                               !(pass->fe_hi[a] > pass->fe_lo[a])))
        return fail(XRT_HIP_ERR_ARG, "figure error: computed knots need a cubic spline and an "
                                     "ascending grid");
    if (pass->surf_kind == XRT_HIP_SURF_USER || pass->surf_kind == XRT_HIP_SURF_ELLIPSE_PARAM)
      return fail(XRT_HIP_ERR_ARG, "figure error on a parametric or user-defined surface is "
                                   "not supported");
    if (material->kind == XRT_HIP_MAT_MULTILAYER)
      return fail(XRT_HIP_ERR_ARG, "figure error under a layered material is not supported");
    if (pass->g_ray_x)
      return fail(XRT_HIP_ERR_ARG, "figure error on a general zone plate is not supported");
    if (pass->no_intersection_search && material->kind == XRT_HIP_MAT_CRYSTAL)
      return fail(XRT_HIP_ERR_ARG, "figure error on a crystal: not without the intersection "
                                   "search");
  }
  if (material->kind == XRT_HIP_MAT_CRYSTAL && material->structure == 2 && !material->cell)
    return fail(XRT_HIP_ERR_ARG, "crystal from a unit cell without its xrt_hip_cell record");
  if (material->kind == XRT_HIP_MAT_MULTILAYER) {
    if (!material->layers)
      return fail(XRT_HIP_ERR_ARG, "multilayer material without its xrt_hip_multilayer record");
    if (material->geom_bragg && !(material->d > 0.))
      return fail(XRT_HIP_ERR_ARG, "multilayer period d must be positive");
    if (pass->grating)
      return fail(XRT_HIP_ERR_ARG, "grating equation on a multilayer material");
  }
  // (crystals on conics, lenses' paraboloids, cones, VFM / DualVFM: the exact sequence of the
  // surface's family serves them -- reflect_pass_launch; a blazed profile has no Bragg planes)
  if (pass->surf_kind == XRT_HIP_SURF_BLAZED && material->kind == XRT_HIP_MAT_CRYSTAL)
    return fail(XRT_HIP_ERR_ARG, "crystals on blazed gratings are not supported");
  // grating_axis: -1 constant vector, 0 / 1 density polynomial along x / y, 2 the groove
  // function of a user-defined surface's unit
  if (pass->grating && (pass->grating_axis < -1 ||
                        pass->grating_axis > (pass->surf_kind == XRT_HIP_SURF_USER ? 2 : 1) ||
                        pass->g_ncoef < 0 || pass->g_ncoef > 8))
    return fail(XRT_HIP_ERR_ARG, "bad grating description");
  if (pass->grating && material->kind == XRT_HIP_MAT_CRYSTAL)
    return fail(XRT_HIP_ERR_ARG, "grating equation on a crystal material");
  if (pass->eff_tab_n < 0 || (pass->eff_tab_n > 0 && (pass->eff_tab_n < 2 || !pass->eff_tab_E ||
                                                     !pass->eff_tab_I || pass->eff_n < 1)))
    return fail(XRT_HIP_ERR_ARG, "efficiency table: needs >= 2 energies, both arrays and "
                                 "eff_n rows");
  if (pass->eff_tab_n > 0 && (material->kind == XRT_HIP_MAT_MULTILAYER ||
                              material->kind == XRT_HIP_MAT_CRYSTAL))
    return fail(XRT_HIP_ERR_ARG, "efficiency table with a layered or crystal material");
  if (pass->grating < 0 || pass->grating > 2 ||
      (pass->grating == 2 && !pass->g_ray_x && (pass->zone_n < 1 || !pass->zone_r)))
    return fail(XRT_HIP_ERR_ARG, "bad zone plate description");
  if ((pass->g_ray_x != nullptr) != (pass->g_ray_y != nullptr) ||
      (pass->g_ray_x && (pass->grating != 2 || pass->surf_kind >= XRT_HIP_SURF_BLAZED ||
                         material->kind == XRT_HIP_MAT_CRYSTAL ||
                         material->kind == XRT_HIP_MAT_MULTILAYER)))
    return fail(XRT_HIP_ERR_ARG, "per-ray groove vectors need both components, grating = 2, a "
                                 "flat / toroidal surface and a non-crystal material");
  if (pass->invert_normal != 1 && pass->invert_normal != -1)
    return fail(XRT_HIP_ERR_ARG, "invert_normal must be +1 or -1");
  if (pass->shape < XRT_HIP_SHAPE_RECT || pass->shape > XRT_HIP_SHAPE_POLYGON)
    return fail(XRT_HIP_ERR_ARG, "unknown shape %d", pass->shape);
  if (pass->shape == XRT_HIP_SHAPE_POLYGON && (pass->poly_n < 0 || (pass->poly_n > 0 && !pass->poly_xy)))
    return fail(XRT_HIP_ERR_ARG, "polygon shape without vertices");
  if (material->kind < XRT_HIP_MAT_NONE || material->kind > XRT_HIP_MAT_MULTILAYER)
    return fail(XRT_HIP_ERR_ARG, "unknown material kind %d", material->kind);
  if (material->kind != XRT_HIP_MAT_NONE && material->kind != XRT_HIP_MAT_MULTILAYER) {
    if (material->nelem < (material->n_fixed ? 0 : 1) || material->nelem > XRT_HIP_MAX_ELEM)
      return fail(XRT_HIP_ERR_ARG, "material needs 1..%d elements", XRT_HIP_MAX_ELEM);
    for (int e = 0; e < material->nelem; ++e)
      if (!material->tab_E[e] || !material->tab_f1[e] || !material->tab_f2[e] ||
          material->tab_n[e] < 2)
        return fail(XRT_HIP_ERR_ARG, "element %d: missing f1/f2 table", e);
    if (material->n_fixed < 0 || material->n_fixed > 2 ||
        (material->n_fixed == 2 && (!material->n_ray || material->kind == XRT_HIP_MAT_CRYSTAL)))
      return fail(XRT_HIP_ERR_ARG, "material: n_fixed %d (0 tables, 1 constant, 2 per ray with "
                                   "n_ray; not for crystals)", material->n_fixed);
  }
  return XRT_HIP_OK;
}

// the events armed by xrt_hip_reflect_time_next_pass belong to the NEXT pass call,
// whatever becomes of it: taken out of the thread-local slots before anything can fail
struct ArmedEvents {
  hipEvent_t e[4];
  ArmedEvents() {
    for (int k = 0; k < 4; ++k) {
      e[k] = g_next_pass_events[k];
      g_next_pass_events[k] = nullptr;
    }
  }
};

// events of a kernel_ms request: created together, destroyed on every way out
struct OwnEvents {
  hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
  hipError_t create() {
    for (hipEvent_t& ev : e) {
      hipError_t rc = hipEventCreate(&ev);
      if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
  }
  ~OwnEvents() {
    for (hipEvent_t ev : e)
      if (ev) (void)hipEventDestroy(ev);
  }
};

static int check_geosource(const xrt_hip_geosource* g);

static int check_plot(const xrt_hip_plot* plot, bool with_c);

static int reflect_pass_impl(const xrt_hip_pass* pass, const xrt_hip_material* material,
                             const xrt_hip_beam* in, const xrt_hip_beam* restore,
                             xrt_hip_beam* out_local, xrt_hip_beam* out_virgin, double* theta,
                             void* workspace, size_t workspace_bytes, void* stream,
                             double* info_host, float* kernel_ms,
                             const xrt_hip_screen* screen, xrt_hip_beam* out_screen,
                             int keep_virgin, int* fused,
                             const xrt_hip_geosource* source = nullptr,
                             const xrt_hip_plot_tail* tail = nullptr, int keep_screen = 1,
                             const xrt::TailApertures* ap = nullptr) {
  const ArmedEvents armed;
  int rc;
  if ((rc = check_pass(pass, material))) return rc;
  xrt_hip_beam no_image;
  if (screen) {
    if (!in) return fail(XRT_HIP_ERR_ARG, "screen without an incoming beam");
    if (tail && !keep_screen && !out_screen) {
      // (a plot behind the screen and nobody else reads the image: no image beam needed)
      memset(&no_image, 0, sizeof(no_image));
      no_image.n = in->n;
      out_screen = &no_image;
    } else {
      if (!out_screen) return fail(XRT_HIP_ERR_ARG, "screen without its image beam");
      if ((rc = check_beam(out_screen, "out_screen", in->n,
                           in->Es_ri != nullptr || in->Ep_ri != nullptr)))
        return rc;
    }
  }
  xrt::PlotTailPlan plan;
  if (tail) {
    if (!screen) return fail(XRT_HIP_ERR_ARG, "a plot in the tail of a pass shows a screen's image");
    if ((rc = check_plot(&tail->plot, false))) return rc;
    if (!xrt_hip_reflect_screen_plot_fusable(pass, material, screen, tail, in->n))
      return fail(XRT_HIP_ERR_ARG, "this pass does not carry screen and plot in its tail "
                                   "(xrt_hip_reflect_screen_plot_fusable)");
    if (xrt::plot_tail_plan(in->n, *tail, &plan, nullptr) != hipSuccess)
      return fail(XRT_HIP_ERR_ARG, "plot tail: accumulators missing or workspace below "
                                   "xrt_hip_plot_tail_workspace_bytes");
  }
  if (pass->is_multi || pass->need_elevation_map)
    return fail(XRT_HIP_ERR_ARG, "is_multi / need_elevation_map: a bounce of multiple_reflect "
                                 "goes through xrt_hip_reflect_bounce_f64_dev");
  if (!in) return fail(XRT_HIP_ERR_ARG, "NULL input beam");
  const int64_t n = in->n;
  if (n < 0) return fail(XRT_HIP_ERR_ARG, "negative ray count");
  const bool amp = in->Es_ri != nullptr || in->Ep_ri != nullptr;
  if ((rc = check_beam(in, "in", n, amp))) return rc;
  if ((rc = check_beam(restore, "restore", n, amp))) return rc;
  // out_local NULL: the beam in the element's local frame is not wanted (the reference's
  // needLocal=False, oes/reflect.py:104-108) -- 100 B per ray less to write
  xrt_hip_beam no_local;
  memset(&no_local, 0, sizeof(no_local));
  no_local.n = n;
  if (!out_local) {
    // (the layered kernels -- Multilayer AND Coated -- are compiled without the test for a
    // missing local beam, reflect_impl.h:optional_local)
    if (material->kind == XRT_HIP_MAT_MULTILAYER)
      return fail(XRT_HIP_ERR_ARG, "passes of layered materials keep their local beam "
                                   "(out_local NULL)");
    out_local = &no_local;
  } else if ((rc = check_beam(out_local, "out_local", n, amp))) {
    return rc;
  }
  if ((rc = check_beam(out_virgin, "out_virgin", n, amp))) return rc;
  if (n == 0) return XRT_HIP_OK;
  if (!workspace || workspace_bytes < xrt::reflect_workspace_bytes(n))
    return fail(XRT_HIP_ERR_NOMEM, "workspace %zu B < required %zu B", workspace_bytes,
                xrt::reflect_workspace_bytes(n));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipEvent_t e0 = nullptr, e1 = nullptr, k0 = nullptr, k1 = nullptr;
  OwnEvents own;
  if (kernel_ms) {
    HIP_TRY(own.create());
    e0 = own.e[0];
    e1 = own.e[1];
    k0 = own.e[2];
    k1 = own.e[3];
  } else {            // armed by xrt_hip_reflect_time_next_pass: record, do not wait
    e0 = armed.e[0];
    e1 = armed.e[1];
    k0 = armed.e[2];
    k1 = armed.e[3];
  }
  // info_host wants the batch statistics, which only the exact sequence collects;
  // XRT_HIP_REFLECT_EXACT=1 switches the optimistic single pass off altogether
  const char* ex = getenv("XRT_HIP_REFLECT_EXACT");
  const bool force_exact = info_host != nullptr || (ex && ex[0] == '1');
  hipError_t e = xrt::reflect_pass_launch(*pass, *material, *in, *restore, *out_local,
                                          *out_virgin, theta, workspace, st, e0, e1, k0, k1,
                                          force_exact, screen, out_screen, keep_virgin != 0,
                                          fused, source, tail ? &plan : nullptr, keep_screen != 0,
                                          ap);
  if (e != hipSuccess) return fail(XRT_HIP_ERR_HIP, "reflect launch: %s", hipGetErrorString(e));
  if (kernel_ms) {
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&kernel_ms[0], e0, e1));
    HIP_TRY(hipEventElapsedTime(&kernel_ms[1], k0, k1));
    xrt::GStat g;
    HIP_TRY(hipMemcpy(&g, workspace, sizeof(g), hipMemcpyDeviceToHost));
    kernel_ms[2] = g.redo ? 1.f : 0.f;
    if (g.hang) return fail(XRT_HIP_ERR_HIP, "reflect: a grid barrier of the exact sequence timed out");
  }
  if (info_host) {
    xrt::GStat g;
    HIP_TRY(hipMemcpyAsync(&g, workspace, sizeof(g), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int j = 0; j < 16; ++j) info_host[j] = 0.;
    info_host[0] = g.axis;
    info_host[1] = g.positive;
    info_host[2] = (g.maxdz2 > g.maxdz1 * 20.) ? 1. : 0.;
    info_host[3] = g.t1min;
    info_host[4] = g.t2max;
    info_host[5] = g.maxdz1;
    info_host[6] = g.maxdz2;
    info_host[7] = (double)g.n_enter;
    info_host[8] = (double)g.n_good1;
    info_host[9] = g.sum_bdn;
    info_host[10] = g.any_neg;
    info_host[11] = g.any_pos;
    if (g.hang) return fail(XRT_HIP_ERR_HIP, "reflect: a grid barrier of the exact sequence timed out");
  }
  return XRT_HIP_OK;
}

int xrt_hip_reflect_pass_f64_dev(const xrt_hip_pass* pass, const xrt_hip_material* material,
                                 const xrt_hip_beam* in, const xrt_hip_beam* restore,
                                 xrt_hip_beam* out_local, xrt_hip_beam* out_virgin,
                                 double* theta, void* workspace, size_t workspace_bytes,
                                 void* stream, double* info_host, float* kernel_ms) {
  return reflect_pass_impl(pass, material, in, restore, out_local, out_virgin, theta, workspace,
                           workspace_bytes, stream, info_host, kernel_ms, nullptr, nullptr, 1,
                           nullptr);
}

int xrt_hip_shine_reflect_screen_f64_dev(
    const xrt_hip_geosource* source, const xrt_hip_pass* pass, const xrt_hip_material* material,
    xrt_hip_beam* source_beam, xrt_hip_beam* out_local, xrt_hip_beam* out_virgin, double* theta,
    const xrt_hip_screen* screen, xrt_hip_beam* out_screen, int keep_virgin, void* workspace,
    size_t workspace_bytes, void* stream, int* fused) {
  int rc;
  if ((rc = check_geosource(source))) return rc;
  if (!screen) return fail(XRT_HIP_ERR_ARG, "NULL screen");
  if (screen->radius != 0. && !keep_virgin)
    return fail(XRT_HIP_ERR_ARG, "a hemispheric screen takes the stored global beam "
                                 "(keep_virgin = 1)");
  if (!pass || !pass->out_to_global || !pass->in_is_global || !source->to_global)
    return fail(XRT_HIP_ERR_ARG, "source, element and screen meet in the global frame "
                                 "(source to_global, pass in_is_global / out_to_global)");
  return reflect_pass_impl(pass, material, source_beam, source_beam, out_local, out_virgin, theta,
                           workspace, workspace_bytes, stream, nullptr, nullptr, screen,
                           out_screen, keep_virgin, fused, source);
}

int xrt_hip_reflect_screen_f64_dev(const xrt_hip_pass* pass, const xrt_hip_material* material,
                                   const xrt_hip_beam* in, const xrt_hip_beam* restore,
                                   xrt_hip_beam* out_local, xrt_hip_beam* out_virgin,
                                   double* theta, const xrt_hip_screen* screen,
                                   xrt_hip_beam* out_screen, int keep_virgin, void* workspace,
                                   size_t workspace_bytes, void* stream, int* fused,
                                   float* kernel_ms) {
  if (!screen) return fail(XRT_HIP_ERR_ARG, "NULL screen");
  if (screen->radius != 0. && !keep_virgin)
    return fail(XRT_HIP_ERR_ARG, "a hemispheric screen takes the stored global beam "
                                 "(keep_virgin = 1)");
  if (!pass || !pass->out_to_global)
    return fail(XRT_HIP_ERR_ARG, "a screen takes the beam in the global frame (out_to_global)");
  return reflect_pass_impl(pass, material, in, restore, out_local, out_virgin, theta, workspace,
                           workspace_bytes, stream, nullptr, kernel_ms, screen, out_screen,
                           keep_virgin, fused);
}

int xrt_hip_plot_tail_workspace_bytes(int64_t nrays, const xrt_hip_plot_tail* tail, size_t* bytes) {
  if (!tail || !bytes) return fail(XRT_HIP_ERR_ARG, "NULL plot tail / result");
  if (nrays < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  int rc;
  if ((rc = check_plot(&tail->plot, false))) return rc;
  *bytes = 0;
  if (nrays > 0) HIP_TRY(xrt::plot_tail_plan(nrays, *tail, nullptr, bytes));
  return XRT_HIP_OK;
}

int xrt_hip_reflect_screen_plot_fusable(const xrt_hip_pass* pass, const xrt_hip_material* material,
                                        const xrt_hip_screen* screen,
                                        const xrt_hip_plot_tail* tail, int64_t nrays) {
  if (!pass || !material || !screen || !tail || nrays <= 0) return 0;
  const char* ex = getenv("XRT_HIP_REFLECT_EXACT");
  if (ex && ex[0] == '1') return 0;
  if (!pass->out_to_global || pass->is_multi || pass->need_elevation_map) return 0;
  if (!xrt::reflect_pass_carries_screen(*pass, *material, *screen)) return 0;
  size_t need = 0;
  if (xrt::plot_tail_plan(nrays, *tail, nullptr, &need) != hipSuccess || need == 0) return 0;
  return 1;
}

int xrt_hip_reflect_screen_plot_f64_dev(
    const xrt_hip_pass* pass, const xrt_hip_material* material, const xrt_hip_beam* in,
    const xrt_hip_beam* restore, xrt_hip_beam* out_local, xrt_hip_beam* out_virgin,
    double* theta, const xrt_hip_screen* screen, xrt_hip_beam* out_screen, int keep_virgin,
    int keep_screen, const xrt_hip_plot_tail* tail, void* workspace, size_t workspace_bytes,
    void* stream, int* fused) {
  if (!screen || !tail) return fail(XRT_HIP_ERR_ARG, "NULL screen / plot tail");
  if (!pass || !pass->out_to_global)
    return fail(XRT_HIP_ERR_ARG, "a screen takes the beam in the global frame (out_to_global)");
  return reflect_pass_impl(pass, material, in, restore, out_local, out_virgin, theta, workspace,
                           workspace_bytes, stream, nullptr, nullptr, screen, out_screen,
                           keep_virgin, fused, nullptr, tail, keep_screen);
}

int xrt_hip_shine_reflect_screen_plot_f64_dev(
    const xrt_hip_geosource* source, const xrt_hip_pass* pass, const xrt_hip_material* material,
    xrt_hip_beam* source_beam, xrt_hip_beam* out_local, xrt_hip_beam* out_virgin, double* theta,
    const xrt_hip_screen* screen, xrt_hip_beam* out_screen, int keep_virgin, int keep_screen,
    const xrt_hip_plot_tail* tail, void* workspace, size_t workspace_bytes, void* stream,
    int* fused) {
  int rc;
  if ((rc = check_geosource(source))) return rc;
  if (!screen || !tail) return fail(XRT_HIP_ERR_ARG, "NULL screen / plot tail");
  if (!pass || !pass->out_to_global || !pass->in_is_global || !source->to_global)
    return fail(XRT_HIP_ERR_ARG, "source, element and screen meet in the global frame "
                                 "(source to_global, pass in_is_global / out_to_global)");
  return reflect_pass_impl(pass, material, source_beam, source_beam, out_local, out_virgin, theta,
                           workspace, workspace_bytes, stream, nullptr, nullptr, screen,
                           out_screen, keep_virgin, fused, source, tail, keep_screen);
}

int xrt_hip_reflect_tail_f64_dev(const xrt_hip_geosource* source, const xrt_hip_pass* pass,
                                 const xrt_hip_material* material, xrt_hip_beam* in,
                                 const xrt_hip_beam* restore, xrt_hip_beam* out_local,
                                 xrt_hip_beam* out_virgin, double* theta, const xrt_hip_tail* tail,
                                 int keep_virgin, void* workspace, size_t workspace_bytes,
                                 void* stream, int* fused) {
  if (!tail) return fail(XRT_HIP_ERR_ARG, "NULL tail");
  if (tail->n_apertures < 0 || tail->n_apertures > XRT_TAIL_APERTURES)
    return fail(XRT_HIP_ERR_ARG, "a tail carries 0 to %d apertures", XRT_TAIL_APERTURES);
  if (!pass || !pass->out_to_global)
    return fail(XRT_HIP_ERR_ARG, "apertures and screens take the beam in the global frame "
                                 "(out_to_global)");
  xrt::TailApertures ap;
  memset(&ap, 0, sizeof(ap));
  ap.n = tail->n_apertures;
  for (int k = 0; k < ap.n; ++k) {
    if (tail->aperture[k].poly_n > 0)
      return fail(XRT_HIP_ERR_ARG, "a polygonal aperture does not ride in a tail");
    ap.a[k] = tail->aperture[k];
  }
  if (tail->plot && !tail->screen) return fail(XRT_HIP_ERR_ARG, "a plot shows a screen's image");
  if (tail->screen && tail->screen->radius != 0. && !keep_virgin)
    return fail(XRT_HIP_ERR_ARG, "a hemispheric screen takes the stored global beam "
                                 "(keep_virgin = 1)");
  int rc;
  if (source) {
    if ((rc = check_geosource(source))) return rc;
    if (!pass->in_is_global || !source->to_global)
      return fail(XRT_HIP_ERR_ARG, "source and element meet in the global frame");
    restore = in;
  }
  return reflect_pass_impl(pass, material, in, restore, out_local, out_virgin, theta, workspace,
                           workspace_bytes, stream, nullptr, nullptr, tail->screen,
                           tail->out_screen, keep_virgin, fused, source, tail->plot,
                           tail->keep_screen, &ap);
}

int xrt_hip_double_reflect_fusable(const xrt_hip_pass* pass1, const xrt_hip_material* material1,
                                   const xrt_hip_pass* pass2,
                                   const xrt_hip_material* material2) {
  if (!pass1 || !material1 || !pass2 || !material2) return 0;
  return xrt::reflect_dcm_fusable(*pass1, *material1, *pass2, *material2) ? 1 : 0;
}

static int double_reflect_impl(const xrt_hip_pass* pass1, const xrt_hip_material* material1,
                               const xrt_hip_pass* pass2, const xrt_hip_material* material2,
                               const xrt_hip_beam* in, xrt_hip_beam* out_local1,
                               xrt_hip_beam* out_local2, xrt_hip_beam* out_global,
                               double* theta1, double* theta2, void* workspace,
                               size_t workspace_bytes, void* stream, float* kernel_ms,
                               const xrt_hip_screen* screen, xrt_hip_beam* out_screen,
                               int keep_global, const xrt::TailApertures* ap, int* fused) {
  const ArmedEvents armed;
  int rc;
  if ((rc = check_pass(pass1, material1))) return rc;
  if ((rc = check_pass(pass2, material2))) return rc;
  if (!xrt::reflect_dcm_fusable(*pass1, *material1, *pass2, *material2))
    return fail(XRT_HIP_ERR_ARG, "double_reflect: this pair of passes needs two "
                                 "xrt_hip_reflect_pass_f64_dev calls (flat Bragg crystals or the faces of a "
                                 "flat plate only)");
  if (!in) return fail(XRT_HIP_ERR_ARG, "NULL input beam");
  const int64_t n = in->n;
  if (n < 0) return fail(XRT_HIP_ERR_ARG, "negative ray count");
  const bool amp = in->Es_ri != nullptr || in->Ep_ri != nullptr;
  if ((rc = check_beam(in, "in", n, amp))) return rc;
  // out_local1 and out_local2 NULL (both or neither): the beams on the two crystals are not
  // wanted -- 200 instead of 416 B per ray
  if ((out_local1 == nullptr) != (out_local2 == nullptr))
    return fail(XRT_HIP_ERR_ARG, "double_reflect: both local beams or neither");
  xrt_hip_beam no_local;
  memset(&no_local, 0, sizeof(no_local));
  no_local.n = n;
  if (!out_local1) {
    out_local1 = out_local2 = &no_local;
    theta1 = theta2 = nullptr;
  } else {
    if ((rc = check_beam(out_local1, "out_local1", n, amp))) return rc;
    if ((rc = check_beam(out_local2, "out_local2", n, amp))) return rc;
  }
  if ((rc = check_beam(out_global, "out_global", n, amp))) return rc;
  if (screen) {
    if (!out_screen) return fail(XRT_HIP_ERR_ARG, "double_reflect: a screen without its image beam");
    if ((rc = check_beam(out_screen, "out_screen", n, amp))) return rc;
    if (screen->radius != 0. && !keep_global)
      return fail(XRT_HIP_ERR_ARG, "a hemispheric screen takes the stored global beam "
                                   "(keep_global = 1)");
  }
  if (fused) *fused = 0;
  if (n == 0) return XRT_HIP_OK;
  if (!workspace || workspace_bytes < xrt::reflect_workspace_bytes(n))
    return fail(XRT_HIP_ERR_NOMEM, "workspace %zu B < required %zu B", workspace_bytes,
                xrt::reflect_workspace_bytes(n));
  for (const xrt_hip_beam* o : {out_local1, out_local2, out_global})
    if (o->x && (o->x == in->x || o->state == in->state))
      return fail(XRT_HIP_ERR_ARG, "double_reflect: outputs must not share arrays with the input");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipEvent_t e0 = nullptr, e1 = nullptr, k0 = nullptr, k1 = nullptr;
  OwnEvents own;
  if (kernel_ms) {
    HIP_TRY(own.create());
    e0 = own.e[0];
    e1 = own.e[1];
    k0 = own.e[2];
    k1 = own.e[3];
  } else {
    e0 = armed.e[0];
    e1 = armed.e[1];
    k0 = armed.e[2];
    k1 = armed.e[3];
  }
  const char* ex = getenv("XRT_HIP_REFLECT_EXACT");
  const bool force_exact = ex && ex[0] == '1';
  hipError_t e = xrt::reflect_dcm_launch(*pass1, *material1, *pass2, *material2, *in,
                                         *out_local1, *out_local2, *out_global, theta1, theta2,
                                         workspace, st, e0, e1, k0, k1, force_exact, screen,
                                         out_screen, keep_global != 0, ap, fused);
  if (e != hipSuccess)
    return fail(XRT_HIP_ERR_HIP, "double_reflect launch: %s", hipGetErrorString(e));
  if (kernel_ms) {
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&kernel_ms[0], e0, e1));
    HIP_TRY(hipEventElapsedTime(&kernel_ms[1], k0, k1));
    xrt::GStat g[2];
    HIP_TRY(hipMemcpy(&g[0], workspace, sizeof(xrt::GStat), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&g[1], reinterpret_cast<char*>(workspace) + 256 + REFLECT_PART_BYTES,
                      sizeof(xrt::GStat), hipMemcpyDeviceToHost));
    kernel_ms[2] = (g[0].redo || g[1].redo) ? 1.f : 0.f;
    if (g[0].hang || g[1].hang)
      return fail(XRT_HIP_ERR_HIP, "double_reflect: a grid barrier of the exact sequence timed out");
  }
  return XRT_HIP_OK;
}

int xrt_hip_double_reflect_f64_dev(const xrt_hip_pass* pass1, const xrt_hip_material* material1,
                                   const xrt_hip_pass* pass2, const xrt_hip_material* material2,
                                   const xrt_hip_beam* in, xrt_hip_beam* out_local1,
                                   xrt_hip_beam* out_local2, xrt_hip_beam* out_global,
                                   double* theta1, double* theta2, void* workspace,
                                   size_t workspace_bytes, void* stream, float* kernel_ms) {
  return double_reflect_impl(pass1, material1, pass2, material2, in, out_local1, out_local2,
                             out_global, theta1, theta2, workspace, workspace_bytes, stream,
                             kernel_ms, nullptr, nullptr, 1, nullptr, nullptr);
}

int xrt_hip_double_reflect_tail_f64_dev(const xrt_hip_pass* pass1, const xrt_hip_material* material1,
                                        const xrt_hip_pass* pass2, const xrt_hip_material* material2,
                                        const xrt_hip_beam* in, xrt_hip_beam* out_local1,
                                        xrt_hip_beam* out_local2, xrt_hip_beam* out_global,
                                        double* theta1, double* theta2, const xrt_hip_tail* tail,
                                        int keep_global, void* workspace, size_t workspace_bytes,
                                        void* stream, int* fused) {
  if (!tail) return fail(XRT_HIP_ERR_ARG, "NULL tail");
  if (tail->n_apertures < 0 || tail->n_apertures > XRT_TAIL_APERTURES)
    return fail(XRT_HIP_ERR_ARG, "a tail carries 0 to %d apertures", XRT_TAIL_APERTURES);
  if (tail->plot)
    return fail(XRT_HIP_ERR_ARG, "double_reflect: a plot does not ride behind the pair "
                                 "(xrt_hip_plot_hist_ws_f64_dev on the image)");
  xrt::TailApertures ap;
  memset(&ap, 0, sizeof(ap));
  ap.n = tail->n_apertures;
  for (int k = 0; k < ap.n; ++k) {
    if (tail->aperture[k].poly_n > 0)
      return fail(XRT_HIP_ERR_ARG, "a polygonal aperture does not ride in a tail");
    ap.a[k] = tail->aperture[k];
  }
  return double_reflect_impl(pass1, material1, pass2, material2, in, out_local1, out_local2,
                             out_global, theta1, theta2, workspace, workspace_bytes, stream,
                             nullptr, tail->screen, tail->out_screen, keep_global, &ap, fused);
}

int xrt_hip_surface_eval_f64_dev(const xrt_hip_pass* pass, int what, int64_t n, const double* u,
                                 const double* v, const double* w, double* out, void* stream) {
  if (!pass) return fail(XRT_HIP_ERR_ARG, "NULL pass");
  if (what < 0 || what > 5) return fail(XRT_HIP_ERR_ARG, "surface_eval: what = %d", what);
  if (what == 5 && pass->shape == XRT_HIP_SHAPE_POLYGON && (pass->poly_n < 0 || !pass->poly_xy))
    return fail(XRT_HIP_ERR_ARG, "polygon shape without vertices");
  if (n < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (n == 0) return XRT_HIP_OK;
  if (!u || !v || !out || ((what == 3 || what == 4) && !w))
    return fail(XRT_HIP_ERR_ARG, "NULL array");
  if (pass->surf_kind < XRT_HIP_SURF_FLAT || pass->surf_kind > XRT_HIP_SURF_USER)
    return fail(XRT_HIP_ERR_ARG, "unknown surface kind %d", pass->surf_kind);
  if (pass->surf_kind == XRT_HIP_SURF_USER && (!pass->user_unit || what == 2 || what == 3 ||
                                               what == 4))
    return fail(XRT_HIP_ERR_ARG, "user-defined surface: no compiled unit, or a function of "
                                 "parametric surfaces asked for");
  HIP_TRY(xrt::surface_eval_launch(*pass, what, n, u, v, w, out,
                                   reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_local_to_global_f64_dev(const xrt_hip_pass* pass, xrt_hip_beam* beam, void* stream) {
  if (!pass || !beam) return fail(XRT_HIP_ERR_ARG, "NULL pass / beam");
  const bool amp = beam->Es_ri != nullptr || beam->Ep_ri != nullptr;
  int rc;
  if ((rc = check_beam(beam, "beam", beam->n, amp))) return rc;
  if (pass->to_virgin.n < 0 || pass->to_virgin.n > XRT_HIP_MAX_ROT)
    return fail(XRT_HIP_ERR_ARG, "rotation sequence longer than %d", XRT_HIP_MAX_ROT);
  HIP_TRY(xrt::beam_to_global_launch(*pass, *beam, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_diffract_pre_f64_dev(const xrt_hip_pass* surface, int is_oe,
                                 const xrt_hip_beam* samples, double* sx, double* sy,
                                 double* sz, double* nx, double* ny, double* nz, double* nl,
                                 double* k, double* Es_ri, double* Ep_ri, void* workspace,
                                 size_t workspace_bytes, void* stream, double* sums_host) {
  if (!surface || !samples || !sums_host) return fail(XRT_HIP_ERR_ARG, "NULL argument");
  const int64_t n = samples->n;
  int rc;
  if ((rc = check_beam(samples, "samples", n, false))) return rc;
  sums_host[0] = sums_host[1] = sums_host[2] = 0.;
  if (n == 0) return XRT_HIP_OK;
  if (!sx || !sy || !sz || !nx || !ny || !nz || !nl || !k || !Es_ri || !Ep_ri)
    return fail(XRT_HIP_ERR_ARG, "NULL output array");
  const size_t need = (size_t)DIFFRACT_PRE_MAX_BLOCKS * 3 * sizeof(double);
  if (!workspace || workspace_bytes < need)
    return fail(XRT_HIP_ERR_NOMEM, "workspace %zu B < required %zu B", workspace_bytes, need);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int nblocks = 0;
  HIP_TRY(xrt::diffract_pre_launch(*surface, is_oe, *samples, sx, sy, sz, nx, ny, nz, nl, k,
                                   Es_ri, Ep_ri, reinterpret_cast<double*>(workspace),
                                   &nblocks, st));
  double part[DIFFRACT_PRE_MAX_BLOCKS * 3];
  HIP_TRY(hipMemcpyAsync(part, workspace, (size_t)nblocks * 3 * sizeof(double),
                         hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for (int b = 0; b < nblocks; ++b)        // in block order: the same sums every time
    for (int j = 0; j < 3; ++j) sums_host[j] += part[3 * b + j];
  return XRT_HIP_OK;
}

int xrt_hip_wave_fields_f64_dev(int64_t n, double* const* fresh_ri, double* const* acc_ri,
                                const double* energy0, double scale, int from_oe,
                                xrt_hip_beam* wave, void* stream) {
  if (n < 0 || !fresh_ri || !acc_ri || !wave) return fail(XRT_HIP_ERR_ARG, "bad argument");
  int rc;
  if ((rc = check_beam(wave, "wave", n, true))) return rc;
  if (n == 0) return XRT_HIP_OK;
  for (int j = 0; j < 5; ++j)
    if (!fresh_ri[j] || !acc_ri[j]) return fail(XRT_HIP_ERR_ARG, "NULL integral array");
  if (!energy0) return fail(XRT_HIP_ERR_ARG, "NULL energy");
  HIP_TRY(xrt::wave_fields_launch(n, fresh_ri, acc_ri, energy0, scale, from_oe, *wave,
                                  reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_basis_to_global_f64_dev(const xrt_hip_screen* frame, xrt_hip_beam* beam,
                                    int with_directions, void* stream) {
  if (!frame || !beam) return fail(XRT_HIP_ERR_ARG, "NULL frame / beam");
  int rc;
  if ((rc = check_beam(beam, "beam", beam->n, false))) return rc;
  HIP_TRY(xrt::basis_to_global_launch(*frame, *beam, with_directions,
                                      reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_wave_receive_f64_dev(const xrt_hip_pass* receiver, int is_oe, xrt_hip_beam* wave,
                                 xrt_hip_beam* glo, void* stream) {
  if (!receiver || !wave || !glo) return fail(XRT_HIP_ERR_ARG, "NULL argument");
  int rc;
  if ((rc = check_beam(wave, "wave", wave->n, true))) return rc;
  if ((rc = check_beam(glo, "glo", wave->n, true))) return rc;
  if (receiver->to_local.n < 0 || receiver->to_local.n > XRT_HIP_MAX_ROT)
    return fail(XRT_HIP_ERR_ARG, "rotation sequence longer than %d", XRT_HIP_MAX_ROT);
  HIP_TRY(xrt::wave_receive_launch(*receiver, is_oe, *wave, *glo,
                                   reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

static int check_material_tables(const xrt_hip_material* m) {
  if (!m) return fail(XRT_HIP_ERR_ARG, "NULL material");
  if (m->nelem < (m->n_fixed ? 0 : 1) || m->nelem > XRT_HIP_MAX_ELEM)
    return fail(XRT_HIP_ERR_ARG, "material needs 1..%d elements", XRT_HIP_MAX_ELEM);
  if (m->n_fixed < 0 || m->n_fixed > 2 || (m->n_fixed == 2 && !m->n_ray))
    return fail(XRT_HIP_ERR_ARG, "material: n_fixed %d (0 tables, 1 constant, 2 per ray with "
                                 "n_ray)", m->n_fixed);
  for (int e = 0; e < m->nelem; ++e)
    if (!m->tab_E[e] || !m->tab_f1[e] || !m->tab_f2[e] || m->tab_n[e] < 2)
      return fail(XRT_HIP_ERR_ARG, "element %d: missing f1/f2 table", e);
  return XRT_HIP_OK;
}

int xrt_hip_material_amplitude_f64_dev(const xrt_hip_material* material, int64_t n,
                                       const double* E, const double* bdn, double* rs_ri,
                                       double* rp_ri, double* mu, double* nk, void* stream) {
  int rc = check_material_tables(material);
  if (rc) return rc;
  if (material->kind < XRT_HIP_MAT_MIRROR || material->kind > XRT_HIP_MAT_PLATE)
    return fail(XRT_HIP_ERR_ARG, "material kind %d has no Fresnel amplitude", material->kind);
  if (n < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (n == 0) return XRT_HIP_OK;
  if (!E || !bdn || !rs_ri || !rp_ri) return fail(XRT_HIP_ERR_ARG, "NULL array");
  HIP_TRY(xrt::material_amplitude_launch(*material, n, E, bdn, rs_ri, rp_ri, mu, nk,
                                         reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_multilayer_amplitude_f64_dev(const xrt_hip_material* material, int64_t n,
                                         const double* E, const double* bdn, double* rs_ri,
                                         double* rp_ri, void* stream) {
  if (!material) return fail(XRT_HIP_ERR_ARG, "NULL material");
  if (material->kind != XRT_HIP_MAT_MULTILAYER || !material->layers)
    return fail(XRT_HIP_ERR_ARG, "material is not a multilayer");
  if (n < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (n == 0) return XRT_HIP_OK;
  if (!E || !bdn || !rs_ri || !rp_ri) return fail(XRT_HIP_ERR_ARG, "NULL array");
  HIP_TRY(xrt::multilayer_amplitude_launch(*material, n, E, bdn, rs_ri, rp_ri,
                                           reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_crystal_amplitude_f64_dev(const xrt_hip_material* material, int64_t n,
                                      const double* E, const double* gamma0,
                                      const double* gammah, const double* hns, double* S_ri,
                                      double* P_ri, void* stream) {
  int rc = check_material_tables(material);
  if (rc) return rc;
  if (material->kind != XRT_HIP_MAT_CRYSTAL)
    return fail(XRT_HIP_ERR_ARG, "material is not a crystal");
  if (material->structure == 2 && !material->cell)
    return fail(XRT_HIP_ERR_ARG, "crystal from a unit cell without its xrt_hip_cell record");
  if (n < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (n == 0) return XRT_HIP_OK;
  if (!E || !gamma0 || !gammah || !hns || !S_ri || !P_ri)
    return fail(XRT_HIP_ERR_ARG, "NULL array");
  HIP_TRY(xrt::crystal_amplitude_launch(*material, n, E, gamma0, gammah, hns, S_ri, P_ri,
                                        reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

size_t xrt_hip_bounce_workspace_bytes(int64_t n) {
  return xrt::bounce_workspace_bytes(n < 0 ? 0 : n);
}

int xrt_hip_reflect_bounce_f64_dev(const xrt_hip_pass* pass, const xrt_hip_material* material,
                                   const xrt_hip_beam* in, xrt_hip_beam* out,
                                   const xrt_hip_bounce* bounce, void* workspace,
                                   size_t workspace_bytes, void* stream, int64_t* counts_host,
                                   double* info_host) {
  int rc;
  if ((rc = check_pass(pass, material))) return rc;
  if (!in || !bounce) return fail(XRT_HIP_ERR_ARG, "NULL input beam / bounce record");
  const int64_t n = in->n;
  if (n < 0) return fail(XRT_HIP_ERR_ARG, "negative ray count");
  const bool amp = in->Es_ri != nullptr || in->Ep_ri != nullptr;
  if ((rc = check_beam(in, "in", n, amp))) return rc;
  if ((rc = check_beam(out, "out", n, amp))) return rc;
  // what the bounce kernels do not hold (see xrt_hip.h)
  if (material->kind == XRT_HIP_MAT_CRYSTAL || material->kind == XRT_HIP_MAT_MULTILAYER)
    return fail(XRT_HIP_ERR_ARG, "multiple_reflect: Bragg crystals and layered materials are "
                                 "not supported (mirrors, plates, gratings, no material)");
  if (pass->surf_kind == XRT_HIP_SURF_BLAZED || pass->surf_kind == XRT_HIP_SURF_BENT_BRAGG ||
      pass->surf_kind == XRT_HIP_SURF_DICED || pass->surf_kind == XRT_HIP_SURF_SAGITTAL)
    return fail(XRT_HIP_ERR_ARG, "multiple_reflect: surface kind %d is not supported",
                pass->surf_kind);
  if (pass->grating == 2 || pass->g_ray_x || pass->state_ray || pass->order_ray)
    return fail(XRT_HIP_ERR_ARG, "multiple_reflect: no zone plates, no per-ray orders");
  if (pass->fe_c) return fail(XRT_HIP_ERR_ARG, "multiple_reflect: no figure error");
  if (pass->no_intersection_search)
    return fail(XRT_HIP_ERR_ARG, "multiple_reflect searches its intersections");
  if (pass->asymmetric) return fail(XRT_HIP_ERR_ARG, "multiple_reflect: no asymmetric cut");
  if (material->n_fixed == 2)
    return fail(XRT_HIP_ERR_ARG, "multiple_reflect: no per-ray refractive index");
  if (pass->surf_kind == XRT_HIP_SURF_USER &&
      !static_cast<const xrt::UserUnit*>(pass->user_unit)->multi)
    return fail(XRT_HIP_ERR_ARG, "multiple_reflect: the surface's unit holds no bounce kernel "
                                 "(layered flavour, or built before round 5)");
  if (pass->is_multi ? (pass->in_is_global || pass->good_mode != 1 || !bounce->nrefl_in)
                     : (pass->good_mode != 0 || bounce->nrefl_in != nullptr))
    return fail(XRT_HIP_ERR_ARG, "bounce: the first takes the beam with state > 0 (good_mode 0, "
                                 "nrefl_in NULL), the others (is_multi) the virgin-local beam "
                                 "of the bounce before with states 1, 2 (good_mode 1, nrefl_in)");
  if (!bounce->nrefl_out || !bounce->theta)
    return fail(XRT_HIP_ERR_ARG, "bounce: nrefl_out and theta are required");
  int have_out = 0, have_in = 0, have_spr = 0;
  for (int k = 0; k < 4; ++k) {
    have_out += bounce->elev_out[k] != nullptr;
    have_in += bounce->elev_in[k] != nullptr;
  }
  for (int k = 0; k < 3; ++k) have_spr += bounce->spr_out[k] != nullptr;
  if ((have_out != 0 && have_out != 4) || (have_in != 0 && have_in != 4) ||
      (have_spr != 0 && have_spr != 3) || (have_out == 4) != (pass->need_elevation_map != 0) ||
      (have_in == 4 && have_out != 4))
    return fail(XRT_HIP_ERR_ARG, "bounce: the elevation arrays come in fours (out iff "
                                 "need_elevation_map), s / phi / r in threes");
  const void* ins[] = {in->x, in->y, in->z, in->a, in->b, in->c, in->path, in->E, in->Jss,
                       in->Jpp, in->Jsp_ri, in->state, in->Es_ri, in->Ep_ri};
  const void* outs[] = {out->x, out->y, out->z, out->a, out->b, out->c, out->path, out->E,
                        out->Jss, out->Jpp, out->Jsp_ri, out->state, out->Es_ri, out->Ep_ri};
  for (const void* u : ins)
    for (const void* v : outs)
      if (u && u == v) return fail(XRT_HIP_ERR_ARG, "bounce: `out` shares an array with `in`");
  if (n == 0) {
    if (counts_host) counts_host[0] = counts_host[1] = 0;
    return XRT_HIP_OK;
  }
  if (!workspace || workspace_bytes < xrt::bounce_workspace_bytes(n))
    return fail(XRT_HIP_ERR_NOMEM, "workspace %zu B < required %zu B", workspace_bytes,
                xrt::bounce_workspace_bytes(n));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = xrt::reflect_bounce_launch(*pass, *material, *in, *out, *bounce, workspace, st,
                                            info_host != nullptr);
  if (e != hipSuccess) return fail(XRT_HIP_ERR_HIP, "bounce launch: %s", hipGetErrorString(e));
  if (counts_host || info_host) {
    unsigned long long head[32];   // counts (16 B) ... diag at byte 128 (16 doubles)
    HIP_TRY(hipMemcpyAsync(head, workspace, sizeof(head), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    xrt::GStat g;
    HIP_TRY(hipMemcpy(&g, reinterpret_cast<char*>(workspace) + 256, sizeof(g),
                      hipMemcpyDeviceToHost));
    if (g.hang) return fail(XRT_HIP_ERR_HIP, "bounce: a grid barrier gave up waiting");
    if (counts_host) {
      counts_host[0] = (int64_t)head[0];
      counts_host[1] = (int64_t)head[1];
    }
    if (info_host) memcpy(info_host, head + 16, 16 * sizeof(double));
    if (bounce->found_host) {
      const double* diag = reinterpret_cast<const double*>(head + 16);
      bounce->found_host[0] = diag[9] != 0.;
      bounce->found_host[1] = diag[10] != 0.;
      bounce->found_host[2] = diag[11] != 0.;
      bounce->found_host[3] = (int32_t)diag[12];
    }
  }
  return XRT_HIP_OK;
}

int xrt_hip_multiple_reflect_out_f64_dev(const xrt_hip_pass* pass, const xrt_hip_beam* last,
                                         const xrt_hip_beam* original, const int32_t* nrefl,
                                         xrt_hip_beam* out_global, void* stream) {
  if (!pass || !last || !nrefl) return fail(XRT_HIP_ERR_ARG, "NULL pass / beam / nrefl");
  const int64_t n = last->n;
  const bool amp = last->Es_ri != nullptr || last->Ep_ri != nullptr;
  int rc;
  if ((rc = check_beam(last, "last", n, amp))) return rc;
  if ((rc = check_beam(original, "original", n, amp))) return rc;
  if ((rc = check_beam(out_global, "out_global", n, amp))) return rc;
  HIP_TRY(xrt::multi_to_global_launch(*pass, *last, *original, nrefl, *out_global,
                                      reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_user_unit_abi(void) { return xrt::user_unit_abi(); }

int xrt_hip_user_surface_load(const char* path, void** handle) {
  if (!path || !handle) return fail(XRT_HIP_ERR_ARG, "NULL path / handle");
  *handle = nullptr;
  void* dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!dl) return fail(XRT_HIP_ERR_ARG, "cannot open the surface unit %s: %s", path, dlerror());
  auto abi = reinterpret_cast<int (*)()>(dlsym(dl, "xrt_user_unit_abi"));
  xrt::UserUnit* u = new xrt::UserUnit();
  u->dl = dl;
  u->fused = reinterpret_cast<int (*)(int, const void*)>(dlsym(dl, "xrt_user_unit_fused"));
  u->exact = reinterpret_cast<int (*)(const void*)>(dlsym(dl, "xrt_user_unit_exact"));
  u->eval = reinterpret_cast<int (*)(const xrt_hip_pass*, int, int64_t, const double*,
                                     const double*, double*, void*)>(
      dlsym(dl, "xrt_user_unit_eval"));
  u->xtal = reinterpret_cast<int (*)(int, const void*)>(dlsym(dl, "xrt_user_unit_xtal"));
  u->multi = reinterpret_cast<int (*)(const void*)>(dlsym(dl, "xrt_user_unit_multi"));
  auto flavour = reinterpret_cast<int (*)()>(dlsym(dl, "xrt_user_unit_layered"));
  u->layered = flavour ? flavour() : 0;
  if (!abi || !u->fused || !u->exact || !u->eval || (u->layered && !u->xtal)) {
    delete u;
    dlclose(dl);
    return fail(XRT_HIP_ERR_ARG, "%s is not a surface unit (entry points missing)", path);
  }
  if (abi() != xrt::user_unit_abi()) {
    const int theirs = abi();
    delete u;
    dlclose(dl);
    return fail(XRT_HIP_ERR_ARG, "%s was built against other headers (unit abi %d, library %d): "
                                 "rebuild it", path, theirs, xrt::user_unit_abi());
  }
  *handle = u;
  return XRT_HIP_OK;
}

int xrt_hip_user_surface_unload(void* handle) {
  if (!handle) return XRT_HIP_OK;
  xrt::UserUnit* u = static_cast<xrt::UserUnit*>(handle);
  // (the unit's code objects stay registered with the HIP runtime: the library is not closed,
  // kernels of it may still be in flight)
  delete u;
  return XRT_HIP_OK;
}

static int check_geosource(const xrt_hip_geosource* g) {
  if (!g) return fail(XRT_HIP_ERR_ARG, "NULL source");
  for (int k = 0; k < 5; ++k)
    if (g->law[k] < 0 || g->law[k] > XRT_HIP_LAW_NORMAL_UNIFORM)
      return fail(XRT_HIP_ERR_ARG, "source: unknown law %d of coordinate %d", g->law[k], k);
  if (g->e_law < 0 || g->e_law > 3) return fail(XRT_HIP_ERR_ARG, "source: unknown energy law");
  if (g->e_law == 3 && (g->n_lines < 1 || g->n_lines > XRT_HIP_MAX_LINES))
    return fail(XRT_HIP_ERR_ARG, "source: %d energy lines (1..%d)", g->n_lines,
                XRT_HIP_MAX_LINES);
  if (g->rot.n < 0 || g->rot.n > XRT_HIP_MAX_ROT)
    return fail(XRT_HIP_ERR_ARG, "source: %d rotation steps", g->rot.n);
  return XRT_HIP_OK;
}

int xrt_hip_geosource_shine_f64_dev(const xrt_hip_geosource* source, xrt_hip_beam* out,
                                    void* stream) {
  int rc;
  if ((rc = check_geosource(source))) return rc;
  if (!out) return fail(XRT_HIP_ERR_ARG, "NULL beam");
  if (out->n < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if ((rc = check_beam(out, "out", out->n, out->Es_ri != nullptr || out->Ep_ri != nullptr)))
    return rc;
  HIP_TRY(xrt::geosource_shine_launch(*source, *out, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_geosource_probe_f64_dev(const xrt_hip_geosource* source, int64_t n,
                                    int32_t* any_above_one, void* stream) {
  int rc;
  if ((rc = check_geosource(source))) return rc;
  if (n < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (!any_above_one) return fail(XRT_HIP_ERR_ARG, "NULL flag");
  HIP_TRY(xrt::geosource_probe_launch(*source, n, any_above_one,
                                      reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_screen_expose_f64_dev(const xrt_hip_screen* screen, const xrt_hip_beam* in,
                                  xrt_hip_beam* out, void* stream) {
  if (!screen || !in) return fail(XRT_HIP_ERR_ARG, "NULL screen / beam");
  const int64_t n = in->n;
  const bool amp = in->Es_ri != nullptr || in->Ep_ri != nullptr;
  int rc;
  if ((rc = check_beam(in, "in", n, amp))) return rc;
  if ((rc = check_beam(out, "out", n, amp))) return rc;
  HIP_TRY(xrt::screen_expose_launch(*screen, *in, *out, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_aperture_propagate_f64_dev(const xrt_hip_aperture* aperture,
                                       xrt_hip_beam* beam_inout, xrt_hip_beam* out_local,
                                       xrt_hip_beam* out_global, void* stream) {
  if (!aperture || !beam_inout) return fail(XRT_HIP_ERR_ARG, "NULL aperture / beam");
  const int64_t n = beam_inout->n;
  const bool amp = beam_inout->Es_ri != nullptr || beam_inout->Ep_ri != nullptr;
  int rc;
  if ((rc = check_beam(beam_inout, "beam", n, amp))) return rc;
  // out_local NULL: only the states of beam_inout are updated (no beam is written)
  if (!out_local && out_global)
    return fail(XRT_HIP_ERR_ARG, "out_global without out_local");
  if (out_local && (rc = check_beam(out_local, "out_local", n, amp))) return rc;
  xrt_hip_beam none;
  memset(&none, 0, sizeof(none));
  if (out_global && (rc = check_beam(out_global, "out_global", n, amp))) return rc;
  HIP_TRY(xrt::aperture_propagate_launch(*aperture, *beam_inout, out_local ? *out_local : none,
                                         out_global ? *out_global : none,
                                         reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_screen_expose_mark_f64_dev(const xrt_hip_screen* screen,
                                       const xrt_hip_aperture* aperture,
                                       xrt_hip_beam* beam_inout, xrt_hip_beam* out_screen,
                                       void* stream) {
  if (!screen || !aperture || !beam_inout)
    return fail(XRT_HIP_ERR_ARG, "NULL screen / aperture / beam");
  if (screen->radius != 0.)
    return fail(XRT_HIP_ERR_ARG, "a hemispheric screen takes its own launch "
                                 "(xrt_hip_screen_expose_f64_dev)");
  if (aperture->poly_n > 0)
    return fail(XRT_HIP_ERR_ARG, "an aperture with an outline of vertices takes its own launch "
                                 "(xrt_hip_aperture_propagate_f64_dev)");
  const int64_t n = beam_inout->n;
  const bool amp = beam_inout->Es_ri != nullptr || beam_inout->Ep_ri != nullptr;
  int rc;
  if ((rc = check_beam(beam_inout, "beam", n, amp))) return rc;
  if ((rc = check_beam(out_screen, "out_screen", n, amp))) return rc;
  HIP_TRY(xrt::screen_expose_mark_launch(*screen, *aperture, *beam_inout, *out_screen,
                                         reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_hist2d_f64_dev(const xrt_hip_beam* beam, const double* x, const double* y,
                           double x_factor, double y_factor, int ray_flags, int flux_kind,
                           double source_weight, int bins_x, double x_lo, double x_hi,
                           int bins_y, double y_lo, double y_hi, double* hist,
                           double* counters, void* stream) {
  if (!beam) return fail(XRT_HIP_ERR_ARG, "NULL beam");
  int rc;
  if ((rc = check_beam(beam, "beam", beam->n, false))) return rc;
  if (bins_x < 1 || bins_y < 1) return fail(XRT_HIP_ERR_ARG, "bins must be >= 1");
  if (!(x_hi > x_lo) || !(y_hi > y_lo)) return fail(XRT_HIP_ERR_ARG, "empty histogram range");
  if (flux_kind < 0 || flux_kind > 5) return fail(XRT_HIP_ERR_ARG, "unknown flux kind");
  if (beam->n > 0 && (!x || !y || !hist)) return fail(XRT_HIP_ERR_ARG, "NULL array");
  HIP_TRY(xrt::hist2d_launch(*beam, x, y, x_factor, y_factor, ray_flags, flux_kind,
                             source_weight, bins_x, x_lo, x_hi, bins_y, y_lo, y_hi, hist,
                             counters, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

static int check_plot(const xrt_hip_plot* plot, bool with_c) {
  if (plot->bins_x < 1 || plot->bins_y < 1 || (with_c && plot->bins_c < 1))
    return fail(XRT_HIP_ERR_ARG, "bins must be >= 1");
  if (!(plot->x_lim[1] > plot->x_lim[0]) || !(plot->y_lim[1] > plot->y_lim[0]) ||
      !(plot->c_lim[1] > plot->c_lim[0]))
    return fail(XRT_HIP_ERR_ARG, "empty histogram range");
  if (plot->flux_kind < 0 || plot->flux_kind > 5) return fail(XRT_HIP_ERR_ARG, "unknown flux kind");
  return XRT_HIP_OK;
}

int xrt_hip_plot_hist_ws_f64_dev(const xrt_hip_beam* beam, const double* x, const double* y,
                                 const double* c, const xrt_hip_plot* plot, double* hist2d,
                                 double* hist2d_rgb, double* hist_x, double* hist_y,
                                 double* hist_c, double* counters, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!beam || !plot) return fail(XRT_HIP_ERR_ARG, "NULL beam / plot");
  int rc;
  if ((rc = check_beam(beam, "beam", beam->n, false))) return rc;
  if ((rc = check_plot(plot, hist_c != nullptr))) return rc;
  if (beam->n > 0 && (!x || !y || !c || !hist2d)) return fail(XRT_HIP_ERR_ARG, "NULL array");
  HIP_TRY(xrt::plot_hist_launch(*beam, x, y, c, *plot, hist2d, hist2d_rgb, hist_x, hist_y,
                                hist_c, counters, reinterpret_cast<hipStream_t>(stream),
                                workspace, workspace ? workspace_bytes : 0, nullptr));
  return XRT_HIP_OK;
}

int xrt_hip_plot_hist_f64_dev(const xrt_hip_beam* beam, const double* x, const double* y,
                              const double* c, const xrt_hip_plot* plot, double* hist2d,
                              double* hist2d_rgb, double* hist_x, double* hist_y,
                              double* hist_c, double* counters, void* stream) {
  return xrt_hip_plot_hist_ws_f64_dev(beam, x, y, c, plot, hist2d, hist2d_rgb, hist_x, hist_y,
                                      hist_c, counters, nullptr, 0, stream);
}

int xrt_hip_plot_hist_workspace_bytes(int64_t nrays, const xrt_hip_plot* plot, int with_rgb,
                                      int with_lines, size_t* bytes) {
  if (!plot || !bytes) return fail(XRT_HIP_ERR_ARG, "NULL plot / result");
  if (nrays < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  int rc;
  if ((rc = check_plot(plot, false))) return rc;
  xrt_hip_beam beam;
  memset(&beam, 0, sizeof(beam));
  beam.n = nrays;
  double* some = reinterpret_cast<double*>(uintptr_t(256));   // (never dereferenced: size only)
  HIP_TRY(xrt::plot_hist_launch(beam, some, some, some, *plot, some, with_rgb ? some : nullptr,
                                with_lines ? some : nullptr, with_lines ? some : nullptr,
                                nullptr, with_lines ? some : nullptr, nullptr, nullptr, 0,
                                bytes));
  return XRT_HIP_OK;
}

static int check_undulator(const xrt_hip_undulator* u, int64_t nrays) {
  if (!u) return fail(XRT_HIP_ERR_ARG, "NULL undulator description");
  if (u->mode < XRT_HIP_UND_FAR || u->mode > XRT_HIP_UND_NF)
    return fail(XRT_HIP_ERR_ARG, "unknown undulator mode %d", u->mode);
  if (u->jend < 0 || nrays < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (u->mode != XRT_HIP_UND_FAR && u->nper < 1)
    return fail(XRT_HIP_ERR_ARG, "nper must be >= 1 for the taper / near-field sums");
  if (u->jend > 0 && (!u->tg || !u->ag || !u->sintg || !u->costg || !u->sintgph || !u->costgph))
    return fail(XRT_HIP_ERR_ARG, "NULL node table");
  return XRT_HIP_OK;
}

size_t xrt_hip_undulator_workspace_bytes(int64_t jend) {
  return ((size_t)(jend < 1 ? 1 : jend) * xrt::UND_NODE_DOUBLES + xrt::UND_TAB_DOUBLES) *
         sizeof(double);
}

int xrt_hip_undulator_f64_dev(const xrt_hip_undulator* u, int64_t nrays, const double* gamma,
                              const double* wu, const double* w, const double* ww1,
                              const double* ddphi, const double* ddpsi, double* Is_ri,
                              double* Ip_ri, void* workspace, size_t workspace_bytes,
                              void* stream, float* kernel_ms) {
  int rc;
  if ((rc = check_undulator(u, nrays))) return rc;
  if (nrays > 0 && (!gamma || !wu || !w || !ww1 || !ddphi || !ddpsi || !Is_ri || !Ip_ri))
    return fail(XRT_HIP_ERR_ARG, "NULL ray array");
  if (!workspace || workspace_bytes < xrt_hip_undulator_workspace_bytes(u->jend))
    return fail(XRT_HIP_ERR_ARG, "workspace too small: %zu < %zu", workspace_bytes,
                xrt_hip_undulator_workspace_bytes(u->jend));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!u->workspace_packed) HIP_TRY(xrt::undulator_pack_launch(*u, workspace, st));
  if (!kernel_ms) {
    HIP_TRY(xrt::undulator_sum_launch(*u, nrays, gamma, wu, w, ww1, ddphi, ddpsi, Is_ri, Ip_ri,
                                      workspace, st));
    return XRT_HIP_OK;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  hipError_t e = hipEventRecord(e0, st);
  if (e == hipSuccess)
    e = xrt::undulator_sum_launch(*u, nrays, gamma, wu, w, ww1, ddphi, ddpsi, Is_ri, Ip_ri,
                                  workspace, st);
  if (e == hipSuccess) e = hipEventRecord(e1, st);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  if (e == hipSuccess) e = hipEventElapsedTime(kernel_ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  HIP_TRY(e);
  return XRT_HIP_OK;
}

int xrt_hip_undulator_imap_f64_dev(const xrt_hip_undulator* u, const xrt_hip_undulator_map* m,
                                   int64_t nrays, const double* w, const double* theta,
                                   const double* psi, const double* gamma, double* I,
                                   double* Es_ri, double* Ep_ri, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  int rc;
  if ((rc = check_undulator(u, nrays))) return rc;
  if (!m) return fail(XRT_HIP_ERR_ARG, "NULL map description");
  if (!(m->L0 > 0) || !(m->gamma0 > 1) || !(m->dstep > 0))
    return fail(XRT_HIP_ERR_ARG, "undulator map: L0, gamma0 and dstep must be positive");
  if (nrays > 0 && (!w || !theta || !psi || !I || !Es_ri || !Ep_ri))
    return fail(XRT_HIP_ERR_ARG, "NULL ray array");
  if (!workspace || workspace_bytes < xrt_hip_undulator_workspace_bytes(u->jend))
    return fail(XRT_HIP_ERR_ARG, "workspace too small: %zu < %zu", workspace_bytes,
                xrt_hip_undulator_workspace_bytes(u->jend));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!u->workspace_packed) HIP_TRY(xrt::undulator_pack_launch(*u, workspace, st));
  HIP_TRY(xrt::undulator_imap_launch(*u, *m, nrays, w, theta, psi, gamma, I, Es_ri, Ep_ri,
                                     workspace, st));
  return XRT_HIP_OK;
}

int xrt_hip_undulator_f64(int device, const xrt_hip_undulator* u, int64_t nrays,
                          const double* gamma, const double* wu, const double* w,
                          const double* ww1, const double* ddphi, const double* ddpsi,
                          double* Is_ri, double* Ip_ri, float* kernel_ms) {
  int rc;
  if ((rc = check_undulator(u, nrays))) return rc;
  if (nrays == 0) return XRT_HIP_OK;
  if (!gamma || !wu || !w || !ww1 || !ddphi || !ddpsi || !Is_ri || !Ip_ri)
    return fail(XRT_HIP_ERR_ARG, "NULL ray array");
  int prev = 0;
  HIP_TRY(hipGetDevice(&prev));
  HIP_TRY(hipSetDevice(device));
  int result = XRT_HIP_OK;
  {
    const size_t rb = (size_t)nrays * sizeof(double), nb = (size_t)u->jend * sizeof(double);
    DevBuf ray[6], tab[6], out[2], ws;
    const double* hray[6] = {gamma, wu, w, ww1, ddphi, ddpsi};
    const double* htab[6] = {u->tg, u->ag, u->sintg, u->costg, u->sintgph, u->costgph};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 6 && e == hipSuccess; ++i) {
      e = ray[i].alloc(rb);
      if (e == hipSuccess) e = hipMemcpy(ray[i].p, hray[i], rb, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = tab[i].alloc(nb);
      if (e == hipSuccess && nb) e = hipMemcpy(tab[i].p, htab[i], nb, hipMemcpyHostToDevice);
    }
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = out[i].alloc(2 * rb);
    const size_t wb = xrt_hip_undulator_workspace_bytes(u->jend);
    if (e == hipSuccess) e = ws.alloc(wb);
    if (e != hipSuccess) {
      result = fail(XRT_HIP_ERR_HIP, "undulator staging failed: %s", hipGetErrorString(e));
    } else {
      xrt_hip_undulator d = *u;
      d.tg = tab[0].as<double>();
      d.ag = tab[1].as<double>();
      d.sintg = tab[2].as<double>();
      d.costg = tab[3].as<double>();
      d.sintgph = tab[4].as<double>();
      d.costgph = tab[5].as<double>();
      float ms = 0.f;
      result = xrt_hip_undulator_f64_dev(
          &d, nrays, ray[0].as<double>(), ray[1].as<double>(), ray[2].as<double>(),
          ray[3].as<double>(), ray[4].as<double>(), ray[5].as<double>(), out[0].as<double>(),
          out[1].as<double>(), ws.p, wb, nullptr, &ms);
      if (kernel_ms) *kernel_ms = ms;
      if (result == XRT_HIP_OK) {
        e = hipMemcpy(Is_ri, out[0].p, 2 * rb, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(Ip_ri, out[1].p, 2 * rb, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
          result = fail(XRT_HIP_ERR_HIP, "undulator copy-back failed: %s", hipGetErrorString(e));
      }
    }
  }
  (void)hipSetDevice(prev);
  return result;
}

static int check_custom(const xrt_hip_custom_field* f, int64_t nrays) {
  if (!f) return fail(XRT_HIP_ERR_ARG, "NULL custom-field description");
  if (f->jend < 0 || nrays < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (f->jend > 0 && (!f->tg || !f->ag || !f->Bx || !f->By || !f->Bz || !f->betax ||
                      !f->betay || !f->trajx || !f->trajy || !f->trajz))
    return fail(XRT_HIP_ERR_ARG, "NULL node table");
  if (f->near_field && !(f->R0 > 0)) return fail(XRT_HIP_ERR_ARG, "near field needs R0 > 0");
  return XRT_HIP_OK;
}

int xrt_hip_custom_field_f64_dev(const xrt_hip_custom_field* f, int64_t nrays,
                                 const double* emcg, const double* gamma, const double* w,
                                 const double* ddphi, const double* ddpsi, double* Is_ri,
                                 double* Ip_ri, void* workspace, size_t workspace_bytes,
                                 void* stream, float* kernel_ms) {
  int rc;
  if ((rc = check_custom(f, nrays))) return rc;
  if (nrays > 0 && (!emcg || !gamma || !w || !ddphi || !ddpsi || !Is_ri || !Ip_ri))
    return fail(XRT_HIP_ERR_ARG, "NULL ray array");
  if (!workspace || workspace_bytes < xrt_hip_undulator_workspace_bytes(f->jend))
    return fail(XRT_HIP_ERR_ARG, "workspace too small: %zu < %zu", workspace_bytes,
                xrt_hip_undulator_workspace_bytes(f->jend));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!kernel_ms) {
    HIP_TRY(xrt::custom_field_launch(*f, nrays, emcg, gamma, w, ddphi, ddpsi, Is_ri, Ip_ri,
                                     workspace, st, nullptr, nullptr));
    return XRT_HIP_OK;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  hipError_t e = xrt::custom_field_launch(*f, nrays, emcg, gamma, w, ddphi, ddpsi, Is_ri,
                                          Ip_ri, workspace, st, e0, e1);
  *kernel_ms = 0.f;
  if (e == hipSuccess && nrays > 0) e = hipEventSynchronize(e1);
  if (e == hipSuccess && nrays > 0) e = hipEventElapsedTime(kernel_ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  HIP_TRY(e);
  return XRT_HIP_OK;
}

int xrt_hip_trajectory_f64_dev(int filament, int64_t n, const double* wt, const double* Bx,
                               const double* By, const double* Bz, double gamma, double emcg,
                               double* betax, double* betay, double* trajx, double* trajy,
                               double* trajz, double* betam, void* stream) {
  if (n < 2) return fail(XRT_HIP_ERR_ARG, "a trajectory needs at least 2 grid points, got %lld",
                         (long long)n);
  if (!wt || !Bx || !By || !Bz || !betax || !betay || !trajx || !trajy || !trajz || !betam)
    return fail(XRT_HIP_ERR_ARG, "NULL trajectory array");
  if (filament && !(gamma > 0)) return fail(XRT_HIP_ERR_ARG, "filament trajectory needs gamma");
  HIP_TRY(xrt::trajectory_launch(filament, n, wt, Bx, By, Bz, gamma, emcg, betax, betay, trajx,
                                 trajy, trajz, betam, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_bend_imap_f64_dev(const xrt_hip_bend* m, int64_t n, const double* E,
                              const double* theta, const double* psi, const double* gamma_ray,
                              double* I, double* Es_ri, double* Ep_ri, void* stream) {
  if (!m) return fail(XRT_HIP_ERR_ARG, "NULL magnet description");
  if (n < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (n > 0 && (!E || !theta || !psi || !I || !Es_ri || !Ep_ri))
    return fail(XRT_HIP_ERR_ARG, "NULL ray array");
  if (m->wiggler && !(m->K != 0.)) return fail(XRT_HIP_ERR_ARG, "a wiggler needs K");
  HIP_TRY(xrt::bend_imap_launch(*m, n, E, theta, psi, gamma_ray, I, Es_ri, Ep_ri,
                                reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_gaussian_beam_f64_dev(const xrt_hip_gauss* g, int64_t n, const double* x,
                                  const double* y, const double* z, const double* E,
                                  const double* dS, double dS_scalar, double* amp_ri, double* a,
                                  double* b, double* c, void* stream) {
  if (!g) return fail(XRT_HIP_ERR_ARG, "NULL beam description");
  if (n < 0) return fail(XRT_HIP_ERR_ARG, "negative size");
  if (n > 0 && (!x || !y || !z || !E || !amp_ri || !a || !b || !c))
    return fail(XRT_HIP_ERR_ARG, "NULL array");
  if (!(g->w0x > 0.) || (g->astigmatic && !(g->w0z > 0.)))
    return fail(XRT_HIP_ERR_ARG, "waist size must be positive");
  if (g->mode < 0 || g->mode > 2 || (g->mode == 1 && (g->p < 0 || g->astigmatic)) ||
      (g->mode == 2 && (g->m < 0 || g->n < 0)))
    return fail(XRT_HIP_ERR_ARG, "bad mode indices");
  HIP_TRY(xrt::gauss_beam_launch(*g, n, x, y, z, E, dS, dS_scalar, amp_ri, a, b, c,
                                 reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_debug_bessel_k_f64_dev(int64_t n, const double* x, double* k13, double* k23,
                                   void* stream) {
  if (n > 0 && (!x || !k13 || !k23)) return fail(XRT_HIP_ERR_ARG, "NULL array");
  HIP_TRY(xrt::bessel_k_probe_launch(n, x, k13, k23, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_custom_field_f64(int device, const xrt_hip_custom_field* f, int64_t nrays,
                             const double* emcg, const double* gamma, const double* w,
                             const double* ddphi, const double* ddpsi, double* Is_ri,
                             double* Ip_ri, float* kernel_ms) {
  int rc;
  if ((rc = check_custom(f, nrays))) return rc;
  if (nrays == 0) return XRT_HIP_OK;
  if (!emcg || !gamma || !w || !ddphi || !ddpsi || !Is_ri || !Ip_ri)
    return fail(XRT_HIP_ERR_ARG, "NULL ray array");
  int prev = 0;
  HIP_TRY(hipGetDevice(&prev));
  HIP_TRY(hipSetDevice(device));
  int result = XRT_HIP_OK;
  {
    const size_t rb = (size_t)nrays * sizeof(double), nb = (size_t)f->jend * sizeof(double);
    DevBuf ray[5], tab[10], out[2], ws;
    const double* hray[5] = {emcg, gamma, w, ddphi, ddpsi};
    const double* htab[10] = {f->tg, f->ag, f->Bx, f->By, f->Bz, f->betax, f->betay,
                              f->trajx, f->trajy, f->trajz};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 5 && e == hipSuccess; ++i) {
      e = ray[i].alloc(rb);
      if (e == hipSuccess) e = hipMemcpy(ray[i].p, hray[i], rb, hipMemcpyHostToDevice);
    }
    for (int i = 0; i < 10 && e == hipSuccess; ++i) {
      e = tab[i].alloc(nb);
      if (e == hipSuccess && nb) e = hipMemcpy(tab[i].p, htab[i], nb, hipMemcpyHostToDevice);
    }
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = out[i].alloc(2 * rb);
    const size_t wb = xrt_hip_undulator_workspace_bytes(f->jend);
    if (e == hipSuccess) e = ws.alloc(wb);
    if (e != hipSuccess) {
      result = fail(XRT_HIP_ERR_HIP, "custom-field staging failed: %s", hipGetErrorString(e));
    } else {
      xrt_hip_custom_field d = *f;
      const double** dst[10] = {&d.tg, &d.ag, &d.Bx, &d.By, &d.Bz, &d.betax, &d.betay,
                                &d.trajx, &d.trajy, &d.trajz};
      for (int i = 0; i < 10; ++i) *dst[i] = tab[i].as<double>();
      float ms = 0.f;
      result = xrt_hip_custom_field_f64_dev(
          &d, nrays, ray[0].as<double>(), ray[1].as<double>(), ray[2].as<double>(),
          ray[3].as<double>(), ray[4].as<double>(), out[0].as<double>(), out[1].as<double>(),
          ws.p, wb, nullptr, &ms);
      if (kernel_ms) *kernel_ms = ms;
      if (result == XRT_HIP_OK) {
        e = hipMemcpy(Is_ri, out[0].p, 2 * rb, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(Ip_ri, out[1].p, 2 * rb, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
          result = fail(XRT_HIP_ERR_HIP, "custom-field copy-back failed: %s",
                        hipGetErrorString(e));
      }
    }
  }
  (void)hipSetDevice(prev);
  return result;
}

int xrt_hip_debug_sqrt_f64_dev(int64_t n, const double* x, double* r, double* rinv,
                               void* stream) {
  if (n <= 0) return XRT_HIP_OK;
  HIP_TRY(xrt::debug_sqrt_launch(n, x, r, rinv, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_debug_sqrt_seeded_f64_dev(int64_t n, const double* x, const double* seed,
                                      double* r, double* hinv, void* stream) {
  if (n <= 0) return XRT_HIP_OK;
  HIP_TRY(xrt::debug_sqrt_seeded_launch(n, x, seed, r, hinv,
                                        reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_kirchhoff_report(const void* workspace, void* stream, unsigned* flags,
                             unsigned* variants, int64_t* row) {
  if (!workspace) return fail(XRT_HIP_ERR_ARG, "NULL workspace");
  xrt::KirchhoffInfo info;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(&info, workspace, sizeof(info), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (flags) *flags = info.flags;
  if (variants) *variants = info.variants;
  if (row) *row = info.not_row ? (int64_t)~info.not_row : 0;
  return XRT_HIP_OK;
}

int xrt_hip_debug_divconst_f64_dev(int64_t n, const double* a, double b, double* q,
                                   void* stream) {
  if (n <= 0) return XRT_HIP_OK;
  HIP_TRY(xrt::debug_divconst_launch(n, a, b, 1.0 / b, q, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_debug_sincos_f64_dev(int64_t n, const double* phi, double* sn, double* cs,
                                 void* stream) {
  if (n <= 0) return XRT_HIP_OK;
  HIP_TRY(xrt::debug_sincos_launch(n, phi, sn, cs, 0, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_debug_sincos_tab_f64_dev(int64_t n, const double* phi, double* sn, double* cs,
                                     void* stream) {
  if (n <= 0) return XRT_HIP_OK;
  HIP_TRY(xrt::debug_sincos_launch(n, phi, sn, cs, 1, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_debug_sincos_tab4k_f64_dev(int64_t n, const double* phi, double* sn, double* cs,
                                     void* stream) {
  if (n <= 0) return XRT_HIP_OK;
  HIP_TRY(xrt::debug_sincos_launch(n, phi, sn, cs, 2, reinterpret_cast<hipStream_t>(stream)));
  return XRT_HIP_OK;
}

int xrt_hip_event_create(void** event) {
  if (!event) return fail(XRT_HIP_ERR_ARG, "NULL event slot");
  hipEvent_t ev;
  HIP_TRY(hipEventCreate(&ev));
  *event = ev;
  return XRT_HIP_OK;
}

int xrt_hip_event_destroy(void* event) {
  if (event) HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
  return XRT_HIP_OK;
}

int xrt_hip_event_elapsed_ms(void* begin, void* end, float* ms) {
  if (!begin || !end || !ms) return fail(XRT_HIP_ERR_ARG, "NULL event / result");
  HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(end)));
  HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(begin),
                              reinterpret_cast<hipEvent_t>(end)));
  return XRT_HIP_OK;
}

int xrt_hip_reflect_time_next_pass(void* pass_begin, void* pass_end, void* kernel_begin,
                                   void* kernel_end) {
  g_next_pass_events[0] = reinterpret_cast<hipEvent_t>(pass_begin);
  g_next_pass_events[1] = reinterpret_cast<hipEvent_t>(pass_end);
  g_next_pass_events[2] = reinterpret_cast<hipEvent_t>(kernel_begin);
  g_next_pass_events[3] = reinterpret_cast<hipEvent_t>(kernel_end);
  return XRT_HIP_OK;
}

}  // extern "C"
