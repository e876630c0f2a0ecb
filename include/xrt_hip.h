/* xrt_hip.h — C ABI of libxrt_hip.so, the MI355X (gfx950) compute backend for the
 * hot path of xrt's raycing engine.
 *
 * Plain C, ctypes/cffi loadable: only pointers, sizes and scalars cross the
 * boundary. Every entry point returns 0 on success and a negative code on
 * failure; xrt_hip_last_error() then returns a thread-local message.
 * "_dev" entry points take DEVICE pointers (data already resident in HBM) and a
 * hipStream_t passed as void* (NULL = default stream); they are asynchronous
 * unless a timing output is requested. Entry points without "_dev" take HOST
 * pointers, are blocking, and do their own staging.
 *
 * Reference interfaces replaced (paths relative to the xrt source tree):
 *   xrt_hip_kirchhoff_f64      <- XRT_CL.run_parallel('integrate_kirchhoff', ...)
 *                                 as marshalled by _diffraction_integral_CL
 *                                 (xrt/backends/raycing/waves.py:854-896,
 *                                  myopencl.py:414-583, cl/diffract.cl:80-151)
 *   xrt_hip_kirchhoff_f64_dev  <- the same integral, numpy form
 *                                 _diffraction_integral_conv (waves.py:834-851),
 *                                 on device-resident SoA arrays
 *   xrt_hip_reflect_pass_f64_dev <- OE._reflect_local + the global<->local
 *                                 transforms around it in OE.reflect
 *                                 (oes/reflect.py:18-163, 551-1139;
 *                                  oes/base.py:801-1048, 1094-1163, 1231-1295)
 *   xrt_hip_reflect_bounce_f64_dev <- one turn of OE.multiple_reflect's loop
 *                                 (oes/reflect.py:165-264; base.py:1279-1289, 842-845)
 *   xrt_hip_screen_expose_f64_dev <- Screen.expose (screens.py:226-302)
 */
#ifndef XRT_HIP_H
#define XRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define XRT_HIP_API __attribute__((visibility("default")))
#else
#define XRT_HIP_API
#endif

#define XRT_HIP_OK 0
#define XRT_HIP_ERR_ARG (-1)     /* bad argument */
#define XRT_HIP_ERR_HIP (-2)     /* HIP runtime error (message has hipGetErrorString) */
#define XRT_HIP_ERR_NODEV (-3)   /* no usable GPU */
#define XRT_HIP_ERR_NOMEM (-4)   /* workspace too small / allocation failed */

/* ---- library ---------------------------------------------------------- */
XRT_HIP_API int xrt_hip_version(void);              /* 100*major + minor */
XRT_HIP_API int xrt_hip_device_count(void);         /* >=0, or negative error */
XRT_HIP_API const char* xrt_hip_last_error(void);   /* thread-local, never NULL */

/* ---- P2: Fresnel-Kirchhoff diffraction integral ------------------------
 * convention 0 = numpy path (+i k/4pi; waves.py:844,847)
 * convention 1 = OpenCL kernel (-i/4pi and the (1+i) factor on the direction
 *                integrals; cl/diffract.cl:136-148)                          */
#define XRT_HIP_KIRCHHOFF_NUMPY 0
#define XRT_HIP_KIRCHHOFF_OPENCL 1

/* Launch plan. nsplit_req<=0 / ppt_req<=0 = choose automatically. ppt_req: receiving
 * points per lane (1, 2 or 4), optionally OR-ed with XRT_HIP_KIRCHHOFF_NO_FAST /
 * _NO_SHARE, which keep the kernel off its specialised loops (tests compare every
 * loop with the oracle that way). Outputs may be NULL. */
#define XRT_HIP_KIRCHHOFF_NO_FAST 0x100   /* no planar / paraxial specialisation */
#define XRT_HIP_KIRCHHOFF_NO_SHARE 0x200  /* no sharing of a receiving-mesh column */
/* Opt-in: the loops for samples with general normals (mirror -> mirror transfers) in a relaxed
 * form -- d.d contracted, the root without its last correction, k folded into the phase
 * reduction: 5 of 60 issue slots per pair less. Results are no longer the doubles numpy's
 * _diffraction_integral_conv (waves.py:836-850) produces but agree with them norm-wise to
 * ~1e-8 (reported by tests/test_gpu_kirchhoff.py and bench.py); the default stays exact. */
#define XRT_HIP_KIRCHHOFF_RELAXED 0x400
XRT_HIP_API int xrt_hip_kirchhoff_plan(int64_t np, int64_t ns, int nsplit_req, int ppt_req,
                           size_t* workspace_bytes, int* nsplit, int* ppt);

/* Device-resident form. All arrays fp64. px,py,pz[np]: receiving points in the
 * diffracting element's local frame. Samples [ns]: position sx,sy,sz; surface
 * normal nx,ny,nz; nl = (ray direction).(normal) ("cosGamma"); wavenumber k
 * [1/mm]; Es, Ep complex interleaved (re,im) [2*ns]. Outputs S,P,A,B,C complex
 * interleaved [2*np] = (Es, Ep, aE, bE, cE) of the reference.
 * workspace: >= workspace_bytes from xrt_hip_kirchhoff_plan with the same
 * (np, ns, nsplit_req, ppt_req). kernel_ms: if not NULL the call synchronises
 * and returns the duration of the main kernel (HIP events on `stream`). */
XRT_HIP_API int xrt_hip_kirchhoff_f64_dev(
    int64_t np, const double* px, const double* py, const double* pz,
    int64_t ns, const double* sx, const double* sy, const double* sz,
    const double* nx, const double* ny, const double* nz, const double* nl,
    const double* k, const double* Es_ri, const double* Ep_ri, int convention,
    double* S_ri, double* P_ri, double* A_ri, double* B_ri, double* C_ri,
    void* workspace, size_t workspace_bytes, int nsplit_req, int ppt_req,
    void* stream, float* kernel_ms);

/* What the last launch on `workspace` found and did (synchronises `stream`):
 * flags: bit 0 some Ep != 0, bit 1 some normal off the y axis, bit 2 more than one
 * wavenumber, bit 3 receiving points not on one plane y = const; variants: bit v =
 * loop variant v of kirchhoff.hip ran (KV_* there); row: receiving-mesh row length
 * found (0: none). Any output may be NULL. */
XRT_HIP_API int xrt_hip_kirchhoff_report(const void* workspace, void* stream,
                                         unsigned* flags, unsigned* variants, int64_t* row);

/* Host form with exactly the reference's OpenCL marshalling
 * (waves.py:860-894): pos_xyzw / nrm_xyzw are ns*4 doubles = ns x [x,y,z,0]
 * (numpy (4,ns) order='F'). The receiving points are split evenly over
 * dev_ids[0..ndev) like XRT_CL.run_parallel_max does over OpenCL devices
 * (myopencl.py:455-533); samples are replicated. Blocking. kernel_ms (optional)
 * receives the slowest device's main-kernel time. */
XRT_HIP_API int xrt_hip_kirchhoff_f64(
    int ndev, const int* dev_ids, int64_t np, const double* px, const double* py,
    const double* pz, int64_t ns, const double* cos_gamma, const double* Es_ri,
    const double* Ep_ri, const double* k, const double* pos_xyzw,
    const double* nrm_xyzw, int convention, double* S_ri, double* P_ri,
    double* A_ri, double* B_ri, double* C_ri, float* kernel_ms);


/* ---- P1: ray-surface intersection + reflect / refract amplitudes ---------
 * One call = one pass of OE._reflect_local (oes/reflect.py:551-1139) over a
 * device-resident beam, including the global<->local transforms that
 * OE.reflect (reflect.py:104-134) and DCM.double_reflect (dcm.py:270-335) wrap
 * around it. All angles arrive as host-computed cos/sin (the reference also
 * takes np.cos/np.sin of the scalar angle, _rotate.py:48-50).               */

/* SoA ray record, sources/beams.py:153-182. Jsp, Es, Ep: interleaved (re,im).
 * Es/Ep may both be NULL (beam without field amplitudes). */
typedef struct xrt_hip_beam {
  int64_t n;
  double *x, *y, *z, *a, *b, *c, *path, *E, *Jss, *Jpp, *Jsp_ri;
  int32_t* state;
  double *Es_ri, *Ep_ri;
} xrt_hip_beam;

#define XRT_HIP_MAX_ROT 8
/* Sequence of plane rotations as rotate_beam applies them (_rotate.py:23-57):
 * axis 0 = x (pitch), 1 = y (roll), 2 = z (yaw); zero angles are left out. */
typedef struct xrt_hip_rotation {
  int32_t n;
  int32_t axis[XRT_HIP_MAX_ROT];
  double cosa[XRT_HIP_MAX_ROT];
  double sina[XRT_HIP_MAX_ROT];
} xrt_hip_rotation;

#define XRT_HIP_SURF_FLAT 0
#define XRT_HIP_SURF_TOROID 1
#define XRT_HIP_SURF_BENTFLAT 2   /* z = (y^2 - y0^2)/2/R, oes/__init__.py:240-303 */
#define XRT_HIP_SURF_BLAZED 3     /* saw-tooth grating, constant line density,
                                     oes/gratings.py:316-535 (own first-facet intersection) */
#define XRT_HIP_SURF_ELLIPSE_PARAM 4  /* Elliptical / Parabolical / HyperbolicMirrorParam,
                                     oes/parametric.py:9-716: parametric (s, phi, r) root
                                     solve, base.py:822-841; surf_p[8] selects the conic */
#define XRT_HIP_SURF_PARABOLOID 5  /* refractive lenses, oes/refractive.py:394-419, 613-617:
                                     z = (x^2 + y^2) / (4 focus), cut off at zmax; a parabolic
                                     cylinder takes x = 0 */
#define XRT_HIP_SURF_CONE 6        /* ConicalMirror, oes/__init__.py:589-636: surf_p = L0,
                                     0.25 t2t^2, redfocus t2t, -0.5 t2t, sign(t2t), redfocus,
                                     t2t, 0.5 t2t with t2t = tan(2 theta) */
#define XRT_HIP_SURF_SAGITTAL 7    /* sagittally bent cylinder z = Rs - sqrt(Rs^2 - x^2), the
                                     second crystal of DCMwithSagittalFocusing
                                     (oes/__init__.py:639-664): surf_p = Rs, Rs^2 */
#define XRT_HIP_SURF_BENT_BRAGG 8  /* bent crystal analysers, oes/bragg.py:104-343. surf_p[0] =
                                     shape: 0 cylinder with a circular cross section (Johann /
                                     JohanssonCylinder), 1 parabolic cylinder, 2 toroid (Johann /
                                     Johansson / GeneralBraggToroid), 3 sphere, 4 paraboloid
                                     (BentLaueSphere, oes/laue.py:478-507); [1] = atomic planes: 0
                                     follow the surface (Johann; turned by alpha), 1 ground
                                     (Johansson: planes of twice the radius), 2 their own radii
                                     (General), 3 across the surface (BentLaueCylinder /
                                     BentLaueSphere), 4 across, ground (GroundBentLaueCylinder);
                                     [2] Rm, [3] Rs, [4] cos(alpha), [5] sin(alpha), [6] alpha
                                     given (!= 0), [7] RmBragg, [8] RsBragg. The pass's
                                     `asymmetric` flag says whether the two normals differ. */
#define XRT_HIP_SURF_VFM 9         /* VFM, oes/__init__.py:417-478: sagittal cylinder r (levelled off
                                     beyond the optical x limits), meridional parabola R with
                                     fixed ends: surf_p = r, r^2, zMax (INFINITY: no optical
                                     limits), limPhysY[0]^2, R, limOptX[0], limOptX[1] */
#define XRT_HIP_SURF_DUALVFM 10    /* DualVFM, oes/__init__.py:480-587: two sagittal cylinders side
                                     by side (x >= 0: r1 about x1, sunk by h1; x < 0: r2, x2, h2),
                                     nowhere above z = 0: surf_p = r1 - h1, r1^2, x1, r2 - h2,
                                     r2^2, x2, limPhysY[0]^2, R */
#define XRT_HIP_SURF_DICED 11      /* DicedOE / DicedJohannToroid / DicedJohanssonToroid,
                                     oes/bragg.py:8-101, 345-375: flat facets (or, Johansson,
                                     facets ground to Rm) tangent to the base surface at the facet
                                     centres, gaps between them absorb. surf_p = base (0 flat, 2
                                     toroid), planes (0 Johann, 1 Johansson), Rm, Rs, cos(alpha),
                                     sin(alpha), alpha given, xStep, yStep, dxFacet / 2,
                                     dyFacet / 2 */
#define XRT_HIP_SURF_USER 12       /* a surface the USER defines, the reference's way of adding
                                     one: an OE subclass whose local_z / local_n are handed to the
                                     accelerated path as source snippets (oes/base.py:69-90
                                     cl_local_z / cl_local_n / cl_plist, spliced into the kernel
                                     at :552-564). Here: two C expressions over (x, y, p[]) compiled
                                     once per class into a unit of their own (hipcc, the ray
                                     kernels instantiated around them) and loaded with
                                     xrt_hip_user_surface_load; surf_p = the class's parameter
                                     list p[0..11]; xrt_hip_pass.user_unit = the handle; with
                                     grating = 1 and grating_axis = 2 the groove vector comes
                                     from the unit's local_g (the reference's cl_local_g) */
#define XRT_HIP_SHAPE_RECT 0
#define XRT_HIP_SHAPE_ROUND 1
#define XRT_HIP_SHAPE_POLYGON 2   /* optical surface outlined by a polygon in the local (x, y)
                                     plane, oes/base.py:1156-1160: matplotlib's
                                     Path.contains_points decides (crossing number with its
                                     edge rules); outside below phys_y[0] = lost, else over */
#define XRT_HIP_OVER_XMIN 1
#define XRT_HIP_OVER_XMAX 2
#define XRT_HIP_OVER_YMIN 4
#define XRT_HIP_OVER_YMAX 8

typedef struct xrt_hip_pass {
  /* entering rays: 0 = state>0 (reflect.py:110), 1 = state in {1,2} (dcm.py:297) */
  int32_t good_mode;
  /* input frame: 1 = global (translate by -center, Rz by the beamline azimuth,
   * beamline.py:230-252), 0 = already virgin local (2nd crystal of a DCM) */
  int32_t in_is_global;
  double center[3];
  double sin_az, cos_az;
  xrt_hip_rotation to_local;   /* reflect.py:617-629 */
  xrt_hip_rotation to_virgin;  /* reflect.py:1122-1132 */
  double shift[3];             /* dx, dy, dz subtracted in the local frame (:630-635) */
  int32_t invert_normal;       /* +1 / -1 (:638-641) */
  int32_t no_intersection_search;
  /* surface */
  int32_t surf_kind;
  double surf_p[12];           /* toroid: R, r, RN(1/R), RN(1/r), flag: 1 = the two
                                  reciprocals may be used (constant-divisor division);
                                  bent-flat: R, limPhysY[0]^2, RN(1/R), -, flag;
                                  blazed: rho_1 (= 1/rho), tanBlaze, tanAntiblaze, sinBlaze,
                                  cosBlaze, sinAntiblaze, cosAntiblaze, 1+tanAntiblaze/tanBlaze,
                                  blaze==pi/2, antiblaze==pi/2;
                                  conic of revolution (XRT_HIP_SURF_ELLIPSE_PARAM): y0, z0,
                                  cosGamma, sinGamma, A, B, isCylindrical, isClosed, conic
                                  (0 ellipse: A, B = ellipseA, ellipseB, parametric.py:143-157;
                                  1 parabola: A = parabParam, :411-425; 2 hyperbola: A, B =
                                  hyperbolaA, hyperbolaB, :611-622);
                                  paraboloid: 4 focus, 2 focus, zmax, 1 = zmax given,
                                  1 = parabolic cylinder */
  double n_const[6];           /* flat: [nH(3), n_surface(3)] (base.py:719-742) */
  int32_t asymmetric;          /* 1: n_const holds two different normals */
  /* limits, base.py:1094-1163 */
  int32_t shape;
  double phys_x[2], phys_y[2];
  int32_t has_opt_x, has_opt_y;
  double opt_x[2], opt_y[2];
  int32_t over_mask;
  int32_t lost_num;
  double roll;                 /* roll (+positionRoll...) used for the coherency
                                  rotation angle roll+atan2(nx,nz) (:948) */
  double cos_roll, sin_roll;   /* host np.cos / np.sin of it */
  /* output frame of the "global" beam */
  int32_t out_to_global;       /* 1: virgin local -> global for rays ending in
                                  state {1,2} (reflect.py:124-130) */
  int32_t only_state1_out;     /* 1: only state 1 (beam createdByDiffract, :121-122) */
  int32_t zero_local_not_entering; /* 1: dcm.py:298-303 (lo2 of rays that missed) */
  int32_t force_lost_out;      /* 1: rays of out_virgin that do not end in state {1,2}
                                  get state lost_num (Plate, dcm.py:304-305, 331-332) */
  /* grating equation instead of specular reflection (material kind 'grating',
   * reflect.py:840-861 with _grating_deflection :451-469, sign -1). The groove
   * vector OE.local_g (base.py:688-717): grating_axis -1 = the constant g_const;
   * 0 / 1 = line density polynomial along x / y:
   * N = g_rho0 * sum_i (i+1) g_coef[i] coord^i, g = N e_axis. */
  int32_t grating;
  int32_t grating_axis;
  int32_t grating_order;
  int32_t g_ncoef;
  double g_rho0;
  double g_coef[8];
  double g_const[3];
  /* XRT_HIP_SHAPE_POLYGON: poly_n vertices, DEVICE array of 2 * poly_n doubles (x0, y0,
   * x1, y1, ...), implicitly closed */
  int32_t poly_n;
  const double* poly_xy;
  /* a sequence of diffraction orders, one drawn per hit ray (reflect.py:455-456):
   * DEVICE array of n int32, the order of ray i at order_ray[i]; NULL = grating_order
   * for every ray. The draw itself (numpy's generator, over the rays that end in state 1)
   * stays with the caller, see xrt_amd/backends/raycing/oes.py:OE._with_ray_orders. */
  const int32_t* order_ray;
  /* grating == 2: a circular Fresnel zone plate in the local (x, y) plane
   * (NormalFZP, oes/gratings.py:10-137): zone_r = DEVICE array of the zone_n + 1 zone
   * radii r_0 = 0 < r_1 < ...; rays in zones of the wrong parity or beyond the last one are
   * lost, the others are deflected by the local zone density 1 / (r_{i+1} - r_{i-1})
   * pointing at the axis (sign +1 in the grating equation, reflect.py:857). zone_black =
   * 1: the central zone is opaque. */
  int32_t zone_n;
  int32_t zone_black;
  const double* zone_r;
  /* grating / zone-plate efficiency per diffraction order in place of the Fresnel
   * amplitudes (Material.get_grating_efficiency, materials/material.py:391-413, constant
   * values): eff_n pairs (order, sqrt(efficiency)); an order that is not listed gets 0.
   * eff_n = 0: the material's own amplitudes. */
  int32_t eff_n;
  int32_t eff_order[8];
  double eff_amp[8];
  /* GeneralFZPin0YZ (oes/gratings.py:140-313): its zones and groove densities follow from
   * statistics over the whole batch (the lowest path difference; per zone the largest |x|,
   * |y| of the rays that fell into it), so the caller works them out between two passes and
   * hands them over per ray: state_ray[i] = what a ray that hit with state 1 becomes (1 or
   * lost_num), g_ray_x / g_ray_y = its groove vector (z component 0), used with
   * grating = 2 and sign +1. All three DEVICE arrays of n, or all NULL. */
  const int32_t* state_ray;
  const double* g_ray_x;
  const double* g_ray_y;
  /* Optional memory of the element across passes (one DEVICE int32, zeroed by the caller
   * once): the root-finding method the batch statistics chose last time (0 secant, 1 Brent,
   * oes/base.py:871). The optimistic single pass assumes that method instead of always the
   * secant; the verdict kernel checks the assumption as before and stores what the
   * statistics say now. Results do not depend on it. NULL: assume the secant. */
  int32_t* method_hint;
  /* surf_kind == XRT_HIP_SURF_USER: the handle xrt_hip_user_surface_load returned. */
  void* user_unit;
  /* Material(efficiency = [[order, column], ...], efficiencyFile = ...) (material.py:335-346,
   * 391-413): the efficiency of order eff_order[k] as a function of energy, np.interp on a
   * table: eff_tab_n energies eff_tab_E[] and the rows eff_tab_I[k * eff_tab_n + j] (DEVICE
   * arrays); the amplitude is its square root. eff_tab_n = 0: the constants eff_amp[]. The
   * caller checks that every energy lies inside the table (the reference raises otherwise). */
  int32_t eff_tab_n;
  const double* eff_tab_E;
  const double* eff_tab_I;
  /* OE(figureError = ...) (oes/base.py:681-684, 744-770, 826-830; figure_error.py:207-265): a
   * height map [nm] on the surface, held as the tensor-product spline scipy's
   * RectBivariateSpline(y1d, x1d, z) makes of it -- first spline axis = the element's y --
   * of degree fe_k (1..3) on both axes. fe_ty [fe_nty] / fe_tx [fe_ntx]: its knots. Its
   * coefficients come as PAIRS of rows: with C [ncy][ncx] (ncy = fe_nty - fe_k - 1, ncx =
   * fe_ntx - fe_k - 1, row = y) fe_c [ncy][ncx][2] holds (C[i][j], C[i + 1][j]), the last
   * row paired with zeros (a kernel reads a 4 x 4 block as 8 loads of 16 bytes); fe_cy / fe_cx
   * hold in the same form the coefficients of the partial derivatives along y / x as FITPACK's
   * parder forms them: [ncy - 1][ncx][2] and [ncy][ncx - 1][2], one degree less on that axis,
   * knots without the first and the last one. All DEVICE arrays, 16-byte aligned.
   * fe_grid[a] (a = 0: y, 1: x) != 0: the knots of that axis are those of an interpolating
   * spline through a numpy linspace of N = n - 4 nodes -- x[0] four times, x[2] .. x[N - 3],
   * x[N - 1] four times with x[j] = j * fe_step[a] + fe_lo[a] (a product, then a sum) and
   * x[N - 1] = fe_hi[a] -- and the kernels compute them instead of loading them (the caller
   * has compared the knots with this formula bit for bit), and fe_inv[a][j - 1] = 1 / (j *
   * fe_step[a]), j = 1..3, stands for the division by a difference of j interior knots; 0: the
   * knots are loaded.
   * The height at (x + fe_shift[0], y + fe_shift[1]) * 1e-6 is added to local_z inside the
   * intersection search (find_dz); at the hit point the normal is turned about x by
   * atan(dz/dy) and then about y by -atan(dz/dx) (reflect.py:767-775). Arguments outside the
   * knot range evaluate at its edge (FITPACK's fpbisp). fe_c NULL: no figure error. Not with
   * parametric surfaces, user-defined surfaces or layered materials. */
  int32_t fe_ntx, fe_nty, fe_k, fe_reserved;
  const double *fe_tx, *fe_ty, *fe_c, *fe_cx, *fe_cy;
  double fe_shift[2];
  int32_t fe_grid[2];
  double fe_lo[2], fe_step[2], fe_hi[2], fe_inv[2][3];
  /* _reflect_local(..., needElevationMap, isMulti) (oes/reflect.py:551-555): a further bounce
   * of OE.multiple_reflect. is_multi = 1: the brackets of _bracketing(isMulti=True)
   * (oes/base.py:1279-1289) -- tMin = the root of ray . normal on [0, tMax], searched with
   * find_dz(derivOrder = 1) (base.py:819-821, 842-845). need_elevation_map = 1: find_dz at
   * that point goes into the beam's elevation fields (reflect.py:651-659). Read by
   * xrt_hip_reflect_bounce_f64_dev only; the single passes require both to be 0. */
  int32_t is_multi, need_elevation_map;
} xrt_hip_pass;

/* ---- user-defined surfaces -------------------------------------------------------------
 * path: a shared library made from csrc/user_unit.hip.in around the user's two snippets
 * (xrt_amd/usersurf.py writes and compiles it: `hipcc --offload-arch=gfx950 -shared`). It is
 * opened with dlopen, checked against this library's build (xrt_hip_user_unit_abi) and kept
 * until xrt_hip_user_surface_unload. The handle is an opaque pointer; it is only valid in the
 * process that loaded it. A unit comes in two flavours: the general one (material kind read at
 * run time: mirrors, plates, gratings, Bragg crystals -- local_n's normal serves the atomic
 * planes as well) and the layered one, compiled around Parratt's recursion for
 * XRT_HIP_MAT_MULTILAYER; a pass whose material does not fit the unit's flavour is refused. */
XRT_HIP_API int xrt_hip_user_unit_abi(void);
XRT_HIP_API int xrt_hip_user_surface_load(const char* path, void** handle);
XRT_HIP_API int xrt_hip_user_surface_unload(void* handle);

#define XRT_HIP_MAT_NONE 0
#define XRT_HIP_MAT_MIRROR 1
#define XRT_HIP_MAT_THIN_MIRROR 2
#define XRT_HIP_MAT_PLATE 3
#define XRT_HIP_MAT_CRYSTAL 4
#define XRT_HIP_MAT_MULTILAYER 5 /* Multilayer / GradedMultilayer / Coated: xrt_hip_material.layers */
#define XRT_HIP_MAX_ELEM 4
#define XRT_HIP_BUCKETS 1280
#define XRT_HIP_BUCKET_SHIFT 46
#define XRT_HIP_BUCKET_KEY0 (0x3FF << 6) /* the bits of 1.0 >> XRT_HIP_BUCKET_SHIFT */

struct xrt_hip_multilayer;

/* CrystalFromCell (crystals_basic.py:424-440): per element e of the material (its atoms in
 * the cell, their fractions w_j and positions r_j) the sums w = sum w_j, s = sum w_j
 * exp(2 pi i r_j.hkl), sm = the same with -r_j, and f0 at sin(theta)/lambda = 1/2d.
 * F0 = factDW sum_e w (Z + f1 + i f2), F_hkl = factDW sum_e (f0 + f1 + i f2) s, F_-h-k-l
 * with sm. */
typedef struct xrt_hip_cell {
  double w[4];
  double f0[4];
  double s[4][2];
  double sm[4][2];
} xrt_hip_cell;

typedef struct xrt_hip_material {
  int32_t kind;
  int32_t from_vacuum;
  /* elements: Z, stoichiometric quantity, device pointers to the tabulated
   * E, f1, f2 (element.py:252-263) */
  int32_t nelem;
  int32_t Z[XRT_HIP_MAX_ELEM];
  int32_t tab_n[XRT_HIP_MAX_ELEM];
  double quantity[XRT_HIP_MAX_ELEM];
  const double* tab_E[XRT_HIP_MAX_ELEM];
  const double* tab_f1[XRT_HIP_MAX_ELEM];
  const double* tab_f2[XRT_HIP_MAX_ELEM];
  /* optional (NULL: plain binary search) -- a coarse index of tab_E that brings the
   * search down to a step or two: tab_bucket[e][k], k = 0..XRT_HIP_BUCKETS, = number of
   * table energies <= the k-th bucket edge; the edges are the doubles whose top 18 bits
   * are XRT_HIP_BUCKET_KEY0 + k and whose other bits are 0 (64 edges per octave from
   * 1 eV to 2^20 eV). */
  const int32_t* tab_bucket[XRT_HIP_MAX_ELEM];
  double f0_hkl;               /* crystals: f0(sin(theta)/lambda = 1/2d) of element 0
                                  (element.py:203-207), a per-crystal constant */
  double d2f_re, d2f_im;       /* crystals: 1 + exp(i pi/2 (h+k+l)), crystals_basic.py:77 */
  double rho, mass, t;         /* g/cm3, g/mol, thickness [mm] (thin mirror) */
  /* crystal (crystal.py:150-226, crystals_basic.py) */
  int32_t structure;           /* 0 fcc, 1 diamond, 2 from the unit cell (cell_* below) */
  int32_t hkl[3];
  int32_t geom_bragg;          /* 1 Bragg, 0 Laue */
  int32_t geom_transmitted;    /* 1 transmitted, 0 reflected */
  int32_t thick;               /* 1: t is None (semi-infinite Bragg) */
  double d, chi_to_f, fact_dw, t_crystal;
  /* kind == XRT_HIP_MAT_MULTILAYER: the stack, a record in DEVICE memory. Of the fields
   * above the pass reads geom_bragg -- 1: kind 'multilayer', deflects like a Bragg crystal
   * of spacing d, the period (reflect.py:865-872), amplitude at the cosine to the surface
   * normal; 0: Coated (kind 'mirror'), mirror direction, amplitude at the cosine to the
   * local normal --, d and geom_transmitted. */
  const struct xrt_hip_multilayer* layers;
  /* structure == 2, CrystalFromCell: the sums over the atoms of the unit cell, a record in
   * DEVICE memory (kept out of this record, which travels as a kernel argument: the DCM
   * kernel takes two of them) */
  const struct xrt_hip_cell* cell;
  /* Material(refractiveIndex = a number) (material.py:240-262, 364-373): n_fixed != 0 -> the
   * refractive index is n_re + i n_im at every energy and the element tables are not
   * consulted (nelem may be 0). */
  int32_t n_fixed;
  double n_re, n_im;
  /* n_fixed == 2: Material(refractiveIndex = a table or a file) (material.py:252-262, 284-330:
   * cubic spline through the tabulated n + ik): the index of every ray at its energy,
   * evaluated by the caller, interleaved (re, im) in DEVICE memory, n of them. Mirrors, plates,
   * gratings; not inside a multilayer stack. */
  const double* n_ray;
} xrt_hip_material;

/* Multilayer / GradedMultilayer / Coated (materials/multilayer.py): npairs periods of a
 * top and a bottom layer on a substrate, Parratt's recursion with Nevot-Croce roughness
 * factors (multilayer.py:257-566). Of the three xrt_hip_material records only the element
 * tables, rho and mass are read; nelem == 0 stands for vacuum (n = 1: a missing layer or
 * substrate). Thicknesses in Angstrom. */
typedef struct xrt_hip_multilayer {
  xrt_hip_material top, bottom, substrate;
  int32_t npairs;
  int32_t transmitted;      /* geom 'transmitted': the stack + substrate of subst_thickness */
  int32_t uniform;          /* every period has dti[0], dbi[0]: the two phase factors are
                               evaluated once per ray */
  const double* dti;        /* [npairs] top-layer thickness per period, vacuum side first */
  const double* dbi;        /* [npairs] bottom-layer thickness */
  double id2;               /* idThickness^2 (interdiffusion / roughness, rms) */
  double bs_rough2;         /* bottom layer - substrate: id2, or substRoughness^2 without a
                               top layer (multilayer.py:353) */
  double subst_thickness;   /* transmitted only; may be INFINITY */
} xrt_hip_multilayer;

/* Scratch needed by xrt_hip_reflect_pass_f64_dev for n rays. */
XRT_HIP_API size_t xrt_hip_reflect_workspace_bytes(int64_t n);

/* sizeof() of the structs above, to let a binding verify its layout:
 * which = 0 beam, 1 rotation, 2 pass, 3 material, 4 screen, 5 aperture, 6 undulator,
 * 7 undulator_map, 8 plot, 9 custom_field, 10 bend, 11 multilayer, 12 gauss, 13 geosource,
 * 14 bounce. */
XRT_HIP_API int xrt_hip_sizeof(int which);

/* in: incoming beam. out_local: "lb" of the reference (true local frame); NULL (mirrors,
 * plates and gratings only -- not crystals, not layered materials of either kind, Multilayer
 * or Coated; with theta NULL as well) = not wanted, the reference's
 * needLocal=False (oes/reflect.py:104-108): the pass then writes 200 B per ray instead of 308.
 * out_virgin: "gb"/"vlb" (virgin local or global, see xrt_hip_pass).
 * restore: beam whose x..E,J are copied into out_virgin for rays that did not
 * end in state {1,2} (reflect.py:131-134; dcm.py:330-335 passes the ORIGINAL
 * beam here on the 2nd crystal); usually == in. theta (optional): lb.theta[n]
 * (reflect.py:793-796). By default the pass first runs on the batch-global decisions
 * a beam along the beamline always produces (axis y, ray 0's sign, secant, clamp
 * inactive), every ray verifying them, and repeats itself through the exact
 * statistics (one more launch, reflect_exact, which otherwise returns at once) only
 * if one was contradicted -- same bits either way; a request for
 * info_host, outputs that alias the inputs, or XRT_HIP_REFLECT_EXACT=1 in the
 * environment take the exact sequence directly.
 * info_host (optional, 16 doubles, forces a sync):
 * [0] bracketing axis, [1] first-ray sign, [2] brent?, [3] t1.min, [4] t2.max,
 * [5] max|dz1|, [6] max|dz2|, [7] entering rays, [8] rays ending in state 1 and
 * [9] sum(beamInDotNormal over them) -- crystals only, and only when the batch
 * had both signs of beamInDotNormal (the exact two-pass redo ran), [10]/[11] the
 * batch held negative / non-negative beamInDotNormal (crystals). kernel_ms
 * (optional, 3 floats, forces a sync): [0] whole pass, [1] the dominant kernel
 * (fused solve+finish), HIP events on `stream`; [2] 1 if the exact sequence did
 * the work (forced, or the single pass was contradicted), else 0. Asynchronous on
 * `stream` otherwise. */
XRT_HIP_API int xrt_hip_reflect_pass_f64_dev(
    const xrt_hip_pass* pass, const xrt_hip_material* material,
    const xrt_hip_beam* in, const xrt_hip_beam* restore, xrt_hip_beam* out_local,
    xrt_hip_beam* out_virgin, double* theta, void* workspace,
    size_t workspace_bytes, void* stream, double* info_host, float* kernel_ms);

/* ---- OE.multiple_reflect (oes/reflect.py:165-264): up to maxReflections bounces off the
 * same surface (capillaries, whispering-gallery mirrors, Montel pairs). One call = one turn
 * of its loop over a device-resident beam: _reflect_local as the loop calls it (`lb is vlb`:
 * the beam arrives and leaves in the element's VIRGIN local frame -- on the first bounce it
 * arrives in the global frame, pass->in_is_global = 1, good_mode = 0; afterwards
 * in_is_global = 0, good_mode = 1, is_multi = 1), the return of rays over the edge (state 3)
 * to where they were (:225-228) and the count of the rays left in state 1 or 2 (:229-230).
 * `out` (n rays, not sharing arrays with `in`) is the beam after the bounce and at the same
 * time this bounce's footprint: lbN of the reference is the `out` beams of all bounces, one
 * after the other, so a caller hands in consecutive n-ray slices of its lbN arrays and feeds
 * each back as the next `in`.
 * Mirrors, plates, gratings with a single order, no material; flat / toroidal / bent-flat
 * surfaces, the parametric conics (capillaries), cone, lens paraboloid, VFM / DualVFM and
 * user-defined surfaces (general flavour). Refused: Bragg crystals and layered materials (their
 * deflection needs a batch mean / their kernels are not instantiated for it), blazed
 * profiles, zone plates, per-ray orders, figure errors, no_intersection_search. */
typedef struct xrt_hip_bounce {
  const int32_t* nrefl_in;   /* lb.nRefl [n] before the bounce; NULL on the first (zeros) */
  int32_t* nrefl_out;        /* [n] after it */
  double* theta;             /* [n] lb.theta of this bounce: 0 where the ray did not hit. (The
                                reference leaves lb.theta untouched by a bounce in which NO ray
                                hits, reflect.py:793: the caller looks at counts[1].) */
  /* elevationD / X / Y / Z (reflect.py:214-218, 651-659), [n] each, or all NULL;
   * elev_in NULL (first bounce) = the initial values -1, -1000, -1000, -1000 */
  const double* elev_in[4];
  double* elev_out[4];
  /* lb.s / phi / r of a parametric surface (reflect.py:1066-1069), [n] each, or NULL */
  double* spr_out[3];
  /* How many rays of `in` still enter (what the previous bounce returned in counts[0]), or 0 =
   * not known. A hint that changes speed only: below a quarter of the rays the bounce runs in
   * its sparse form (an index of the entering rays, lanes that take the next ray when theirs
   * is done), whose results are bit for bit those of the dense one. */
  int64_t entering_hint;
  /* A full bounce runs in an OPTIMISTIC form: no batch statistics, the decisions the reference
   * takes from the whole batch are assumed -- bracketing axis and sign from the head of the
   * beam, and for each of the two searches of a bounce (tangency point, hit) secant or Brent as
   * given HERE (1 = Brent; what the same bounce found last time is a good guess, or the bounce
   * before) -- and verified by every ray; a contradicted bounce is redone exactly inside the
   * call. assume_hit_brent = -1: the exact form at once. Results are the same bits either way.
   * found_host (optional, 4 int32 of HOST memory, filled when counts_host is given): [0] the hit
   * search of this batch takes Brent, [1] the tangency search does, [2] the bounce was redone,
   * [3] the form it ran in (0 exact, 1 optimistic, 2 sparse). */
  int32_t assume_hit_brent;
  int32_t assume_tangency_brent;
  int32_t* found_host;
} xrt_hip_bounce;

XRT_HIP_API size_t xrt_hip_bounce_workspace_bytes(int64_t n);

/* counts_host (optional, 2 int64, forces a sync): rays in state 1 or 2 after the bounce
 * (0: the loop ends), rays in state 1. Without it the two numbers stay in the first 16
 * bytes of `workspace` (uint64) for the caller to fetch. info_host (optional, 16 doubles,
 * forces a sync): [0] bracketing axis, [1] first-ray sign, [2] Brent? for the hit search,
 * [3] entering rays, [4] Brent? for the tangency search, [5] / [6] its clamp range,
 * [7] / [8] the clamp range t1.min(), t2.max() of the hit search. */
XRT_HIP_API int xrt_hip_reflect_bounce_f64_dev(
    const xrt_hip_pass* pass, const xrt_hip_material* material, const xrt_hip_beam* in,
    xrt_hip_beam* out, const xrt_hip_bounce* bounce, void* workspace, size_t workspace_bytes,
    void* stream, int64_t* counts_host, double* info_host);

/* The end of multiple_reflect (reflect.py:246-255): rays with nrefl > 0 leave with state 1
 * in the global frame (pass: sin_az / cos_az, center), the others are the rays of `original`
 * with the state they ended in. `last` = the beam after the last bounce. */
XRT_HIP_API int xrt_hip_multiple_reflect_out_f64_dev(
    const xrt_hip_pass* pass, const xrt_hip_beam* last, const xrt_hip_beam* original,
    const int32_t* nrefl, xrt_hip_beam* out_global, void* stream);

/* GeometricSource.shine (device generator, below) -> OE.reflect -> Screen.expose as ONE pass:
 * xrt_hip_reflect_screen_f64_dev whose incoming rays are made in registers from the source's
 * record instead of being read (sources/geoms.py:420-535 followed by oes/reflect.py:18-163 and
 * screens.py:226-302). The generator is counter-based: the same record makes the same rays
 * again whenever somebody wants the source's beam itself. `source_beam` (n rays, with Es / Ep
 * iff amplitudes are wanted): scratch that is written only if the exact sequence has to redo
 * the pass, or when the pass is not one of the lean mirror kernels (then the generator's own
 * launch fills it first). *fused (optional): bit 0 = the screen was in the tail of the pass,
 * bit 1 = the source was in its head (source_beam holds nothing). 208 B per ray cross HBM
 * (108 local beam + 100 image) instead of 608. */
struct xrt_hip_geosource;
struct xrt_hip_screen;
XRT_HIP_API int xrt_hip_shine_reflect_screen_f64_dev(
    const struct xrt_hip_geosource* source, const xrt_hip_pass* pass,
    const xrt_hip_material* material, xrt_hip_beam* source_beam, xrt_hip_beam* out_local,
    xrt_hip_beam* out_virgin, double* theta, const struct xrt_hip_screen* screen,
    xrt_hip_beam* out_screen, int keep_virgin, void* workspace, size_t workspace_bytes,
    void* stream, int* fused);

/* DCM.double_reflect (oes/dcm.py:248-354) as ONE pass over the beam: both crystals
 * per ray, the beam between them never goes to memory (416 B of HBM traffic per ray
 * instead of 716). pass1 / pass2 are what two xrt_hip_reflect_pass_f64_dev calls
 * would take (pass1.out_to_global = 0; pass2.in_is_global = 0, good_mode 1,
 * out_to_global = 1, zero_local_not_entering = 1); `in` is also the beam that rays
 * lost on the way are restored from (dcm.py:330-335). Flat Bragg-reflecting crystals
 * of one thickness class only: xrt_hip_double_reflect_fusable tells (1 / 0), other
 * pairs take two pass calls. Outputs must not share arrays with `in`. The batch
 * decisions are assumed as in the single pass -- the second crystal's from the head
 * ray mirrored at the first -- and verified per ray; contradicted, both passes are
 * redone exactly inside the same call. kernel_ms as above. */
XRT_HIP_API int xrt_hip_double_reflect_fusable(const xrt_hip_pass* pass1,
                                               const xrt_hip_material* material1,
                                               const xrt_hip_pass* pass2,
                                               const xrt_hip_material* material2);
XRT_HIP_API int xrt_hip_double_reflect_f64_dev(
    const xrt_hip_pass* pass1, const xrt_hip_material* material1,
    const xrt_hip_pass* pass2, const xrt_hip_material* material2, const xrt_hip_beam* in,
    xrt_hip_beam* out_local1, xrt_hip_beam* out_local2, xrt_hip_beam* out_global,
    double* theta1, double* theta2, void* workspace, size_t workspace_bytes, void* stream,
    float* kernel_ms);

/* The surface functions of an element on device arrays -- what the reference's OE
 * classes expose as local_z / local_n / local_r / xyz_to_param / param_to_xyz
 * (oes/base.py:675-742, oes/__init__.py:398-411, oes/parametric.py:213-247), evaluated
 * by the code the ray kernels use. Only the surface part of `pass` is read.
 *   what 0: (u, v) = (x, y) -> z           1: (u, v) = (x, y) | (s, phi) -> n[0..5]
 *        2: (u, v) = (s, phi) -> r         3: (u, v, w) = (x, y, z) -> (s, phi, r)
 *        4: (u, v, w) = (s, phi, r) -> (x, y, z)
 *        5: (u, v) = (x, y) -> the state rays_good gives a hit there (as a double)
 * out: k-th output of point i at out[k * n + i] (1, 6, 1, 3, 3, 1 outputs). */
#define XRT_HIP_SURF_EVAL_Z 0
#define XRT_HIP_SURF_EVAL_N 1
#define XRT_HIP_SURF_EVAL_R 2
#define XRT_HIP_SURF_EVAL_TO_PARAM 3
#define XRT_HIP_SURF_EVAL_FROM_PARAM 4
#define XRT_HIP_SURF_EVAL_STATE 5
XRT_HIP_API int xrt_hip_surface_eval_f64_dev(const xrt_hip_pass* pass, int what, int64_t n,
                                             const double* u, const double* v,
                                             const double* w, double* out, void* stream);

/* OE.local_to_global (oes/base.py:1165-1229) on a device-resident beam, IN PLACE: true
 * local frame -> global frame (pass: shift, to_virgin, sin_az / cos_az, center,
 * out_to_global), coherency matrix and amplitudes turned by roll + atan2(n_x, n_z)
 * (pass: cos_roll, sin_roll and the surface). Every ray, whatever its state. */
XRT_HIP_API int xrt_hip_local_to_global_f64_dev(const xrt_hip_pass* pass, xrt_hip_beam* beam,
                                                void* stream);

/* waves.diffract around the Kirchhoff integral (waves.py:606-831), on device arrays.
 *
 * diffract_pre: the samples on the diffracting element -> the inputs of
 * xrt_hip_kirchhoff_f64_dev (positions, normals -- the element's surface normal at each
 * sample if is_oe, else (0, 1, 0) --, nl = direction . normal, k = E / CHBAR 1e7, Es / Ep
 * with every sample not in state 1 switched off, waves.py:674-689, :841) and
 * sums_host[0..2] = sum(Jss + Jpp), sum((Jss + Jpp) nl), count over the lit samples
 * (:690-696). workspace >= 6144 bytes. Synchronises.
 *
 * wave_fields: acc += fresh (the five integrals S, P, A, B, C of this call into the
 * wave's accumulators), then amplitudes, coherency matrix, direction (phase of the
 * dominant direction integral removed) and energy of `wave` from the accumulators,
 * intensities times `scale`, amplitudes times sqrt(scale) (:707-749).
 *
 * basis_to_global: beam positions (and directions) out of a frame given by the
 * origin / axes of an xrt_hip_screen record (screens, apertures), in place.
 *
 * wave_receive: the diffracted field `glo` (global frame) into the local frame of the
 * element the samples `wave` lie on: receiver = its pass record (azimuth, to_local, roll,
 * surface); is_oe = 0 for screens / apertures (azimuth only); obliquity applied to both
 * beams (:773-824). */
XRT_HIP_API int xrt_hip_diffract_pre_f64_dev(
    const xrt_hip_pass* surface, int is_oe, const xrt_hip_beam* samples, double* sx,
    double* sy, double* sz, double* nx, double* ny, double* nz, double* nl, double* k,
    double* Es_ri, double* Ep_ri, void* workspace, size_t workspace_bytes, void* stream,
    double* sums_host);
XRT_HIP_API int xrt_hip_wave_fields_f64_dev(int64_t n, double* const* fresh_ri,
                                            double* const* acc_ri, const double* energy0,
                                            double scale, int from_oe, xrt_hip_beam* wave,
                                            void* stream);
struct xrt_hip_screen;
XRT_HIP_API int xrt_hip_basis_to_global_f64_dev(const struct xrt_hip_screen* frame,
                                                xrt_hip_beam* beam, int with_directions,
                                                void* stream);
XRT_HIP_API int xrt_hip_wave_receive_f64_dev(const xrt_hip_pass* receiver, int is_oe,
                                             xrt_hip_beam* wave, xrt_hip_beam* glo,
                                             void* stream);

/* Stand-alone amplitude evaluation on device arrays (what the reference exposes
 * as Material.get_amplitude(E, beamInDotNormal, fromVacuum) -> rs, rp, mu, n'k
 * (materials/material.py:415-493) and Crystal.get_amplitude(E, beamInDotNormal,
 * beamOutDotNormal, beamInDotHNormal) -> curveS, curveP (crystal.py:492-645)).
 * rs/rp, S/P: complex interleaved [2n]; mu, nk: [n] (may be NULL). */
XRT_HIP_API int xrt_hip_material_amplitude_f64_dev(
    const xrt_hip_material* material, int64_t n, const double* E, const double* bdn,
    double* rs_ri, double* rp_ri, double* mu, double* nk, void* stream);
XRT_HIP_API int xrt_hip_crystal_amplitude_f64_dev(
    const xrt_hip_material* material, int64_t n, const double* E, const double* gamma0,
    const double* gammah, const double* hns, double* S_ri, double* P_ri, void* stream);
/* Multilayer.get_amplitude(E, beamInDotNormal) -> (ri_s, ri_p) or, transmitted, (ti_s,
 * ti_p) (materials/multilayer.py:257-566; the reference's OpenCL twins are
 * get_amplitude_graded_multilayer{,_tran}, cl/materials.cl). material->kind must be
 * XRT_HIP_MAT_MULTILAYER. */
XRT_HIP_API int xrt_hip_multilayer_amplitude_f64_dev(
    const xrt_hip_material* material, int64_t n, const double* E, const double* bdn,
    double* rs_ri, double* rp_ri, void* stream);

/* ---- Screen.expose (screens.py:226-302) on a device-resident beam ---------
 * ex, ey, ez: the screen's local axes in the global frame (beamline.py:288-316);
 * lost_num = -ordinal-2000 (screens.py:77). */
typedef struct xrt_hip_screen {
  double center[3];
  double ex[3], ey[3], ez[3];
  double compress_x, compress_z; /* 0 = none */
  int32_t lost_num;
  int32_t only_positive_path;
  /* HemisphericScreen (screens.py:422-559): radius != 0 -> the rays are carried to the
   * sphere of that radius about `center` (the far intersection), positions go into the
   * screen's axes, directions stay global, and the two angles theta = asin(z / R) -
   * theta_offset, phi = atan2(y, x) - phi_offset are written to out_theta / out_phi
   * (device arrays of n doubles, may be NULL). */
  double radius, theta_offset, phi_offset;
  double* out_theta;
  double* out_phi;
} xrt_hip_screen;

XRT_HIP_API int xrt_hip_screen_expose_f64_dev(const xrt_hip_screen* screen,
                                              const xrt_hip_beam* in, xrt_hip_beam* out,
                                              void* stream);

/* OE.reflect whose global beam goes straight into Screen.expose, as ONE pass over the beam
 * (N1 of SURVEY 8f, "fused aperture / screen ops"; oes/reflect.py:18-163 followed by
 * screens.py:226-302): xrt_hip_reflect_pass_f64_dev with the screen's image `out_screen` made in
 * the tail of the pass, from the registers that hold the outgoing ray. keep_virgin = 0: nobody
 * else reads the global beam -- out_virgin (still a full beam: scratch) is then written only
 * if the exact sequence has to redo the pass, and its contents are undefined on return
 * otherwise: 308 B per ray cross HBM (100 in, 100 + 8 local, 100 image) instead of the 508 of
 * the two passes. The lean mirror / plate kernels carry the screen (flat screens); any other
 * pass is followed by the screen's own launch inside this call -- *fused (optional) tells
 * which (1 / 0). Same bits as the two calls one after the other. */
XRT_HIP_API int xrt_hip_reflect_screen_f64_dev(
    const xrt_hip_pass* pass, const xrt_hip_material* material, const xrt_hip_beam* in,
    const xrt_hip_beam* restore, xrt_hip_beam* out_local, xrt_hip_beam* out_virgin,
    double* theta, const xrt_hip_screen* screen, xrt_hip_beam* out_screen, int keep_virgin,
    void* workspace, size_t workspace_bytes, void* stream, int* fused, float* kernel_ms);

/* ---- RectangularAperture.propagate / RoundAperture.propagate (apertures.py:334-413, 770-846)
 * Rays with state > 0 are taken to the aperture plane (local y = 0); those
 * outside the blades (inside, for a beam stop) get state lost_num = -ordinal-1000
 * IN THE INCOMING BEAM TOO (the reference mutates beam.state, :373). blade_mask:
 * bit 0 left, 1 right, 2 bottom, 3 top (which blades exist). out_global may be
 * NULL (needNewGlobal=False). */
typedef struct xrt_hip_aperture {
  double center[3];
  double ex[3], ey[3], ez[3];
  double sin_az, cos_az;       /* for the optional global output, beamline.py:267-287 */
  double blade[4];             /* left, right, bottom, top */
  int32_t blade_mask;
  int32_t is_beam_stop;
  int32_t lost_num;
  int32_t round;               /* 1: RoundAperture (apertures.py:770-846): stopped where
                                  sqrt(x^2 + z^2) > radius instead of by blades */
  double radius;
  /* DoubleSlit (apertures.py:931-1021): an opaque band shade[0] < z < shade[1] between the
   * bottom and the top blade; its new global beam carries the incoming path twice (:1013) */
  int32_t has_shade;
  int32_t glo_adds_path;
  double shade[2];
  /* PolygonalAperture (apertures.py:1035-1310): open inside the polygon of poly_n
   * vertices (DEVICE array x0, z0, x1, z1, ...; matplotlib's Path.contains_points). As in
   * the reference the test also runs on the rays that do NOT enter (state <= 0), on their
   * untransformed coordinates, and relabels them in the incoming beam (:1198-1203). */
  int32_t poly_n;
  int32_t own_marks;           /* 1: the states of beam_inout already carry THIS aperture's marks
                                  (the call that makes the beam in the aperture's frame after a
                                  states-only call on the same arrays): a ray in state lost_num
                                  was alive when it arrived. Not with a polygon (see above). */
  const double* poly_xz;
} xrt_hip_aperture;

/* out_local may be NULL (then out_global must be too): only the states in beam_inout are
 * updated -- 52 B read and <= 4 B written per ray; the beam in the aperture's frame can be made
 * later by a second call on the same arrays, with a copy of the states as they were or, the
 * states as this call left them, with own_marks = 1. */
XRT_HIP_API int xrt_hip_aperture_propagate_f64_dev(const xrt_hip_aperture* aperture,
                                                   xrt_hip_beam* beam_inout,
                                                   xrt_hip_beam* out_local,
                                                   xrt_hip_beam* out_global, void* stream);

/* Screen.expose of a resident beam followed by aperture.propagate of the SAME beam (a front-end
 * monitor and the mask behind it: screens.py:226-302, then apertures.py:334-413) as one pass
 * over the rays: out_screen = what xrt_hip_screen_expose_f64_dev makes from the states as they
 * are on entry, then the marks of xrt_hip_aperture_propagate_f64_dev (out_local NULL) in
 * beam_inout. Flat screens (radius == 0) and apertures without vertices (poly_n == 0); refused
 * (XRT_HIP_ERR_ARG) otherwise. Same bits as the two calls. */
XRT_HIP_API int xrt_hip_screen_expose_mark_f64_dev(const xrt_hip_screen* screen,
                                                   const xrt_hip_aperture* aperture,
                                                   xrt_hip_beam* beam_inout,
                                                   xrt_hip_beam* out_screen, void* stream);

/* ---- GeometricSource.shine on the device ------------
 * Replaces the host sampling of GeometricSource.shine (sources/geoms.py:420-535) with one
 * kernel that writes the 13 (15) SoA arrays of the beam straight into HBM. The laws are the
 * reference's (_apply_distribution geoms.py:370-407, _set_annulus :409-418, make_energy
 * :16-60, make_polarization :63-179, b from a and c :497-505, rotate_beam and
 * virgin_local_to_global :514-518); the random numbers are Philox4x32-10 blocks
 * (Salmon et al., SC'11) addressed by (ray index, slot, call number) under a 64-bit seed --
 * the stream layout is written out in oracle/geosource_np.py, the CPU restatement the GPU
 * tests compare with. law[k] / p0[k] / p1[k] for k = y, x, z, x', z':
 *   NORMAL: p0 = sigma;  FLAT: [p0, p1);  NORMAL_UNIFORM (uniformRayDensity): sigma p0,
 *   uniform in [-p1, p1], the Gaussian goes into Jss, Jpp, Jsp (and its root into Es, Ep).
 * annulus_xz / annulus_ac: that pair is (r, phi) uniform over the ring ann_* = {rMin, rMax,
 * phiMin, phiMax}. e_law: 0 = every ray has E = e_p0; 1 normal (e_p0, e_p1);
 * 2 flat [e_p0, e_p1); 3 one of n_lines values e_lines[] by the cumulative weights e_cdf[].
 * slopes != 0: a, c are slopes, (a, c, 1) / sqrt(1 + a^2 + c^2) (the reference takes this
 * branch when ANY ray has a^2 + c^2 > 1; xrt_hip_geosource_probe_f64_dev answers that). */
#define XRT_HIP_LAW_NONE 0
#define XRT_HIP_LAW_NORMAL 1
#define XRT_HIP_LAW_FLAT 2
#define XRT_HIP_LAW_NORMAL_UNIFORM 3
#define XRT_HIP_MAX_LINES 16
typedef struct xrt_hip_geosource {
  uint64_t seed;
  uint32_t call;
  int32_t slopes;
  int32_t law[5];
  double p0[5], p1[5];
  int32_t annulus_xz, annulus_ac;
  double ann_xz[4], ann_ac[4];
  int32_t e_law, filament, n_lines, random_ep;
  double e_p0, e_p1;
  double e_lines[XRT_HIP_MAX_LINES], e_cdf[XRT_HIP_MAX_LINES];
  double Jss, Jpp, Jsp[2], Es[2], Ep[2];
  xrt_hip_rotation rot;         /* pitch / roll / yaw of the source, as rotate_beam applies them */
  int32_t to_global;            /* undo the beamline azimuth, add center */
  int32_t state;                /* the state every ray gets (1) */
  double sin_az, cos_az, center[3];
  /* NULL, or a DEVICE cell whose value is added to `call` (modulo 2^32) when the kernel runs: a
   * launch captured in a HIP graph is replayed with the same record, the owner of the graph
   * increments the cell between replays and every replay draws new rays */
  const uint32_t* call_dev;
} xrt_hip_geosource;

/* out: device arrays of out->n rays, all overwritten (Es_ri / Ep_ri NULL = no amplitudes). */
XRT_HIP_API int xrt_hip_geosource_shine_f64_dev(const xrt_hip_geosource* source,
                                                xrt_hip_beam* out, void* stream);
/* any_above_one (device int32, zeroed by the caller) becomes 1 if some ray of the n would have
 * a^2 + c^2 > 1 with this source. */
XRT_HIP_API int xrt_hip_geosource_probe_f64_dev(const xrt_hip_geosource* source, int64_t n,
                                                int32_t* any_above_one, void* stream);

/* ---- weighted 2-D histogram of a device-resident beam ---------------------
 * The reduce step of every run_ray_tracing iteration: raycing.get_output
 * (raycing/__init__.py:170-300) selects rays by state and forms the intensity,
 * multipro.do_hist2d (xrt/multipro.py:111-177) bins it with
 * np.histogram2d(y, x, bins, range, weights). Binning follows numpy: edges =
 * linspace(lo, hi, bins+1), right-open bins, last bin right-closed.
 * x, y: device arrays [n] (any beam field or derived quantity), multiplied by
 * x_factor / y_factor. ray_flags: bit0 state==1, bit1 ==2, bit2 ==3, bit3 <0
 * (lost), bit4 >0 (rayFlag 4). flux_kind: 0 total (Jss+Jpp), 1 s, 2 p,
 * 3 +/-45 (2 Re Jsp), 4 left-right (2 Im Jsp), 5 power ((Jss+Jpp) E e0).
 * hist: bins_y * bins_x doubles, row-major [iy][ix], ACCUMULATED into (zero it
 * first for a fresh histogram). counters (optional, 8 doubles, accumulated):
 * [0] selected rays, [1] sum of weights of all selected, [2] of those inside the
 * range, [3] alive (state>0), [4] good, [5] out, [6] over, [7] dead. */
XRT_HIP_API int xrt_hip_hist2d_f64_dev(
    const xrt_hip_beam* beam, const double* x, const double* y, double x_factor,
    double y_factor, int ray_flags, int flux_kind, double source_weight,
    int bins_x, double x_lo, double x_hi, int bins_y, double y_lo, double y_hi,
    double* hist, double* counters, void* stream);

/* ---- all histograms of one XYCPlot in one pass ----------------------------
 * multipro.py:316-361: cData01 = clip((c - c_lim0) colorFactor/(c_lim1 - c_lim0));
 * RGB = hsv_to_rgb(cData01, colorSaturation, flux); then np.histogram on x, y, c
 * (weights flux and R, G, B) and np.histogram2d(y, x) (weights intensity and
 * R, G, B). flux = intensity for the supported flux kinds.
 * c: device array of the colour datum (e.g. beam.E). Outputs are ACCUMULATED:
 * hist2d [by][bx], hist2d_rgb [by][bx][3] (may be NULL), hist_x [bx][4] =
 * (flux, R, G, B) per bin, hist_y [by][4], hist_c [bc][4] (each may be NULL).
 * counters as in xrt_hip_hist2d_f64_dev. */
typedef struct xrt_hip_plot {
  double x_factor, y_factor, c_factor;
  double source_weight;
  double x_lim[2], y_lim[2], c_lim[2];
  double color_factor, color_saturation;
  int32_t bins_x, bins_y, bins_c;
  int32_t ray_flags, flux_kind;
} xrt_hip_plot;

XRT_HIP_API int xrt_hip_plot_hist_f64_dev(
    const xrt_hip_beam* beam, const double* x, const double* y, const double* c,
    const xrt_hip_plot* plot, double* hist2d, double* hist2d_rgb, double* hist_x,
    double* hist_y, double* hist_c, double* counters, void* stream);
/* The same with scratch of the caller (DEVICE memory, `workspace_bytes` from
 * xrt_hip_plot_hist_workspace_bytes for this ray count, plot and set of outputs; used in stream
 * order). xrt_hip_plot_hist_f64_dev takes its scratch from the device's stream-ordered pool
 * (hipMallocAsync / hipFreeAsync per call); this form allocates nothing, which is also what a
 * call recorded into a HIP graph needs. A workspace that is NULL or too small falls back to
 * the pool. with_lines: any of hist_x / hist_y / hist_c / counters is asked for. */
XRT_HIP_API int xrt_hip_plot_hist_workspace_bytes(int64_t nrays, const xrt_hip_plot* plot,
                                                  int with_rgb, int with_lines, size_t* bytes);
XRT_HIP_API int xrt_hip_plot_hist_ws_f64_dev(
    const xrt_hip_beam* beam, const double* x, const double* y, const double* c,
    const xrt_hip_plot* plot, double* hist2d, double* hist2d_rgb, double* hist_x,
    double* hist_y, double* hist_c, double* counters, void* workspace, size_t workspace_bytes,
    void* stream);

/* ---- a plot in the tail of a pass (round 6) -------------------------------------------------
 * run_ray_tracing adds the image of a screen to an XYCPlot right after the pass that made it
 * (xrt/multipro.py:316-361 after raycing/__init__.py:170-300: get_output, do_hist2d). When
 * nothing else reads that image the plot joins the pass as Screen.expose does: the tail of
 * the ray kernel takes the image record from its registers, forms weight, hue and bins, every
 * wave sorts its 64 rays by tile of the 2-D histogram and writes 20-B records; two small
 * kernels inside the same call add them up into the accumulators. The image itself (100 B per
 * ray) is written only on request (keep_screen); what xrt_hip_plot_hist_ws_f64_dev would have
 * read (44 B per ray) and its sorting pass disappear.
 * x_field / y_field / c_field: which quantity of the IMAGE each axis shows. Accumulators as in
 * xrt_hip_plot_hist_ws_f64_dev, all of hist2d, hist2d_rgb, hist_x, hist_y and counters
 * present (hist_c optional). workspace: DEVICE scratch of xrt_hip_plot_tail_workspace_bytes
 * (0 bytes = this plot cannot ride a pass: more than 2046 bins along x or y, more than 4094 colour
 * bins, more than 56 tiles). Sums are formed in another order than by the stand-alone kernels:
 * equal to ~1e-15 relative, bins and counts identical. */
enum {
  XRT_HIP_FIELD_X = 0, XRT_HIP_FIELD_Y = 1, XRT_HIP_FIELD_Z = 2, XRT_HIP_FIELD_A = 3,
  XRT_HIP_FIELD_B = 4, XRT_HIP_FIELD_C = 5, XRT_HIP_FIELD_PATH = 6, XRT_HIP_FIELD_E = 7,
  XRT_HIP_FIELD_XPRIME = 8, XRT_HIP_FIELD_ZPRIME = 9
};
typedef struct xrt_hip_plot_tail {
  xrt_hip_plot plot;
  int32_t x_field, y_field, c_field;
  int32_t reserved;
  double *hist2d, *hist2d_rgb, *hist_x, *hist_y, *hist_c, *counters;
  void* workspace;
  size_t workspace_bytes;
} xrt_hip_plot_tail;
XRT_HIP_API int xrt_hip_plot_tail_workspace_bytes(int64_t nrays, const xrt_hip_plot_tail* tail,
                                                  size_t* bytes);
/* 1: this pass would carry screen and plot in its tail (a lean mirror / plate kernel, a flat
 * screen, a plot the tail can serve); 0: use xrt_hip_reflect_screen_f64_dev and the histogram
 * call. */
XRT_HIP_API int xrt_hip_reflect_screen_plot_fusable(const xrt_hip_pass* pass,
                                                    const xrt_hip_material* material,
                                                    const struct xrt_hip_screen* screen,
                                                    const xrt_hip_plot_tail* tail, int64_t nrays);
/* xrt_hip_reflect_screen_f64_dev / xrt_hip_shine_reflect_screen_f64_dev with the plot behind
 * the screen. keep_screen = 0: nobody else reads the image -- out_screen may be NULL.
 * Refused (XRT_HIP_ERR_ARG) where xrt_hip_reflect_screen_plot_fusable says 0. *fused: bit 0
 * screen, bit 1 source, bit 2 plot. A contradicted optimistic pass is redone inside the call
 * and the records are made from the real image (reflect_redo_scr). */
XRT_HIP_API int xrt_hip_reflect_screen_plot_f64_dev(
    const xrt_hip_pass* pass, const xrt_hip_material* material, const xrt_hip_beam* in,
    const xrt_hip_beam* restore, xrt_hip_beam* out_local, xrt_hip_beam* out_virgin,
    double* theta, const struct xrt_hip_screen* screen, xrt_hip_beam* out_screen,
    int keep_virgin, int keep_screen, const xrt_hip_plot_tail* tail, void* workspace,
    size_t workspace_bytes, void* stream, int* fused);
XRT_HIP_API int xrt_hip_shine_reflect_screen_plot_f64_dev(
    const struct xrt_hip_geosource* source, const xrt_hip_pass* pass,
    const xrt_hip_material* material, xrt_hip_beam* source_beam, xrt_hip_beam* out_local,
    xrt_hip_beam* out_virgin, double* theta, const struct xrt_hip_screen* screen,
    xrt_hip_beam* out_screen, int keep_virgin, int keep_screen, const xrt_hip_plot_tail* tail,
    void* workspace, size_t workspace_bytes, void* stream, int* fused);

/* ---- everything that may ride in the tail of a pass, in one record (round 6) ----------------
 * Up to two apertures that follow the element directly (RectangularAperture, RoundAperture,
 * DoubleSlit and their beam stops, apertures.py:334-413; no polygons) in the order the beam meets
 * them, then optionally a screen, then optionally the plot of its image. An aperture in the
 * tail is its states-only call (xrt_hip_aperture_propagate_f64_dev with out_local NULL) made on
 * the outgoing record while it is in registers: the state written to out_virgin -- and seen by
 * the screen -- is what aperture.propagate(gb) leaves in gb.state; the beam in the aperture's
 * frame is made later, if anybody wants it, by the full call on out_virgin with own_marks = 1.
 * The lean mirror / plate kernels carry the tail, single flat Bragg crystals its apertures and
 * screen; any other pass is followed by the apertures'
 * and the screen's own launches inside the call (*fused: bit 0 screen, 1 source, 2 plot,
 * 3 apertures in the tail). source NULL: the rays are read from `in`; else they are made from
 * the source's record and `in` is the scratch of xrt_hip_shine_reflect_screen_f64_dev. */
typedef struct xrt_hip_tail {
  int32_t n_apertures;
  int32_t keep_screen;             /* 0: nobody else reads the image (with a plot only) */
  xrt_hip_aperture aperture[2];
  const struct xrt_hip_screen* screen;      /* or NULL */
  xrt_hip_beam* out_screen;                 /* or NULL */
  const xrt_hip_plot_tail* plot;            /* or NULL (needs the screen) */
} xrt_hip_tail;
XRT_HIP_API int xrt_hip_reflect_tail_f64_dev(
    const struct xrt_hip_geosource* source, const xrt_hip_pass* pass,
    const xrt_hip_material* material, xrt_hip_beam* in, const xrt_hip_beam* restore,
    xrt_hip_beam* out_local, xrt_hip_beam* out_virgin, double* theta, const xrt_hip_tail* tail,
    int keep_virgin, void* workspace, size_t workspace_bytes, void* stream, int* fused);

/* ... and behind a DCM (round 6): xrt_hip_double_reflect_f64_dev with the apertures and the flat
 * screen that follow the monochromator directly in the tail of its fused kernel (dcm.py:248-354 ->
 * apertures.py:334-413 -> screens.py:226-302). tail->plot must be NULL. keep_global = 0 (with a
 * screen): out_global is scratch, written only if the pass has to be redone. The pair of plate
 * faces and a forced exact sequence are followed by the apertures' and the screen's own launches
 * inside the call (*fused: bit 0 screen, bit 3 apertures in the tail). */
XRT_HIP_API int xrt_hip_double_reflect_tail_f64_dev(
    const xrt_hip_pass* pass1, const xrt_hip_material* material1, const xrt_hip_pass* pass2,
    const xrt_hip_material* material2, const xrt_hip_beam* in, xrt_hip_beam* out_local1,
    xrt_hip_beam* out_local2, xrt_hip_beam* out_global, double* theta1, double* theta2,
    const xrt_hip_tail* tail, int keep_global, void* workspace, size_t workspace_bytes,
    void* stream, int* fused);




/* ---- undulator field integral -------------------------
 * Replaces run_parallel('undulator' | 'undulator_taper' | 'undulator_nf', ...)
 * as issued by Undulator._build_I_map_CL (sources/synchr.py:2110-2176; kernels
 * cl/undulator.cl:54-300): per ray, the sum over the quadrature nodes of one
 * period (far field) or of all `nper` periods (taper / near field). Arithmetic
 * follows the reference's numpy path Undulator._sp_sum (synchr.py:1930-2038),
 * including its two quirks (tapered phase uses sintg for the Kx term; the
 * near-field carrier is sin/cos(R0z) without w/wu) — see DESIGN.md.
 * Arguments as marshalled at synchr.py:2132-2160:
 *   scalarArgs  -> alpha_s (= _taperVal / E2WC, mode 1) | r0z (= R0*2pi/L0,
 *                  mode 2), Kx, Ky, jend, nper (= Np, modes 1 and 2)
 *   slicedRO    -> gamma, wu, w, ww1, ddphi (theta), ddpsi  [nrays]
 *   nonSlicedRO -> tg, ag, sintg, costg, sintgph, costgph   [jend]
 *   slicedRW    -> Is, Ip: complex128 [nrays] (interleaved re,im), overwritten */
enum { XRT_HIP_UND_FAR = 0, XRT_HIP_UND_TAPER = 1, XRT_HIP_UND_NF = 2 };

typedef struct xrt_hip_undulator {
  int32_t mode;
  int32_t nper;
  double Kx, Ky;
  double alpha_s;
  double r0z;
  int64_t jend;
  const double *tg, *ag, *sintg, *costg, *sintgph, *costgph;
  /* nonzero: `workspace` still holds the node records an earlier _dev call on this stream made
   * from these tables, Kx, Ky and jend (Undulator.shine calls build_I_map many times with one
   * set of tables): the pack launch in front of the sum is skipped */
  int32_t workspace_packed;
  int32_t reserved;
} xrt_hip_undulator;

/* bytes of device scratch the _dev entry point needs for `jend` nodes */
XRT_HIP_API size_t xrt_hip_undulator_workspace_bytes(int64_t jend);

/* all pointers (also those inside `u`) are device pointers; asynchronous on
 * `stream` unless kernel_ms is given (then it synchronises and returns the
 * duration of the summation kernel, HIP events on that stream) */
XRT_HIP_API int xrt_hip_undulator_f64_dev(
    const xrt_hip_undulator* u, int64_t nrays, const double* gamma, const double* wu,
    const double* w, const double* ww1, const double* ddphi, const double* ddpsi,
    double* Is_ri, double* Ip_ri, void* workspace, size_t workspace_bytes, void* stream,
    float* kernel_ms);

/* host pointers everywhere (what XRT_CL.run_parallel is handed); blocking */
XRT_HIP_API int xrt_hip_undulator_f64(
    int device, const xrt_hip_undulator* u, int64_t nrays, const double* gamma,
    const double* wu, const double* w, const double* ww1, const double* ddphi,
    const double* ddpsi, double* Is_ri, double* Ip_ri, float* kernel_ms);

/* The whole Undulator._build_I_map_conv (synchr.py:2050-2108) in one launch:
 * pre-factors wu, ww1, ab from (w, theta, psi, gamma), the node sum above, the
 * optional harmonic window (synchr.py:2094-2098) and the Amp2Flux scaling.
 * gamma: per-ray array (energy spread, synchr.py:2054-2059) or NULL = gamma0.
 * Outputs: I [n] (flux density), Es, Ep complex128 [n]. u->r0z must be
 * R0*2*pi/L0 for mode 2, u->alpha_s = _taperVal/E2WC for mode 1. */
typedef struct xrt_hip_undulator_map {
  double L0;          /* period [mm] */
  double Np;          /* number of periods (enters sin(pi Np ww1)/sin(pi ww1)) */
  double gamma0;
  double eI;          /* ring current [A] */
  double dstep;       /* 2 pi / gIntervals */
  double harmonic;
  int32_t has_harmonic;
  int32_t dist_bw;    /* 1: distE == 'BW' (bwFact = 0.001), 0: 1/w */
} xrt_hip_undulator_map;

XRT_HIP_API int xrt_hip_undulator_imap_f64_dev(
    const xrt_hip_undulator* u, const xrt_hip_undulator_map* m, int64_t nrays,
    const double* w, const double* theta, const double* psi, const double* gamma,
    double* I, double* Es_ri, double* Ep_ri, void* workspace, size_t workspace_bytes,
    void* stream);

/* ---- field sums of a source with a tabulated magnetic field ----------------
 * Replaces run_parallel('custom_field' | 'custom_field_filament', ...) of
 * SourceFromField._build_I_map_custom_field_CL (sources/synchr.py:1157-1272;
 * kernels cl/undulator.cl:822-1103); arithmetic of the numpy path
 * SourceFromField._sp_sum (synchr.py:888-973). Node tables on the integration
 * grid (built by the caller's trajectory integration): tg, ag, Bx, By, Bz, betax,
 * betay, trajx, trajy, trajz [jend]. Per ray: emcg (= SIE0/SIM0/C/10/gamma),
 * gamma, w, ddphi, ddpsi. betam = betazav[-1]; near_field: screen at R0 [mm].
 * Outputs Is, Ip complex128 [nrays]. Workspace as for the undulator sums. */
typedef struct xrt_hip_custom_field {
  int32_t filament;
  int32_t near_field;
  double betam;
  double R0;
  double wc;        /* filament only: if > 0, the carrier w E2WC / betam given by the caller
                       (what the OpenCL kernel receives, synchr.py:1206); 0: derived
                       from w, gamma, betam as _sp_sum does */
  int64_t jend;
  const double *tg, *ag, *Bx, *By, *Bz, *betax, *betay, *trajx, *trajy, *trajz;
  int32_t carrier_form; /* 0: the carrier of the summing form _sp_sum (what the reference takes
                           for more than 10 rays); 1: that of its vectorised form _sp (10 rays
                           or fewer: the node-number search), synchr.py:813-816 vs :901-902 */
  int32_t reserved;
} xrt_hip_custom_field;

XRT_HIP_API int xrt_hip_custom_field_f64_dev(
    const xrt_hip_custom_field* f, int64_t nrays, const double* emcg, const double* gamma,
    const double* w, const double* ddphi, const double* ddpsi, double* Is_ri, double* Ip_ri,
    void* workspace, size_t workspace_bytes, void* stream, float* kernel_ms);

/* host pointers everywhere; blocking */
XRT_HIP_API int xrt_hip_custom_field_f64(
    int device, const xrt_hip_custom_field* f, int64_t nrays, const double* emcg,
    const double* gamma, const double* w, const double* ddphi, const double* ddpsi,
    double* Is_ri, double* Ip_ri, float* kernel_ms);

/* Electron trajectory through a tabulated field, SourceFromField._build_trajectory_conv
 * (sources/synchr.py:1049-1147) = run_parallel('get_trajectory' |
 * 'get_trajectory_filament', ...) of _build_trajectory_CL (:1011-1047; kernels
 * cl/undulator.cl:733, 918): Runge-Kutta along the grid wt[n] [mm] with the field [T] on
 * the half-step grid (Bx, By, Bz [2n - 1]); the mean velocity and mean position are
 * removed. filament = 0: per unit emcg (pass emcg = 1, gamma unused); 1: in units of c for
 * the electron `gamma`, emcg = SIE0/SIM0/C/10/gamma. Outputs on the grid: betax, betay,
 * trajx, trajy, trajz [n]; betam[1] = the mean longitudinal term (betazav[-1]). All
 * pointers are device pointers. */
XRT_HIP_API int xrt_hip_trajectory_f64_dev(int filament, int64_t n, const double* wt,
                                           const double* Bx, const double* By,
                                           const double* Bz, double gamma, double emcg,
                                           double* betax, double* betay, double* trajx,
                                           double* trajy, double* trajz, double* betam,
                                           void* stream);

/* Intensity and amplitude map of a bending magnet or a wiggler, BendingMagnet.build_I_map
 * (sources/synchr.py:185-227; the reference has no accelerator path for it): per ray the
 * photon energy E [eV] and the observation angles theta, psi [rad] -> flux I and the two
 * field amplitudes (Es imaginary, Ep real, as the reference's). gamma_ray: per-ray electron
 * gamma (energy spread) or NULL for the nominal one. */
typedef struct xrt_hip_bend {
  double gamma;            /* nominal Lorentz factor */
  double B;                /* peak field [T] */
  double K;                /* wiggler: deflection parameter */
  double poles;            /* 2 Np (bending magnet: Np = 0.5 -> 1) */
  double eI;               /* ring current [A] */
  int32_t wiggler;         /* 1: critical energy varies with theta (isMPW) */
  int32_t per_bandwidth;   /* 1: distE = 'BW' (per 0.1 % bandwidth), 0: per eV */
} xrt_hip_bend;
XRT_HIP_API int xrt_hip_bend_imap_f64_dev(const xrt_hip_bend* m, int64_t n, const double* E,
                                          const double* theta, const double* psi,
                                          const double* gamma_ray, double* I, double* Es_ri,
                                          double* Ep_ri, void* stream);
/* ---- GaussianBeam / LaguerreGaussianBeam / HermiteGaussianBeam.shine(wave=...)
 * (sources/geoms.py:538-850): the analytical mode field on the points of a wave.
 * x, y, z: the points in the source's frame (wave.xDiffr ...), E [eV] per point, dS: cell
 * area per point (NULL: dS_scalar for all). amp_ri[2n]: field per point times sqrt(dS)
 * (multiplies Es, Ep; |amp|^2 the coherency matrix); a, b, c: unit direction along the
 * wavefront normal. */
typedef struct xrt_hip_gauss {
  double w0x, w0z;       /* waist; both used if astigmatic (w0 given as a pair) */
  int32_t astigmatic;
  int32_t mode;          /* 0 Gaussian, 1 Laguerre-Gauss (l, p), 2 Hermite-Gauss (m, n) */
  int32_t l, p, m, n;
  double clp;            /* mode normalisation: sqrt(p! / (|l| + p)!) or
                            (2^(m+n) m! n!)^(-1/2) */
} xrt_hip_gauss;
XRT_HIP_API int xrt_hip_gaussian_beam_f64_dev(const xrt_hip_gauss* g, int64_t n, const double* x,
                                              const double* y, const double* z,
                                              const double* E, const double* dS,
                                              double dS_scalar, double* amp_ri, double* a,
                                              double* b, double* c, void* stream);
/* building block (GPU tests): modified Bessel functions K_{1/3}, K_{2/3} */
XRT_HIP_API int xrt_hip_debug_bessel_k_f64_dev(int64_t n, const double* x, double* k13,
                                               double* k23, void* stream);

/* ---- timing without a host sync ------------------------------------------
 * xrt_hip_reflect_time_next_pass arms the NEXT xrt_hip_reflect_pass_f64_dev call of
 * this thread: it records pass_begin / pass_end around the whole pass and
 * kernel_begin / kernel_end around its dominant kernel (the fused solve+finish) on
 * the pass's stream and returns without waiting -- for measuring launch durations
 * inside a pipelined loop (bench.py). Events come from xrt_hip_event_create;
 * xrt_hip_event_elapsed_ms waits for `end`. Any of the four may be NULL. */
XRT_HIP_API int xrt_hip_event_create(void** event);
XRT_HIP_API int xrt_hip_event_destroy(void* event);
XRT_HIP_API int xrt_hip_event_elapsed_ms(void* begin, void* end, float* ms);
XRT_HIP_API int xrt_hip_reflect_time_next_pass(void* pass_begin, void* pass_end,
                                   void* kernel_begin, void* kernel_end);

/* ---- building-block checks (used by the GPU tests only) ---------------- */
XRT_HIP_API int xrt_hip_debug_sqrt_f64_dev(int64_t n, const double* x, double* r, double* rinv,
                               void* stream);
/* the Kirchhoff loop's root from a caller-supplied seed ~ 1/sqrt(x); hinv = 1/(2 sqrt x) */
XRT_HIP_API int xrt_hip_debug_sqrt_seeded_f64_dev(int64_t n, const double* x,
                                                  const double* seed, double* r,
                                                  double* hinv, void* stream);
/* q[i] = a[i] / b through the constant-divisor sequence of the reflect kernels */
XRT_HIP_API int xrt_hip_debug_divconst_f64_dev(int64_t n, const double* a, double b, double* q,
                                   void* stream);
XRT_HIP_API int xrt_hip_debug_sincos_f64_dev(int64_t n, const double* phi, double* sn, double* cs,
                                 void* stream);
/* the LDS-table forms the Kirchhoff kernel uses when |k r| < 2^42: 2048 entries
 * (4e-16) and, with four receiving points per lane, 4096 entries (1.5e-14; 2.6e-14 at
 * the largest phases) */
XRT_HIP_API int xrt_hip_debug_sincos_tab_f64_dev(int64_t n, const double* phi, double* sn,
                                     double* cs, void* stream);
XRT_HIP_API int xrt_hip_debug_sincos_tab4k_f64_dev(int64_t n, const double* phi, double* sn,
                                                   double* cs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XRT_HIP_H */
