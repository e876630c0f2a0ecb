/* TEST INFRASTRUCTURE ONLY (never linked into libxrt_hip.so).
 *
 * C/OpenMP restatement of the reference's ray-surface pass for the subset BASELINE cfg2
 * needs -- OE.reflect on a flat or toroidal mirror with a Fresnel coating, beams
 * without field amplitudes -- so that bench.py has an ALL-CORES CPU baseline for the
 * ray-tracing metric. One ray per loop iteration, the reference's
 * batch-global decisions as OpenMP reductions in front. Validated against the numpy
 * restatement oracle/reflect_np.py (the one pinned to the reference's golden vectors)
 * by tests/test_oracle_reflect_c.py: states equal, geometry 1e-12, intensities 1e-9.
 *
 * Reference anchors (xrt/backends/raycing/):
 *   frames           oes/reflect.py:104-134, 617-635, 1115-1132; beamline.py:230-287
 *   bracket          oes/base.py:1231-1295 (axis from max|a|,|b|,|c| over state-1 rays,
 *                    formula from the FIRST entering ray)
 *   find_dz/local_z  oes/base.py:801-846; oes/__init__.py:398-401 (toroid)
 *   find_intersection oes/base.py:848-885, secant :933-959 (Brent is not restated:
 *                    a batch that would take it is refused with -2)
 *   rays_good        oes/base.py:1094-1163 (rectangular limits)
 *   local_n          oes/__init__.py:403-411
 *   Fresnel          materials/material.py:348-378, 415-493; element.py:252-263
 *   J update         oes/reflect.py:948-1064; sources/beams.py:448-479
 *
 * gcc -O2 -fopenmp -shared -fPIC reflect_c.c -lm
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXROT 8
#define MAXELEM 4

typedef struct {
  int n;
  int axis[MAXROT];        /* 0 x, 1 y, 2 z */
  double c[MAXROT], s[MAXROT];
} rot_t;

typedef struct {
  double center[3], sin_az, cos_az;
  rot_t to_local, to_virgin;
  double dx;               /* shift along local x */
  int surf;                /* 0 flat, 1 toroid */
  double R, r;
  double phys_x[2], phys_y[2];
  int has_opt_x, has_opt_y;
  double opt_x[2], opt_y[2];
  int over_mask;           /* 1 xmin, 2 xmax, 4 ymin, 8 ymax */
  int lost_num;
  double roll;
  /* coating */
  int nelem;
  int Z[MAXELEM], tab_n[MAXELEM];
  double quantity[MAXELEM];
  const double *tab_E[MAXELEM], *tab_f1[MAXELEM], *tab_f2[MAXELEM];
  double rho, mass;
} oe_t;

typedef struct {
  double *x, *y, *z, *a, *b, *c, *path, *E, *Jss, *Jpp, *Jsp; /* Jsp interleaved */
  int32_t* state;
} beam_t;

static const double zEps = 1e-12, dT = 1e-5, maxHalf = 1000., maxDepth = 100.;
static const double CH = 12398.419297617678, R0e = 2.817940285e-5, AVOG = 6.02214199e23;

int xrt_oracle_reflect_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static void rot3(const rot_t* R, double* x, double* y, double* z) {
  for (int i = 0; i < R->n; ++i) {
    const double c = R->c[i], s = R->s[i];
    if (R->axis[i] == 2) {
      const double xn = c * *x - s * *y, yn = s * *x + c * *y;
      *x = xn; *y = yn;
    } else if (R->axis[i] == 1) {
      const double xn = c * *x + s * *z, zn = -s * *x + c * *z;
      *x = xn; *z = zn;
    } else {
      const double yn = c * *y - s * *z, zn = s * *y + c * *z;
      *y = yn; *z = zn;
    }
  }
}

static void to_local(const oe_t* o, const beam_t* in, int64_t i, double* p, double* d) {
  double x = in->x[i] - o->center[0], y = in->y[i] - o->center[1], z = in->z[i] - o->center[2];
  double a = in->a[i], b = in->b[i], c = in->c[i];
  if (o->sin_az != 0.) {
    const double an = o->cos_az * a - o->sin_az * b, bn = o->sin_az * a + o->cos_az * b;
    a = an; b = bn;
    const double xn = o->cos_az * x - o->sin_az * y, yn = o->sin_az * x + o->cos_az * y;
    x = xn; y = yn;
  }
  rot3(&o->to_local, &x, &y, &z);
  rot3(&o->to_local, &a, &b, &c);
  x -= o->dx;
  p[0] = x; p[1] = y; p[2] = z;
  d[0] = a; d[1] = b; d[2] = c;
}

static double surf_z(const oe_t* o, double x, double y) {
  if (o->surf == 1) {
    double rx = 1 - (x / o->r) * (x / o->r);
    if (rx < 0) rx = 0;
    return y * y / 2.0 / o->R + o->r * (1 - sqrt(rx));
  }
  return 0.;
}

static double find_dz(const oe_t* o, double t, const double* p, const double* d, double* q) {
  q[0] = p[0] + d[0] * t;
  q[1] = p[1] + d[1] * t;
  q[2] = p[2] + d[2] * t;
  double s = surf_z(o, q[0], q[1]);
  if (isnan(s)) s = 0.;
  return (q[2] - s) * 1. * 1.;
}

static void bracket(const oe_t* o, int axis, int positive, const double* p, const double* d,
                    double* tMin, double* tMax) {
  double lo, hi;
  if (axis == 0) {
    lo = o->phys_x[0] > -INFINITY ? o->phys_x[0] : -maxHalf;
    hi = o->phys_x[1] < INFINITY ? o->phys_x[1] : maxHalf;
  } else if (axis == 1) {
    lo = o->phys_y[0] > -INFINITY ? o->phys_y[0] : -maxHalf;
    hi = o->phys_y[1] < INFINITY ? o->phys_y[1] : maxHalf;
  } else {
    lo = -maxDepth;
    hi = maxDepth;
  }
  if (positive) {
    *tMin = (lo - p[axis]) / d[axis] - dT;
    *tMax = (hi - p[axis]) / d[axis] + dT;
  } else {
    *tMin = (hi - p[axis]) / d[axis] - dT;
    *tMax = (lo - p[axis]) / d[axis] + dT;
  }
  if (*tMin < -1e6 * zEps) *tMin = -1e6 * zEps;
}

static int sgn(double v) { return (v > 0.) - (v < 0.); }

static int rays_good(const oe_t* o, double x, double y) {
  int st = 1;
  if (o->has_opt_x && ((o->phys_x[0] <= x && x < o->opt_x[0]) ||
                       (o->opt_x[1] <= x && x < o->phys_x[1])))
    st = 2;
  if (o->has_opt_y && ((o->phys_y[0] <= y && y < o->opt_y[0]) ||
                       (o->opt_y[1] <= y && y < o->phys_y[1])))
    st = 2;
  if (x < o->phys_x[0] || x > o->phys_x[1] || y < o->phys_y[0] || y > o->phys_y[1])
    st = o->lost_num;
  if (((o->over_mask & 1) && x < o->phys_x[0]) || ((o->over_mask & 2) && x > o->phys_x[1]) ||
      ((o->over_mask & 4) && y < o->phys_y[0]) || ((o->over_mask & 8) && y > o->phys_y[1]))
    st = 3;
  return st;
}

static double complex interp_f(const oe_t* o, int e, double E) {
  const double* tE = o->tab_E[e];
  const int n = o->tab_n[e];
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (E >= tE[mid]) lo = mid + 1; else hi = mid;
  }
  int j = lo - 1;
  if (j < 0) j = 0;
  double f1, f2;
  if (j >= n - 1) {
    f1 = o->tab_f1[e][n - 1];
    f2 = o->tab_f2[e][n - 1];
  } else if (tE[j] == E) {
    f1 = o->tab_f1[e][j];
    f2 = o->tab_f2[e][j];
  } else {                      /* numpy's interp: slope * (x - x0) + y0 */
    const double dx = tE[j + 1] - tE[j];
    f1 = (o->tab_f1[e][j + 1] - o->tab_f1[e][j]) / dx * (E - tE[j]) + o->tab_f1[e][j];
    f2 = (o->tab_f2[e][j + 1] - o->tab_f2[e][j]) / dx * (E - tE[j]) + o->tab_f2[e][j];
  }
  return f1 + I * f2;
}

/* -> 0, or -2: the batch would take Brent (not restated), -1: bad arguments */
int xrt_oracle_reflect(const oe_t* o, int64_t n, const beam_t* in, beam_t* gb, beam_t* lb,
                       double* theta) {
  if (!o || !in || !gb || !lb || n < 0) return -1;
  const double PI2 = 6.283185307179586476925286766559;
  /* ---- batch-global decisions ---- */
  double ma = 0, mb = 0, mc = 0;
  int64_t first = n, nmain = 0;
#pragma omp parallel for reduction(max : ma, mb, mc) reduction(min : first) reduction(+ : nmain)
  for (int64_t i = 0; i < n; ++i) {
    if (in->state[i] <= 0) continue;
    if (i < first) first = i;
    if (in->state[i] != 1) continue;
    double p[3], d[3];
    to_local(o, in, i, p, d);
    if (fabs(d[0]) > ma) ma = fabs(d[0]);
    if (fabs(d[1]) > mb) mb = fabs(d[1]);
    if (fabs(d[2]) > mc) mc = fabs(d[2]);
    ++nmain;
  }
  if (nmain == 0) { ma = 0; mb = 1; mc = 0; }
  const double mm = fmax(fmax(ma, mb), mc);
  const int axis = mm == ma ? 0 : (mm == mb ? 1 : 2);
  int positive = 1;
  if (first < n) {
    double p[3], d[3];
    to_local(o, in, first, p, d);
    positive = d[axis] > 0.;
  }
  double t1min = INFINITY, t2max = -INFINITY, d1max = 0, d2max = 0;
#pragma omp parallel for reduction(min : t1min) reduction(max : t2max, d1max, d2max)
  for (int64_t i = 0; i < n; ++i) {
    if (in->state[i] <= 0) continue;
    double p[3], d[3], q[3], t1, t2;
    to_local(o, in, i, p, d);
    bracket(o, axis, positive, p, d, &t1, &t2);
    const double dz1 = find_dz(o, t1, p, d, q);
    double dz2 = find_dz(o, t2, p, d, q);
    if (dz1 <= 0. || dz2 >= 0.) dz2 = 0.;
    if (t1 < t1min) t1min = t1;
    if (t2 > t2max) t2max = t2;
    if (fabs(dz1) > d1max) d1max = fabs(dz1);
    if (fabs(dz2) > d2max) d2max = fabs(dz2);
  }
  if (d2max > d1max * 20.) return -2;
  const double pre = 1e-24 * AVOG * R0e / PI2;
  /* ---- the rays ---- */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const int st0 = in->state[i];
    double Jss = in->Jss[i], Jpp = in->Jpp[i];
    double complex Jsp = in->Jsp[2 * i] + I * in->Jsp[2 * i + 1];
    if (st0 <= 0) {              /* not entering: both outputs are copies */
      beam_t* outs[2] = {gb, lb};
      for (int k = 0; k < 2; ++k) {
        beam_t* b = outs[k];
        b->x[i] = in->x[i]; b->y[i] = in->y[i]; b->z[i] = in->z[i];
        b->a[i] = in->a[i]; b->b[i] = in->b[i]; b->c[i] = in->c[i];
        b->path[i] = in->path[i]; b->E[i] = in->E[i];
        b->Jss[i] = Jss; b->Jpp[i] = Jpp;
        b->Jsp[2 * i] = creal(Jsp); b->Jsp[2 * i + 1] = cimag(Jsp);
        b->state[i] = st0;
      }
      theta[i] = 0.;
      continue;
    }
    double p[3], d[3], q1[3], q2[3], t1, t2;
    to_local(o, in, i, p, d);
    bracket(o, axis, positive, p, d, &t1, &t2);
    double dz1 = find_dz(o, t1, p, d, q1), dz2 = find_dz(o, t2, p, d, q2);
    double t, h[3];
    int lost = 0;
    if (dz1 <= 0.) {
      lost = 1; t = t1; h[0] = q1[0]; h[1] = q1[1]; h[2] = q1[2];
    } else if (dz2 >= 0.) {
      t = t2; h[0] = q2[0]; h[1] = q2[1]; h[2] = q2[2];
    } else {                     /* bracket-keeping secant, base.py:933-959 */
      int numit = 2, active = 1;
      while (active && numit < 100) {
        const double tt = t1, dz = dz1;
        t1 = t2; dz1 = dz2;
        t2 = tt - (t1 - tt) * dz / (dz1 - dz);
        if (t2 < t1min) t2 = t1min;
        if (t2 > t2max) t2 = t2max;
        dz2 = find_dz(o, t2, p, d, q2);
        if (!isnan(dz2) && !isnan(dz1) && sgn(dz2) == sgn(dz1)) { t1 = tt; dz1 = dz; }
        active = fabs(dz2) > zEps;
        ++numit;
      }
      t = t2; h[0] = q2[0]; h[1] = q2[1]; h[2] = q2[2];
    }
    int st = rays_good(o, h[0], h[1]);
    if (lost) st = o->lost_num;
    double oa = d[0], ob = d[1], oc = d[2], th = 0., path = in->path[i];
    double vJss = Jss, vJpp = Jpp;
    double complex vJsp = Jsp;
    if (st == 1) {
      path += t;
      double nx = 0, ny = 0, nz = 1;
      if (o->surf == 1) {        /* oes/__init__.py:403-411 */
        const double rx = 1 - (h[0] / o->r) * (h[0] / o->r);
        const double ax = rx < 0 ? 0 : pow(rx, -0.5);
        const double na = -h[0] / o->r * ax, nb = -h[1] / o->R;
        const double norm = sqrt(na * na + nb * nb + 1);
        nx = na / norm; ny = nb / norm; nz = 1. / norm;
      }
      double bdn = d[0] * nx + d[1] * ny + d[2] * nz;
      if (bdn < -1) bdn = -1;
      if (bdn > 1) bdn = 1;
      th = acos(bdn) - M_PI / 2;
      oa = d[0] - nx * 2 * bdn;
      ob = d[1] - ny * 2 * bdn;
      oc = d[2] - nz * 2 * bdn;
      const double ang = o->roll + atan2(nx, nz);
      /* coherency matrix into the local s/p frame: rotate by -ang */
      double c = cos(-ang), s = sin(-ang);
      double c2 = c * c, s2 = s * s, cs = c * s;
      double lJss = Jss * c2 + Jpp * s2 + 2 * creal(Jsp) * cs;
      double lJpp = Jss * s2 + Jpp * c2 - 2 * creal(Jsp) * cs;
      double complex lJsp = (Jpp - Jss) * cs + creal(Jsp) * (c2 - s2) + cimag(Jsp) * I;
      /* Fresnel amplitudes, material.py:415-493 */
      const double E = in->E[i];
      double complex xf = 0;
      for (int e = 0; e < o->nelem; ++e) xf += (o->Z[e] + interp_f(o, e, E)) * o->quantity[e];
      const double complex n2 = 1 - pre * (CH / E) * (CH / E) * o->rho * xf / o->mass;
      const double cosA = fabs(bdn);
      double sinA2 = 1 - bdn * bdn;
      if (sinA2 < 0) sinA2 = 0;
      const double complex cosB = csqrt(1 - (1. / n2) * (1. / n2) * sinA2);
      const double complex n2cosB = n2 * cosB;
      double complex rs = (cosA - n2cosB) / (cosA + n2cosB);
      double complex rp = (n2 * cosA - cosB) / (n2 * cosA + cosB);
      if (isnan(creal(rs)) || isnan(cimag(rs))) rs = 0;
      if (isnan(creal(rp)) || isnan(cimag(rp))) rp = 0;
      Jss = creal(lJss * rs * conj(rs));
      Jpp = creal(lJpp * rp * conj(rp));
      Jsp = lJsp * rs * conj(rp);
      /* and back for the outgoing beam: rotate by +ang */
      c = cos(ang); s = sin(ang);
      c2 = c * c; s2 = s * s; cs = c * s;
      vJss = Jss * c2 + Jpp * s2 + 2 * creal(Jsp) * cs;
      vJpp = Jss * s2 + Jpp * c2 - 2 * creal(Jsp) * cs;
      vJsp = (Jpp - Jss) * cs + creal(Jsp) * (c2 - s2) + cimag(Jsp) * I;
    }
    theta[i] = th;
    lb->x[i] = h[0]; lb->y[i] = h[1]; lb->z[i] = h[2];
    lb->a[i] = oa; lb->b[i] = ob; lb->c[i] = oc;
    lb->path[i] = path; lb->E[i] = in->E[i];
    lb->Jss[i] = Jss; lb->Jpp[i] = Jpp;
    lb->Jsp[2 * i] = creal(Jsp); lb->Jsp[2 * i + 1] = cimag(Jsp);
    lb->state[i] = st;
    if (st != 1 && st != 2) {     /* reflect.py:131-134 */
      gb->x[i] = in->x[i]; gb->y[i] = in->y[i]; gb->z[i] = in->z[i];
      gb->a[i] = in->a[i]; gb->b[i] = in->b[i]; gb->c[i] = in->c[i];
      gb->path[i] = in->path[i]; gb->E[i] = in->E[i];
      gb->Jss[i] = in->Jss[i]; gb->Jpp[i] = in->Jpp[i];
      gb->Jsp[2 * i] = in->Jsp[2 * i]; gb->Jsp[2 * i + 1] = in->Jsp[2 * i + 1];
      gb->state[i] = st;
      continue;
    }
    double x = h[0] + o->dx, y = h[1], z = h[2];
    rot3(&o->to_virgin, &x, &y, &z);
    rot3(&o->to_virgin, &oa, &ob, &oc);
    if (o->sin_az != 0.) {
      const double an = o->cos_az * oa + o->sin_az * ob, bn = -o->sin_az * oa + o->cos_az * ob;
      oa = an; ob = bn;
      const double xn = o->cos_az * x + o->sin_az * y, yn = -o->sin_az * x + o->cos_az * y;
      x = xn; y = yn;
    }
    gb->x[i] = x + o->center[0]; gb->y[i] = y + o->center[1]; gb->z[i] = z + o->center[2];
    gb->a[i] = oa; gb->b[i] = ob; gb->c[i] = oc;
    gb->path[i] = path; gb->E[i] = in->E[i];
    gb->Jss[i] = vJss; gb->Jpp[i] = vJpp;
    gb->Jsp[2 * i] = creal(vJsp); gb->Jsp[2 * i + 1] = cimag(vJsp);
    gb->state[i] = st;
  }
  return 0;
}
