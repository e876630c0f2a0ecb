/* TEST INFRASTRUCTURE ONLY (never linked into libxrt_hip.so).
 *
 * C/OpenMP restatement of the reference's numpy Fresnel-Kirchhoff kernel
 * _diffraction_integral_conv (xrt/backends/raycing/waves.py:834-851), one output
 * pixel per loop iteration, samples summed in order. It exists to give bench.py a
 * CPU baseline that uses all host cores; it is validated
 * against the numpy restatement oracle/kirchhoff_np.py, which is the one pinned
 * to the reference's golden vectors (tests/test_oracle_p2_golden.py).
 *
 *   U    = i k/(4 pi) (nl + d.n/r) exp(i k r) / r              waves.py:840-844
 *   Es'  = sum Es U,  Ep' = sum Ep U                            :845-846
 *   abc' = sum k^2/(4 pi) (Es+Ep) U / r * (dx, dy, dz)          :847-850
 *
 * gcc -O2 -fopenmp -shared -fPIC kirchhoff_c.c -lm
 */
#include <math.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int xrt_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* n: nx, ny, nz arrays [ns]; Es, Ep, out*: interleaved (re, im) */
void xrt_oracle_kirchhoff(int64_t np, const double* px, const double* py, const double* pz,
                          int64_t ns, const double* sx, const double* sy, const double* sz,
                          const double* nx, const double* ny, const double* nz,
                          const double* nl, const double* k, const double* Es,
                          const double* Ep, double* oS, double* oP, double* oA, double* oB,
                          double* oC) {
  const double four_pi = 4 * 3.14159265358979323846;
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < np; ++p) {
    double S[2] = {0, 0}, P[2] = {0, 0}, A[2] = {0, 0}, B[2] = {0, 0}, C[2] = {0, 0};
    const double x = px[p], y = py[p], z = pz[p];
    for (int64_t s = 0; s < ns; ++s) {
      const double a = x - sx[s], b = y - sy[s], c = z - sz[s];
      const double r = sqrt(a * a + b * b + c * c);
      const double cosn = (a * nx[s] + b * ny[s] + c * nz[s]) / r;
      const double ph = k[s] * r;
      const double amp = k[s] / four_pi * (nl[s] + cosn) / r;
      /* U = i * amp * (cos ph + i sin ph) */
      const double ur = -amp * sin(ph), ui = amp * cos(ph);
      const double esr = Es[2 * s], esi = Es[2 * s + 1];
      const double epr = Ep[2 * s], epi = Ep[2 * s + 1];
      S[0] += esr * ur - esi * ui;
      S[1] += esr * ui + esi * ur;
      P[0] += epr * ur - epi * ui;
      P[1] += epr * ui + epi * ur;
      const double f = k[s] * k[s] / four_pi / r;
      const double qr = esr + epr, qi = esi + epi;
      const double wr = f * (qr * ur - qi * ui), wi = f * (qr * ui + qi * ur);
      A[0] += wr * a;
      A[1] += wi * a;
      B[0] += wr * b;
      B[1] += wi * b;
      C[0] += wr * c;
      C[1] += wi * c;
    }
    oS[2 * p] = S[0];
    oS[2 * p + 1] = S[1];
    oP[2 * p] = P[0];
    oP[2 * p + 1] = P[1];
    oA[2 * p] = A[0];
    oA[2 * p + 1] = A[1];
    oB[2 * p] = B[0];
    oB[2 * p + 1] = B[1];
    oC[2 * p] = C[0];
    oC[2 * p + 1] = C[1];
  }
}
