#define _GNU_SOURCE /* sincos */
/*
 * ife_oracle.c -- CPU restatement (plain C) of the reference's per-voxel Hessian
 * feature path.  TEST INFRASTRUCTURE ONLY: see ife_oracle.h for who may load it
 * and for the parity status of each part (eigen solver pinned by the reference's
 * known-answer tests; ITK-wired stages "parity unpinned").
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: the reference is built for generic x86-64 (no FMA),
 * so every multiply and add below rounds separately, and the HIP path is built the
 * same way so that it can reproduce these results bit for bit.
 *
 * All paths below are relative to /root/reference.
 */
#include "ife_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void ife_or_set_threads(int n) {
  g_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
  omp_set_num_threads(g_threads);
#endif
}
int ife_or_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------------- */
/* a1: include/ife/Numerics/Symmetric3x3EigenvalueSolver.h:33-132             */
/* ------------------------------------------------------------------------- */

#define IFE_DIAG_SORT(T, ABS)                                                       \
  /* :45-83 nested strict '>' tree; on ties the else arm wins */                     \
  if (ABS(A11) > ABS(A22)) {                                                        \
    if (ABS(A11) > ABS(A33)) {                                                      \
      ev[0] = A11;                                                                  \
      if (ABS(A22) > ABS(A33)) { ev[1] = A22; ev[2] = A33; }                        \
      else                     { ev[1] = A33; ev[2] = A22; }                        \
    } else { ev[0] = A33; ev[1] = A11; ev[2] = A22; }                               \
  } else {                                                                          \
    if (ABS(A22) > ABS(A33)) {                                                      \
      ev[0] = A22;                                                                  \
      if (ABS(A11) > ABS(A33)) { ev[1] = A11; ev[2] = A33; }                        \
      else                     { ev[1] = A33; ev[2] = A11; }                        \
    } else { ev[0] = A33; ev[1] = A22; ev[2] = A11; }                               \
  }

void ife_or_eig3_f64(const double A[6], double ev[3]) {
  double A11 = A[0], A12 = A[1], A13 = A[2], A22 = A[3], A23 = A[4], A33 = A[5]; /* :37-42 */
  double p = A12 * A12 + A13 * A13 + A23 * A23;                                   /* :44 */
  if (p == 0) {
    IFE_DIAG_SORT(double, fabs)
  } else {
    double q = (A11 + A22 + A33) / 3;                                             /* :85 */
    p = (A11 - q) * (A11 - q) + (A22 - q) * (A22 - q) + (A33 - q) * (A33 - q) + 2 * p;
    p = sqrt(p / 6);                                                              /* :88 */
    double B11 = (A11 - q) / p, B12 = A12 / p, B13 = A13 / p;
    double B22 = (A22 - q) / p, B23 = A23 / p, B33 = (A33 - q) / p;               /* :92-97 */
    double r = (B11 * B22 * B33 + 2 * B12 * B13 * B23 - B23 * B23 * B11 - B13 * B13 * B22 -
                B12 * B12 * B33) / 2.0;                                           /* :98-103 */
    double phi;
    if (r <= -1) phi = M_PI / 3;
    else if (r >= 1) phi = 0;
    else phi = acos(r) / 3;                                                       /* :107-116 */
    ev[0] = q + 2 * p * cos(phi);
    ev[2] = q + 2 * p * cos(phi + M_PI * (2.0 / 3.0));
    ev[1] = 3 * q - ev[0] - ev[2];                                                /* :119-121 */
    if (fabs(ev[0]) < fabs(ev[2])) { double t = ev[0]; ev[0] = ev[2]; ev[2] = t; } /* :123-125 */
    if (fabs(ev[1]) < fabs(ev[2])) { double t = ev[1]; ev[1] = ev[2]; ev[2] = t; } /* :127-129 */
  }
}

void ife_or_eig3_f32(const float A[6], float ev[3], int trig_mode) {
  float A11 = A[0], A12 = A[1], A13 = A[2], A22 = A[3], A23 = A[4], A33 = A[5];
  float p = A12 * A12 + A13 * A13 + A23 * A23;
  if (p == 0) {
    IFE_DIAG_SORT(float, fabsf)
  } else {
    float q = (A11 + A22 + A33) / 3;
    p = (A11 - q) * (A11 - q) + (A22 - q) * (A22 - q) + (A33 - q) * (A33 - q) + 2 * p;
    /* :88 unqualified sqrt: ::sqrt(double) with <cmath> only, the float overload
     * once <math.h> has pulled std::sqrt into the global namespace. */
    if (trig_mode == IFE_OR_TRIG_CMATH) p = (float)sqrt((double)(p / 6));
    else p = sqrtf(p / 6);
    float B11 = (A11 - q) / p, B12 = A12 / p, B13 = A13 / p;
    float B22 = (A22 - q) / p, B23 = A23 / p, B33 = (A33 - q) / p;
    /* float expression, then "/ 2.0" in double (exact), rounded back to float */
    float r = (float)((double)(B11 * B22 * B33 + 2 * B12 * B13 * B23 - B23 * B23 * B11 -
                               B13 * B13 * B22 - B12 * B12 * B33) / 2.0);
    float phi;
    if (r <= -1) phi = (float)(M_PI / 3);
    else if (r >= 1) phi = 0;
    else if (trig_mode == IFE_OR_TRIG_CMATH) phi = (float)(acos((double)r) / 3);
    else phi = acosf(r) / 3;
    if (trig_mode == IFE_OR_TRIG_CMATH)
      ev[0] = (float)((double)q + (double)(2 * p) * cos((double)phi));
    else
      ev[0] = q + 2 * p * cosf(phi);
    /* phi + M_PI*(2.0/3.0) is double in both contexts, so cos is the double one */
    ev[2] = (float)((double)q + (double)(2 * p) * cos((double)phi + M_PI * (2.0 / 3.0)));
    ev[1] = 3 * q - ev[0] - ev[2];
    if (fabsf(ev[0]) < fabsf(ev[2])) { float t = ev[0]; ev[0] = ev[2]; ev[2] = t; }
    if (fabsf(ev[1]) < fabsf(ev[2])) { float t = ev[1]; ev[1] = ev[2]; ev[2] = t; }
  }
}

/* ------------------------------------------------------------------------- */
/* a2: include/ife/Numerics/EigenvalueFeaturesFunctor.h:20-31                  */
/* ------------------------------------------------------------------------- */
void ife_or_eigfeat_f64(const double A[6], double f[6]) {
  double ev[3];
  ife_or_eig3_f64(A, ev);
  f[0] = ev[0]; f[1] = ev[1]; f[2] = ev[2];
  f[3] = ev[0] + ev[1] + ev[2];
  f[4] = ev[0] * ev[1] * ev[2];
  f[5] = sqrt(ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2]);
}

void ife_or_eigfeat_f32(const float A[6], float f[6], int trig_mode) {
  float ev[3];
  ife_or_eig3_f32(A, ev, trig_mode);
  f[0] = ev[0]; f[1] = ev[1]; f[2] = ev[2];
  f[3] = ev[0] + ev[1] + ev[2];
  f[4] = ev[0] * ev[1] * ev[2];
  f[5] = sqrtf(ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2]); /* std::sqrt(float) */
}

void ife_or_eig3_batch_f32(const float *A6, int64_t n, float *ev3, int trig_mode) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) ife_or_eig3_f32(A6 + 6 * i, ev3 + 3 * i, trig_mode);
}
void ife_or_eigfeat_batch_f32(const float *A6, int64_t n, float *f6, int trig_mode) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) ife_or_eigfeat_f32(A6 + 6 * i, f6 + 6 * i, trig_mode);
}
void ife_or_eig3_batch_f64(const double *A6, int64_t n, double *ev3) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) ife_or_eig3_f64(A6 + 6 * i, ev3 + 3 * i);
}

/* sin and cos of the filter's pole angles.  ITK writes std::sin(x) and std::cos(x) side by
 * side (itkRecursiveGaussianImageFilter.hxx, ComputeNCoefficients / ComputeDCoefficients); GCC,
 * the compiler of the reference's Linux build, turns such a pair into ONE sincos(x) call, and
 * glibc's sincos differs from its cos in the last bit for some arguments (W1 / sigma_d with
 * sigma_d = 9.92: 0x1.fed70ab75a41dp-1 against ...41cp-1).  The cancellation in 1 + sum(D)
 * amplifies that bit to 1e-13 of an output sample, i.e. to one float ulp at about one sample
 * in 10^8-10^9.  Both sides of the parity tests therefore call sincos explicitly: which of
 * the two a build gets no longer depends on a compiler's optimiser (found by
 * tests/test_gpu_fuzz.py: the device library's host code is compiled by clang, which keeps
 * the two calls apart). */
static void pole_sincos(double x, double *s, double *c) { sincos(x, s, c); }

/* ------------------------------------------------------------------------- */
/* a4 pieces [ITK-upstream, parity unpinned]: itk::RecursiveGaussianImageFilter */
/* ::SetUp / ComputeNCoefficients / ComputeDCoefficients /                      */
/* ComputeRemainingCoefficients (ZeroOrder, NormalizeAcrossScale off), called   */
/* through NormalizedGaussianConvolutionImageFilter.hxx:51-55.                  */
/* ------------------------------------------------------------------------- */
int ife_or_gauss_coeffs_zero_order(double sigma, double spacing, ife_or_gauss_coeffs *c) {
  const double A1 = 1.3530, B1 = 1.8151, W1 = 0.6681, L1 = -1.3932;
  const double A2 = -0.3531, B2 = 0.0902, W2 = 2.0787, L2 = -1.3732;
  if (spacing < 0.0) spacing = -spacing;
  if (spacing < 1e-8) return -1;
  const double sigmad = sigma / spacing;

  {
    double Sin1, Sin2, Cos1, Cos2;
    pole_sincos(W1 / sigmad, &Sin1, &Cos1);
    pole_sincos(W2 / sigmad, &Sin2, &Cos2);
    (void)Sin1; (void)Sin2;
    const double Exp1 = exp(L1 / sigmad), Exp2 = exp(L2 / sigmad);
    c->D4 = Exp1 * Exp1 * Exp2 * Exp2;
    c->D3 = -2 * Cos1 * Exp1 * Exp2 * Exp2;
    c->D3 += -2 * Cos2 * Exp2 * Exp1 * Exp1;
    c->D2 = 4 * Cos2 * Cos1 * Exp1 * Exp2;
    c->D2 += Exp1 * Exp1 + Exp2 * Exp2;
    c->D1 = -2 * (Exp2 * Cos2 + Exp1 * Cos1);
  }
  const double SD = 1.0 + c->D1 + c->D2 + c->D3 + c->D4;
  double SN;
  {
    double Sin1, Sin2, Cos1, Cos2;
    pole_sincos(W1 / sigmad, &Sin1, &Cos1);
    pole_sincos(W2 / sigmad, &Sin2, &Cos2);
    const double Exp1 = exp(L1 / sigmad), Exp2 = exp(L2 / sigmad);
    c->N0 = A1 + A2;
    c->N1 = Exp2 * (B2 * Sin2 - (A2 + 2 * A1) * Cos2);
    c->N1 += Exp1 * (B1 * Sin1 - (A1 + 2 * A2) * Cos1);
    c->N2 = (A1 + A2) * Cos2 * Cos1;
    c->N2 -= B1 * Cos2 * Sin1 + B2 * Cos1 * Sin2;
    c->N2 *= 2 * Exp1 * Exp2;
    c->N2 += A2 * Exp1 * Exp1 + A1 * Exp2 * Exp2;
    c->N3 = Exp2 * Exp1 * Exp1 * (B2 * Sin2 - A2 * Cos2);
    c->N3 += Exp1 * Exp2 * Exp2 * (B1 * Sin1 - A1 * Cos1);
    SN = c->N0 + c->N1 + c->N2 + c->N3;
  }
  const double alpha0 = 2 * SN / SD - c->N0;
  c->N0 *= 1.0 / alpha0;
  c->N1 *= 1.0 / alpha0;
  c->N2 *= 1.0 / alpha0;
  c->N3 *= 1.0 / alpha0;

  /* symmetric case */
  c->M1 = c->N1 - c->D1 * c->N0;
  c->M2 = c->N2 - c->D2 * c->N0;
  c->M3 = c->N3 - c->D3 * c->N0;
  c->M4 = -c->D4 * c->N0;

  /* edge-extension boundary coefficients */
  const double SN2 = c->N0 + c->N1 + c->N2 + c->N3;
  const double SM2 = c->M1 + c->M2 + c->M3 + c->M4;
  const double SD2 = 1.0 + c->D1 + c->D2 + c->D3 + c->D4;
  c->BN1 = c->D1 * SN2 / SD2;
  c->BN2 = c->D2 * SN2 / SD2;
  c->BN3 = c->D3 * SN2 / SD2;
  c->BN4 = c->D4 * SN2 / SD2;
  c->BM1 = c->D1 * SM2 / SD2;
  c->BM2 = c->D2 * SM2 / SD2;
  c->BM3 = c->D3 * SM2 / SD2;
  c->BM4 = c->D4 * SM2 / SD2;
  return 0;
}

/* [ITK-upstream] itk::RecursiveSeparableImageFilter::FilterDataArray, RealType=double.
 * ln must be >= 4 (ITK throws otherwise). */
void ife_or_iir_line(const double *data, double *outs, double *s, int64_t ln,
                     const ife_or_gauss_coeffs *c) {
  const double N0 = c->N0, N1 = c->N1, N2 = c->N2, N3 = c->N3;
  const double D1 = c->D1, D2 = c->D2, D3 = c->D3, D4 = c->D4;
  const double M1 = c->M1, M2 = c->M2, M3 = c->M3, M4 = c->M4;
  const double BN1 = c->BN1, BN2 = c->BN2, BN3 = c->BN3, BN4 = c->BN4;
  const double BM1 = c->BM1, BM2 = c->BM2, BM3 = c->BM3, BM4 = c->BM4;

  const double outV1 = data[0];
  s[0] = outV1 * N0 + outV1 * N1 + outV1 * N2 + outV1 * N3;
  s[1] = data[1] * N0 + outV1 * N1 + outV1 * N2 + outV1 * N3;
  s[2] = data[2] * N0 + data[1] * N1 + outV1 * N2 + outV1 * N3;
  s[3] = data[3] * N0 + data[2] * N1 + data[1] * N2 + outV1 * N3;

  s[0] -= outV1 * BN1 + outV1 * BN2 + outV1 * BN3 + outV1 * BN4;
  s[1] -= s[0] * D1 + outV1 * BN2 + outV1 * BN3 + outV1 * BN4;
  s[2] -= s[1] * D1 + s[0] * D2 + outV1 * BN3 + outV1 * BN4;
  s[3] -= s[2] * D1 + s[1] * D2 + s[0] * D3 + outV1 * BN4;

  for (int64_t i = 4; i < ln; i++) {
    s[i] = data[i] * N0 + data[i - 1] * N1 + data[i - 2] * N2 + data[i - 3] * N3;
    s[i] -= s[i - 1] * D1 + s[i - 2] * D2 + s[i - 3] * D3 + s[i - 4] * D4;
  }
  for (int64_t i = 0; i < ln; i++) outs[i] = s[i];

  const double outV2 = data[ln - 1];
  s[ln - 1] = outV2 * M1 + outV2 * M2 + outV2 * M3 + outV2 * M4;
  s[ln - 2] = data[ln - 1] * M1 + outV2 * M2 + outV2 * M3 + outV2 * M4;
  s[ln - 3] = data[ln - 2] * M1 + data[ln - 1] * M2 + outV2 * M3 + outV2 * M4;
  s[ln - 4] = data[ln - 3] * M1 + data[ln - 2] * M2 + data[ln - 1] * M3 + outV2 * M4;

  s[ln - 1] -= outV2 * BM1 + outV2 * BM2 + outV2 * BM3 + outV2 * BM4;
  s[ln - 2] -= s[ln - 1] * D1 + outV2 * BM2 + outV2 * BM3 + outV2 * BM4;
  s[ln - 3] -= s[ln - 2] * D1 + s[ln - 1] * D2 + outV2 * BM3 + outV2 * BM4;
  s[ln - 4] -= s[ln - 3] * D1 + s[ln - 2] * D2 + s[ln - 1] * D3 + outV2 * BM4;

  for (int64_t i = ln - 4; i > 0; i--) {
    s[i - 1] = data[i] * M1 + data[i + 1] * M2 + data[i + 2] * M3 + data[i + 3] * M4;
    s[i - 1] -= s[i] * D1 + s[i + 1] * D2 + s[i + 2] * D3 + s[i + 3] * D4;
  }
  for (int64_t i = 0; i < ln; i++) outs[i] += s[i];
}

/* ------------------------------------------------------------------------- */
/* Row f4.  [ITK-upstream, parity unpinned] itk::RecursiveGaussianImageFilter::SetUp for   */
/* the three orders (ZeroOrder above is the order-0 case of this).  Constants of the       */
/* exponential series per order, ComputeNCoefficients / ComputeDCoefficients, the         */
/* normalisations alpha0 / alpha1 / alpha2 (unit response to a constant, a unit ramp, a    */
/* unit parabola, in PIXEL units) and ComputeRemainingCoefficients(symmetric): the first   */
/* order is antisymmetric (M = -(N - D N0), M4 = +D4 N0).  NormalizeAcrossScale off.       */
/* The reference never instantiates orders 1 and 2: it only sketches the differential      */
/* normalized convolution in a comment (NormalizedGaussianConvolutionImageFilter.h:28-44). */
/* ------------------------------------------------------------------------- */
static void n_coefficients(double sigmad, double A1, double B1, double W1, double L1, double A2,
                           double B2, double W2, double L2, double *N0, double *N1, double *N2,
                           double *N3, double *SN, double *DN, double *EN) {
  double Sin1, Sin2, Cos1, Cos2;
  pole_sincos(W1 / sigmad, &Sin1, &Cos1);
  pole_sincos(W2 / sigmad, &Sin2, &Cos2);
  const double Exp1 = exp(L1 / sigmad), Exp2 = exp(L2 / sigmad);
  *N0 = A1 + A2;
  *N1 = Exp2 * (B2 * Sin2 - (A2 + 2 * A1) * Cos2);
  *N1 += Exp1 * (B1 * Sin1 - (A1 + 2 * A2) * Cos1);
  *N2 = (A1 + A2) * Cos2 * Cos1;
  *N2 -= B1 * Cos2 * Sin1 + B2 * Cos1 * Sin2;
  *N2 *= 2 * Exp1 * Exp2;
  *N2 += A2 * Exp1 * Exp1 + A1 * Exp2 * Exp2;
  *N3 = Exp2 * Exp1 * Exp1 * (B2 * Sin2 - A2 * Cos2);
  *N3 += Exp1 * Exp2 * Exp2 * (B1 * Sin1 - A1 * Cos1);
  *SN = *N0 + *N1 + *N2 + *N3;
  *DN = *N1 + 2 * *N2 + 3 * *N3;
  *EN = *N1 + 4 * *N2 + 9 * *N3;
}

int ife_or_gauss_coeffs_order(double sigma, double spacing, int order, ife_or_gauss_coeffs *c) {
  const double A1[3] = {1.3530, -0.6724, -1.3563}, B1[3] = {1.8151, -3.4327, 5.2318};
  const double A2[3] = {-0.3531, 0.6724, 0.3446}, B2[3] = {0.0902, 0.6100, -2.2355};
  const double W1 = 0.6681, L1 = -1.3932, W2 = 2.0787, L2 = -1.3732;
  if (order == 0) return ife_or_gauss_coeffs_zero_order(sigma, spacing, c);
  if (order != 1 && order != 2) return -1;
  double direction = 1.0;
  if (spacing < 0.0) { direction = -1.0; spacing = -spacing; }
  if (spacing < 1e-8) return -1;
  const double sigmad = sigma / spacing;
  {
    double Sin1, Sin2, Cos1, Cos2;
    pole_sincos(W1 / sigmad, &Sin1, &Cos1);
    pole_sincos(W2 / sigmad, &Sin2, &Cos2);
    (void)Sin1; (void)Sin2;
    const double Exp1 = exp(L1 / sigmad), Exp2 = exp(L2 / sigmad);
    c->D4 = Exp1 * Exp1 * Exp2 * Exp2;
    c->D3 = -2 * Cos1 * Exp1 * Exp2 * Exp2;
    c->D3 += -2 * Cos2 * Exp2 * Exp1 * Exp1;
    c->D2 = 4 * Cos2 * Cos1 * Exp1 * Exp2;
    c->D2 += Exp1 * Exp1 + Exp2 * Exp2;
    c->D1 = -2 * (Exp2 * Cos2 + Exp1 * Cos1);
  }
  const double SD = 1.0 + c->D1 + c->D2 + c->D3 + c->D4;
  const double DD = c->D1 + 2 * c->D2 + 3 * c->D3 + 4 * c->D4;
  const double ED = c->D1 + 4 * c->D2 + 9 * c->D3 + 16 * c->D4;
  int symmetric;
  if (order == 1) {
    double SN, DN, EN;
    n_coefficients(sigmad, A1[1], B1[1], W1, L1, A2[1], B2[1], W2, L2, &c->N0, &c->N1, &c->N2,
                   &c->N3, &SN, &DN, &EN);
    double alpha1 = 2 * (SN * DD - DN * SD) / (SD * SD);
    alpha1 *= direction;
    c->N0 *= 1.0 / alpha1; c->N1 *= 1.0 / alpha1; c->N2 *= 1.0 / alpha1; c->N3 *= 1.0 / alpha1;
    symmetric = 0;
  } else {
    double N0_0, N1_0, N2_0, N3_0, N0_2, N1_2, N2_2, N3_2, SN0, DN0, EN0, SN2, DN2, EN2;
    n_coefficients(sigmad, A1[0], B1[0], W1, L1, A2[0], B2[0], W2, L2, &N0_0, &N1_0, &N2_0, &N3_0,
                   &SN0, &DN0, &EN0);
    n_coefficients(sigmad, A1[2], B1[2], W1, L1, A2[2], B2[2], W2, L2, &N0_2, &N1_2, &N2_2, &N3_2,
                   &SN2, &DN2, &EN2);
    const double beta = -(2 * SN2 - SD * N0_2) / (2 * SN0 - SD * N0_0);
    const double N0 = N0_2 + beta * N0_0, N1 = N1_2 + beta * N1_0;
    const double N2 = N2_2 + beta * N2_0, N3 = N3_2 + beta * N3_0;
    const double SN = SN2 + beta * SN0, DN = DN2 + beta * DN0, EN = EN2 + beta * EN0;
    const double alpha2 = (EN * SD * SD - ED * SN * SD - 2 * DN * DD * SD + 2 * DD * DD * SN) /
                          (SD * SD * SD);
    c->N0 = N0 * (1.0 / alpha2); c->N1 = N1 * (1.0 / alpha2);
    c->N2 = N2 * (1.0 / alpha2); c->N3 = N3 * (1.0 / alpha2);
    symmetric = 1;
  }
  if (symmetric) {
    c->M1 = c->N1 - c->D1 * c->N0;
    c->M2 = c->N2 - c->D2 * c->N0;
    c->M3 = c->N3 - c->D3 * c->N0;
    c->M4 = -c->D4 * c->N0;
  } else {
    c->M1 = -(c->N1 - c->D1 * c->N0);
    c->M2 = -(c->N2 - c->D2 * c->N0);
    c->M3 = -(c->N3 - c->D3 * c->N0);
    c->M4 = c->D4 * c->N0;
  }
  const double SN2b = c->N0 + c->N1 + c->N2 + c->N3;
  const double SM2b = c->M1 + c->M2 + c->M3 + c->M4;
  const double SD2b = 1.0 + c->D1 + c->D2 + c->D3 + c->D4;
  c->BN1 = c->D1 * SN2b / SD2b; c->BN2 = c->D2 * SN2b / SD2b;
  c->BN3 = c->D3 * SN2b / SD2b; c->BN4 = c->D4 * SN2b / SD2b;
  c->BM1 = c->D1 * SM2b / SD2b; c->BM2 = c->D2 * SM2b / SD2b;
  c->BM3 = c->D3 * SM2b / SD2b; c->BM4 = c->D4 * SM2b / SD2b;
  return 0;
}

static int64_t axis_len(const ife_or_dims *d, int a) { return a == 0 ? d->nx : a == 1 ? d->ny : d->nz; }
static double axis_sp(const ife_or_dims *d, int a) { return a == 0 ? d->sx : a == 1 ? d->sy : d->sz; }
static int64_t axis_stride(const ife_or_dims *d, int a) {
  return a == 0 ? 1 : a == 1 ? d->nx : d->nx * d->ny;
}

/* [ITK-upstream] RecursiveSeparableImageFilter::ThreadedGenerateData: copy a line to
 * double, FilterDataArray, cast each sample back to the (float) output pixel type. */
int ife_or_recursive_gaussian_axis(const float *in, float *out, const ife_or_dims *d, int axis,
                                   double sigma) {
  return ife_or_recursive_gaussian_axis_order(in, out, d, axis, sigma, 0);
}

int ife_or_recursive_gaussian_axis_order(const float *in, float *out, const ife_or_dims *d, int axis,
                                         double sigma, int order) {
  const int64_t ln = axis_len(d, axis);
  if (ln < 4) return -2;
  ife_or_gauss_coeffs c;
  if (ife_or_gauss_coeffs_order(sigma, axis_sp(d, axis), order, &c)) return -1;
  const int64_t st = axis_stride(d, axis);
  const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
  const int64_t n1 = axis_len(d, a1), n2 = axis_len(d, a2);
  const int64_t s1 = axis_stride(d, a1), s2 = axis_stride(d, a2);
  int err = 0;
#pragma omp parallel
  {
    double *buf = (double *)malloc(sizeof(double) * 3 * (size_t)ln);
    if (!buf) {
#pragma omp atomic write
      err = 1;
    } else {
      double *inps = buf, *outs = buf + ln, *scr = buf + 2 * ln;
#pragma omp for collapse(2) schedule(static)
      for (int64_t j = 0; j < n2; ++j)
        for (int64_t i = 0; i < n1; ++i) {
          const int64_t base = i * s1 + j * s2;
          for (int64_t k = 0; k < ln; ++k) inps[k] = in[base + k * st];
          ife_or_iir_line(inps, outs, scr, ln, &c);
          for (int64_t k = 0; k < ln; ++k) out[base + k * st] = (float)outs[k];
        }
      free(buf);
    }
  }
  return err ? -3 : 0;
}

/* [ITK-upstream] itk::SmoothingRecursiveGaussianImageFilter: first filter along
 * direction ImageDimension-1 (Z), then directions 0 (X) and 1 (Y); float images
 * between the axis passes, final cast to float. */
int ife_or_smoothing_recursive_gaussian(const float *in, float *out, const ife_or_dims *d,
                                        double sigma) {
  const int64_t n = d->nx * d->ny * d->nz;
  float *tmp = (float *)malloc(sizeof(float) * (size_t)n);
  if (!tmp) return -3;
  int rc = ife_or_recursive_gaussian_axis(in, tmp, d, 2, sigma);
  if (!rc) rc = ife_or_recursive_gaussian_axis(tmp, out, d, 0, sigma);
  if (!rc) {
    memcpy(tmp, out, sizeof(float) * (size_t)n);
    rc = ife_or_recursive_gaussian_axis(tmp, out, d, 1, sigma);
  }
  free(tmp);
  return rc;
}

/* a4: NormalizedGaussianConvolutionImageFilter.hxx:40-63.
 * Multiply (:48-49), two smoothings (:51-55), Divide (:57-61) with
 * [ITK-upstream] Functor::Div: B != 0 ? A / B : NumericTraits<float>::max(). */
int ife_or_normalized_gaussian_convolution(const float *image, const float *certainty,
                                           float *out, const ife_or_dims *d, double sigma) {
  const int64_t n = d->nx * d->ny * d->nz;
  float *tc = (float *)malloc(sizeof(float) * (size_t)n);
  float *g1 = (float *)malloc(sizeof(float) * (size_t)n);
  float *g2 = (float *)malloc(sizeof(float) * (size_t)n);
  int rc = (tc && g1 && g2) ? 0 : -3;
  if (!rc) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) tc[i] = image[i] * certainty[i];
    rc = ife_or_smoothing_recursive_gaussian(tc, g1, d, sigma);
  }
  if (!rc) rc = ife_or_smoothing_recursive_gaussian(certainty, g2, d, sigma);
  if (!rc) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = (g2[i] != 0.0f) ? g1[i] / g2[i] : FLT_MAX;
  }
  free(tc); free(g1); free(g2);
  return rc;
}

/* Row f4: the differential normalized convolution the reference sketches at
 * NormalizedGaussianConvolutionImageFilter.h:28-44:
 *   d/dx [{a*cT}/{a*c}] = ({a_x*cT}{a*c} - {a_x*c}{a*cT}) / {a*c}^2
 * with a the Gaussian of the 0th-order filter and a_x the same separable filter with the
 * FirstOrder recursive Gaussian along `axis` (the axes run Z, X, Y like
 * SmoothingRecursiveGaussianImageFilter; float images between the passes).  The first-order
 * filter answers in pixel units; dividing by the spacing of the axis gives physical units
 * (as itk::GradientRecursiveGaussianImageFilter does).  Float arithmetic, evaluated as
 * written; a zero denominator gives NumericTraits<float>::max() like the Div functor. */
static int smooth_with_order(const float *in, float *out, const ife_or_dims *d, double sigma,
                             int deriv_axis) {
  const int64_t n = d->nx * d->ny * d->nz;
  float *tmp = (float *)malloc(sizeof(float) * (size_t)n);
  if (!tmp) return -3;
  int rc = ife_or_recursive_gaussian_axis_order(in, tmp, d, 2, sigma, deriv_axis == 2);
  if (!rc) rc = ife_or_recursive_gaussian_axis_order(tmp, out, d, 0, sigma, deriv_axis == 0);
  if (!rc) {
    memcpy(tmp, out, sizeof(float) * (size_t)n);
    rc = ife_or_recursive_gaussian_axis_order(tmp, out, d, 1, sigma, deriv_axis == 1);
  }
  free(tmp);
  return rc;
}

int ife_or_differential_normalized_convolution(const float *image, const float *certainty,
                                               float *out, const ife_or_dims *d, double sigma,
                                               int axis) {
  if (axis < 0 || axis > 2) return -1;
  const int64_t n = d->nx * d->ny * d->nz;
  float *tc = (float *)malloc(sizeof(float) * (size_t)n);
  float *g[4] = {0, 0, 0, 0};
  int rc = tc ? 0 : -3;
  for (int k = 0; k < 4 && !rc; ++k)
    if (!(g[k] = (float *)malloc(sizeof(float) * (size_t)n))) rc = -3;
  if (!rc) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) tc[i] = image[i] * certainty[i];
    rc = smooth_with_order(tc, g[0], d, sigma, -1);                 /* a * cT   */
  }
  if (!rc) rc = smooth_with_order(certainty, g[1], d, sigma, -1);   /* a * c    */
  if (!rc) rc = smooth_with_order(tc, g[2], d, sigma, axis);        /* a_x * cT */
  if (!rc) rc = smooth_with_order(certainty, g[3], d, sigma, axis); /* a_x * c  */
  if (!rc) {
    const float inv_sp = (float)(1.0 / axis_sp(d, axis));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      const float den = g[1][i] * g[1][i];
      const float num = g[2][i] * g[1][i] - g[3][i] * g[0][i];
      out[i] = (den != 0.0f) ? (num / den) * inv_sp : FLT_MAX;
    }
  }
  free(tc);
  for (int k = 0; k < 4; ++k) free(g[k]);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* a3 pieces [ITK-upstream, parity unpinned]: itk::DerivativeOperator           */
/* ::GenerateCoefficients, FlipAxes, ScaleCoefficients and the                  */
/* NeighborhoodOperatorImageFilter inner product (double accumulate, start at   */
/* 0, ZeroFluxNeumann = index clamp), result cast to float.                     */
/* ------------------------------------------------------------------------- */
static void derivative_operator(int order, double scale, double coeff[3]) {
  /* w = 2*((order+1)/2)+1 = 3 for order 1 and 2 */
  coeff[0] = 0.0; coeff[1] = 1.0; coeff[2] = 0.0;
  const int w = 3;
  double previous, next;
  int i, j;
  for (i = 0; i < order / 2; i++) {
    previous = coeff[1] - 2 * coeff[0];
    for (j = 1; j < w - 1; j++) {
      next = coeff[j - 1] + coeff[j + 1] - 2 * coeff[j];
      coeff[j - 1] = previous;
      previous = next;
    }
    next = coeff[j - 1] - 2 * coeff[j];
    coeff[j - 1] = previous;
    coeff[j] = next;
  }
  for (i = 0; i < order % 2; i++) {
    previous = 0.5 * coeff[1];
    for (j = 1; j < w - 1; j++) {
      next = -0.5 * coeff[j - 1] + 0.5 * coeff[j + 1];
      coeff[j - 1] = previous;
      previous = next;
    }
    next = -0.5 * coeff[j - 1];
    coeff[j - 1] = previous;
    coeff[j] = next;
  }
  /* FlipAxes */
  double t = coeff[0]; coeff[0] = coeff[2]; coeff[2] = t;
  /* ScaleCoefficients */
  for (i = 0; i < 3; ++i) coeff[i] = coeff[i] * scale;
}

static double deriv_scale(double spacing, int order, int dscale_mode) {
  double s = 1.0 / spacing;
  if (dscale_mode == IFE_OR_DSCALE_POW && order == 2) s = s * s;
  return s;
}

static inline int64_t clampi(int64_t v, int64_t hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

int ife_or_derivative(const float *in, float *out, const ife_or_dims *d, int order,
                      int direction, int dscale_mode) {
  if (order < 1 || order > 2 || direction < 0 || direction > 2) return -1;
  const double sp = axis_sp(d, direction);
  if (sp == 0.0) return -1;
  double co[3];
  derivative_operator(order, deriv_scale(sp, order, dscale_mode), co);
  const int64_t nx = d->nx, ny = d->ny, nz = d->nz;
  const int64_t st = axis_stride(d, direction), len = axis_len(d, direction);
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t z = 0; z < nz; ++z)
    for (int64_t y = 0; y < ny; ++y)
      for (int64_t x = 0; x < nx; ++x) {
        const int64_t idx = x + nx * (y + ny * z);
        const int64_t p = direction == 0 ? x : direction == 1 ? y : z;
        const int64_t im = idx + (clampi(p - 1, len - 1) - p) * st;
        const int64_t ip = idx + (clampi(p + 1, len - 1) - p) * st;
        double sum = 0.0;
        sum += co[0] * (double)in[im];
        sum += co[1] * (double)in[idx];
        sum += co[2] * (double)in[ip];
        out[idx] = (float)sum;
      }
  return 0;
}

/* a3: Hessian3DImageFilter.hxx:13-60 wiring; component order xx,xy,xz,yy,yz,zz (:54-59) */
int ife_or_hessian3d(const float *in, float *out6, const ife_or_dims *d, int dscale_mode) {
  const int64_t n = d->nx * d->ny * d->nz;
  float *t = (float *)malloc(sizeof(float) * (size_t)n * 3);
  if (!t) return -3;
  float *dx = t, *dy = t + n, *c = t + 2 * n;
  int rc = ife_or_derivative(in, dx, d, 1, 0, dscale_mode);           /* :31-34 */
  if (!rc) rc = ife_or_derivative(in, dy, d, 1, 1, dscale_mode);      /* :35-37 */
  struct { const float *src; int order, dir, comp; } plan[6] = {
      {in, 2, 0, 0},  /* Dxx :19-22 */
      {dx, 1, 1, 1},  /* Dxy = D_y(Dx) :39-43 */
      {dx, 1, 2, 2},  /* Dxz = D_z(Dx) :44-47 */
      {in, 2, 1, 3},  /* Dyy */
      {dy, 1, 2, 4},  /* Dyz = D_z(Dy) :48-51 */
      {in, 2, 2, 5},  /* Dzz */
  };
  for (int k = 0; k < 6 && !rc; ++k) {
    rc = ife_or_derivative(plan[k].src, c, d, plan[k].order, plan[k].dir, dscale_mode);
    if (!rc) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) out6[i * 6 + plan[k].comp] = c[i];
    }
  }
  free(t);
  return rc;
}

/* [ITK-upstream] itk::GradientMagnitudeImageFilter::ThreadedGenerateData:
 * RealType=double, per axis g = inner product with the order-1 operator scaled by
 * 1/spacing, a += g*g in axis order 0,1,2, out = float(std::sqrt(a)). */
int ife_or_gradient_magnitude(const float *in, float *out, const ife_or_dims *d) {
  double co[3][3];
  for (int a = 0; a < 3; ++a) {
    if (axis_sp(d, a) == 0.0) return -1;
    derivative_operator(1, 1.0 / axis_sp(d, a), co[a]);
  }
  const int64_t nx = d->nx, ny = d->ny, nz = d->nz;
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t z = 0; z < nz; ++z)
    for (int64_t y = 0; y < ny; ++y)
      for (int64_t x = 0; x < nx; ++x) {
        const int64_t idx = x + nx * (y + ny * z);
        const int64_t pos[3] = {x, y, z};
        double a = 0.0;
        for (int ax = 0; ax < 3; ++ax) {
          const int64_t st = axis_stride(d, ax), len = axis_len(d, ax);
          const int64_t im = idx + (clampi(pos[ax] - 1, len - 1) - pos[ax]) * st;
          const int64_t ip = idx + (clampi(pos[ax] + 1, len - 1) - pos[ax]) * st;
          double g = 0.0;
          g += co[ax][0] * (double)in[im];
          g += co[ax][1] * (double)in[idx];
          g += co[ax][2] * (double)in[ip];
          a += g * g;
        }
        out[idx] = (float)sqrt(a);
      }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* a5: ImageToEmphysemaFeaturesFilter.hxx:15-55 (wiring), :99-121 (GenerateData) */
/* out comps: [S, |grad S|, ev1, ev2, ev3, LoG, product, Frobenius], each through */
/* MaskImageFilter (mask != 0 ? v : 0).                                          */
/* ------------------------------------------------------------------------- */
int ife_or_emphysema_features(const float *image, const uint8_t *mask, float *out8,
                              const ife_or_dims *d, double sigma, int trig_mode,
                              int dscale_mode) {
  const int64_t n = d->nx * d->ny * d->nz;
  float *cert = (float *)malloc(sizeof(float) * (size_t)n);
  float *S = (float *)malloc(sizeof(float) * (size_t)n);
  float *G = (float *)malloc(sizeof(float) * (size_t)n);
  float *H = (float *)malloc(sizeof(float) * (size_t)n * 6);
  int rc = (cert && S && G && H) ? 0 : -3;
  if (!rc) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) cert[i] = (float)mask[i]; /* CastImageFilter .hxx:21,110 */
    rc = ife_or_normalized_gaussian_convolution(image, cert, S, d, sigma);
  }
  if (!rc) rc = ife_or_gradient_magnitude(S, G, d);
  if (!rc) rc = ife_or_hessian3d(S, H, d, dscale_mode);
  if (!rc) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      float f[6];
      ife_or_eigfeat_f32(H + 6 * i, f, trig_mode); /* every voxel, also outside the mask */
      float *o = out8 + 8 * i;
      if (mask[i] != 0) {
        o[0] = S[i]; o[1] = G[i];
        for (int k = 0; k < 6; ++k) o[2 + k] = f[k];
      } else {
        for (int k = 0; k < 8; ++k) o[k] = 0.0f;
      }
    }
  }
  free(cert); free(S); free(G); free(H);
  return rc;
}

/* a6: tools/FiniteDifference_HessianFeatures.cxx:126-229 (dead tool); normative
 * definition = a3 o a2 o mask, WITHOUT the :155 direction bug.  mask==0 -> six 0. */
int ife_or_fd_hessian_features(const float *image, const uint8_t *mask, float *out6,
                               const ife_or_dims *d, int trig_mode, int dscale_mode) {
  const int64_t n = d->nx * d->ny * d->nz;
  int rc = ife_or_hessian3d(image, out6, d, dscale_mode);
  if (rc) return rc;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float *o = out6 + 6 * i;
    if (mask && mask[i] == 0) {
      for (int k = 0; k < 6; ++k) o[k] = 0.0f;
    } else {
      float f[6];
      ife_or_eigfeat_f32(o, f, trig_mode);
      for (int k = 0; k < 6; ++k) o[k] = f[k];
    }
  }
  return 0;
}

/* a7: tools/FiniteDifference_GradientFeatures.cxx:104-113; the mask is a float image there */
int ife_or_fd_gradient_features(const float *image, const float *mask, float *out,
                                const ife_or_dims *d) {
  const int64_t n = d->nx * d->ny * d->nz;
  int rc = ife_or_gradient_magnitude(image, out, d);
  if (rc) return rc;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    if (mask[i] == 0.0f) out[i] = 0.0f;
  return 0;
}

/* a8: tools/MaskedImageFilter.cxx:75-93 */
void ife_or_mask_image_f64(const double *image, const double *mask, double outside, double *out,
                           int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) out[i] = (mask[i] != 0.0) ? image[i] : outside;
}

/* ------------------------------------------------------------------------------------ */
/* f1: include/ife/Statistics/DetermineEdgesForEqualizedHistogram.h:21-137, on indices   */
/* ------------------------------------------------------------------------------------ */
#define IFE_OR_EDGES(NAME, T)                                                               \
  int NAME(const T *v, int64_t n, int64_t nbins, T *edges) {                                \
    if (n < 0) return 2;               /* :31-33 */                                         \
    if (n < nbins) return 1;           /* :36-38 */                                         \
    if (nbins < 1) return 1;                                                                \
    const int64_t per_bin = n / nbins; /* :40 */                                            \
    int64_t surplus = n - per_bin * nbins, deficit = 0, nedge = 0, it = 0; /* :41-44 */     \
    while (nedge + 1 < nbins) {        /* :45 */                                            \
      int64_t index = per_bin;                                                              \
      if (surplus) {                   /* :50-58 */                                         \
        int64_t take = surplus / (nbins - nedge);                                           \
        if (take == 0) take = 1;                                                            \
        index += take;                                                                      \
        surplus -= take;                                                                    \
      } else if (deficit) {            /* :59-67 */                                         \
        int64_t take = deficit / (nbins - nedge);                                           \
        if (take == 0) take = 1;                                                            \
        index -= take;                                                                      \
        deficit -= take;                                                                    \
      }                                                                                     \
      if (!(n - it > index)) return 3; /* assert :74 */                                     \
      it += index;                     /* :75 */                                            \
      const T x = v[it];                                                                    \
      int64_t lo = 0, hi = it;         /* lower_bound(first, it, *it) :82 */                \
      while (lo < hi) {                                                                     \
        const int64_t mid = lo + (hi - lo) / 2;                                             \
        if (v[mid] < x) lo = mid + 1; else hi = mid;                                        \
      }                                                                                     \
      const int64_t lb = lo;                                                                \
      if (lb != it) {                  /* :86 */                                            \
        lo = it; hi = n;               /* upper_bound(it, last, *it) :88 */                 \
        while (lo < hi) {                                                                   \
          const int64_t mid = lo + (hi - lo) / 2;                                           \
          if (!(x < v[mid])) lo = mid + 1; else hi = mid;                                   \
        }                                                                                   \
        const int64_t ub = lo;                                                              \
        if (ub == n) {                 /* :90-95 */                                         \
          it = lb;                                                                          \
        } else {                                                                            \
          const int64_t lbdist = it - lb, ubdist = ub - it; /* :102-108 */                  \
          if (lbdist < ubdist || (lbdist == ubdist && deficit)) { /* :116-124 */            \
            it = lb;                                                                        \
            if (lbdist > deficit) { surplus = lbdist - deficit; deficit = 0; }              \
            else deficit -= lbdist;                                                         \
          } else {                     /* :125-134 */                                       \
            it = ub;                                                                        \
            if (ubdist > surplus) { deficit = ubdist - surplus; surplus = 0; }              \
            else surplus -= ubdist;                                                         \
          }                                                                                 \
        }                                                                                   \
      }                                                                                     \
      edges[nedge++] = v[it];          /* :138-139 */                                       \
    }                                                                                       \
    return 0;                                                                               \
  }
IFE_OR_EDGES(ife_or_equalized_edges_f32, float)
IFE_OR_EDGES(ife_or_equalized_edges_f64, double)

static int cmp_f32(const void *a, const void *b) {
  const float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}
/* tools/DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures.cxx:284 */
void ife_or_sort_f32(float *v, int64_t n) { qsort(v, (size_t)n, sizeof(float), cmp_f32); }

/* same tool, :221-236 (nSamples == 0): raster order, first matching foreground value wins */
int64_t ife_or_gather_foreground(const float *features, int ncomp, const uint8_t *mask,
                                 int64_t nvox, const uint32_t *fg, int nfg, float **columns) {
  int64_t m = 0;
  for (int64_t i = 0; i < nvox; ++i) {
    int hit = 0;
    for (int k = 0; k < nfg && !hit; ++k) hit = ((uint32_t)mask[i] == fg[k]);
    if (!hit) continue;
    if (columns)
      for (int c = 0; c < ncomp; ++c) columns[c][m] = features[i * ncomp + c];
    ++m;
  }
  return m;
}

/* f2: include/ife/Statistics/DenseHistogram.h:47-60 */
void ife_or_dense_histogram_f32(const float *edges, int64_t nedges, const float *values,
                                int64_t n, uint32_t *counts, float *freqs) {
  for (int64_t b = 0; b <= nedges; ++b) counts[b] = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t lo = 0, hi = nedges; /* lower_bound(edges, value): first edge >= value (:48) */
    while (lo < hi) {
      const int64_t mid = lo + (hi - lo) / 2;
      if (edges[mid] < values[i]) lo = mid + 1; else hi = mid;
    }
    ++counts[lo];
  }
  /* getFrequencies :55-60: the sum accumulates in int (the literal 0), then becomes float */
  int sum = 0;
  for (int64_t b = 0; b <= nedges; ++b) sum = (int)(sum + counts[b]);
  const float fsum = (float)sum;
  for (int64_t b = 0; b <= nedges; ++b) freqs[b] = (float)counts[b] / fsum;
}

/* f2: tools/MakeBag.cxx:405-472 */
int ife_or_roi_histograms(const float *features, int ncomp, const uint8_t *mask,
                          const ife_or_dims *d, const int64_t *rois, int nrois,
                          const float *edges, int64_t nedges, uint32_t *counts, float *freqs) {
  const int64_t nb = nedges + 1;
  for (int r = 0; r < nrois; ++r) {
    const int64_t *q = rois + 6 * r;
    if (q[0] < 0 || q[1] < 0 || q[2] < 0 || q[3] < 0 || q[4] < 0 || q[5] < 0 ||
        q[0] + q[3] > d->nx || q[1] + q[4] > d->ny || q[2] + q[5] > d->nz)
      return 4;
  }
  for (int r = 0; r < nrois; ++r) {
    const int64_t *q = rois + 6 * r;
    uint32_t *cr = counts + (int64_t)r * ncomp * nb;
    for (int64_t k = 0; k < ncomp * nb; ++k) cr[k] = 0;
    for (int64_t z = q[2]; z < q[2] + q[5]; ++z)
      for (int64_t y = q[1]; y < q[1] + q[4]; ++y)
        for (int64_t x = q[0]; x < q[0] + q[3]; ++x) {
          const int64_t i = x + d->nx * (y + d->ny * z);
          if (!mask[i]) continue; /* :438 */
          for (int c = 0; c < ncomp; ++c) {
            const float v = features[i * ncomp + c];
            const float *e = edges + (int64_t)c * nedges;
            int64_t lo = 0, hi = nedges;
            while (lo < hi) {
              const int64_t mid = lo + (hi - lo) / 2;
              if (e[mid] < v) lo = mid + 1; else hi = mid;
            }
            ++cr[c * nb + lo];
          }
        }
    for (int c = 0; c < ncomp; ++c) { /* getFrequencies, DenseHistogram.h:55-60 */
      int sum = 0;
      for (int64_t b = 0; b < nb; ++b) sum = (int)(sum + cr[c * nb + b]);
      const float fsum = (float)sum;
      for (int64_t b = 0; b < nb; ++b)
        freqs[((int64_t)r * ncomp + c) * nb + b] = (float)cr[c * nb + b] / fsum;
    }
  }
  return 0;
}
