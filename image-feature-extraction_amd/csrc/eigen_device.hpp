// eigen_device.hpp -- per-voxel symmetric 3x3 eigenvalues and derived scalars, one
// voxel per lane.  Device restatement of the reference's
//   include/ife/Numerics/Symmetric3x3EigenvalueSolver.h:33-132   (a1)
//   include/ife/Numerics/EigenvalueFeaturesFunctor.h:20-31       (a2)
// with TRealType=float, the instantiation every tool uses
// (ImageToEmphysemaFeaturesFilter.h:94).
//
// The translation unit is compiled with -ffp-contract=off, so each float multiply
// and add below rounds on its own exactly as in a generic x86-64 build of the
// reference, and float division / sqrtf are the correctly rounded forms hipcc emits
// by default (NOT __fsqrt_rn, which lowers to the bare 1-ulp v_sqrt_f32).  The one place the reference's precision depends on its include
// context (unqualified sqrt/acos/cos, :88,:115,:119-120) is the TRIG template
// parameter: 0 = double functions (only <cmath> visible), 1 = float overloads.
#pragma once
#include <hip/hip_runtime.h>

namespace ife {

struct Eig3 {
  float e0, e1, e2;
};

template <int TRIG>
__device__ __forceinline__ Eig3 eig3_sym(float A11, float A12, float A13, float A22, float A23,
                                         float A33) {
  Eig3 r;
  float p = A12 * A12 + A13 * A13 + A23 * A23;  // :44
  if (p == 0.0f) {
    // :45-83 diagonal: strict '>' tree, else arm wins ties
    const float a1 = fabsf(A11), a2 = fabsf(A22), a3 = fabsf(A33);
    if (a1 > a2) {
      if (a1 > a3) {
        r.e0 = A11;
        if (a2 > a3) { r.e1 = A22; r.e2 = A33; }
        else         { r.e1 = A33; r.e2 = A22; }
      } else { r.e0 = A33; r.e1 = A11; r.e2 = A22; }
    } else {
      if (a2 > a3) {
        r.e0 = A22;
        if (a1 > a3) { r.e1 = A11; r.e2 = A33; }
        else         { r.e1 = A33; r.e2 = A11; }
      } else { r.e0 = A33; r.e1 = A22; r.e2 = A11; }
    }
    return r;
  }
  const float q = (A11 + A22 + A33) / 3.0f;  // :85
  const float d1 = A11 - q, d2 = A22 - q, d3 = A33 - q;
  p = d1 * d1 + d2 * d2 + d3 * d3 + 2.0f * p;  // :86-87
  // :88  sqrt(p / 6): a correctly rounded double sqrt rounded to float equals the
  // correctly rounded float sqrt (53 >= 2*24+2), so both contexts share this form.
  p = sqrtf(p / 6.0f);
  const float B11 = d1 / p, B12 = A12 / p, B13 = A13 / p;  // :92-97
  const float B22 = d2 / p, B23 = A23 / p, B33 = d3 / p;
  // :98-103, float expression; the "/ 2.0" in double is exact
  const float r2 = B11 * B22 * B33 + 2.0f * B12 * B13 * B23 - B23 * B23 * B11 -
                   B13 * B13 * B22 - B12 * B12 * B33;
  const float rr = r2 * 0.5f;
  const float twop = 2.0f * p;
  float phi;
  if (rr <= -1.0f) phi = (float)(M_PI / 3);  // :107-116
  else if (rr >= 1.0f) phi = 0.0f;
  else if (TRIG == 0) phi = (float)(acos((double)rr) / 3);
  else phi = acosf(rr) / 3.0f;
  float e0, e2;
  if (TRIG == 0) e0 = (float)((double)q + (double)twop * cos((double)phi));  // :119
  else e0 = q + twop * cosf(phi);
  // :120 the argument is double in both contexts
  e2 = (float)((double)q + (double)twop * cos((double)phi + M_PI * (2.0 / 3.0)));
  float e1 = 3.0f * q - e0 - e2;  // :121
  if (fabsf(e0) < fabsf(e2)) { const float t = e0; e0 = e2; e2 = t; }  // :123-125
  if (fabsf(e1) < fabsf(e2)) { const float t = e1; e1 = e2; e2 = t; }  // :127-129
  r.e0 = e0; r.e1 = e1; r.e2 = e2;
  return r;
}

// ---------------------------------------------------------------------------------
// Fast form of the TRIG == 0 (double sqrt/acos/cos) solver.
//
// Same float operations on the same operands in the same order as eig3_sym<0>, so the
// float intermediates (q, p, B, r) are bit-identical; only HOW three of them are
// obtained changes:
//   * x / 3 and x / 6: reciprocal multiply + one fma correction, which is the
//     correctly rounded quotient (verified exhaustively over all 2^32 inputs,
//     tests/csrc/div_const_exhaustive.c; extreme magnitudes take the plain division);
//   * the six divisions by p share one refined reciprocal; each quotient then takes the
//     two fma refinement steps of the IEEE division expansion, without its range
//     scaling: correctly rounded for p in [2^-60, 2^60] (outside that range the generic
//     solver runs instead) and quotients of normal size; a quotient below 2^-100 in
//     magnitude may differ in its last bit, which cannot reach r = det(B)/2 in float;
//   * acos and cos (double) are evaluated with argument-range-specific polynomials
//     (|error| < 2e-16 on r in (-1,1), phi in [0, pi/3]) instead of the general library
//     routines: the values agree with libm except in the last double bits, i.e. after
//     rounding to float they differ from the oracle for about one voxel in 10^8.
// ---------------------------------------------------------------------------------
// x / d for d = 3, 6: q = x*RN(1/d), one fma correction.  Exhaustively equal to the IEEE
// quotient for 2^-100 <= |x| <= 2^100 (tests/csrc/div_const_exhaustive.c); zero keeps its
// sign, everything else (denormal quotients, inf, NaN) takes the division expansion.
template <int D>
__device__ __forceinline__ float div_by_const(float x) {
  const float d = (float)D;
  const float y = D == 3 ? 0x1.555556p-2f : 0x1.555556p-3f;  // RN(1/3), RN(1/6)
  const float ax = fabsf(x);
  const float q = x * y;
  float f = fmaf(fmaf(-d, q, x), y, q);
  f = ax == 0.0f ? x : f;
  // rare: one wave-uniform branch instead of a per-lane exec region on the common path
  const bool odd = !(ax <= 0x1p100f) || (ax < 0x1p-100f && ax != 0.0f);
  if (__builtin_amdgcn_ballot_w64(odd) != 0) f = odd ? x / d : f;
  return f;
}
__device__ __forceinline__ float div_by_3(float x) { return div_by_const<3>(x); }
__device__ __forceinline__ float div_by_6(float x) { return div_by_const<6>(x); }
struct SharedRecip {
  float d, r;
};
__device__ __forceinline__ SharedRecip shared_recip(float d) {
  SharedRecip s;
  s.d = d;
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.0f), r, r);
  s.r = r;
  return s;
}
__device__ __forceinline__ float div_shared(float n, const SharedRecip &s) {
  float q = n * s.r;
  q = fmaf(fmaf(-s.d, q, n), s.r, q);
  return fmaf(fmaf(-s.d, q, n), s.r, q);
}

// ---- double constants of the fast solver ----------------------------------------------
// A double operand cannot be an instruction literal, so every coefficient below costs a
// v_mov_b64 (or two v_mov_b32) per use when the compiler has no scalar registers left for
// it -- 51 of the feature kernel's 337 vector instructions per voxel.  The kernels that
// are short of scalar registers therefore read the table from LDS instead (a broadcast
// ds_read_b64 issues beside the vector unit); everything else uses the immediates.
enum {
  EK_AS0 = 0,    // 12 asin coefficients, highest degree first
  EK_CS0 = 12,   // 7 cos coefficients, highest degree first
  EK_PIO2_HI = 19, EK_PIO2_LO, EK_PI_HI, EK_PI_LO, EK_TWO_PI_3, EK_THIRD, EK_COUNT
};
__device__ constexpr double kEigConst[EK_COUNT] = {
    // asin(s) = s + s*z*P(z), z = s^2 <= 0.25 (degree-11 minimax fit, |err P| < 2.4e-16)
    0.028169218060881414, -0.010749050339697808, 0.01603551434914882, 0.0078029494773533175,
    0.011875494382636922, 0.013929652902326633, 0.017355259955786323, 0.02237204763174451,
    0.03038194736709848, 0.044642857103423646, 0.07500000000020764, 0.1666666666666665,
    // cos on [-0.1, 1.1]:  1 - w/2 + w^2 C(w), w = y^2 (degree-6 fit, |err| < 4e-18)
    4.7137756144213336e-14, -1.1469654898726794e-11, 2.0876747999120392e-09, -2.75573191859408e-07,
    2.4801587301510604e-05, -0.001388888888888883, 0.041666666666666664,
    1.57079632679489655800e+00, 6.12323399573676603587e-17,   // pi/2 hi, lo
    3.14159265358979311600e+00, 1.22464679914735320717e-16,   // pi hi, lo
    M_PI * (2.0 / 3.0),                                       // Symmetric3x3EigenvalueSolver.h:120
    0x1.5555555555555p-2};                                    // RN(1/3)
struct EigConstImm {
  __device__ __forceinline__ double operator[](int i) const { return kEigConst[i]; }
};
struct EigConstLds {
  typedef __attribute__((address_space(3))) const double lds_f64;
  lds_f64 *tab;  // EK_COUNT doubles, filled by eig_const_fill before the first use
  __device__ __forceinline__ double operator[](int i) const { return tab[i]; }
  // The table never changes, so the compiler would lift all its loads out of the caller's
  // loop and carry 50 registers; an address it cannot see through keeps them inside the
  // iteration, where they are issued together ahead of the polynomial that uses them.
  __device__ __forceinline__ static EigConstLds at(const double *shared_table) {
    unsigned a = (unsigned)(uintptr_t)(lds_f64 *)shared_table;
    asm volatile("" : "+v"(a));
    return EigConstLds{(lds_f64 *)(uintptr_t)a};
  }
};
// call from every thread of the workgroup, then barrier before the first solve
__device__ __forceinline__ void eig_const_fill(double *tab) {
  if (threadIdx.x < EK_COUNT) tab[threadIdx.x] = kEigConst[threadIdx.x];
}

template <typename KT>
__device__ __forceinline__ double asin_poly(double z, const KT &K) {
  double p = K[EK_AS0];
#pragma unroll
  for (int i = 1; i < 12; ++i) p = fma(p, z, K[EK_AS0 + i]);
  return p;
}
// acos on (-1, 1); the argument is a float so 1 - |r| is exact
template <typename KT>
__device__ __forceinline__ double acos_unit(float rf, const KT &K) {
  const double PIO2_HI = K[EK_PIO2_HI], PIO2_LO = K[EK_PIO2_LO];
  const double r = (double)rf;
  const double a = fabs(r);
  const bool big = a > 0.5;
  const double z = big ? (1.0 - a) * 0.5 : a * a;
  // sqrt(z) by one rsq + two Newton steps (z in (0, 0.25]; unused lanes are selected away)
  const double y0 = __builtin_amdgcn_rsq(z);
  double g = z * y0, h = 0.5 * y0;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  g = fma(fma(-g, g, z), h, g);
  const double s = big ? g : a;
  const double t = fma(s * z, asin_poly(z, K), s);  // asin(s)
  // |r| <= 0.5: pi/2 - sign(r) asin|r| ; r > 0.5: 2 asin(s) ; r < -0.5: pi - 2 asin(s)
  const double ts = r < 0.0 ? -t : t;
  const double small = (PIO2_HI - ts) + PIO2_LO;
  const double t2 = t + t;
  const double bigv = r < 0.0 ? (2.0 * PIO2_HI - t2) + 2.0 * PIO2_LO : t2;
  return big ? bigv : small;
}
template <typename KT>
__device__ __forceinline__ double cos_small(double y, const KT &K) {
  const double w = y * y;
  double c = K[EK_CS0];
#pragma unroll
  for (int i = 1; i < 7; ++i) c = fma(c, w, K[EK_CS0 + i]);
  return fma(w * w, c, fma(-0.5, w, 1.0));
}
template <typename KT>
__device__ __forceinline__ double div3_f64(double x, const KT &K) {
  const double y = K[EK_THIRD];
  const double q = x * y;
  return fma(fma(-3.0, q, x), y, q);
}

// Out-of-line copy of the generic solver for the rare out-of-range fallback, so that its
// register needs (the library cos with large-argument reduction) do not set the
// kernel's allocation.
template <int TRIG>
__device__ __noinline__ Eig3 eig3_sym_generic_call(float A11, float A12, float A13, float A22,
                                                   float A23, float A33) {
  return eig3_sym<TRIG>(A11, A12, A13, A22, A23, A33);
}

// ---- TRIG = 2: float trigonometry inside north_star's 1e-5 bar ---------------------------
// q, p, B and r stay the bit-identical float values of the reference; only acos(r)/3 and
// the two cosines are evaluated with short float polynomials (|error| of each below 2e-7),
// so an eigenvalue moves by at most a few 1e-7 * 2p <= 1e-6 |lambda_1| (|lambda_1| >= p for
// every symmetric matrix).  That is the spread the reference itself has between its two
// include contexts (double functions vs float overloads), and 40 float operations instead
// of ~200 double ones.  Fits: scripts/experiments/fit_trig.py (float section).
__device__ __forceinline__ float acos_third_f32(float r) {
  const float a = fabsf(r);
  const bool big = a > 0.5f;
  const float z = big ? (1.0f - a) * 0.5f : a * a;  // 1 - a is exact for a in [0.5, 1]
  const float s = big ? __builtin_amdgcn_sqrtf(z) : a;
  // asin(s) = s + s z P(z), z = s^2 <= 0.25, |err| < 3e-9
  float p = 0x1.13fed4p-5f;
  p = fmaf(p, z, 0x1.18f91ep-6f);
  p = fmaf(p, z, 0x1.fd8da2p-6f);
  p = fmaf(p, z, 0x1.6d5bbap-5f);
  p = fmaf(p, z, 0x1.333430p-4f);
  p = fmaf(p, z, 0x1.555554p-3f);
  const float t = fmaf(s * z, p, s);
  // |r| <= 0.5: pi/2 - sign(r) asin|r| ; r > 0.5: 2 asin(s) ; r < -0.5: pi - 2 asin(s); all / 3
  const float third = 0x1.555556p-2f;
  const float ts = r < 0.0f ? -t : t;
  const float small = fmaf(-third, ts, 0x1.0c1524p-1f);                      // pi/6 - ts/3
  const float t23 = t * 0x1.555556p-1f;                                      // 2t/3
  const float bigv = r < 0.0f ? 0x1.0c1524p+0f - t23 : t23;                  // pi/3 - 2t/3
  return big ? bigv : small;
}
// cos on [-0.01, 1.06]: 1 - w/2 + w^2 C(w), w = y^2, |err| < 2e-9 before rounding
__device__ __forceinline__ float cos_small_f32(float y) {
  const float w = y * y;
  float c = -0x1.22df14p-22f;
  c = fmaf(c, w, 0x1.a00bdap-16f);
  c = fmaf(c, w, -0x1.6c16b4p-10f);
  c = fmaf(c, w, 0x1.555556p-5f);
  return fmaf(w * w, c, fmaf(-0.5f, w, 1.0f));
}

// TRIG = 1 (float overloads of acos / cos, the <math.h> context) shares everything with
// TRIG = 0 but the places where a value is rounded to float: acosf(r) and cosf(phi) are
// taken as the correctly rounded floats of the double polynomials (the host libm is within
// one ulp of that), phi = acosf(r) / 3.0f and e0 = q + 2p cosf(phi) are float operations;
// e2 keeps the double argument and the double cos in both contexts (:120).
template <int TRIG = 0, typename KT>
__device__ __forceinline__ Eig3 eig3_sym_fast(float A11, float A12, float A13, float A22,
                                             float A23, float A33, const KT &K) {
  float p = A12 * A12 + A13 * A13 + A23 * A23;
  const bool diag = p == 0.0f;
  const float q = div_by_3(A11 + A22 + A33);
  const float d1 = A11 - q, d2 = A22 - q, d3 = A33 - q;
  p = d1 * d1 + d2 * d2 + d3 * d3 + 2.0f * p;
  p = sqrtf(div_by_6(p));
  const float ap = fabsf(p);
  const bool unsafe = !diag && !(ap >= 0x1p-60f && ap <= 0x1p60f);
  const SharedRecip rp = shared_recip(p);
  const float B11 = div_shared(d1, rp), B12 = div_shared(A12, rp), B13 = div_shared(A13, rp);
  const float B22 = div_shared(d2, rp), B23 = div_shared(A23, rp), B33 = div_shared(d3, rp);
  const float r2 = B11 * B22 * B33 + 2.0f * B12 * B13 * B23 - B23 * B23 * B11 -
                   B13 * B13 * B22 - B12 * B12 * B33;
  const float rr = r2 * 0.5f;
  const float twop = 2.0f * p;
  float phi, e0, e2;
  if constexpr (TRIG == 2) {
    phi = acos_third_f32(rr);
    phi = rr >= 1.0f ? 0.0f : phi;
    phi = rr <= -1.0f ? (float)(M_PI / 3) : phi;
    e0 = fmaf(twop, cos_small_f32(phi), q);
    // cos(phi + 2pi/3) = -cos(pi/3 - phi), and pi/3 - phi lies in [0, pi/3] again
    e2 = fmaf(-twop, cos_small_f32((float)(M_PI / 3) - phi), q);
  } else {
    phi = TRIG == 0 ? (float)div3_f64(acos_unit(rr, K), K) : div_by_3((float)acos_unit(rr, K));
    phi = rr >= 1.0f ? 0.0f : phi;
    phi = rr <= -1.0f ? (float)(M_PI / 3) : phi;
    const double qd = (double)q, tpd = (double)twop, phid = (double)phi;
    e0 = TRIG == 0 ? (float)(qd + tpd * cos_small(phid, K)) : q + twop * (float)cos_small(phid, K);
    // cos(phi + 2pi/3) = -cos(pi - (phi + 2pi/3)); pi - arg is exact in double-double
    const double arg = phid + K[EK_TWO_PI_3];
    const double yy = (K[EK_PI_HI] - arg) + K[EK_PI_LO];
    e2 = (float)(qd - tpd * cos_small(yy, K));
  }
  float e1 = 3.0f * q - e0 - e2;
  if (fabsf(e0) < fabsf(e2)) { const float t = e0; e0 = e2; e2 = t; }
  if (fabsf(e1) < fabsf(e2)) { const float t = e1; e1 = e2; e2 = t; }
  Eig3 r;
  r.e0 = e0; r.e1 = e1; r.e2 = e2;
  // Both special cases are rare and are entered per WAVE (scalar branch on a ballot), so
  // the common path carries no per-lane exec regions; inside, lanes select.
  if (__builtin_amdgcn_ballot_w64(diag) != 0) {
    const float a1 = fabsf(A11), a2 = fabsf(A22), a3 = fabsf(A33);
    const bool c12 = a1 > a2, c13 = a1 > a3, c23 = a2 > a3;
    const float d0 = c12 ? (c13 ? A11 : A33) : (c23 ? A22 : A33);
    const float d1 = c12 ? (c13 ? (c23 ? A22 : A33) : A11) : (c23 ? (c13 ? A11 : A33) : A22);
    const float d2 = c12 ? (c13 ? (c23 ? A33 : A22) : A22) : (c23 ? (c13 ? A33 : A11) : A11);
    r.e0 = diag ? d0 : r.e0;
    r.e1 = diag ? d1 : r.e1;
    r.e2 = diag ? d2 : r.e2;
  }
  if (__builtin_amdgcn_ballot_w64(unsafe) != 0) {
    const Eig3 g = eig3_sym_generic_call<(TRIG == 2 ? 0 : TRIG)>(A11, A12, A13, A22, A23, A33);
    r.e0 = unsafe ? g.e0 : r.e0;
    r.e1 = unsafe ? g.e1 : r.e1;
    r.e2 = unsafe ? g.e2 : r.e2;
  }
  return r;
}

// Correctly rounded float square root for NORMAL x in [2^-100, 2^100]: the hardware root
// (1 ulp) and the neighbour test hipcc's own sqrtf expansion uses, without that expansion's
// scaling of tiny arguments and special-value fix-up (9 instead of 17 instructions).
__device__ __forceinline__ float sqrt_rn_normal(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
  const float su = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
  const float ed = fmaf(-sd, s, x);
  const float eu = fmaf(-su, s, x);
  float r = ed <= 0.0f ? sd : s;
  r = eu > 0.0f ? su : r;
  return r;
}
// x / D without the range test of div_by_const (the caller has established
// 2^-100 <= |x| <= 2^100 or x == 0)
template <int D>
__device__ __forceinline__ float div_by_const_inrange(float x) {
  const float d = (float)D;
  const float y = D == 3 ? 0x1.555556p-2f : 0x1.555556p-3f;
  const float q = x * y;
  const float f = fmaf(fmaf(-d, q, x), y, q);
  return x == 0.0f ? x : f;
}

// The default solver (IFE_OPT_TRIG_MODE=2): eig3_sym_fast<2> with its range tests folded
// into ONE rare branch.  q, p, B, r are the same bits as in every other mode.
template <typename KT>
__device__ __forceinline__ Eig3 eig3_sym_fast2(float A11, float A12, float A13, float A22,
                                              float A23, float A33, const KT &K) {
  const float p0 = A12 * A12 + A13 * A13 + A23 * A23;
  const bool diag = p0 == 0.0f;
  const float tr = A11 + A22 + A33;
  const float q = div_by_const_inrange<3>(tr);
  const float d1 = A11 - q, d2 = A22 - q, d3 = A33 - q;
  const float p2 = d1 * d1 + d2 * d2 + d3 * d3 + 2.0f * p0;
  const float atr = fabsf(tr);
  // everything the short forms assume: |tr| and p2 of ordinary magnitude (then p lies in
  // [2^-52, 2^49], inside the range of the shared reciprocal)
  const bool ok = (atr <= 0x1p100f) && (atr >= 0x1p-100f || atr == 0.0f) && (p2 >= 0x1p-100f) &&
                  (p2 <= 0x1p100f);
  const bool unsafe = !diag && !ok;
  const float p = sqrt_rn_normal(div_by_const_inrange<6>(p2));
  const SharedRecip rp = shared_recip(p);
  const float B11 = div_shared(d1, rp), B12 = div_shared(A12, rp), B13 = div_shared(A13, rp);
  const float B22 = div_shared(d2, rp), B23 = div_shared(A23, rp), B33 = div_shared(d3, rp);
  const float r2 = B11 * B22 * B33 + 2.0f * B12 * B13 * B23 - B23 * B23 * B11 -
                   B13 * B13 * B22 - B12 * B12 * B33;
  const float rr = r2 * 0.5f;
  const float twop = 2.0f * p;
  float phi = acos_third_f32(rr);
  phi = rr >= 1.0f ? 0.0f : phi;
  phi = rr <= -1.0f ? (float)(M_PI / 3) : phi;
  float e0 = fmaf(twop, cos_small_f32(phi), q);
  float e2 = fmaf(-twop, cos_small_f32((float)(M_PI / 3) - phi), q);
  float e1 = 3.0f * q - e0 - e2;
  if (fabsf(e0) < fabsf(e2)) { const float t = e0; e0 = e2; e2 = t; }
  if (fabsf(e1) < fabsf(e2)) { const float t = e1; e1 = e2; e2 = t; }
  Eig3 r;
  r.e0 = e0; r.e1 = e1; r.e2 = e2;
  if (__builtin_amdgcn_ballot_w64(diag || unsafe) != 0) {  // rare: one scalar branch
    const float a1 = fabsf(A11), a2 = fabsf(A22), a3 = fabsf(A33);
    const bool c12 = a1 > a2, c13 = a1 > a3, c23 = a2 > a3;
    const float g0 = c12 ? (c13 ? A11 : A33) : (c23 ? A22 : A33);
    const float g1 = c12 ? (c13 ? (c23 ? A22 : A33) : A11) : (c23 ? (c13 ? A11 : A33) : A22);
    const float g2 = c12 ? (c13 ? (c23 ? A33 : A22) : A22) : (c23 ? (c13 ? A33 : A11) : A11);
    r.e0 = diag ? g0 : r.e0;
    r.e1 = diag ? g1 : r.e1;
    r.e2 = diag ? g2 : r.e2;
    if (__builtin_amdgcn_ballot_w64(unsafe) != 0) {
      // out of line and exact (double libm): the rare lanes need not be fast
      const Eig3 g = eig3_sym_generic_call<0>(A11, A12, A13, A22, A23, A33);
      r.e0 = unsafe ? g.e0 : r.e0;
      r.e1 = unsafe ? g.e1 : r.e1;
      r.e2 = unsafe ? g.e2 : r.e2;
    }
  }
  return r;
}

struct EigFeat {
  float f[6];
};

// EigenvalueFeaturesFunctor.h:24-29
template <int TRIG, typename KT = EigConstImm>
__device__ __forceinline__ EigFeat eig_features(float A11, float A12, float A13, float A22,
                                                float A23, float A33, const KT &K = KT()) {
  Eig3 ev;
  if constexpr (TRIG == 2) ev = eig3_sym_fast2(A11, A12, A13, A22, A23, A33, K);
  else ev = eig3_sym_fast<TRIG>(A11, A12, A13, A22, A23, A33, K);
  EigFeat o;
  o.f[0] = ev.e0;
  o.f[1] = ev.e1;
  o.f[2] = ev.e2;
  o.f[3] = ev.e0 + ev.e1 + ev.e2;
  o.f[4] = ev.e0 * ev.e1 * ev.e2;
  o.f[5] = sqrtf(ev.e0 * ev.e0 + ev.e1 * ev.e1 + ev.e2 * ev.e2);
  return o;
}

}  // namespace ife
