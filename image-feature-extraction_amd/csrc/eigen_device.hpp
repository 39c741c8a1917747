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

struct EigFeat {
  float f[6];
};

// EigenvalueFeaturesFunctor.h:24-29
template <int TRIG>
__device__ __forceinline__ EigFeat eig_features(float A11, float A12, float A13, float A22,
                                                float A23, float A33) {
  const Eig3 ev = eig3_sym<TRIG>(A11, A12, A13, A22, A23, A33);
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
