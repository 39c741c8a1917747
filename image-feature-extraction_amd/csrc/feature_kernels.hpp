// feature_kernels.hpp -- fused finite-difference Hessian + gradient magnitude +
// eigenvalue features + mask, one pass over the (smoothed) volume.
//
// Fuses what the reference runs as ~25 separate ITK filters per scale:
//   DivideImageFilter            NormalizedGaussianConvolutionImageFilter.hxx:57-61
//   GradientMagnitudeImageFilter ImageToEmphysemaFeaturesFilter.hxx:27-28
//   Hessian3DImageFilter         Hessian3DImageFilter.hxx:13-60 (8 DerivativeImageFilters)
//   UnaryFunctorImageFilter<EigenvalueFeaturesFunctor>  ImageToEmphysemaFeaturesFilter.hxx:33-35
//   6x VectorIndexSelectionCast, 8x MaskImageFilter, Compose   :37-54
// and, for the un-smoothed tool body, tools/FiniteDifference_HessianFeatures.cxx:126-229.
//
// Arithmetic restated from ITK ("parity unpinned"):
// every DerivativeImageFilter is a 3-tap inner product accumulated in double
// starting from 0.0, replicate (ZeroFluxNeumann) boundary, result cast to float; the
// cross terms are chained first differences with the float round trip in between;
// gradient magnitude accumulates g*g in double and casts sqrt to float.
//
// Mapping: a workgroup of 512 threads owns a 64 x 8 XY tile and marches along z.
// Each plane is staged once (with a one-voxel replicate halo) in an LDS tile,
// double buffered so that one barrier per plane suffices; the global loads of plane
// p+1 are issued before plane p is consumed.  Per thread a three-plane ring of
// centre values and first differences lives in registers, so every smoothed value is
// read from HBM once per z-chunk (plus halo) and every output written once.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "eigen_device.hpp"

namespace ife {

enum FeatMode {
  FEAT_FEATURES8 = 0,  // S, |grad|, ev1..3, LoG, product, Frobenius (a5)
  FEAT_EIG6 = 1,       // ev1..3, LoG, product, Frobenius (a6)
  FEAT_HESSIAN6 = 2,   // xx, xy, xz, yy, yz, zz (a3)
  FEAT_GRADMAG = 3     // |grad| (a7)
};

template <int MODE>
struct FeatNOut {
  static constexpr int value = MODE == FEAT_FEATURES8 ? 8 : (MODE == FEAT_GRADMAG ? 1 : 6);
};

struct FeatGeom {
  int nx, ny, nz;
  int zchunk;
  int64_t plane;  // nx*ny
  int64_t nvox;
};

// Operator coefficients after FlipAxes and ScaleCoefficients (double), per axis.
struct DerivCoef {
  double m1[3], p1[3];  // order 1: coefficient of f[i-1] and of f[i+1]
  double a2[3], b2[3], c2[3];  // order 2: coefficients of f[i-1], f[i], f[i+1]
};

// value sources -------------------------------------------------------------------
// Smoothed field S = num/den with ITK's Div functor (B != 0 ? A/B : max()); den may
// be null when the certainty is identically one (then den == 1.0f exactly).
struct ValSmooth {
  const float *num;
  const float *den;
  __device__ __forceinline__ float ld(int64_t i) const {
    const float a = num[i];
    if (den == nullptr) return a;
    const float b = den[i];
    return b != 0.0f ? a / b : FLT_MAX;
  }
};
template <typename TI>
struct ValRaw {
  const TI *img;
  __device__ __forceinline__ float ld(int64_t i) const { return (float)img[i]; }
};

__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// 3-tap operators, accumulated the way NeighborhoodInnerProduct does (sum starts at 0).
__device__ __forceinline__ double d1(double cm, double cp, float fm, float fp) {
  double s = 0.0;
  s += cm * (double)fm;
  s += cp * (double)fp;
  return s;
}
__device__ __forceinline__ float d2(double ca, double cb, double cc, float fm, float f0,
                                    float fp) {
  double s = 0.0;
  s += ca * (double)fm;
  s += cb * (double)f0;
  s += cc * (double)fp;
  return (float)s;
}

constexpr int FT_TX = 64;
constexpr int FT_TY = 8;
constexpr int FT_HX = FT_TX + 2;
constexpr int FT_HY = FT_TY + 2;
constexpr int FT_NE = FT_HX * FT_HY;  // 660 staged elements per plane
constexpr int FT_THREADS = FT_TX * FT_TY;

template <int MODE, typename VAL, typename TM>
__global__ __launch_bounds__(FT_THREADS) void features_kernel(VAL val, const TM *__restrict__ mask,
                                                              float *__restrict__ out, FeatGeom g,
                                                              DerivCoef dc, int planar, int trig) {
  __shared__ float tile[2][FT_HY][FT_HX];
  // mask tile of the output plane, staged as dwords (per-lane sub-dword global loads are
  // slow on gfx950); row pitch = 64 mask elements
  constexpr int MT_DW = FT_TY * FT_TX * (int)sizeof(TM) / 4;  // dwords per plane tile
  constexpr int MT_DWROW = FT_TX * (int)sizeof(TM) / 4;       // dwords per tile row
  __shared__ uint32_t mtile[2][MT_DW];
  constexpr int NOUT = FeatNOut<MODE>::value;
  constexpr bool NEED_H = MODE != FEAT_GRADMAG;
  constexpr bool NEED_G = MODE == FEAT_FEATURES8 || MODE == FEAT_GRADMAG;

  const int tid = threadIdx.x;
  const int tx = tid & 63, ty = tid >> 6;
  const int x = blockIdx.x * FT_TX + tx, y = blockIdx.y * FT_TY + ty;
  const int z0 = blockIdx.z * g.zchunk;
  const int z1 = min(z0 + g.zchunk, g.nz);
  const bool inb = x < g.nx && y < g.ny;

  // staging assignment: elements tid and tid+512 of the (HY x HX) halo tile
  const int e0 = tid, e1 = tid + FT_THREADS;
  const bool has1 = e1 < FT_NE;
  const int e0y = e0 / FT_HX, e0x = e0 % FT_HX;
  const int e1y = has1 ? e1 / FT_HX : 0, e1x = has1 ? e1 % FT_HX : 0;
  const int64_t off0 = (int64_t)clampi(blockIdx.x * FT_TX - 1 + e0x, g.nx - 1) +
                       (int64_t)g.nx * clampi(blockIdx.y * FT_TY - 1 + e0y, g.ny - 1);
  const int64_t off1 = (int64_t)clampi(blockIdx.x * FT_TX - 1 + e1x, g.nx - 1) +
                       (int64_t)g.nx * clampi(blockIdx.y * FT_TY - 1 + e1y, g.ny - 1);

  float c_m = 0, c_0 = 0, c_p = 0;
  float dx_m = 0, dx_0 = 0, dx_p = 0, dy_m = 0, dy_0 = 0, dy_p = 0;
  float dxx_0 = 0, dxx_p = 0, dyy_0 = 0, dyy_p = 0, dxy_0 = 0, dxy_p = 0;
  double axy_0 = 0, axy_p = 0;

  float r0, r1 = 0.0f;
  {
    const int64_t pb = (int64_t)clampi(z0 - 1, g.nz - 1) * g.plane;
    r0 = val.ld(pb + off0);
    if (has1) r1 = val.ld(pb + off1);
  }
  // mask staging: thread t < MT_DW owns dword (t % MT_DWROW) of tile row (t / MT_DWROW).
  // Vector form needs rows that are dword multiples and a dword-aligned base.
  const bool mvec = mask != nullptr && ((int64_t)g.nx * (int64_t)sizeof(TM)) % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(mask) & 3) == 0;
  const int mrow = tid / MT_DWROW, mcol = tid % MT_DWROW;
  const int mx = blockIdx.x * FT_TX + mcol * (4 / (int)sizeof(TM));
  const int my = blockIdx.y * FT_TY + mrow;
  const bool mine = mvec && tid < MT_DW && mx < g.nx && my < g.ny;
  const int64_t moff = (int64_t)mx + (int64_t)g.nx * my;  // element offset inside a plane
  uint32_t mr = 0;

  for (int p = z0 - 1; p <= z1; ++p) {
    const int buf = (p - (z0 - 1)) & 1;
    tile[buf][e0y][e0x] = r0;
    if (has1) tile[buf][e1y][e1x] = r1;
    if (mine) mtile[buf][tid] = mr;  // mask of plane p-1, loaded one iteration ago
    if (p < z1) {  // issue the next plane's loads before consuming this one
      const int64_t pb = (int64_t)clampi(p + 1, g.nz - 1) * g.plane;
      r0 = val.ld(pb + off0);
      if (has1) r1 = val.ld(pb + off1);
      // plane p is the output plane of the next iteration
      if (mine && p >= z0)
        mr = *reinterpret_cast<const uint32_t *>(mask + (int64_t)p * g.plane + moff);
    }
    __syncthreads();

    // in-plane quantities of plane p at (x, y); tile coordinates are +1
    const float (*t)[FT_HX] = tile[buf];
    const float fc = t[ty + 1][tx + 1];
    const float fxm = t[ty + 1][tx], fxp = t[ty + 1][tx + 2];
    const float fym = t[ty][tx + 1], fyp = t[ty + 2][tx + 1];
    const double gx = d1(dc.m1[0], dc.p1[0], fxm, fxp);
    const double gy = d1(dc.m1[1], dc.p1[1], fym, fyp);

    c_m = c_0; c_0 = c_p; c_p = fc;
    dx_m = dx_0; dx_0 = dx_p; dx_p = (float)gx;
    dy_m = dy_0; dy_0 = dy_p; dy_p = (float)gy;
    if (NEED_G) {
      axy_0 = axy_p;
      double a = 0.0;
      a += gx * gx;
      a += gy * gy;
      axy_p = a;
    }
    if (NEED_H) {
      dxx_0 = dxx_p; dyy_0 = dyy_p; dxy_0 = dxy_p;
      dxx_p = d2(dc.a2[0], dc.b2[0], dc.c2[0], fxm, fc, fxp);
      dyy_p = d2(dc.a2[1], dc.b2[1], dc.c2[1], fym, fc, fyp);
      // Dxy = D_y(Dx): Dx (as a float image) at rows y-1 and y+1
      const float dx_ym = (float)d1(dc.m1[0], dc.p1[0], t[ty][tx], t[ty][tx + 2]);
      const float dx_yp = (float)d1(dc.m1[0], dc.p1[0], t[ty + 2][tx], t[ty + 2][tx + 2]);
      dxy_p = (float)d1(dc.m1[1], dc.p1[1], dx_ym, dx_yp);
    }

    const int z = p - 1;
    if (z >= z0 && inb) {
      const int64_t idx = (int64_t)x + (int64_t)g.nx * ((int64_t)y + (int64_t)g.ny * z);
      bool keep = true;
      if (mvec)
        keep = reinterpret_cast<const TM *>(mtile[buf])[ty * FT_TX + tx] != (TM)0;
      else if (mask != nullptr)
        keep = mask[idx] != (TM)0;
      float o[NOUT];
#pragma unroll
      for (int k = 0; k < NOUT; ++k) o[k] = 0.0f;
      if (keep) {
        float G = 0.0f;
        if (NEED_G) {
          const double gz = d1(dc.m1[2], dc.p1[2], c_m, c_p);
          double a = axy_0;
          a += gz * gz;
          G = (float)sqrt(a);
        }
        if constexpr (MODE == FEAT_GRADMAG) {
          o[0] = G;
        } else {
          const float dzz = d2(dc.a2[2], dc.b2[2], dc.c2[2], c_m, c_0, c_p);
          const float dxz = (float)d1(dc.m1[2], dc.p1[2], dx_m, dx_p);
          const float dyz = (float)d1(dc.m1[2], dc.p1[2], dy_m, dy_p);
          if constexpr (MODE == FEAT_HESSIAN6) {
            o[0] = dxx_0; o[1] = dxy_0; o[2] = dxz; o[3] = dyy_0; o[4] = dyz; o[5] = dzz;
          } else {
            EigFeat ef;
            if (trig == 0) ef = eig_features<0>(dxx_0, dxy_0, dxz, dyy_0, dyz, dzz);
            else ef = eig_features<1>(dxx_0, dxy_0, dxz, dyy_0, dyz, dzz);
            if constexpr (MODE == FEAT_FEATURES8) {
              o[0] = c_0; o[1] = G;
#pragma unroll
              for (int k = 0; k < 6; ++k) o[2 + k] = ef.f[k];
            } else {
#pragma unroll
              for (int k = 0; k < 6; ++k) o[k] = ef.f[k];
            }
          }
        }
      }
      if (planar) {
#pragma unroll
        for (int k = 0; k < NOUT; ++k) out[(int64_t)k * g.nvox + idx] = o[k];
      } else if constexpr (NOUT == 8) {
        float4 *q = reinterpret_cast<float4 *>(out + idx * 8);
        q[0] = make_float4(o[0], o[1], o[2], o[3]);
        q[1] = make_float4(o[4], o[5], o[6], o[7]);
      } else if constexpr (NOUT == 6) {
        float2 *q = reinterpret_cast<float2 *>(out + idx * 6);
        q[0] = make_float2(o[0], o[1]);
        q[1] = make_float2(o[2], o[3]);
        q[2] = make_float2(o[4], o[5]);
      } else {
        out[idx] = o[0];
      }
    }
  }
}

// ---- per-voxel numerics on a flat batch (a1 / a2 parity hooks) -----------------------
template <int NOUT>
__global__ __launch_bounds__(256) void eig_batch_kernel(const float *__restrict__ A6,
                                                        float *__restrict__ outv, int64_t n,
                                                        int trig) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *a = A6 + i * 6;
  if (NOUT == 3) {
    Eig3 e = trig == 0 ? eig3_sym_fast(a[0], a[1], a[2], a[3], a[4], a[5])
                       : eig3_sym<1>(a[0], a[1], a[2], a[3], a[4], a[5]);
    outv[i * 3 + 0] = e.e0; outv[i * 3 + 1] = e.e1; outv[i * 3 + 2] = e.e2;
  } else {
    EigFeat f = trig == 0 ? eig_features<0>(a[0], a[1], a[2], a[3], a[4], a[5])
                          : eig_features<1>(a[0], a[1], a[2], a[3], a[4], a[5]);
#pragma unroll
    for (int k = 0; k < 6; ++k) outv[i * 6 + k] = f.f[k];
  }
}

// tools/MaskedImageFilter.cxx:75-93 (double pixels)
__global__ __launch_bounds__(256) void mask_f64_kernel(const double *__restrict__ img,
                                                       const double *__restrict__ msk,
                                                       double outside, double *__restrict__ outv,
                                                       int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) outv[i] = msk[i] != 0.0 ? img[i] : outside;
}

// Multiply + Cast prepass: tc = float(image) * float(mask), cf = float(mask)
// (CastImageFilter ImageToEmphysemaFeaturesFilter.hxx:21,110 and MultiplyImageFilter
// NormalizedGaussianConvolutionImageFilter.hxx:48-49), once per call for all scales.
// It exists because per-lane sub-dword loads are slow on gfx950 (a wave-instruction of
// byte loads was measured at ~64 cycles in the TA): here every lane moves 4 voxels
// with one vector load per input, and the line kernels then only read floats.
// msk == nullptr: tc = float(image) only.  cf == nullptr: not written.
template <typename TI, typename TM>
__global__ __launch_bounds__(256) void prep_kernel_vec4(const TI *__restrict__ img,
                                                        const TM *__restrict__ msk,
                                                        float *__restrict__ tc,
                                                        float *__restrict__ cf, int64_t n4) {
  typedef TI TI4 __attribute__((ext_vector_type(4)));
  typedef TM TM4 __attribute__((ext_vector_type(4)));
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    const TI4 a = reinterpret_cast<const TI4 *>(img)[i];
    float4 t = make_float4((float)a.x, (float)a.y, (float)a.z, (float)a.w);
    if (msk != nullptr) {
      const TM4 m = reinterpret_cast<const TM4 *>(msk)[i];
      const float4 c = make_float4((float)m.x, (float)m.y, (float)m.z, (float)m.w);
      t.x *= c.x; t.y *= c.y; t.z *= c.z; t.w *= c.w;
      if (cf != nullptr) reinterpret_cast<float4 *>(cf)[i] = c;
    }
    reinterpret_cast<float4 *>(tc)[i] = t;
  }
}
template <typename TI, typename TM>
__global__ __launch_bounds__(256) void prep_kernel_scalar(const TI *__restrict__ img,
                                                          const TM *__restrict__ msk,
                                                          float *__restrict__ tc,
                                                          float *__restrict__ cf, int64_t i0,
                                                          int64_t n) {
  int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float t = (float)img[i];
    if (msk != nullptr) {
      const float c = (float)msk[i];
      t *= c;
      if (cf != nullptr) cf[i] = c;
    }
    tc[i] = t;
  }
}

// DivideImageFilter on its own, for the NormalizedGaussianConvolution entry point
__global__ __launch_bounds__(256) void divide_kernel(const float *__restrict__ a,
                                                     const float *__restrict__ b,
                                                     float *__restrict__ outv, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float d = b[i];
    outv[i] = d != 0.0f ? a[i] / d : FLT_MAX;
  }
}

}  // namespace ife
