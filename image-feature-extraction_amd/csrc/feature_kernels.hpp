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
// Each plane is staged once (with a one-voxel replicate halo) in an LDS tile; a ring
// of four tiles keeps planes z-1, z, z+1 resident while plane z+2 is in flight, so
// every smoothed value is read from HBM once per z-chunk (plus halo) and every output
// written once.
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
  FEAT_GRADMAG = 3,    // |grad| (a7)
  // the eight features of FEAT_FEATURES8, written only at sampled voxels and compacted in
  // raster order into eight sample columns (row f1: the feature volume is never stored).
  // The mask argument is then a code per voxel: bit 0 = sample here, bit 1 = label non-zero
  // (features are zero where it is clear, as MaskImageFilter leaves them).
  FEAT_SAMPLES8 = 4
};

template <int MODE>
struct FeatNOut {
  static constexpr int value =
      (MODE == FEAT_FEATURES8 || MODE == FEAT_SAMPLES8) ? 8 : (MODE == FEAT_GRADMAG ? 1 : 6);
};

struct FeatGeom {
  int nx, ny, nz;
  int zchunk;
  int64_t plane;  // nx*ny
  int64_t nvox;
  // Z-slab support: the value buffer may carry one halo plane in front (zoff = 1) and
  // one behind; value plane of output plane z is clamp(z + zoff, 0, zc_hi).  Whole
  // volume: zoff = 0, zc_hi = nz - 1 (replicate boundary at both ends).
  int zoff, zc_hi;
  int gx, gy, gz;  // tiles along x, y and z-chunks; the launch grid is 1-D (gx*gy*gz)
  // FEAT_SAMPLES8 only: nvox is then the element stride between the eight columns;
  // seg_base[bx + gx*(y + ny*z)] = samples in front of that 64-voxel row segment
  const uint32_t *seg_base;
  int64_t col_offset;
};

// Operator coefficients after FlipAxes and ScaleCoefficients (double), per axis.
struct DerivCoef {
  double m1[3], p1[3];  // order 1: coefficient of f[i-1] and of f[i+1]
  double a2[3], b2[3], c2[3];  // order 2: coefficients of f[i-1], f[i], f[i+1]
};

// value sources -------------------------------------------------------------------
// Smoothed field S = num/den with ITK's Div functor (B != 0 ? A/B : max()); den may
// be null when the certainty is identically one (then den == 1.0f exactly).
// A source is read in two steps so that the loads of plane z+2 stay in flight across the
// arithmetic of plane z: fetch() only issues loads, finish() (called one plane later, just
// before the LDS write) turns the raw registers into the float value.
struct ValSmooth {
  const float *num;
  const float *den;
  struct Raw { float a, b; };
  __device__ __forceinline__ Raw fetch(int64_t i) const {
    Raw r;
    r.a = num[i];
    r.b = den != nullptr ? den[i] : 1.0f;
    return r;
  }
  __device__ __forceinline__ float finish(const Raw &r) const {
    if (den == nullptr) return r.a;
    // A/1 == A: a wave whose certainties are all exactly one (the interior of a mask,
    // DESIGN.md "all-ones certainty") skips the division (scalar branch)
    if (__builtin_amdgcn_ballot_w64(r.b != 1.0f) == 0) return r.a;
    return r.b != 0.0f ? r.a / r.b : FLT_MAX;
  }
};
// The smoothed field itself (the last axis pass has divided already, or the certainty is
// identically one): one float per voxel and no run-time choice inside the kernel.
struct ValS {
  const float *s;
  struct Raw { float a; };
  __device__ __forceinline__ Raw fetch(int64_t i) const { return Raw{s[i]}; }
  __device__ __forceinline__ float finish(const Raw &r) const { return r.a; }
};
template <typename TI>
struct ValRaw {
  const TI *img;
  struct Raw { TI v; };
  __device__ __forceinline__ Raw fetch(int64_t i) const { return Raw{img[i]}; }
  __device__ __forceinline__ float finish(const Raw &r) const { return (float)r.v; }
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
#ifndef IFE_FT_TY
#define IFE_FT_TY 8
#endif
constexpr int FT_TY = IFE_FT_TY;
constexpr int FT_HX = FT_TX + 2;
constexpr int FT_HY = FT_TY + 2;
constexpr int FT_NE = FT_HX * FT_HY;  // staged elements per plane (tile + halo)
constexpr int FT_THREADS = FT_TX * FT_TY;
constexpr int FT_NLD = (FT_NE + FT_THREADS - 1) / FT_THREADS;  // staged elements per thread

// The arithmetic of one voxel: its 19 stencil points from the three staged planes (row pitch
// FT_HX, one-voxel halo) to the NOUT outputs of MODE, before masking.  Shared by both forms
// of the feature kernel, so that they cannot differ in a rounding.
template <int MODE, bool UNIT, int TRIG, bool KLDS>
__device__ __forceinline__ void feat_point(const float (*tm)[FT_HX], const float (*t0)[FT_HX],
                                           const float (*tp)[FT_HX], int ty, int tx,
                                           const DerivCoef &dc, const double *ktab,
                                           float (&o)[FeatNOut<MODE>::value]) {
  constexpr bool F8 = MODE == FEAT_FEATURES8 || MODE == FEAT_SAMPLES8;
  constexpr bool NEED_H = MODE != FEAT_GRADMAG;
  constexpr bool NEED_G = F8 || MODE == FEAT_GRADMAG;
  const float c = t0[ty + 1][tx + 1];
  const float xm = t0[ty + 1][tx], xp = t0[ty + 1][tx + 2];
  const float ym = t0[ty][tx + 1], yp = t0[ty + 2][tx + 1];
  const float zm = tm[ty + 1][tx + 1], zp = tp[ty + 1][tx + 1];
  float G = 0.0f;
  if (NEED_G) {
    double gx, gy, gz;
    if (UNIT) {
      gx = 0.5 * ((double)xp - (double)xm);
      gy = 0.5 * ((double)yp - (double)ym);
      gz = 0.5 * ((double)zp - (double)zm);
    } else {
      gx = d1(dc.m1[0], dc.p1[0], xm, xp);
      gy = d1(dc.m1[1], dc.p1[1], ym, yp);
      gz = d1(dc.m1[2], dc.p1[2], zm, zp);
    }
    double a = gx * gx;
    a += gy * gy;
    a += gz * gz;
#if defined(IFE_DIAG_NO_GSQRT)
    G = (float)a;
#else
    G = (float)sqrt(a);
#endif
  }
  if constexpr (MODE == FEAT_GRADMAG) {
    o[0] = G;
  } else if (NEED_H) {
    // first differences as float images at the six neighbours, then chained
    const float cmm = t0[ty][tx], cpm = t0[ty][tx + 2];
    const float cmp = t0[ty + 2][tx], cpp = t0[ty + 2][tx + 2];
    const float zm_xm = tm[ty + 1][tx], zm_xp = tm[ty + 1][tx + 2];
    const float zm_ym = tm[ty][tx + 1], zm_yp = tm[ty + 2][tx + 1];
    const float zp_xm = tp[ty + 1][tx], zp_xp = tp[ty + 1][tx + 2];
    const float zp_ym = tp[ty][tx + 1], zp_yp = tp[ty + 2][tx + 1];
    float dxx, dyy, dzz, dxy, dxz, dyz;
    if (UNIT) {
      const float dx_ym = 0.5f * (cpm - cmm), dx_yp = 0.5f * (cpp - cmp);
      const float dx_zm = 0.5f * (zm_xp - zm_xm), dx_zp = 0.5f * (zp_xp - zp_xm);
      const float dy_zm = 0.5f * (zm_yp - zm_ym), dy_zp = 0.5f * (zp_yp - zp_ym);
      dxy = 0.5f * (dx_yp - dx_ym);
      dxz = 0.5f * (dx_zp - dx_zm);
      dyz = 0.5f * (dy_zp - dy_zm);
      const double cd = (double)c;
      dxx = (float)(fma(-2.0, cd, (double)xm) + (double)xp);
      dyy = (float)(fma(-2.0, cd, (double)ym) + (double)yp);
      dzz = (float)(fma(-2.0, cd, (double)zm) + (double)zp);
    } else {
      const float dx_ym = (float)d1(dc.m1[0], dc.p1[0], cmm, cpm);
      const float dx_yp = (float)d1(dc.m1[0], dc.p1[0], cmp, cpp);
      const float dx_zm = (float)d1(dc.m1[0], dc.p1[0], zm_xm, zm_xp);
      const float dx_zp = (float)d1(dc.m1[0], dc.p1[0], zp_xm, zp_xp);
      const float dy_zm = (float)d1(dc.m1[1], dc.p1[1], zm_ym, zm_yp);
      const float dy_zp = (float)d1(dc.m1[1], dc.p1[1], zp_ym, zp_yp);
      dxy = (float)d1(dc.m1[1], dc.p1[1], dx_ym, dx_yp);
      dxz = (float)d1(dc.m1[2], dc.p1[2], dx_zm, dx_zp);
      dyz = (float)d1(dc.m1[2], dc.p1[2], dy_zm, dy_zp);
      dxx = d2(dc.a2[0], dc.b2[0], dc.c2[0], xm, c, xp);
      dyy = d2(dc.a2[1], dc.b2[1], dc.c2[1], ym, c, yp);
      dzz = d2(dc.a2[2], dc.b2[2], dc.c2[2], zm, c, zp);
    }
    if constexpr (MODE == FEAT_HESSIAN6) {
      o[0] = dxx; o[1] = dxy; o[2] = dxz; o[3] = dyy; o[4] = dyz; o[5] = dzz;
    } else {
      EigFeat ef;
#if defined(IFE_DIAG_NO_SOLVER)  // timing diagnostics only: results are wrong by construction
      ef.f[0] = dxx; ef.f[1] = dxy; ef.f[2] = dxz; ef.f[3] = dyy; ef.f[4] = dyz; ef.f[5] = dzz;
#else
      if constexpr (KLDS)
        ef = eig_features<TRIG>(dxx, dxy, dxz, dyy, dyz, dzz, EigConstLds::at(ktab));
      else
        ef = eig_features<TRIG>(dxx, dxy, dxz, dyy, dyz, dzz);
#endif
      if constexpr (F8) {
        o[0] = c; o[1] = G;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[2 + k] = ef.f[k];
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = ef.f[k];
      }
    }
  }
}

// One workgroup owns a 64 x FT_TY XY tile and marches along z.  Four LDS slots hold the
// tile (with replicate halo) of planes z-1, z, z+1 and the plane being written for the
// next step, so one barrier per plane suffices; a thread reads its 19 stencil points
// from LDS and carries nothing from plane to plane (short live ranges: the eigen solve
// is the register peak, not a ring of per-plane state).
//
// UNIT = all spacings are exactly 1: the operator coefficients are +-0.5 and 1,-2,1, so
//  * a first difference kept as a float image, float(0.5*b - 0.5*a) accumulated in double,
//    equals 0.5f*(b - a) in float (b - a rounds once; a 53-bit intermediate can never sit
//    on a float rounding boundary unless it is exact; scaling by 0.5 is exact);
//  * a second difference float((a - 2c) + b) takes one fma (2c is exact) and one add.
// Both forms are bit-identical to the generic inner products, except that the sign of
// an exact zero is not tracked (+0/-0 compare equal and never reach a non-zero output).
// output stores of the feature kernel: 0 plain, 1 non-temporal.  Round 1 (1.54 ms kernel):
// no difference.  Round 2, with the kernel at 1.25 ms and bound by its stores: 1.274 -> 1.237
// ms per launch and the prepass that follows the last launch 0.349 -> 0.317 (it no longer
// shares the memory system with 4 GB of dirty lines being written back): step 10.17 -> 10.00
// on one box, twice.  Nothing reads the outputs again inside the call.
#ifndef IFE_FT_NT
#define IFE_FT_NT 1
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#if IFE_FT_NT
#define IFE_FT_STORE(ptr, ...) __builtin_nontemporal_store((__VA_ARGS__), (ptr))
#else
#define IFE_FT_STORE(ptr, ...) (*(ptr) = (__VA_ARGS__))
#endif
#ifndef IFE_FT_KLDS
#define IFE_FT_KLDS 1  // 1: solver constants from LDS, 0: immediates
#endif
// Waves per SIMD the register allocation aims at.  The default solver (TRIG 2) fits three
// 512-thread workgroups per CU (6 waves per SIMD, 80 VGPRs: measured 1.39 -> 1.30 ms per launch);
// the double-precision solvers and the general-spacing forms need more registers.
#ifndef IFE_FT_WAVES
#define IFE_FT_WAVES(MODE, UNIT, TRIG) ((TRIG) == 2 && (UNIT) ? 6 : 1)
#endif
template <int MODE, bool UNIT, int TRIG, bool PLANAR, typename VAL, typename TM>
__global__ __launch_bounds__(FT_THREADS, IFE_FT_WAVES(MODE, UNIT, TRIG)) void features_kernel(VAL val, const TM *__restrict__ mask,
                                                              float *__restrict__ out, FeatGeom g,
                                                              DerivCoef dc) {
  __shared__ float tile[4][FT_HY][FT_HX];
  // mask tile of the output plane, staged as dwords (per-lane sub-dword global loads are
  // slow on gfx950); row pitch = 64 mask elements
  constexpr int MT_DW = FT_TY * FT_TX * (int)sizeof(TM) / 4;  // dwords per plane tile
  constexpr int MT_DWROW = FT_TX * (int)sizeof(TM) / 4;       // dwords per tile row
  __shared__ uint32_t mtile[2][MT_DW];
  // double constants of the solver, read from LDS (eigen_device.hpp "double constants")
  constexpr bool F8 = MODE == FEAT_FEATURES8 || MODE == FEAT_SAMPLES8;
  constexpr bool SAMPLES = MODE == FEAT_SAMPLES8;
  static_assert(!SAMPLES || PLANAR, "sample columns are written through the planar store");
  constexpr bool KLDS = (F8 || MODE == FEAT_EIG6) && IFE_FT_KLDS && TRIG != 2;
  __shared__ double ktab[KLDS ? EK_COUNT : 1];
  if (KLDS) eig_const_fill(ktab);  // the barrier of the first plane iteration covers it
  constexpr int NOUT = FeatNOut<MODE>::value;

  const int tid = threadIdx.x;
  const int tx = tid & 63, ty = tid >> 6;
  // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (linear id
  // % 8 labels the XCD), and neighbouring tiles share halo rows/columns and z-planes, so
  // each XCD gets one contiguous run of the (z-chunk, tile-row, tile-column) order: the
  // halo re-reads then hit that XCD's own L2 instead of going to the fabric as whole
  // 128-B lines.  Placement only affects speed, never results.
  int bx, by, bz;
  {
    const uint32_t nb = (uint32_t)g.gx * (uint32_t)g.gy * (uint32_t)g.gz;
    const uint32_t lin = blockIdx.x;
    const uint32_t per = nb / 8u, rem = nb % 8u;  // XCD c owns per (+1 if c < rem) tiles
    const uint32_t c = lin % 8u, i = lin / 8u;
    const uint32_t t = c * per + (c < rem ? c : rem) + i;
    bx = (int)(t % (uint32_t)g.gx);
    by = (int)((t / (uint32_t)g.gx) % (uint32_t)g.gy);
    bz = (int)(t / ((uint32_t)g.gx * (uint32_t)g.gy));
  }
  const int x = bx * FT_TX + tx, y = by * FT_TY + ty;
  const int z0 = bz * g.zchunk;
  const int z1 = min(z0 + g.zchunk, g.nz);
  const bool inb = x < g.nx && y < g.ny;

  // staging assignment: elements tid + k*FT_THREADS of the (HY x HX) halo tile
  int64_t off[FT_NLD];
  int eidx[FT_NLD];
  bool has[FT_NLD];
#pragma unroll
  for (int k = 0; k < FT_NLD; ++k) {
    const int e = tid + k * FT_THREADS;
    has[k] = e < FT_NE;
    const int ey = has[k] ? e / FT_HX : 0, ex = has[k] ? e % FT_HX : 0;
    eidx[k] = ey * FT_HX + ex;
    off[k] = (int64_t)clampi(bx * FT_TX - 1 + ex, g.nx - 1) +
             (int64_t)g.nx * clampi(by * FT_TY - 1 + ey, g.ny - 1);
  }
  // mask staging: thread t < MT_DW owns dword (t % MT_DWROW) of tile row (t / MT_DWROW).
  // Vector form needs rows that are dword multiples and a dword-aligned base.
  const bool mvec = mask != nullptr && ((int64_t)g.nx * (int64_t)sizeof(TM)) % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(mask) & 3) == 0;
  const int mrow = tid / MT_DWROW, mcol = tid % MT_DWROW;
  const int mx = bx * FT_TX + mcol * (4 / (int)sizeof(TM));
  const int my = by * FT_TY + mrow;
  const bool mine = mvec && tid < MT_DW && mx < g.nx && my < g.ny;
  const int64_t moff = (int64_t)mx + (int64_t)g.nx * my;  // element offset inside a plane

  typename VAL::Raw r[FT_NLD];
  uint32_t mr = 0;
  auto load_plane = [&](int p) {
    const int64_t pb = (int64_t)clampi(p + g.zoff, g.zc_hi) * g.plane;
#pragma unroll
    for (int k = 0; k < FT_NLD; ++k)
      if (has[k]) r[k] = val.fetch(pb + off[k]);
  };
  auto store_plane = [&](int p) {
    float *t = &tile[(p - (z0 - 1)) & 3][0][0];
#pragma unroll
    for (int k = 0; k < FT_NLD; ++k)
      if (has[k]) t[eidx[k]] = val.finish(r[k]);
  };

  load_plane(z0 - 1);
  store_plane(z0 - 1);
  load_plane(z0);
  store_plane(z0);
  load_plane(z0 + 1);
  if (mine) mr = *reinterpret_cast<const uint32_t *>(mask + (int64_t)z0 * g.plane + moff);

  // The outputs of plane z are stored at the top of iteration z+1, after that iteration's
  // wait for its staged loads: the wait (vmcnt counts loads and stores in order) then never
  // sits behind stores that were issued a moment ago.
  //
  // Interleaved outputs: a lane's record is NOUT consecutive floats, so a store instruction
  // of the plain form writes 16 of every 32 bytes and the two instructions of a voxel row
  // interleave (measured: 2.7 TB/s, and no gain from a sparse mask).  The records of a wave
  // (one tile row = 64 x-consecutive voxels = one contiguous run of memory) are therefore
  // turned through a wave-private LDS strip so that store k of lane l carries piece
  // k*64 + l of that run: every instruction writes one contiguous kilobyte.
  constexpr bool XPOSE = !PLANAR && (NOUT == 8 || NOUT == 6);
  constexpr int XW = NOUT == 8 ? 4 : 2;            // floats per store
  constexpr int XN = XPOSE ? NOUT / XW : 1;        // stores per lane
  __shared__ float xstrip[XPOSE ? FT_THREADS / 64 : 1][XPOSE ? 64 * NOUT : 1];
  const int row_valid = min(FT_TX, g.nx - bx * FT_TX);  // in-bounds voxels of this wave's row
  float po[NOUT];
  int64_t pidx = -1;  // XPOSE: index of the row's first voxel; else this lane's voxel
  auto flush = [&]() {
    if (pidx < 0) return;
    if constexpr (PLANAR) {
#pragma unroll
      for (int k = 0; k < NOUT; ++k) out[(int64_t)k * g.nvox + pidx] = po[k];
    } else if constexpr (XPOSE) {
      float *q = out + pidx * NOUT;
#pragma unroll
      for (int k = 0; k < XN; ++k) {
        const int e = (k * 64 + tx) * XW;  // first float of this lane's piece
        if (e < row_valid * NOUT) {
          if constexpr (XW == 4)
            IFE_FT_STORE(reinterpret_cast<f32x4 *>(q + e), (f32x4){po[4 * k], po[4 * k + 1], po[4 * k + 2], po[4 * k + 3]});
          else
            IFE_FT_STORE(reinterpret_cast<f32x2 *>(q + e), (f32x2){po[2 * k], po[2 * k + 1]});
        }
      }
    } else if constexpr (NOUT == 8) {
      float4 *q = reinterpret_cast<float4 *>(out + pidx * 8);
      q[0] = make_float4(po[0], po[1], po[2], po[3]);
      q[1] = make_float4(po[4], po[5], po[6], po[7]);
    } else if constexpr (NOUT == 6) {
      float2 *q = reinterpret_cast<float2 *>(out + pidx * 6);
      q[0] = make_float2(po[0], po[1]);
      q[1] = make_float2(po[2], po[3]);
      q[2] = make_float2(po[4], po[5]);
    } else {
      out[pidx] = po[0];
    }
  };

  for (int z = z0; z < z1; ++z) {
    store_plane(z + 1);
    if (mine) mtile[z & 1][tid] = mr;
    flush();
    pidx = -1;
    if (z + 1 < z1) {  // issue the next step's loads before consuming this one
      load_plane(z + 2);
      if (mine) mr = *reinterpret_cast<const uint32_t *>(mask + (int64_t)(z + 1) * g.plane + moff);
    }
    __syncthreads();
    if (XPOSE ? (y >= g.ny || row_valid <= 0) : !inb) continue;  // XPOSE: whole waves only

    const int64_t idx = (int64_t)x + (int64_t)g.nx * ((int64_t)y + (int64_t)g.ny * z);
    bool keep = inb;  // lanes past the row end (XPOSE) compute on clamped values, store nothing
    bool samp = false;
    if (inb) {
      TM mval = (TM)3;
      if (mvec)
        mval = reinterpret_cast<const TM *>(mtile[z & 1])[ty * FT_TX + tx];
      else if (mask != nullptr)
        mval = mask[idx];
      if constexpr (SAMPLES) {
        samp = ((int)mval & 1) != 0;
        keep = ((int)mval & 3) == 3;
      } else {
        keep = mval != (TM)0;
      }
    }
    float (&o)[NOUT] = po;  // results are built in the carried registers
#pragma unroll
    for (int k = 0; k < NOUT; ++k) o[k] = 0.0f;
    // A wave whose 64 voxels are all outside the mask skips the arithmetic (scalar branch);
    // otherwise every lane computes and masked lanes are zeroed at the end.
    if (__builtin_amdgcn_ballot_w64(keep) != 0) {
      feat_point<MODE, UNIT, TRIG, KLDS>(tile[(z - 1 - (z0 - 1)) & 3], tile[(z - (z0 - 1)) & 3],
                                         tile[(z + 1 - (z0 - 1)) & 3], ty, tx, dc, ktab, o);
    }
    if (!keep) {
#pragma unroll
      for (int k = 0; k < NOUT; ++k) o[k] = 0.0f;
    }
    if constexpr (XPOSE) {
      float *xs = xstrip[tid >> 6];
      if constexpr (XW == 4) {
        *reinterpret_cast<float4 *>(xs + tx * 8) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4 *>(xs + tx * 8 + 4) = make_float4(o[4], o[5], o[6], o[7]);
      } else {
#pragma unroll
        for (int k = 0; k < XN; ++k)
          *reinterpret_cast<float2 *>(xs + tx * NOUT + 2 * k) = make_float2(o[2 * k], o[2 * k + 1]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < XN; ++k) {
        if constexpr (XW == 4) {
          const float4 v = *reinterpret_cast<const float4 *>(xs + (k * 64 + tx) * 4);
          o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w;
        } else {
          const float2 v = *reinterpret_cast<const float2 *>(xs + (k * 64 + tx) * 2);
          o[2 * k] = v.x; o[2 * k + 1] = v.y;
        }
      }
      __builtin_amdgcn_wave_barrier();
      pidx = idx - tx;  // first voxel of the row
    } else if constexpr (SAMPLES) {
      const uint64_t sm = __builtin_amdgcn_ballot_w64(samp);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
      const int64_t seg = (int64_t)bx + (int64_t)g.gx * ((int64_t)y + (int64_t)g.ny * z);
      pidx = samp ? g.col_offset + (int64_t)g.seg_base[seg] + rank : -1;
    } else {
      pidx = idx;
    }
  }
  flush();
}

// =====================================================================================
// Ring form of the feature kernel: same tile, same arithmetic (feat_point), different plumbing.
// =====================================================================================
// For a single float field (the smoothed value S after the last axis pass, or a raw float
// image) the staged planes need no arithmetic, so they go from memory straight into LDS
// (`buffer_load_dword ... lds`): no staging registers, no ds_write, and -- because a plane in
// flight costs LDS, not registers -- plane z+FR_P is requested while plane z is computed.  The
// requests are counted by hand (the compiler does not see them): before the barrier of step z
// a wave waits until at most the operations younger than ITS pieces of plane z+1 are
// outstanding (vmcnt counts loads and stores together, in issue order), the barrier makes
// every wave's pieces visible, and the output stores of the last FR_P-1 steps stay in flight
// across it.  Round 2's kernel waited for vmcnt(0) in every step (a register it had spilled
// came back through a scratch load behind the freshly issued prefetch), i.e. each wave sat
// out the full memory latency once per plane.
//
// Every global access is a buffer access with a wave-uniform base in scalar registers and a
// 32-bit per-lane offset that never changes (in-plane offsets, computed once); stores are
// bounded by the descriptor's size instead of by EXEC (a row that ends inside the tile, a row
// below the volume, a voxel that is not sampled: the hardware drops the access), so every
// wave issues the same number of memory operations in every step, which is what makes the
// counted waits exact.
constexpr int FR_NS = 8;                   // ring slots (planes resident or in flight)
#ifndef IFE_FR_P
#define IFE_FR_P 4
#endif
constexpr int FR_P = IFE_FR_P;             // plane z+FR_P is requested in step z (FR_P + 2 <= FR_NS)
constexpr int FR_VPIECES = (FT_NE + 63) / 64;   // 64-dword pieces of a staged plane (11)
constexpr int FR_VDW = FR_VPIECES * 64;    // dwords of a slot reserved for the values
constexpr int FR_MDW_MAX = FT_TY * FT_TX * 2 / 4;  // mask dwords of a slot (2-byte masks)
constexpr int FR_SLOT_DW = FR_VDW + FR_MDW_MAX;
static_assert(FR_P >= 2 && FR_P + 2 <= FR_NS, "ring too small for the prefetch distance");
static_assert(FT_THREADS == 512 && FR_VPIECES <= 16, "piece assignment assumes eight waves, two pieces each at most");

typedef uint32_t ft_u32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor (stride 0, byte range `bytes`) as four scalars for inline asm
__device__ __forceinline__ ft_u32x4 ft_rsrc_words(const void *base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  ft_u32x4 r;  // all four words wave-uniform: the asm's "s" constraint refuses anything else
  r.x = (uint32_t)a;
  r.y = (uint32_t)(a >> 32) & 0xffffu;
  r.z = bytes;
  r.w = 0x00020000u;
  return r;
}
// One wave-instruction: lane l loads the dword at base + voff[l] into LDS dword lds_addr/4 + l.
// M0 carries the LDS address; it is compiler-reserved, so it is saved and restored inside the
// same statement.  Not counted by the compiler: see ft_wait_vmcnt.
__device__ __forceinline__ void ft_dma_dword(ft_u32x4 rsrc, uint32_t voff, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dword %1, %2, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_addr)
      : "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate is an instruction field)
__device__ __forceinline__ void ft_wait_vmcnt(int n) {
#define IFE_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    IFE_W(0) IFE_W(1) IFE_W(2) IFE_W(3) IFE_W(4) IFE_W(5) IFE_W(6) IFE_W(7) IFE_W(8) IFE_W(9)
    IFE_W(10) IFE_W(11) IFE_W(12) IFE_W(13) IFE_W(14) IFE_W(15) IFE_W(16) IFE_W(17) IFE_W(18)
    IFE_W(19) IFE_W(20) IFE_W(21) IFE_W(22) IFE_W(23) IFE_W(24) IFE_W(25) IFE_W(26) IFE_W(27)
    IFE_W(28) IFE_W(29) IFE_W(30) IFE_W(31) IFE_W(32) IFE_W(33) IFE_W(34) IFE_W(35) IFE_W(36)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef IFE_W
}
__device__ __forceinline__ void ft_lds_barrier() {
  // LDS only: global loads and stores in flight stay in flight across the barrier
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#ifndef IFE_FT_NT_AUX
#define IFE_FT_NT_AUX (IFE_FT_NT ? 2 : 0)
#endif

template <int MODE, bool UNIT, int TRIG, bool PLANAR, typename TM>
__global__ __launch_bounds__(FT_THREADS, IFE_FT_WAVES(MODE, UNIT, TRIG)) void features_ring_kernel(
    const float *__restrict__ val, const TM *__restrict__ mask, float *__restrict__ out, FeatGeom g,
    DerivCoef dc) {
  static_assert(sizeof(TM) <= 2, "mask tiles of wider types do not fit the slot");
  using rsrc_t = __amdgpu_buffer_rsrc_t;
  constexpr int NOUT = FeatNOut<MODE>::value;
  constexpr bool F8 = MODE == FEAT_FEATURES8 || MODE == FEAT_SAMPLES8;
  constexpr bool SAMPLES = MODE == FEAT_SAMPLES8;
  static_assert(!SAMPLES || PLANAR, "sample columns are written through the planar store");
  constexpr bool KLDS = (F8 || MODE == FEAT_EIG6) && IFE_FT_KLDS && TRIG != 2;
  constexpr bool XPOSE = !PLANAR && (NOUT == 8 || NOUT == 6);
  constexpr int XW = NOUT == 8 ? 4 : 2;       // floats per store of the interleaved form
  constexpr int XN = XPOSE ? NOUT / XW : 1;   // stores per lane of the interleaved form
  constexpr int NST = XPOSE ? XN : NOUT;      // stores every wave issues in every step
  constexpr int MT_DW = FT_TY * FT_TX * (int)sizeof(TM) / 4;  // mask dwords per plane tile
  constexpr int MT_DWROW = FT_TX * (int)sizeof(TM) / 4;       // mask dwords per tile row
  constexpr int MPIECES = MT_DW / 64;

  __shared__ __attribute__((aligned(16))) uint32_t ring[FR_NS * FR_SLOT_DW];
  __shared__ float xstrip[XPOSE ? FT_THREADS / 64 : 1][XPOSE ? 64 * NOUT : 1];
  __shared__ double ktab[KLDS ? EK_COUNT : 1];
  if (KLDS) eig_const_fill(ktab);  // the barrier of the first step covers it

  const int tid = threadIdx.x;
  const int tx = tid & 63;
  const int w = (int)__builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);  // wave = tile row
  int bx, by, bz;
  {  // XCD-aware tile order, as in features_kernel
    const uint32_t nb = (uint32_t)g.gx * (uint32_t)g.gy * (uint32_t)g.gz;
    const uint32_t lin = blockIdx.x;
    const uint32_t per = nb / 8u, rem = nb % 8u;
    const uint32_t c = lin % 8u, i = lin / 8u;
    const uint32_t t = c * per + (c < rem ? c : rem) + i;
    bx = (int)(t % (uint32_t)g.gx);
    by = (int)((t / (uint32_t)g.gx) % (uint32_t)g.gy);
    bz = (int)(t / ((uint32_t)g.gx * (uint32_t)g.gy));
  }
  const int x = bx * FT_TX + tx, y = by * FT_TY + w;
  const int z0 = bz * g.zchunk;
  const int z1 = min(z0 + g.zchunk, g.nz);
  const bool has_mask = mask != nullptr;
  const int row_valid = y < g.ny ? min(FT_TX, g.nx - bx * FT_TX) : 0;  // stored voxels of this wave's row

  // ---- staging assignment: piece A = elements 64w.. of the halo tile; piece B = elements
  // 512 + 64w.. (waves 0-2) or 64 dwords of the mask tile (the next MPIECES waves) ----
  auto val_off = [&](int e) -> uint32_t {
    if (e >= FT_NE) e = 0;  // the pad of the last piece repeats element 0 (never read)
    const int ey = e / FT_HX, ex = e % FT_HX;
    return 4u * ((uint32_t)clampi(bx * FT_TX - 1 + ex, g.nx - 1) +
                 (uint32_t)g.nx * (uint32_t)clampi(by * FT_TY - 1 + ey, g.ny - 1));
  };
  const uint32_t offA = val_off(tid);
  const bool b_val = w < FR_VPIECES - 8;
  const bool b_mask = !b_val && has_mask && (w - (FR_VPIECES - 8)) < MPIECES;
  const int n_dma = 1 + ((b_val || b_mask) ? 1 : 0);
  uint32_t offB = 0;
  if (b_val) {
    offB = val_off(FT_THREADS + tid);
  } else if (b_mask) {
    const int d = (w - (FR_VPIECES - 8)) * 64 + tx;  // dword of the mask tile
    int mx = bx * FT_TX + (d % MT_DWROW) * (4 / (int)sizeof(TM));
    int my = by * FT_TY + d / MT_DWROW;
    if (mx >= g.nx || my >= g.ny) { mx = bx * FT_TX; my = by * FT_TY; }  // any valid dword: never read
    offB = (uint32_t)sizeof(TM) * ((uint32_t)mx + (uint32_t)g.nx * (uint32_t)my);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)ring;
  const uint32_t dstA = (uint32_t)w * 256u;
  const uint32_t dstB = b_val ? (uint32_t)(8 + w) * 256u
                              : (uint32_t)FR_VDW * 4u + (uint32_t)(w - (FR_VPIECES - 8)) * 256u;
  auto slot_of = [&](int p) { return (uint32_t)(p - (z0 - 1)) & (uint32_t)(FR_NS - 1); };
  auto request = [&](int p) {  // plane p of the values and of the mask into slot_of(p)
    const uint32_t sb = lds0 + slot_of(p) * (uint32_t)(FR_SLOT_DW * 4);
    const ft_u32x4 rv = ft_rsrc_words(val + (int64_t)clampi(p + g.zoff, g.zc_hi) * g.plane, 0xffffffffu);
    ft_dma_dword(rv, offA, sb + dstA);
    if (b_val) {
      ft_dma_dword(rv, offB, sb + dstB);
    } else if (b_mask) {
      const ft_u32x4 rm = ft_rsrc_words(mask + (int64_t)clampi(p, g.nz - 1) * g.plane, 0xffffffffu);
      ft_dma_dword(rm, offB, sb + dstB);
    }
  };

  for (int p = z0 - 1; p <= min(z0 + FR_P - 1, z1); ++p) request(p);

  for (int z = z0; z < z1; ++z) {
    // this wave's pieces of plane z+1 have landed once at most the younger operations are
    // outstanding: the requests for planes z+2 .. min(z+FR_P-1, z1) and the stores of the
    // steps since that request was issued (step z+1-FR_P, or the prologue)
    if (z - z0 >= FR_P - 1 && z1 - 1 - z >= FR_P - 2) {  // steady state: one compare pair, no table
      if (n_dma == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((FR_P - 2) * 2 + (FR_P - 1) * NST) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((FR_P - 2) * 1 + (FR_P - 1) * NST) : "memory");
    } else {
      ft_wait_vmcnt(min(FR_P - 2, z1 - 1 - z) * n_dma + min(z - z0, FR_P - 1) * NST);
    }
    ft_lds_barrier();
    if (z + FR_P <= z1) request(z + FR_P);

    const float(*tm)[FT_HX] = reinterpret_cast<const float(*)[FT_HX]>(ring + slot_of(z - 1) * FR_SLOT_DW);
    const float(*t0)[FT_HX] = reinterpret_cast<const float(*)[FT_HX]>(ring + slot_of(z) * FR_SLOT_DW);
    const float(*tp)[FT_HX] = reinterpret_cast<const float(*)[FT_HX]>(ring + slot_of(z + 1) * FR_SLOT_DW);
    bool keep = x < g.nx && row_valid > 0;
    bool samp = false;
    if (has_mask) {
      const TM mval = reinterpret_cast<const TM *>(ring + slot_of(z) * FR_SLOT_DW + FR_VDW)[tid];
      if constexpr (SAMPLES) {
        samp = keep && ((int)mval & 1) != 0;
        keep = keep && ((int)mval & 3) == 3;
      } else {
        keep = keep && mval != (TM)0;
      }
    } else if constexpr (SAMPLES) {
      samp = keep;
    }
    float o[NOUT];
    // a wave whose voxels are all outside the mask skips the arithmetic, one whose voxels are
    // all inside skips the masking (scalar branches; the interior of a mask pays neither)
    const uint64_t kept = __builtin_amdgcn_ballot_w64(keep);
    if (kept != 0) {
      feat_point<MODE, UNIT, TRIG, KLDS>(tm, t0, tp, w, tx, dc, ktab, o);
      if (kept != __builtin_amdgcn_ballot_w64(true)) {
#pragma unroll
        for (int k = 0; k < NOUT; ++k) o[k] = keep ? o[k] : 0.0f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < NOUT; ++k) o[k] = 0.0f;
    }
    // ---- stores: bounded by the descriptor, issued by every wave in every step ----
    const int64_t row0 = (int64_t)bx * FT_TX + (int64_t)g.nx * ((int64_t)y + (int64_t)g.ny * z);
    if constexpr (XPOSE) {
      // the 64 records of the row are one contiguous run: turned through a wave-private LDS
      // strip so that store k of lane l carries piece k*64 + l of it
      float *xs = xstrip[w];
      if constexpr (XW == 4) {
        *reinterpret_cast<float4 *>(xs + tx * 8) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4 *>(xs + tx * 8 + 4) = make_float4(o[4], o[5], o[6], o[7]);
      } else {
#pragma unroll
        for (int k = 0; k < XN; ++k)
          *reinterpret_cast<float2 *>(xs + tx * NOUT + 2 * k) = make_float2(o[2 * k], o[2 * k + 1]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + row0 * NOUT, 0, row_valid * NOUT * 4, 0x00020000);
#pragma unroll
      for (int k = 0; k < XN; ++k) {
        if constexpr (XW == 4) {
          const ft_u32x4 v = *reinterpret_cast<const ft_u32x4 *>(xs + (k * 64 + tx) * 4);
          __builtin_amdgcn_raw_buffer_store_b128(v, ro, (uint32_t)(k * 64 + tx) * 16u, 0, IFE_FT_NT_AUX);
        } else {
          typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 v = *reinterpret_cast<const u32x2 *>(xs + (k * 64 + tx) * 2);
          __builtin_amdgcn_raw_buffer_store_b64(v, ro, (uint32_t)(k * 64 + tx) * 8u, 0, IFE_FT_NT_AUX);
        }
      }
      __builtin_amdgcn_wave_barrier();  // the strip is written again in the next step
    } else if constexpr (SAMPLES) {
      // eight sample columns; the sampled voxels of this row segment go to consecutive slots
      const uint64_t sm = __builtin_amdgcn_ballot_w64(samp);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
      const int64_t seg = (int64_t)bx + (int64_t)g.gx * ((int64_t)y + (int64_t)g.ny * z);
      int64_t first = 0;
      if (sm != 0) first = g.col_offset + (int64_t)g.seg_base[seg];  // uniform: a scalar load
      const uint32_t voff = samp ? rank * 4u : 0x7ffffff0u;
      const int cnt = __builtin_popcountll(sm);
#pragma unroll
      for (int k = 0; k < NOUT; ++k) {
        const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (int64_t)k * g.nvox + first, 0, cnt * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o[k]), ro, voff, 0, IFE_FT_NT_AUX);
      }
    } else {
      // planar components (or the single output of FEAT_GRADMAG): 64 consecutive floats per store
#pragma unroll
      for (int k = 0; k < NOUT; ++k) {
        const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (int64_t)k * g.nvox + row0, 0, row_valid * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o[k]), ro, (uint32_t)tx * 4u, 0, IFE_FT_NT_AUX);
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
    Eig3 e = trig == 0   ? eig3_sym_fast<0>(a[0], a[1], a[2], a[3], a[4], a[5], EigConstImm())
             : trig == 1 ? eig3_sym_fast<1>(a[0], a[1], a[2], a[3], a[4], a[5], EigConstImm())
                         : eig3_sym_fast2(a[0], a[1], a[2], a[3], a[4], a[5], EigConstImm());
    outv[i * 3 + 0] = e.e0; outv[i * 3 + 1] = e.e1; outv[i * 3 + 2] = e.e2;
  } else {
    EigFeat f = trig == 0   ? eig_features<0>(a[0], a[1], a[2], a[3], a[4], a[5])
                : trig == 1 ? eig_features<1>(a[0], a[1], a[2], a[3], a[4], a[5])
                            : eig_features<2>(a[0], a[1], a[2], a[3], a[4], a[5]);
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
// Output placement: plain (chunks <= 1), or the Y-chunked order [chunks][nz][ny/chunks][nx]
// an all-to-all sends from (chunk h = rows of the Y-slab of rank h), so that a slab host
// needs no separate packing pass.
struct PrepGeom {
  int64_t nx, ny, nz;
  int64_t chunks, nyl;  // nyl = ny / chunks
};
__device__ __forceinline__ int64_t prep_out_index(const PrepGeom &g, int64_t i) {
  if (g.chunks <= 1) return i;
  const int64_t x = i % g.nx, y = (i / g.nx) % g.ny, z = i / (g.nx * g.ny);
  const int64_t h = y / g.nyl, yy = y % g.nyl;
  return ((h * g.nz + z) * g.nyl + yy) * g.nx + x;
}
// cache policy of the prepass (experiments README): its outputs are read by the Z pass next,
// its inputs by nobody before the next call
#ifndef IFE_PREP_ST_NT
#define IFE_PREP_ST_NT 0
#endif
#ifndef IFE_PREP_LD_NT
#define IFE_PREP_LD_NT 0
#endif
#if IFE_PREP_ST_NT
#define IFE_PREP_STORE(ptr, v) \
  __builtin_nontemporal_store(f32x4{(v).x, (v).y, (v).z, (v).w}, reinterpret_cast<f32x4 *>(ptr))
#else
#define IFE_PREP_STORE(ptr, v) (*(ptr) = (v))
#endif
template <typename TI, typename TM>
__global__ __launch_bounds__(256) void prep_kernel_vec4(const TI *__restrict__ img,
                                                        const TM *__restrict__ msk,
                                                        float *__restrict__ tc,
                                                        float *__restrict__ cf, int64_t n4,
                                                        PrepGeom g) {
  typedef TI TI4 __attribute__((ext_vector_type(4)));
  typedef TM TM4 __attribute__((ext_vector_type(4)));
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
#if IFE_PREP_LD_NT
    const TI4 a = __builtin_nontemporal_load(reinterpret_cast<const TI4 *>(img) + i);
#else
    const TI4 a = reinterpret_cast<const TI4 *>(img)[i];
#endif
    float4 t = make_float4((float)a.x, (float)a.y, (float)a.z, (float)a.w);
    const int64_t o = prep_out_index(g, 4 * i) / 4;  // nx % 4 == 0 on this path
    if (msk != nullptr) {
      const TM4 m = reinterpret_cast<const TM4 *>(msk)[i];
      const float4 c = make_float4((float)m.x, (float)m.y, (float)m.z, (float)m.w);
      t.x *= c.x; t.y *= c.y; t.z *= c.z; t.w *= c.w;
      if (cf != nullptr) IFE_PREP_STORE(reinterpret_cast<float4 *>(cf) + o, c);
    }
    IFE_PREP_STORE(reinterpret_cast<float4 *>(tc) + o, t);
  }
}
template <typename TI, typename TM>
__global__ __launch_bounds__(256) void prep_kernel_scalar(const TI *__restrict__ img,
                                                          const TM *__restrict__ msk,
                                                          float *__restrict__ tc,
                                                          float *__restrict__ cf, int64_t i0,
                                                          int64_t n, PrepGeom g) {
  int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float t = (float)img[i];
    const int64_t o = prep_out_index(g, i);
    if (msk != nullptr) {
      const float c = (float)msk[i];
      t *= c;
      if (cf != nullptr) cf[o] = c;
    }
    tc[o] = t;
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

// Row f4: ({a_x*cT}{a*c} - {a_x*c}{a*cT}) / {a*c}^2, the differential normalized convolution
// sketched at NormalizedGaussianConvolutionImageFilter.h:28-44, in float as written, scaled
// to physical units; a zero denominator gives the Div functor's max().
__global__ __launch_bounds__(256) void diffconv_kernel(const float *__restrict__ g_ct,
                                                       const float *__restrict__ g_c,
                                                       const float *__restrict__ gx_ct,
                                                       const float *__restrict__ gx_c, float inv_sp,
                                                       float *__restrict__ outv, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float den = g_c[i] * g_c[i];
    const float num = gx_ct[i] * g_c[i] - gx_c[i] * g_ct[i];
    outv[i] = den != 0.0f ? (num / den) * inv_sp : FLT_MAX;
  }
}

}  // namespace ife
