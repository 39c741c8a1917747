// iir_kernels.hpp -- ITK's 4th-order recursive (Deriche) Gaussian as line kernels.
//
// Restates, per line, itk::RecursiveSeparableImageFilter::FilterDataArray with the
// ZeroOrder coefficients of itk::RecursiveGaussianImageFilter, which the reference
// reaches through NormalizedGaussianConvolutionImageFilter.hxx:51-55
// (SmoothingRecursiveGaussianImageFilter: axis order Z, X, Y; float images between
// axes; double line state).  SURVEY.md section 8 row a4.
//
// Mapping to the machine: one line per lane, 64 adjacent lines per wave.  A line is
// walked in register blocks of K samples.
//   forward sweep : causal recursion; at every block start the state
//                   (y[i-1..i-4] as double, x[i-1..i-3] as float) is written to a
//                   checkpoint array (44 B per line per K samples).
//   backward sweep: per block, reload the checkpoint, recompute the K causal values
//                   into registers, run the anticausal recursion over the same K
//                   samples, and store float(causal + anticausal).
// This keeps the exact sequential double arithmetic of the reference (every
// multiply and add rounds separately: the TU is built with -ffp-contract=off) while
// the only extra HBM traffic is the checkpoint array.  The causal partial sums never
// leave registers.
//
// Axis variants:
//   strided (Z, Y): lanes own adjacent x, samples are `sstride` elements apart;
//                   every load/store of a wave is one contiguous 256-B row.
//   contig (X)    : lines are contiguous in memory; a wave stages 64 lines x K
//                   samples through LDS (16-B global accesses, transposed so that a
//                   lane again owns one line).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ife {

struct IirCoef {
  double N0, N1, N2, N3;
  double D1, D2, D3, D4;
  double M1, M2, M3, M4;
  double BN1, BN2, BN3, BN4;
  double BM1, BM2, BM3, BM4;
};

struct IirGeom {
  int64_t n;        // samples per line
  int64_t nlines;   // number of lines
  int64_t sstride;  // element stride between consecutive samples of a line
  int64_t inner;    // strided: lines per contiguous row (line L -> base (L%inner)+(L/inner)*outer)
  int64_t outer;    // strided: element stride between rows of lines; contig: line pitch
  // contiguous-axis kernel only: the INPUT may be the Y-chunked image of a Z-slab as an
  // all-to-all leaves it, [in_w][nz][in_group][nx] (chunk h holds rows h*in_group ..), while
  // the output is the plain [nz][ny][nx] slab.  in_w <= 1: plain input.  Needs 64 | in_group
  // so that the 64 lines of a wave stay contiguous.
  int64_t in_w, in_group, in_nz;
};

struct CausalState {
  double x1, x2, x3;      // x[i-1], x[i-2], x[i-3]
  double y1, y2, y3, y4;  // y[i-1..i-4]
};
struct AntiState {
  double x1, x2, x3, x4;  // x[i+1..i+4]
  double y1, y2, y3, y4;  // y[i+1..i+4]
};

// scratch[i] = data[i]*N0 + data[i-1]*N1 + data[i-2]*N2 + data[i-3]*N3
// scratch[i] -= scratch[i-1]*D1 + scratch[i-2]*D2 + scratch[i-3]*D3 + scratch[i-4]*D4
__device__ __forceinline__ double causal_step(CausalState &s, double xin, const IirCoef &c) {
  const double a = xin * c.N0 + s.x1 * c.N1 + s.x2 * c.N2 + s.x3 * c.N3;
  const double t = s.y1 * c.D1 + s.y2 * c.D2 + s.y3 * c.D3 + s.y4 * c.D4;
  const double y = a - t;
  s.x3 = s.x2; s.x2 = s.x1; s.x1 = xin;
  s.y4 = s.y3; s.y3 = s.y2; s.y2 = s.y1; s.y1 = y;
  return y;
}
// Border form: for i < 4 the taps that reach before the line use outV1*BN_k, which is
// what the state holds (y_k initialised to outV1) times the boundary coefficient.
__device__ __forceinline__ double causal_step_edge(CausalState &s, double xin, const IirCoef &c,
                                                   int64_t i) {
  const double d1 = i < 1 ? c.BN1 : c.D1;
  const double d2 = i < 2 ? c.BN2 : c.D2;
  const double d3 = i < 3 ? c.BN3 : c.D3;
  const double d4 = i < 4 ? c.BN4 : c.D4;
  const double a = xin * c.N0 + s.x1 * c.N1 + s.x2 * c.N2 + s.x3 * c.N3;
  const double t = s.y1 * d1 + s.y2 * d2 + s.y3 * d3 + s.y4 * d4;
  const double y = a - t;
  s.x3 = s.x2; s.x2 = s.x1; s.x1 = xin;
  s.y4 = s.y3; s.y3 = s.y2; s.y2 = s.y1; s.y1 = y;
  return y;
}
// scratch[i] = data[i+1]*M1 + data[i+2]*M2 + data[i+3]*M3 + data[i+4]*M4
// scratch[i] -= scratch[i+1]*D1 + scratch[i+2]*D2 + scratch[i+3]*D3 + scratch[i+4]*D4
__device__ __forceinline__ double anti_step(AntiState &s, double xin, const IirCoef &c) {
  const double a = s.x1 * c.M1 + s.x2 * c.M2 + s.x3 * c.M3 + s.x4 * c.M4;
  const double t = s.y1 * c.D1 + s.y2 * c.D2 + s.y3 * c.D3 + s.y4 * c.D4;
  const double y = a - t;
  s.x4 = s.x3; s.x3 = s.x2; s.x2 = s.x1; s.x1 = xin;
  s.y4 = s.y3; s.y3 = s.y2; s.y2 = s.y1; s.y1 = y;
  return y;
}
__device__ __forceinline__ double anti_step_edge(AntiState &s, double xin, const IirCoef &c,
                                                 int64_t i, int64_t n) {
  const double d1 = i + 1 >= n ? c.BM1 : c.D1;
  const double d2 = i + 2 >= n ? c.BM2 : c.D2;
  const double d3 = i + 3 >= n ? c.BM3 : c.D3;
  const double d4 = i + 4 >= n ? c.BM4 : c.D4;
  const double a = s.x1 * c.M1 + s.x2 * c.M2 + s.x3 * c.M3 + s.x4 * c.M4;
  const double t = s.y1 * d1 + s.y2 * d2 + s.y3 * d3 + s.y4 * d4;
  const double y = a - t;
  s.x4 = s.x3; s.x3 = s.x2; s.x2 = s.x1; s.x1 = xin;
  s.y4 = s.y3; s.y3 = s.y2; s.y2 = s.y1; s.y1 = y;
  return y;
}

// ---- addressing ------------------------------------------------------------------
// Every global access below is a raw buffer access: a wave-uniform 128-bit resource
// (base pointer rebuilt per register block from scalars), a constant 32-bit per-lane
// byte offset, and a scalar per-sample offset.  The per-sample address arithmetic is
// scalar, no 64-bit address lives in VGPRs, and SGPR pressure stays at one resource
// per stream (the 20 double coefficients already take 40 SGPRs).
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void *ubase) {
  // stride 0, no range limit: offsets are validated on the host (ife_capi.hip)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ubase), 0, -1, 0x00020000);
}
// Timing-only diagnostics (MI355X guide, "price ONE buffer's traffic"): a resource with zero
// records drops every access through it while the instruction stream stays.  Never defined
// in the product build; results are wrong by construction.
__device__ __forceinline__ rsrc_t make_rsrc_null(const void *ubase) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ubase), 0, 0, 0x00020000);
}
#if defined(IFE_DIAG_NO_OUT)
#define IFE_OUT_RSRC make_rsrc_null
#else
#define IFE_OUT_RSRC make_rsrc
#endif
#if defined(IFE_DIAG_NO_CK)
#define IFE_CK_RSRC make_rsrc_null
#else
#define IFE_CK_RSRC make_rsrc
#endif
#if defined(IFE_DIAG_NO_IN)
#define IFE_IN_RSRC make_rsrc_null
#else
#define IFE_IN_RSRC make_rsrc
#endif
__device__ __forceinline__ float buf_ld_f32(rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st_f32(rsrc_t r, uint32_t voff, uint32_t soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, 0);
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double buf_ld_f64(rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st_f64(rsrc_t r, uint32_t voff, uint32_t soff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, 0);
}
template <typename T>
__device__ __forceinline__ float buf_ld_as_f32(rsrc_t r, uint32_t voff, uint32_t soff);
template <>
__device__ __forceinline__ float buf_ld_as_f32<float>(rsrc_t r, uint32_t voff, uint32_t soff) {
  return buf_ld_f32(r, voff, soff);
}
template <>
__device__ __forceinline__ float buf_ld_as_f32<uint8_t>(rsrc_t r, uint32_t voff, uint32_t soff) {
  return (float)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0);
}
template <>
__device__ __forceinline__ float buf_ld_as_f32<uint16_t>(rsrc_t r, uint32_t voff, uint32_t soff) {
  return (float)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
}
template <>
__device__ __forceinline__ float buf_ld_as_f32<int16_t>(rsrc_t r, uint32_t voff, uint32_t soff) {
  return (float)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
}

// ---- jobs ------------------------------------------------------------------------
// One launch runs up to IIR_MAX_JOBS independent line filters of the same geometry
// (numerator and denominator of up to three scales): blockIdx.y selects the job.  All
// line kernels read and write float volumes (the cast/multiply prepass feeds the first).
constexpr int IIR_MAX_JOBS = 8;
struct IirJob {
  const float *in;
  float *out;
  double *ck_y;  // [npairs][4][nlines]: y[i-1..i-4] at the start of every second block
  float *ck_x;   // [npairs][3][nlines]: x[i-1..i-3], contiguous-axis kernel only
  IirCoef c;
};
struct IirJobs {
  IirJob j[IIR_MAX_JOBS];
};

struct SrcF32 {
  const float *p;
  struct At { rsrc_t a; };
  __device__ __forceinline__ At at(int64_t u) const { return At{IFE_IN_RSRC(p + u)}; }
  __device__ __forceinline__ float ld(const At &b, uint32_t v, uint32_t s) const {
    return buf_ld_f32(b.a, v * 4u, s * 4u);
  }
};

struct Checkpoint {
  double *y;
  float *x;
};

// The recursion state is saved once per PAIR of register blocks (every 2K samples); the
// three input samples in front of the pair are re-read from the input (they sit in rows
// the sweep loads anyway).  That is 32 B per line per 32 samples each way.
__device__ __forceinline__ void ck_store(const Checkpoint &ck, int64_t pair, int64_t nl,
                                         int64_t Lw, uint32_t lane, const CausalState &s) {
  const rsrc_t ry = IFE_CK_RSRC(ck.y + (pair * 4) * nl + Lw);
  const uint32_t sy = (uint32_t)nl * 8u;
  buf_st_f64(ry, lane * 8u, 0u, s.y1);
  buf_st_f64(ry, lane * 8u, sy, s.y2);
  buf_st_f64(ry, lane * 8u, 2u * sy, s.y3);
  buf_st_f64(ry, lane * 8u, 3u * sy, s.y4);
}
__device__ __forceinline__ void ck_load(const Checkpoint &ck, int64_t pair, int64_t nl,
                                        int64_t Lw, uint32_t lane, CausalState &s) {
  const rsrc_t ry = IFE_CK_RSRC(ck.y + (pair * 4) * nl + Lw);
  const uint32_t sy = (uint32_t)nl * 8u;
  s.y1 = buf_ld_f64(ry, lane * 8u, 0u);
  s.y2 = buf_ld_f64(ry, lane * 8u, sy);
  s.y3 = buf_ld_f64(ry, lane * 8u, 2u * sy);
  s.y4 = buf_ld_f64(ry, lane * 8u, 3u * sy);
}

// The contiguous-axis kernel cannot re-read the x history cheaply (it would be 64
// scattered 4-B loads per wave), so it keeps it beside the state: 12 B per line per pair.
__device__ __forceinline__ void ck_store_x(const Checkpoint &ck, int64_t pair, int64_t nl,
                                           int64_t Lw, uint32_t lane, const CausalState &s) {
  const rsrc_t rx = make_rsrc(ck.x + (pair * 3) * nl + Lw);
  const uint32_t sx = (uint32_t)nl * 4u;
  buf_st_f32(rx, lane * 4u, 0u, (float)s.x1);
  buf_st_f32(rx, lane * 4u, sx, (float)s.x2);
  buf_st_f32(rx, lane * 4u, 2u * sx, (float)s.x3);
}
__device__ __forceinline__ void ck_load_x(const Checkpoint &ck, int64_t pair, int64_t nl,
                                          int64_t Lw, uint32_t lane, CausalState &s) {
  const rsrc_t rx = make_rsrc(ck.x + (pair * 3) * nl + Lw);
  const uint32_t sx = (uint32_t)nl * 4u;
  s.x1 = (double)buf_ld_f32(rx, lane * 4u, 0u);
  s.x2 = (double)buf_ld_f32(rx, lane * 4u, sx);
  s.x3 = (double)buf_ld_f32(rx, lane * 4u, 2u * sx);
}

// The first block of a pair is run twice from the same state (once to reach the second
// block, once for its own values).  Left alone, the compiler merges the two runs and keeps
// all K values of the first one alive across the second block: exactly the register
// footprint the recomputation is there to avoid.  Passing the state through an empty asm
// makes the second run opaque.
__device__ __forceinline__ void opaque_state(CausalState &s) {
  asm volatile("" : "+v"(s.y1), "+v"(s.y2), "+v"(s.y3), "+v"(s.y4));
}

// Causal recursion over one register block starting at sample i0.  `edge` blocks (the
// first one, and any block that may run past the line end) take the border form and skip
// samples >= n.  STORE keeps the K values for the anticausal pass.
template <int K, bool STORE>
__device__ __forceinline__ void causal_run(CausalState &s, const float (&x)[K], double (&cz)[K],
                                           const IirCoef &c, int64_t i0, int64_t n, bool edge) {
  if (!edge) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const double y = causal_step(s, (double)x[j], c);
      if (STORE) cz[j] = y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (i0 + j < n) {
        const double y = causal_step_edge(s, (double)x[j], c, i0 + j);
        if (STORE) cz[j] = y;
      }
  }
}
// Anticausal recursion over the same block, leaving float(causal + anticausal) in x.
template <int K>
__device__ __forceinline__ void anti_run(AntiState &a, float (&x)[K], const double (&cz)[K],
                                         const IirCoef &c, int64_t i0, int64_t n, bool edge) {
  if (!edge) {
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
      const double y = anti_step(a, (double)x[j], c);
      x[j] = (float)(cz[j] + y);
    }
  } else {
#pragma unroll
    for (int j = K - 1; j >= 0; --j)
      if (i0 + j < n) {
        const double y = anti_step_edge(a, (double)x[j], c, i0 + j, n);
        x[j] = (float)(cz[j] + y);
      }
  }
}

// =================================================================================
// strided axes (Z and Y)
// =================================================================================
template <int K>
__global__ __launch_bounds__(256) void iir_strided_kernel(IirJobs jobs, IirGeom g) {
  const IirJob &J = jobs.j[blockIdx.y];
  const SrcF32 src{J.in};
  float *__restrict__ out = J.out;
  const IirCoef c = J.c;
  const Checkpoint ck{J.ck_y, J.ck_x};
  const uint32_t lane = threadIdx.x & 63u;
  const int64_t Lw =
      uniform64((int64_t)blockIdx.x * blockDim.x + (int64_t)(threadIdx.x & ~63u));
  if (Lw >= g.nlines) return;  // wave-uniform
  const int64_t n = g.n, st = g.sstride, nl = g.nlines;
  int64_t L = Lw + lane;
  const bool live = L < nl;
  if (!live) L = nl - 1;  // compute on a valid line, never store
  const int64_t base = (L % g.inner) + (L / g.inner) * g.outer;
  const int64_t wbase = uniform64(base);          // lane 0 of the wave
  const uint32_t voff = (uint32_t)(base - wbase);  // elements (one row jump at most)
  const uint32_t sst = (uint32_t)st;               // 2K*st*4 < 2^31 is checked on the host
  const int64_t nb = (n + K - 1) / K;               // register blocks
  const int64_t np = (nb + 1) / 2;                  // block pairs (checkpoint granularity)

  // ---------------- forward sweep: a checkpoint in front of every pair >= 1 ----------------
  {
    float xb[K], xn[K];
    CausalState s;
    const int64_t last = 2 * (np - 1);  // first block of the last pair: nothing needed beyond
    if (last > 0) {
      const auto B = src.at(wbase);
#pragma unroll
      for (int j = 0; j < K; ++j) xb[j] = src.ld(B, voff, (uint32_t)j * sst);
    }
    for (int64_t b = 0; b < last; ++b) {
      const int64_t i0 = b * K;
      if (b + 1 < last) {  // prefetch the next (full) block
        const auto B = src.at(wbase + (i0 + K) * st);
#pragma unroll
        for (int j = 0; j < K; ++j) xn[j] = src.ld(B, voff, (uint32_t)j * sst);
      }
      if (b > 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) causal_step(s, (double)xb[j], c);
      } else {
        const double x0 = (double)xb[0];
        s.x1 = s.x2 = s.x3 = x0;
        s.y1 = s.y2 = s.y3 = s.y4 = x0;
#pragma unroll
        for (int j = 0; j < K; ++j) causal_step_edge(s, (double)xb[j], c, j);
      }
      if ((b & 1) && live) ck_store(ck, (b + 1) / 2, nl, Lw, lane, s);
#pragma unroll
      for (int j = 0; j < K; ++j) xb[j] = xn[j];
    }
  }

  // ---------------- backward sweep, one pair of blocks per iteration ----------------
  {
    float xa[K], xb[K];  // blocks 2p and 2p+1 of the current pair
    float na[K], nb2[K];  // the next pair (p-1), in flight
    float h1 = 0.f, h2 = 0.f, h3 = 0.f;     // x[i-1..i-3] in front of the current pair
    float nh1 = 0.f, nh2 = 0.f, nh3 = 0.f;  // the same for the next pair
    AntiState a;
    auto load_pair = [&](int64_t p, float (&pa)[K], float (&pb)[K], bool clampn) {
      const int64_t i0 = 2 * p * K;
      const auto B = src.at(wbase + i0 * st);
      if (!clampn) {
#pragma unroll
        for (int j = 0; j < K; ++j) pa[j] = src.ld(B, voff, (uint32_t)j * sst);
#pragma unroll
        for (int j = 0; j < K; ++j) pb[j] = src.ld(B, voff, (uint32_t)(K + j) * sst);
      } else {
        const uint32_t lastj = (uint32_t)(n - 1 - i0);  // 0 .. 2K-1
#pragma unroll
        for (int j = 0; j < K; ++j)
          pa[j] = src.ld(B, voff, ((uint32_t)j < lastj ? (uint32_t)j : lastj) * sst);
#pragma unroll
        for (int j = 0; j < K; ++j)
          pb[j] = src.ld(B, voff, ((uint32_t)(K + j) < lastj ? (uint32_t)(K + j) : lastj) * sst);
      }
    };
    auto load_hist = [&](int64_t p, float &q1, float &q2, float &q3) {
      const auto B = src.at(wbase + (2 * p * K - 3) * st);  // p >= 1
      q3 = src.ld(B, voff, 0u);
      q2 = src.ld(B, voff, sst);
      q1 = src.ld(B, voff, 2u * sst);
    };
    // checkpoint of the current pair and, in flight, of the next one: all loads of an
    // iteration are issued before its stores, so the wait for them (vmcnt counts in
    // order) can leave the 2K stores of the iteration outstanding
    CausalState sc, sn;
    sc.y1 = sc.y2 = sc.y3 = sc.y4 = 0.0;
    sn = sc;
    load_pair(np - 1, xa, xb, true);
    if (np > 1) {
      load_hist(np - 1, h1, h2, h3);
      ck_load(ck, np - 1, nl, Lw, live ? lane : 0u, sc);
    }
    {
      // x[n-1]: the clamped loads make every slot past the line end hold it
      const double xN = (double)xb[K - 1];
      a.x1 = a.x2 = a.x3 = a.x4 = xN;
      a.y1 = a.y2 = a.y3 = a.y4 = xN;
    }
    for (int64_t p = np - 1; p >= 0; --p) {
      const int64_t i0 = 2 * p * K, i1 = i0 + K;
      if (p > 0) {
        load_pair(p - 1, na, nb2, false);
        if (p > 1) {
          load_hist(p - 1, nh1, nh2, nh3);
          ck_load(ck, p - 1, nl, Lw, live ? lane : 0u, sn);
        }
      }
      CausalState s0;
      if (p > 0) {
        s0.y1 = sc.y1; s0.y2 = sc.y2; s0.y3 = sc.y3; s0.y4 = sc.y4;
        s0.x1 = (double)h1; s0.x2 = (double)h2; s0.x3 = (double)h3;
      } else {
        const double x0 = (double)xa[0];
        s0.x1 = s0.x2 = s0.x3 = x0;
        s0.y1 = s0.y2 = s0.y3 = s0.y4 = x0;
      }
      const bool head = p == 0;                 // border form at the line start
      const bool tail = i1 + K + 4 > n;         // second block touches the last 4 samples
      const rsrc_t ro = IFE_OUT_RSRC(out + wbase + i0 * st);
      double cz[K];
      if (i1 < n) {  // second block of the pair exists: run through the first to reach it
        CausalState s = s0;
        causal_run<K, false>(s, xa, cz, c, i0, n, head);
        causal_run<K, true>(s, xb, cz, c, i1, n, tail);
        anti_run<K>(a, xb, cz, c, i1, n, tail);
        if (live) {
#pragma unroll
          for (int j = 0; j < K; ++j)
            if (!tail || i1 + j < n) buf_st_f32(ro, voff * 4u, (uint32_t)(K + j) * sst * 4u, xb[j]);
        }
      }
      {
        const bool edge = head || (i0 + K + 4 > n);
        CausalState s = s0;
        opaque_state(s);
        causal_run<K, true>(s, xa, cz, c, i0, n, edge);
        anti_run<K>(a, xa, cz, c, i0, n, edge);
        if (live) {
#pragma unroll
          for (int j = 0; j < K; ++j)
            if (!edge || i0 + j < n) buf_st_f32(ro, voff * 4u, (uint32_t)j * sst * 4u, xa[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < K; ++j) { xa[j] = na[j]; xb[j] = nb2[j]; }
      h1 = nh1; h2 = nh2; h3 = nh3;
      sc.y1 = sn.y1; sc.y2 = sn.y2; sc.y3 = sn.y3; sc.y4 = sn.y4;
    }
  }
}

// ---------------------------------------------------------------------------------
// Strided variant with a checkpoint in front of EVERY register block: 16 B/voxel more
// checkpoint traffic than the pair form, but no block is recomputed twice (the pair form
// spends 8 of its 58 double operations per sample on that, and the kernel is bound by
// double-precision issue, not by HBM: with all memory traffic removed it still takes 75 %
// of its time).
// ---------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void iir_strided1_kernel(IirJobs jobs, IirGeom g) {
  const IirJob &J = jobs.j[blockIdx.y];
  const SrcF32 src{J.in};
  float *__restrict__ out = J.out;
  const IirCoef c = J.c;
  const Checkpoint ck{J.ck_y, J.ck_x};
  const uint32_t lane = threadIdx.x & 63u;
  const int64_t Lw =
      uniform64((int64_t)blockIdx.x * blockDim.x + (int64_t)(threadIdx.x & ~63u));
  if (Lw >= g.nlines) return;  // wave-uniform
  const int64_t n = g.n, st = g.sstride, nl = g.nlines;
  int64_t L = Lw + lane;
  const bool live = L < nl;
  if (!live) L = nl - 1;
  const int64_t base = (L % g.inner) + (L / g.inner) * g.outer;
  const int64_t wbase = uniform64(base);
  const uint32_t voff = (uint32_t)(base - wbase);
  const uint32_t sst = (uint32_t)st;
  const int64_t nb = (n + K - 1) / K;

  // ---------------- forward sweep: checkpoint b+1 after block b, b <= nb-2 ----------------
  {
    float xb[K], xn[K];
    CausalState s;
    if (nb > 1) {
      const auto B = src.at(wbase);
#pragma unroll
      for (int j = 0; j < K; ++j) xb[j] = src.ld(B, voff, (uint32_t)j * sst);
    }
    for (int64_t b = 0; b + 1 < nb; ++b) {
      const int64_t i0 = b * K;
      if (b + 2 < nb) {
        const auto B = src.at(wbase + (i0 + K) * st);
#pragma unroll
        for (int j = 0; j < K; ++j) xn[j] = src.ld(B, voff, (uint32_t)j * sst);
      }
      if (b > 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) causal_step(s, (double)xb[j], c);
      } else {
        const double x0 = (double)xb[0];
        s.x1 = s.x2 = s.x3 = x0;
        s.y1 = s.y2 = s.y3 = s.y4 = x0;
#pragma unroll
        for (int j = 0; j < K; ++j) causal_step_edge(s, (double)xb[j], c, j);
      }
      if (live) ck_store(ck, b + 1, nl, Lw, lane, s);
#pragma unroll
      for (int j = 0; j < K; ++j) xb[j] = xn[j];
    }
  }

  // ---------------- backward sweep, one block per iteration ----------------
  {
    float xa[K], na[K];
    float h1 = 0.f, h2 = 0.f, h3 = 0.f, nh1 = 0.f, nh2 = 0.f, nh3 = 0.f;
    AntiState a;
    CausalState sc, sn;
    sc.y1 = sc.y2 = sc.y3 = sc.y4 = 0.0;
    sn = sc;
    auto load_hist = [&](int64_t b, float &q1, float &q2, float &q3) {
      const auto B = src.at(wbase + (b * K - 3) * st);  // b >= 1
      q3 = src.ld(B, voff, 0u);
      q2 = src.ld(B, voff, sst);
      q1 = src.ld(B, voff, 2u * sst);
    };
    {
      const int64_t i0 = (nb - 1) * K;
      const auto B = src.at(wbase + i0 * st);
      const uint32_t lastj = (uint32_t)(n - 1 - i0);
#pragma unroll
      for (int j = 0; j < K; ++j)
        xa[j] = src.ld(B, voff, ((uint32_t)j < lastj ? (uint32_t)j : lastj) * sst);
      if (nb > 1) {
        load_hist(nb - 1, h1, h2, h3);
        ck_load(ck, nb - 1, nl, Lw, live ? lane : 0u, sc);
      }
      const double xN = (double)xa[K - 1];
      a.x1 = a.x2 = a.x3 = a.x4 = xN;
      a.y1 = a.y2 = a.y3 = a.y4 = xN;
    }
    for (int64_t b = nb - 1; b >= 0; --b) {
      const int64_t i0 = b * K;
      if (b > 0) {  // all loads of the iteration come before its stores
        const auto B = src.at(wbase + (i0 - K) * st);
#pragma unroll
        for (int j = 0; j < K; ++j) na[j] = src.ld(B, voff, (uint32_t)j * sst);
        if (b > 1) {
          load_hist(b - 1, nh1, nh2, nh3);
          ck_load(ck, b - 1, nl, Lw, live ? lane : 0u, sn);
        }
      }
      CausalState s;
      if (b > 0) {
        s.y1 = sc.y1; s.y2 = sc.y2; s.y3 = sc.y3; s.y4 = sc.y4;
        s.x1 = (double)h1; s.x2 = (double)h2; s.x3 = (double)h3;
      } else {
        const double x0 = (double)xa[0];
        s.x1 = s.x2 = s.x3 = x0;
        s.y1 = s.y2 = s.y3 = s.y4 = x0;
      }
      const bool edge = (b == 0) || (i0 + K + 4 > n);
      double cz[K];
      causal_run<K, true>(s, xa, cz, c, i0, n, edge);
      anti_run<K>(a, xa, cz, c, i0, n, edge);
      if (live) {
        const rsrc_t ro = IFE_OUT_RSRC(out + wbase + i0 * st);
#pragma unroll
        for (int j = 0; j < K; ++j)
          if (!edge || i0 + j < n) buf_st_f32(ro, voff * 4u, (uint32_t)j * sst * 4u, xa[j]);
      }
#pragma unroll
      for (int j = 0; j < K; ++j) xa[j] = na[j];
      h1 = nh1; h2 = nh2; h3 = nh3;
      sc.y1 = sn.y1; sc.y2 = sn.y2; sc.y3 = sn.y3; sc.y4 = sn.y4;
    }
  }
}

// =================================================================================
// contiguous axis (X): 64 lines x 2K samples staged through LDS per wave
// =================================================================================
// A tile is two register blocks wide (2K = 32 samples = 128 B per line), so every global
// access of a line is one whole 128-B cache line; with K-wide tiles each line was
// fetched twice (PMC: 46 % over-fetch on the X pass).
template <int W>
struct XTile {
  static constexpr int PITCH = W + 4;  // floats; keeps rows 16-B aligned, conflict-free b128 reads
  static constexpr int FLOATS = 64 * PITCH;
};

// A wave's DS instructions execute in order, so a tile written and then read by the
// same wave needs no barrier: only the compiler must be kept from moving LDS
// accesses across this point (the "memory" clobber), and global traffic in flight
// is left alone (no vmcnt wait, unlike a workgroup-scope fence).
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Fill the wave's LDS tile with samples [i0, i0+W) of its 64 lines.  Samples past the
// line end repeat x[n-1].  `rows` points at sample 0 of the wave's first line (uniform).
template <int W>
__device__ __forceinline__ void xtile_fill(const float *rows, float *tile, uint32_t lane,
                                           int64_t nrows, int64_t pitch, int64_t i0, int64_t n,
                                           bool vec_ok) {
  constexpr int P = XTile<W>::PITCH;
  constexpr int V = W / 4;     // float4 per line
  constexpr int LPI = 64 / V;  // lines covered per wave instruction
  if (vec_ok && i0 + W <= n) {
    const rsrc_t r4 = make_rsrc(rows + i0);
    u32x4 v[V];
#pragma unroll
    for (int r = 0; r < V; ++r) {
      uint32_t row = (lane / V) + r * LPI;
      if ((int64_t)row >= nrows) row = (uint32_t)(nrows - 1);
      const uint32_t off = (row * (uint32_t)pitch + 4u * (lane % V)) * 4u;
      v[r] = __builtin_amdgcn_raw_buffer_load_b128(r4, off, 0u, 0);
    }
#pragma unroll
    for (int r = 0; r < V; ++r)
      *reinterpret_cast<u32x4 *>(tile + ((lane / V) + r * LPI) * P + 4 * (lane % V)) = v[r];
  } else {  // unaligned or ragged tile: rolled loop (rare path, keep its footprint small)
#pragma clang loop unroll(disable)
    for (int e = 0; e < W; ++e) {
      const uint32_t idx = e * 64 + lane;
      uint32_t row = idx / W;
      const uint32_t col = idx % W;
      const uint32_t trow = row;
      if ((int64_t)row >= nrows) row = (uint32_t)(nrows - 1);
      int64_t i = i0 + col;
      if (i >= n) i = n - 1;
      tile[trow * P + col] = rows[(int64_t)row * pitch + i];
    }
  }
  wave_lds_sync();
}

// Register block h (K samples) of the lane's own line, out of / back into the tile.
template <int W, int K>
__device__ __forceinline__ void xtile_get(const float *tile, uint32_t lane, int h, float (&xb)[K]) {
  constexpr int P = XTile<W>::PITCH;
#pragma unroll
  for (int q = 0; q < K / 4; ++q) {
    const float4 t = *reinterpret_cast<const float4 *>(tile + lane * P + h * K + 4 * q);
    xb[4 * q + 0] = t.x; xb[4 * q + 1] = t.y; xb[4 * q + 2] = t.z; xb[4 * q + 3] = t.w;
  }
}
template <int W, int K>
__device__ __forceinline__ void xtile_put(float *tile, uint32_t lane, int h, const float (&xb)[K]) {
  constexpr int P = XTile<W>::PITCH;
#pragma unroll
  for (int q = 0; q < K / 4; ++q)
    *reinterpret_cast<float4 *>(tile + lane * P + h * K + 4 * q) =
        make_float4(xb[4 * q + 0], xb[4 * q + 1], xb[4 * q + 2], xb[4 * q + 3]);
}

// Write the tile back to samples [i0, i0+W) of the lines (valid samples only).
template <int W>
__device__ __forceinline__ void xtile_drain(float *rows, float *tile, uint32_t lane,
                                            int64_t nrows, int64_t pitch, int64_t i0, int64_t n,
                                            bool vec_ok) {
  constexpr int P = XTile<W>::PITCH;
  constexpr int V = W / 4;
  constexpr int LPI = 64 / V;
  wave_lds_sync();
  if (vec_ok && i0 + W <= n) {
    const rsrc_t r4 = make_rsrc(rows + i0);
#pragma unroll
    for (int r = 0; r < V; ++r) {
      const uint32_t row = (lane / V) + r * LPI;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(tile + row * P + 4 * (lane % V));
      const uint32_t off = (row * (uint32_t)pitch + 4u * (lane % V)) * 4u;
      if ((int64_t)row < nrows) __builtin_amdgcn_raw_buffer_store_b128(v, r4, off, 0u, 0);
    }
  } else {
#pragma clang loop unroll(disable)
    for (int e = 0; e < W; ++e) {
      const uint32_t idx = e * 64 + lane;
      const uint32_t row = idx / W, col = idx % W;
      const int64_t i = i0 + col;
      if ((int64_t)row < nrows && i < n) rows[(int64_t)row * pitch + i] = tile[row * P + col];
    }
  }
  wave_lds_sync();
}

template <int K>
__global__ __launch_bounds__(256, 2) void iir_contig_kernel(IirJobs jobs, IirGeom g) {
  const IirJob &J = jobs.j[blockIdx.y];
  const float *__restrict__ in = J.in;
  float *__restrict__ out = J.out;
  const IirCoef c = J.c;
  const Checkpoint ck{J.ck_y, J.ck_x};
  constexpr int W = 2 * K;  // tile width: two register blocks
  __shared__ __attribute__((aligned(16))) float lds[4 * XTile<W>::FLOATS];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *tile = lds + wave * XTile<W>::FLOATS;
  const int64_t line0 = uniform64(((int64_t)blockIdx.x * 4 + wave) * 64);
  if (line0 >= g.nlines) return;
  const int64_t n = g.n, nl = g.nlines, pitch = g.outer;
  const int64_t nrows = nl - line0 < 64 ? nl - line0 : 64;  // live lines of this wave
  const int64_t nt = (n + W - 1) / W;
  // 16-B accesses need aligned rows; the per-lane byte offset must fit 32 bits
  const bool vec_ok = (pitch % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                      (pitch * 64 * 4 < (int64_t)0x7fffffff);
  const bool live = (int64_t)lane < nrows;
  int64_t in_line0 = line0;
  if (g.in_w > 1) {
    const int64_t gi = line0 / g.in_group, within = line0 % g.in_group;  // gi = z*in_w + h
    in_line0 = ((gi % g.in_w) * g.in_nz + gi / g.in_w) * g.in_group + within;
  }
  const float *rows_in = in + in_line0 * pitch;
  float *rows_out = out + line0 * pitch;

  float xb[K];

  // ---------------- forward sweep: a checkpoint in front of every tile >= 1 ----------------
  {
    CausalState s;
    for (int64_t t = 0; t + 1 < nt; ++t) {
      xtile_fill<W>(rows_in, tile, lane, nrows, pitch, t * W, n, vec_ok);
#pragma clang loop unroll(disable)
      for (int h = 0; h < 2; ++h) {
        xtile_get<W, K>(tile, lane, h, xb);
        if (t > 0 || h > 0) {
#pragma unroll
          for (int j = 0; j < K; ++j) causal_step(s, (double)xb[j], c);
        } else {
          const double x0 = (double)xb[0];
          s.x1 = s.x2 = s.x3 = x0;
          s.y1 = s.y2 = s.y3 = s.y4 = x0;
#pragma unroll
          for (int j = 0; j < K; ++j) causal_step_edge(s, (double)xb[j], c, j);
        }
      }
      if (live) {
        ck_store(ck, t + 1, nl, line0, lane, s);
        ck_store_x(ck, t + 1, nl, line0, lane, s);
      }
      wave_lds_sync();  // reads of this tile are done before the next fill overwrites it
    }
  }

  // ---------------- backward sweep, one tile (pair of blocks) per iteration ----------------
  {
    AntiState a;
    for (int64_t t = nt - 1; t >= 0; --t) {
      const int64_t i0 = t * W, i1 = i0 + K;
      xtile_fill<W>(rows_in, tile, lane, nrows, pitch, i0, n, vec_ok);
      if (t == nt - 1) {
        // x[n-1]: the clamped fill makes every slot past the line end hold it
        const double xN = (double)tile[lane * XTile<W>::PITCH + W - 1];
        a.x1 = a.x2 = a.x3 = a.x4 = xN;
        a.y1 = a.y2 = a.y3 = a.y4 = xN;
      }
      CausalState s0;
      float xa[K];
      xtile_get<W, K>(tile, lane, 0, xa);
      if (t > 0) {
        ck_load(ck, t, nl, line0, live ? lane : 0u, s0);
        ck_load_x(ck, t, nl, line0, live ? lane : 0u, s0);
      } else {
        const double x0 = (double)xa[0];
        s0.x1 = s0.x2 = s0.x3 = x0;
        s0.y1 = s0.y2 = s0.y3 = s0.y4 = x0;
      }
      const bool head = t == 0;
      double cz[K];
      if (i1 < n) {
        const bool tail = i1 + K + 4 > n;
        CausalState s = s0;
        causal_run<K, false>(s, xa, cz, c, i0, n, head);
        xtile_get<W, K>(tile, lane, 1, xb);
        causal_run<K, true>(s, xb, cz, c, i1, n, tail);
        anti_run<K>(a, xb, cz, c, i1, n, tail);
        xtile_put<W, K>(tile, lane, 1, xb);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const bool edge = head || (i0 + K + 4 > n);
        CausalState s = s0;
        opaque_state(s);
        xtile_get<W, K>(tile, lane, 0, xa);
        causal_run<K, true>(s, xa, cz, c, i0, n, edge);
        anti_run<K>(a, xa, cz, c, i0, n, edge);
        xtile_put<W, K>(tile, lane, 0, xa);
      }
      xtile_drain<W>(rows_out, tile, lane, nrows, pitch, i0, n, vec_ok);
    }
  }
}

}  // namespace ife
