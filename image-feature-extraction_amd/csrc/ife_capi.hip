// ife_capi.hip -- C-ABI (include/ife_hip.h) over the HIP kernels.  gfx950 only.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// (-ffp-contract=off is part of the numerical contract, see iir_kernels.inc).
#include "../../include/ife_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "feature_kernels.hpp"
#include "iir_types.hpp"
#include "stats_kernels.hpp"
#define IFE_IIR_NS iir_exact
#define IFE_IIR_FMA 0
#include "iir_kernels.inc"
#undef IFE_IIR_NS
#undef IFE_IIR_FMA
#define IFE_IIR_NS iir_fma
#define IFE_IIR_FMA 1
#include "iir_kernels.inc"
#undef IFE_IIR_NS
#undef IFE_IIR_FMA

using namespace ife;

namespace {

enum KernelKind {
  KK_IIR_Z = 0,
  KK_IIR_X,
  KK_IIR_Y,
  KK_ZSLAB_SWEEP,
  KK_ZSLAB_COMBINE,
  KK_FEATURES,
  KK_EIG_BATCH,
  KK_DIVIDE,
  KK_MASK,
  KK_PREP,
  KK_SORT_HIST,
  KK_SORT_SCAN,
  KK_SORT_SCATTER,
  KK_GATHER,
  KK_EDGES,
  KK_HIST,
  KK_COUNT
};
const char *kKindNames[KK_COUNT] = {"iir_z", "iir_x", "iir_y", "zslab_sweep", "zslab_combine", "features", "eig_batch",
                                    "divide", "mask_f64", "prep", "sort_hist", "sort_scan",
                                    "sort_scatter", "gather", "edges", "dense_histogram"};

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};
constexpr int IFE_MAX_SLOTS = 3;

struct ProfRec {
  int kind;
  hipEvent_t a, b;
};

thread_local std::string g_create_error;

}  // namespace

struct ife_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int trig_mode = 2;  // IFE_OPT_TRIG_MODE: float trigonometry inside the 1e-5 bar (ife_hip.h)
  int dscale_mode = 0;
  int profile = 0;
  int zchunk = 64;
  int iir_block = 0;   // 0: per axis (z 12, y 12, x 16); else 8 | 10 | 12 | 16
  int iir_fma = 0;   // 1: fused multiply-add in the line recurrences (opt-in, not bit-exact)
  int iir_ckpt = 2;  // register blocks per checkpoint of the strided line kernel: 1 or 2
  int fused_divide = 1;  // last axis pass stores numerator / denominator (sibling waves), not two fields
  int const_lines = 1;   // lines of one repeated 0 or 1 are copied, not filtered, where verified exact
  int feat_ring = 1;     // feature kernel: planes by LDS-DMA into a ring where the source is one float field
  std::map<std::tuple<double, double, int64_t>, uint32_t> const_flags;  // (sigma, spacing, length) -> IirJob::const_lines
  // per scale slot: numerator ping/pong, denominator ping/pong (up to three scales run
  // through the line kernels together)
  DevBuf fld[IFE_MAX_SLOTS][4];
  DevBuf pre[2];  // image*certainty and certainty as float (prepass, shared by all scales)
  DevBuf ck_y[IIR_MAX_JOBS], ck_x[IIR_MAX_JOBS];  // one checkpoint area per concurrent job
  DevBuf st_img, st_mask, st_aux, st_out;  // HOST-mode staging
  // streaming form of the scale loop (ife_emphysema_features_begin / _fetch / _end)
  DevBuf sc_out;                     // all scales, device resident
  std::vector<hipEvent_t> sc_done;   // one per scale: its feature launch has finished
  hipStream_t sc_copy = nullptr;     // device-to-host copies beside the kernels of later scales
  size_t sc_scale_bytes = 0;
  std::vector<hipEvent_t> *scale_events = nullptr;  // emphysema_typed records into this when set
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  double acc_ms[KK_COUNT] = {0};
  int64_t acc_n[KK_COUNT] = {0};
};

namespace {

int fail(ife_ctx *ctx, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  else g_create_error = buf;
  return code;
}

#define IFE_HIP(ctx, call)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(ctx, e_ == hipErrorOutOfMemory ? IFE_E_NOMEM : IFE_E_HIP, "%s: %s", #call, \
                  hipGetErrorString(e_));                                                   \
  } while (0)

int ensure(ife_ctx *ctx, DevBuf &b, size_t bytes) {
  if (b.cap >= bytes) return IFE_OK;
  if (b.p) {
    IFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    IFE_HIP(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  IFE_HIP(ctx, hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return IFE_OK;
}

int check_vol(ife_ctx *ctx, const ife_volume_desc *v, bool need_iir) {
  if (!v) return fail(ctx, IFE_E_ARG, "volume descriptor is null");
  if (v->nx <= 0 || v->ny <= 0 || v->nz <= 0)
    return fail(ctx, IFE_E_SIZE, "volume size must be positive (got %lld x %lld x %lld)",
                (long long)v->nx, (long long)v->ny, (long long)v->nz);
  if (v->nx > 0x7fffffff || v->ny > 0x7fffffff || v->nz > 0x7fffffff)
    return fail(ctx, IFE_E_SIZE, "axis length exceeds 2^31-1");
  if (!(v->sx > 0.0) || !(v->sy > 0.0) || !(v->sz > 0.0))
    return fail(ctx, IFE_E_ARG, "spacing must be positive");
  if (need_iir && (v->nx < 4 || v->ny < 4 || v->nz < 4))
    return fail(ctx, IFE_E_SIZE,
                "the recursive Gaussian needs at least 4 voxels along every axis "
                "(got %lld x %lld x %lld)",
                (long long)v->nx, (long long)v->ny, (long long)v->nz);
  return IFE_OK;
}

// ---- profiling -------------------------------------------------------------------
struct ProfScope {
  ife_ctx *ctx;
  ProfRec rec;
  bool on;
  ProfScope(ife_ctx *c, int kind) : ctx(c), on(c->profile != 0) {
    if (!on) return;
    rec.kind = kind;
    auto get = [&]() {
      hipEvent_t e;
      if (!ctx->ev_pool.empty()) {
        e = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
      } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
      }
      return e;
    };
    rec.a = get();
    rec.b = get();
    if (!rec.a || !rec.b) { on = false; return; }
    (void)hipEventRecord(rec.a, ctx->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(rec.b, ctx->stream);
    ctx->prof.push_back(rec);
  }
};

int drain_profile(ife_ctx *ctx) {
  if (ctx->prof.empty()) return IFE_OK;
  IFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto &r : ctx->prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      ctx->acc_ms[r.kind] += ms;
      ctx->acc_n[r.kind] += 1;
    }
    ctx->ev_pool.push_back(r.a);
    ctx->ev_pool.push_back(r.b);
  }
  ctx->prof.clear();
  return IFE_OK;
}

// ---- recursive Gaussian coefficients ------------------------------------------------
// [ITK-upstream] itk::RecursiveGaussianImageFilter::SetUp, ZeroOrder,
// NormalizeAcrossScale off.  Host-side, double.
// sin and cos of the pole angles through ONE sincos call, as GCC compiles ITK's adjacent
// std::sin / std::cos: glibc's sincos and cos differ in the last bit for some arguments, and
// 1 + sum(D) amplifies that to one float ulp of an output at one sample in 10^8-10^9
// (oracle/ife_oracle.c: pole_sincos; found by tests/test_gpu_fuzz.py).
void pole_sincos(double x, double &s, double &c) { ::sincos(x, &s, &c); }

int gauss_coeffs(double sigma, double spacing, IirCoef *c) {
  const double A1 = 1.3530, B1 = 1.8151, W1 = 0.6681, L1 = -1.3932;
  const double A2 = -0.3531, B2 = 0.0902, W2 = 2.0787, L2 = -1.3732;
  if (spacing < 0.0) spacing = -spacing;
  if (spacing < 1e-8) return -1;
  const double sd = sigma / spacing;
  double Sin1, Sin2, Cos1, Cos2;
  pole_sincos(W1 / sd, Sin1, Cos1);
  pole_sincos(W2 / sd, Sin2, Cos2);
  const double Exp1 = std::exp(L1 / sd), Exp2 = std::exp(L2 / sd);
  c->D4 = Exp1 * Exp1 * Exp2 * Exp2;
  c->D3 = -2 * Cos1 * Exp1 * Exp2 * Exp2;
  c->D3 += -2 * Cos2 * Exp2 * Exp1 * Exp1;
  c->D2 = 4 * Cos2 * Cos1 * Exp1 * Exp2;
  c->D2 += Exp1 * Exp1 + Exp2 * Exp2;
  c->D1 = -2 * (Exp2 * Cos2 + Exp1 * Cos1);
  const double SD = 1.0 + c->D1 + c->D2 + c->D3 + c->D4;
  c->N0 = A1 + A2;
  c->N1 = Exp2 * (B2 * Sin2 - (A2 + 2 * A1) * Cos2);
  c->N1 += Exp1 * (B1 * Sin1 - (A1 + 2 * A2) * Cos1);
  c->N2 = (A1 + A2) * Cos2 * Cos1;
  c->N2 -= B1 * Cos2 * Sin1 + B2 * Cos1 * Sin2;
  c->N2 *= 2 * Exp1 * Exp2;
  c->N2 += A2 * Exp1 * Exp1 + A1 * Exp2 * Exp2;
  c->N3 = Exp2 * Exp1 * Exp1 * (B2 * Sin2 - A2 * Cos2);
  c->N3 += Exp1 * Exp2 * Exp2 * (B1 * Sin1 - A1 * Cos1);
  const double SN = c->N0 + c->N1 + c->N2 + c->N3;
  const double alpha0 = 2 * SN / SD - c->N0;
  const double nrm = 1.0 / alpha0;
  c->N0 *= nrm; c->N1 *= nrm; c->N2 *= nrm; c->N3 *= nrm;
  c->M1 = c->N1 - c->D1 * c->N0;
  c->M2 = c->N2 - c->D2 * c->N0;
  c->M3 = c->N3 - c->D3 * c->N0;
  c->M4 = -c->D4 * c->N0;
  const double SN2 = c->N0 + c->N1 + c->N2 + c->N3;
  const double SM2 = c->M1 + c->M2 + c->M3 + c->M4;
  const double SD2 = 1.0 + c->D1 + c->D2 + c->D3 + c->D4;
  c->BN1 = c->D1 * SN2 / SD2; c->BN2 = c->D2 * SN2 / SD2;
  c->BN3 = c->D3 * SN2 / SD2; c->BN4 = c->D4 * SN2 / SD2;
  c->BM1 = c->D1 * SM2 / SD2; c->BM2 = c->D2 * SM2 / SD2;
  c->BM3 = c->D3 * SM2 / SD2; c->BM4 = c->D4 * SM2 / SD2;
  return 0;
}

// Orders 1 and 2 of the same routine (row f4; the reference only sketches their use,
// NormalizedGaussianConvolutionImageFilter.h:28-44): per-order constants of the exponential
// series, normalisation to a unit response on a unit ramp / unit parabola in pixel units,
// and the antisymmetric form of the anticausal coefficients for the first order.
void n_coefficients(double sd, double A1, double B1, double W1, double L1, double A2, double B2,
                    double W2, double L2, double &N0, double &N1, double &N2, double &N3, double &SN,
                    double &DN, double &EN) {
  double Sin1, Sin2, Cos1, Cos2;
  pole_sincos(W1 / sd, Sin1, Cos1);
  pole_sincos(W2 / sd, Sin2, Cos2);
  const double Exp1 = std::exp(L1 / sd), Exp2 = std::exp(L2 / sd);
  N0 = A1 + A2;
  N1 = Exp2 * (B2 * Sin2 - (A2 + 2 * A1) * Cos2);
  N1 += Exp1 * (B1 * Sin1 - (A1 + 2 * A2) * Cos1);
  N2 = (A1 + A2) * Cos2 * Cos1;
  N2 -= B1 * Cos2 * Sin1 + B2 * Cos1 * Sin2;
  N2 *= 2 * Exp1 * Exp2;
  N2 += A2 * Exp1 * Exp1 + A1 * Exp2 * Exp2;
  N3 = Exp2 * Exp1 * Exp1 * (B2 * Sin2 - A2 * Cos2);
  N3 += Exp1 * Exp2 * Exp2 * (B1 * Sin1 - A1 * Cos1);
  SN = N0 + N1 + N2 + N3;
  DN = N1 + 2 * N2 + 3 * N3;
  EN = N1 + 4 * N2 + 9 * N3;
}
int gauss_coeffs_order(double sigma, double spacing, int order, IirCoef *c) {
  if (order == 0) return gauss_coeffs(sigma, spacing, c);
  const double A1[3] = {1.3530, -0.6724, -1.3563}, B1[3] = {1.8151, -3.4327, 5.2318};
  const double A2[3] = {-0.3531, 0.6724, 0.3446}, B2[3] = {0.0902, 0.6100, -2.2355};
  const double W1 = 0.6681, L1 = -1.3932, W2 = 2.0787, L2 = -1.3732;
  if (order != 1 && order != 2) return -1;
  double direction = 1.0;
  if (spacing < 0.0) { direction = -1.0; spacing = -spacing; }
  if (spacing < 1e-8) return -1;
  const double sd = sigma / spacing;
  {
    double Sin1, Sin2, Cos1, Cos2;
    pole_sincos(W1 / sd, Sin1, Cos1);
    pole_sincos(W2 / sd, Sin2, Cos2);
    (void)Sin1; (void)Sin2;
    const double Exp1 = std::exp(L1 / sd), Exp2 = std::exp(L2 / sd);
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
  bool symmetric;
  if (order == 1) {
    double SN, DN, EN;
    n_coefficients(sd, A1[1], B1[1], W1, L1, A2[1], B2[1], W2, L2, c->N0, c->N1, c->N2, c->N3, SN, DN, EN);
    double alpha1 = 2 * (SN * DD - DN * SD) / (SD * SD);
    alpha1 *= direction;
    c->N0 *= 1.0 / alpha1; c->N1 *= 1.0 / alpha1; c->N2 *= 1.0 / alpha1; c->N3 *= 1.0 / alpha1;
    symmetric = false;
  } else {
    double N0_0, N1_0, N2_0, N3_0, N0_2, N1_2, N2_2, N3_2, SN0, DN0, EN0, SN2, DN2, EN2;
    n_coefficients(sd, A1[0], B1[0], W1, L1, A2[0], B2[0], W2, L2, N0_0, N1_0, N2_0, N3_0, SN0, DN0, EN0);
    n_coefficients(sd, A1[2], B1[2], W1, L1, A2[2], B2[2], W2, L2, N0_2, N1_2, N2_2, N3_2, SN2, DN2, EN2);
    const double beta = -(2 * SN2 - SD * N0_2) / (2 * SN0 - SD * N0_0);
    const double N0 = N0_2 + beta * N0_0, N1 = N1_2 + beta * N1_0;
    const double N2 = N2_2 + beta * N2_0, N3 = N3_2 + beta * N3_0;
    const double SN = SN2 + beta * SN0, DN = DN2 + beta * DN0, EN = EN2 + beta * EN0;
    const double alpha2 = (EN * SD * SD - ED * SN * SD - 2 * DN * DD * SD + 2 * DD * DD * SN) / (SD * SD * SD);
    c->N0 = N0 * (1.0 / alpha2); c->N1 = N1 * (1.0 / alpha2);
    c->N2 = N2 * (1.0 / alpha2); c->N3 = N3 * (1.0 / alpha2);
    symmetric = true;
  }
  if (symmetric) {
    c->M1 = c->N1 - c->D1 * c->N0; c->M2 = c->N2 - c->D2 * c->N0;
    c->M3 = c->N3 - c->D3 * c->N0; c->M4 = -c->D4 * c->N0;
  } else {
    c->M1 = -(c->N1 - c->D1 * c->N0); c->M2 = -(c->N2 - c->D2 * c->N0);
    c->M3 = -(c->N3 - c->D3 * c->N0); c->M4 = c->D4 * c->N0;
  }
  const double SN2 = c->N0 + c->N1 + c->N2 + c->N3;
  const double SM2 = c->M1 + c->M2 + c->M3 + c->M4;
  const double SD2 = 1.0 + c->D1 + c->D2 + c->D3 + c->D4;
  c->BN1 = c->D1 * SN2 / SD2; c->BN2 = c->D2 * SN2 / SD2;
  c->BN3 = c->D3 * SN2 / SD2; c->BN4 = c->D4 * SN2 / SD2;
  c->BM1 = c->D1 * SM2 / SD2; c->BM2 = c->D2 * SM2 / SD2;
  c->BM3 = c->D3 * SM2 / SD2; c->BM4 = c->D4 * SM2 / SD2;
  return 0;
}

// Constant lines (IFE_OPT_CONST_LINES): what does the recursive Gaussian of this axis make of a
// line whose n samples all have the value xin?  Simulated here in the arithmetic of the line
// kernels -- the same double operations in the same order (this translation unit is built with
// -ffp-contract=off on both sides), signs of zero included -- so the answer found here holds
// there.  True when every output has the same bit pattern, returned in *resp.
// (Restates FilterDataArray like iir_kernels.inc, for one line.)
bool const_response(const IirCoef &c, int64_t n, double xin, float *resp) {
  if (n < 4 || n > 65536) return false;
  std::vector<double> ca((size_t)n);
  double x1 = xin, x2 = xin, x3 = xin, y1 = xin, y2 = xin, y3 = xin, y4 = xin;
  for (int64_t i = 0; i < n; ++i) {
    const double d1 = i < 1 ? c.BN1 : c.D1, d2 = i < 2 ? c.BN2 : c.D2;
    const double d3 = i < 3 ? c.BN3 : c.D3, d4 = i < 4 ? c.BN4 : c.D4;
    const double a = x3 * c.N3 + (x2 * c.N2 + (x1 * c.N1 + xin * c.N0));
    const double t = y4 * d4 + (y3 * d3 + (y2 * d2 + y1 * d1));
    const double y = a - t;
    x3 = x2; x2 = x1; x1 = xin;
    y4 = y3; y3 = y2; y2 = y1; y1 = y;
    ca[(size_t)i] = y;
  }
  double u1 = xin, u2 = xin, u3 = xin, u4 = xin;
  x1 = x2 = x3 = xin;
  double x4 = xin;
  uint32_t first = 0;
  for (int64_t i = n - 1; i >= 0; --i) {
    const double d1 = i + 1 >= n ? c.BM1 : c.D1, d2 = i + 2 >= n ? c.BM2 : c.D2;
    const double d3 = i + 3 >= n ? c.BM3 : c.D3, d4 = i + 4 >= n ? c.BM4 : c.D4;
    const double a = x4 * c.M4 + (x3 * c.M3 + (x2 * c.M2 + x1 * c.M1));
    const double t = u4 * d4 + (u3 * d3 + (u2 * d2 + u1 * d1));
    const double y = a - t;
    x4 = x3; x3 = x2; x2 = x1; x1 = xin;
    u4 = u3; u3 = u2; u2 = u1; u1 = y;
    const float o = (float)(ca[(size_t)i] + y);
    uint32_t bits;
    memcpy(&bits, &o, 4);
    if (i == n - 1) first = bits;
    else if (bits != first) return false;
  }
  memcpy(resp, &first, 4);
  return true;
}
// IirJob::const_lines for one job: bit 0: a line of +0 comes out as +0; bit 1: a line of 1.0f as
// 1.0f; bit 2 / bit 3: a line of -0 comes out as -0 / as +0 (T * 0 is -0 wherever T < 0, so the
// exterior of a CT numerator is made of such lines).  Anything else is filtered.
uint32_t const_line_flags(const IirCoef &c, int64_t n) {
  uint32_t f = 0, bits;
  float r;
  if (const_response(c, n, 0.0, &r)) { memcpy(&bits, &r, 4); if (bits == 0u) f |= 1u; }
  if (const_response(c, n, 1.0, &r) && r == 1.0f) f |= 2u;
  if (const_response(c, n, -0.0, &r)) {
    memcpy(&bits, &r, 4);
    if (bits == 0x80000000u) f |= 4u;
    else if (bits == 0u) f |= 8u;
  }
  return f;
}

// [ITK-upstream] DerivativeOperator coefficients after FlipAxes + ScaleCoefficients:
// order 1 -> {-0.5, 0, 0.5} * s ; order 2 -> {1, -2, 1} * s  (s = 1/spacing, or
// 1/spacing^2 for order 2 with IFE_OPT_DSCALE_MODE=1).
DerivCoef deriv_coeffs(const ife_volume_desc *v, int dscale_mode) {
  DerivCoef d;
  const double sp[3] = {v->sx, v->sy, v->sz};
  for (int a = 0; a < 3; ++a) {
    const double s1 = 1.0 / sp[a];
    const double s2 = dscale_mode == 1 ? s1 * s1 : s1;
    d.m1[a] = -0.5 * s1;
    d.p1[a] = 0.5 * s1;
    d.a2[a] = 1.0 * s2;
    d.b2[a] = -2.0 * s2;
    d.c2[a] = 1.0 * s2;
  }
  return d;
}

IirGeom geom_for_axis(const ife_volume_desc *v, int axis) {
  IirGeom g;
  g.in_w = 0; g.in_group = 0; g.in_nz = 0;
  const int64_t nx = v->nx, ny = v->ny, nz = v->nz;
  if (axis == 2) {
    g.n = nz; g.nlines = nx * ny; g.sstride = nx * ny; g.inner = nx * ny; g.outer = 0;
  } else if (axis == 1) {
    g.n = ny; g.nlines = nx * nz; g.sstride = nx; g.inner = nx; g.outer = nx * ny;
  } else {
    g.n = nx; g.nlines = ny * nz; g.sstride = 1; g.inner = 1; g.outer = nx;
  }
  return g;
}

// checkpoint slots per line: one per register block covers both granularities
size_t ck_pairs(int64_t n, int K) { return (size_t)((n + K - 1) / K); }

int ensure_ck(ife_ctx *ctx, const ife_volume_desc *v, int njobs) {
  const int K = 8;  // sized for the smallest register block any axis may run with
  size_t need_y = 0, need_x = 0;
  for (int a = 0; a < 3; ++a) {
    IirGeom g = geom_for_axis(v, a);
    const size_t np = ck_pairs(g.n, K);
    need_y = std::max(need_y, np * 4 * (size_t)g.nlines * sizeof(double));
    need_x = std::max(need_x, np * 3 * (size_t)g.nlines * sizeof(float));
  }
  for (int j = 0; j < njobs; ++j) {
    int rc = ensure(ctx, ctx->ck_y[j], need_y);
    if (!rc) rc = ensure(ctx, ctx->ck_x[j], need_x);
    if (rc) return rc;
  }
  return IFE_OK;
}

int ensure_slots(ife_ctx *ctx, const ife_volume_desc *v, int nslots) {
  const size_t nb = (size_t)(v->nx * v->ny * v->nz) * sizeof(float);
  for (int s = 0; s < nslots; ++s)
    for (int i = 0; i < 4; ++i) {
      int rc = ensure(ctx, ctx->fld[s][i], nb);
      if (rc) return rc;
    }
  return IFE_OK;
}

// One launch of the line kernel along `axis` over njobs independent float volumes
// (jobs = numerator / denominator of up to three scales), each with its own sigma.
// `in2` (strided axes, pair checkpoints only): jobs are PAIRED -- in[j] numerator, in2[j]
// denominator, out[j] their quotient; job j uses checkpoint areas j and njobs + j.
int launch_iir(ife_ctx *ctx, const ife_volume_desc *v, int axis, int njobs,
               const float *const *in, float *const *out, const double *sigma,
               int in_y_chunks = 1, const int *order = nullptr /* per job: 0 (default), 1, 2 */,
               const float *const *in2 = nullptr) {
  if (njobs < 1 || njobs > IIR_MAX_JOBS) return fail(ctx, IFE_E_ARG, "bad job count %d", njobs);
  IirGeom g = geom_for_axis(v, axis);
  if (in_y_chunks > 1) {
    if (axis != 0) return fail(ctx, IFE_E_ARG, "Y-chunked input is only read by the x pass");
    if (v->ny % in_y_chunks || (v->ny / in_y_chunks) % 64)
      return fail(ctx, IFE_E_SIZE, "Y-chunked input needs ny/chunks to be a multiple of 64");
    g.in_w = in_y_chunks;
    g.in_group = v->ny / in_y_chunks;
    g.in_nz = v->nz;
  }
  // 32-bit offsets of the buffer accesses (iir_kernels.inc "addressing")
  if ((int64_t)2 * 16 * g.sstride * 4 >= (int64_t)1 << 31 ||
      g.outer * 4 >= (int64_t)1 << 32 || g.nlines * 8 * 3 >= (int64_t)1 << 32)
    return fail(ctx, IFE_E_SIZE, "volume too large for the 32-bit offsets of the line kernels");
  if (in2 && (axis == 0 || 2 * njobs > IIR_MAX_JOBS || ctx->iir_ckpt != 2))
    return fail(ctx, IFE_E_ARG, "the paired form runs on the strided axes with at most %d jobs", IIR_MAX_JOBS / 2);
  int rc = ensure_ck(ctx, v, in2 ? 2 * njobs : njobs);
  if (rc) return rc;
  const double sp = axis == 0 ? v->sx : axis == 1 ? v->sy : v->sz;
  IirJobs jobs;
  memset(&jobs, 0, sizeof jobs);
  for (int j = 0; j < njobs; ++j) {
    if (!in[j] || !out[j] || in[j] == out[j]) return fail(ctx, IFE_E_ARG, "bad job buffers");
    if ((reinterpret_cast<uintptr_t>(in[j]) | reinterpret_cast<uintptr_t>(out[j])) % 4)
      return fail(ctx, IFE_E_ARG, "field pointers must be aligned to 4 bytes");
    jobs.j[j].in = in[j];
    jobs.j[j].out = out[j];
    jobs.j[j].ck_y = (double *)ctx->ck_y[j].p;
    jobs.j[j].ck_x = (float *)ctx->ck_x[j].p;
    if (in2) {
      if (!in2[j] || in2[j] == out[j] || reinterpret_cast<uintptr_t>(in2[j]) % 4)
        return fail(ctx, IFE_E_ARG, "bad denominator buffer");
      jobs.j[j].in2 = in2[j];
      jobs.j[j].ck_y2 = (double *)ctx->ck_y[njobs + j].p;
    }
    if (gauss_coeffs_order(sigma[j], sp, order ? order[j] : 0, &jobs.j[j].c))
      return fail(ctx, IFE_E_ARG, "spacing is suspiciously small");
    // constant lines are copied where this filter provably maps the constant to a constant
    // (const_line_flags: simulated on the host for these coefficients and this line length,
    // remembered per context; zero-order only)
    jobs.j[j].const_lines = 0;
    if (ctx->const_lines && !ctx->iir_fma && (!order || order[j] == 0)) {
      const int64_t len = axis == 0 ? v->nx : axis == 1 ? v->ny : v->nz;
      const std::tuple<double, double, int64_t> key(sigma[j], sp, len);
      auto it = ctx->const_flags.find(key);
      if (it == ctx->const_flags.end()) {
        if (ctx->const_flags.size() > 256) ctx->const_flags.clear();
        it = ctx->const_flags.emplace(key, const_line_flags(jobs.j[j].c, len)).first;
      }
      jobs.j[j].const_lines = it->second;
    }
  }
  g.njobs = njobs;
  g.ngroups = (int32_t)((g.nlines + (in2 ? 127 : 255)) / (in2 ? 128 : 256));  // paired: 128 lines x 2 fields
  const dim3 grid((unsigned)((g.ngroups + 7) / 8 * 8 * njobs), 1, 1);  // job-fastest, padded
  ProfScope ps(ctx, axis == 2 ? KK_IIR_Z : axis == 1 ? KK_IIR_Y : KK_IIR_X);
  // register block of the strided axes: 12 for both (three waves per SIMD, 2.7 B of checkpoints
  // per sample).  Round 2 ran z with blocks of 10 for a fourth wave per SIMD; since the waits
  // are exact (iir_kernels.inc "Memory operations and waits") three waves hide the latency and
  // the fewer checkpoints win: z 1.85 -> 1.74 ms, y 1.87 with 12 against 2.06 with 16.
  const int sblock = ctx->iir_block ? ctx->iir_block : 12;
#define IFE_LAUNCH_IIR(NS)                                                                      \
  do {                                                                                          \
    if (axis == 0) {                                                                            \
      if (ctx->iir_block == 8)                                                                  \
        hipLaunchKernelGGL((NS::iir_contig_kernel<8>), grid, dim3(256), 0, ctx->stream, jobs, g); \
      else                                                                                      \
        hipLaunchKernelGGL((NS::iir_contig_kernel<16>), grid, dim3(256), 0, ctx->stream, jobs, g); \
    } else if (ctx->iir_ckpt == 1) {                                                            \
      if (ctx->iir_block == 8)                                                                  \
        hipLaunchKernelGGL((NS::iir_strided1_kernel<8>), grid, dim3(256), 0, ctx->stream, jobs, g); \
      else                                                                                      \
        hipLaunchKernelGGL((NS::iir_strided1_kernel<16>), grid, dim3(256), 0, ctx->stream, jobs, g); \
    } else {                                                                                    \
      if (sblock == 8)                                                                          \
        hipLaunchKernelGGL((NS::iir_strided_kernel<8>), grid, dim3(256), 0, ctx->stream, jobs, g); \
      else if (sblock == 10)                                                                    \
        hipLaunchKernelGGL((NS::iir_strided_kernel<10>), grid, dim3(256), 0, ctx->stream, jobs, g); \
      else if (sblock == 12)                                                                    \
        hipLaunchKernelGGL((NS::iir_strided_kernel<12>), grid, dim3(256), 0, ctx->stream, jobs, g); \
      else                                                                                      \
        hipLaunchKernelGGL((NS::iir_strided_kernel<16>), grid, dim3(256), 0, ctx->stream, jobs, g); \
    }                                                                                           \
  } while (0)
  if (in2) {
#define IFE_LAUNCH_PAIRED(NS)                                                                          \
  do {                                                                                                 \
    if (sblock == 8)                                                                                   \
      hipLaunchKernelGGL((NS::iir_strided_kernel<8, true>), grid, dim3(256), 0, ctx->stream, jobs, g);  \
    else if (sblock == 10)                                                                             \
      hipLaunchKernelGGL((NS::iir_strided_kernel<10, true>), grid, dim3(256), 0, ctx->stream, jobs, g); \
    else if (sblock == 12)                                                                             \
      hipLaunchKernelGGL((NS::iir_strided_kernel<12, true>), grid, dim3(256), 0, ctx->stream, jobs, g); \
    else                                                                                               \
      hipLaunchKernelGGL((NS::iir_strided_kernel<16, true>), grid, dim3(256), 0, ctx->stream, jobs, g); \
  } while (0)
    if (ctx->iir_fma) IFE_LAUNCH_PAIRED(iir_fma);
    else IFE_LAUNCH_PAIRED(iir_exact);
#undef IFE_LAUNCH_PAIRED
  } else if (ctx->iir_fma) IFE_LAUNCH_IIR(iir_fma);
  else IFE_LAUNCH_IIR(iir_exact);
#undef IFE_LAUNCH_IIR
  IFE_HIP(ctx, hipGetLastError());
  return IFE_OK;
}

// ---- Z pass of a Z-slab (iir_types.hpp "ZSlabJob") -----------------------------------
#ifndef IFE_ZSLAB_K
#define IFE_ZSLAB_K 12
#endif
constexpr int ZSLAB_K = IFE_ZSLAB_K;  // register block of the slab kernels (fixes the checkpoint layout)
int64_t zslab_pairs(int64_t n) { return ((n + ZSLAB_K - 1) / ZSLAB_K + 1) / 2; }
size_t zslab_ck_bytes(const ife_volume_desc *v) {
  const size_t L = (size_t)(v->nx * v->ny);
  return (size_t)zslab_pairs(v->nz) * 4 * L * 2 * sizeof(double);
}
// phase 0: causal sweep, 1: anticausal sweep, 2: combine, 3 / 4: fused with the causal /
// anticausal recursion carried through the slab (zslab_fused_kernel)
int launch_zslab(ife_ctx *ctx, int phase, int njobs, const float *const *in, float *const *out,
                 const ife_volume_desc *v, int64_t line0, int64_t nlines, const double *sigmas,
                 int has_lo, int has_hi, const void *state_in, void *state_out, void *const *ck) {
  if (njobs < 1 || njobs > IIR_MAX_JOBS) return fail(ctx, IFE_E_ARG, "bad job count %d", njobs);
  const int64_t L = v->nx * v->ny, n = v->nz;
  if (line0 < 0 || nlines < 1 || line0 + nlines > L)
    return fail(ctx, IFE_E_ARG, "line range [%lld, %lld) outside the %lld lines of the slab",
                (long long)line0, (long long)(line0 + nlines), (long long)L);
  if (n < 4) return fail(ctx, IFE_E_SIZE, "a Z-slab needs at least 4 planes (got %lld)", (long long)n);
  // 32-bit offsets of the buffer accesses (iir_kernels.inc "addressing")
  if ((int64_t)2 * ZSLAB_K * L * 4 >= (int64_t)1 << 31 || L * 8 * 3 >= (int64_t)1 << 32)
    return fail(ctx, IFE_E_SIZE, "slab too large for the 32-bit offsets of the line kernels");
  if (!in || !sigmas || !ck) return fail(ctx, IFE_E_ARG, "null pointer");
  const bool need_in = (phase == 0 || phase == 3) ? has_lo != 0 : (phase == 1 || phase == 4) ? has_hi != 0 : false;
  if ((phase != 2 && !state_out) || (need_in && !state_in)) return fail(ctx, IFE_E_ARG, "null state buffer");
  if ((reinterpret_cast<uintptr_t>(state_in) | reinterpret_cast<uintptr_t>(state_out)) % 8)
    return fail(ctx, IFE_E_ARG, "state buffers must be aligned to 8 bytes");
  const int64_t np = zslab_pairs(n);
  ZSlabJobs jobs;
  memset(&jobs, 0, sizeof jobs);
  for (int j = 0; j < njobs; ++j) {
    ZSlabJob &J = jobs.j[j];
    if (!in[j] || !ck[j] || (phase >= 2 && (!out || !out[j] || out[j] == in[j])))
      return fail(ctx, IFE_E_ARG, "bad job buffers");
    if (reinterpret_cast<uintptr_t>(ck[j]) % 8 || reinterpret_cast<uintptr_t>(in[j]) % 4)
      return fail(ctx, IFE_E_ARG, "job buffers are misaligned");
    J.in = in[j] + line0;
    J.out = phase >= 2 ? out[j] + line0 : nullptr;
    double *cy = (double *)ck[j];
    double *ay = cy + np * 4 * L;
    J.cy = cy + line0; J.ay = ay + line0;
    const size_t rec = (size_t)nlines * IFE_Z_STATE_BYTES;  // [4][nlines] doubles per job
    if (state_in) J.sin_y = (const double *)((const char *)state_in + (size_t)j * rec);
    if (state_out) J.sout_y = (double *)((char *)state_out + (size_t)j * rec);
    if (!(sigmas[j] > 0.0) || gauss_coeffs(sigmas[j], v->sz, &J.c))
      return fail(ctx, IFE_E_ARG, "bad sigma or spacing");
  }
  ZSlabGeom g;
  g.n = n; g.nlines = nlines; g.sstride = L; g.ck_stride = L;
  g.has_lo = has_lo ? 1 : 0; g.has_hi = has_hi ? 1 : 0;
  g.njobs = njobs;
  g.ngroups = (int32_t)((nlines + 255) / 256);
  const dim3 grid((unsigned)((g.ngroups + 7) / 8 * 8 * njobs), 1, 1);
  ProfScope ps(ctx, phase >= 2 ? KK_ZSLAB_COMBINE : KK_ZSLAB_SWEEP);
#define IFE_LAUNCH_ZSLAB(NS)                                                                        \
  do {                                                                                              \
    if (phase == 0)                                                                                 \
      hipLaunchKernelGGL((NS::zslab_causal_kernel<ZSLAB_K>), grid, dim3(256), 0, ctx->stream, jobs, g); \
    else if (phase == 1)                                                                            \
      hipLaunchKernelGGL((NS::zslab_anti_kernel<ZSLAB_K>), grid, dim3(256), 0, ctx->stream, jobs, g);   \
    else if (phase == 3)                                                                            \
      hipLaunchKernelGGL((NS::zslab_fused_kernel<ZSLAB_K, 0>), grid, dim3(256), 0, ctx->stream, jobs, g); \
    else if (phase == 4)                                                                            \
      hipLaunchKernelGGL((NS::zslab_fused_kernel<ZSLAB_K, 1>), grid, dim3(256), 0, ctx->stream, jobs, g); \
    else                                                                                            \
      hipLaunchKernelGGL((NS::zslab_combine_kernel<ZSLAB_K>), grid, dim3(256), 0, ctx->stream, jobs, g); \
  } while (0)
  if (ctx->iir_fma) IFE_LAUNCH_ZSLAB(iir_fma);
  else IFE_LAUNCH_ZSLAB(iir_exact);
#undef IFE_LAUNCH_ZSLAB
  IFE_HIP(ctx, hipGetLastError());
  return IFE_OK;
}

template <typename TI, typename TM>
int launch_prep(ife_ctx *ctx, const TI *img, const TM *msk, float *tc, float *cf, int64_t n,
                PrepGeom pg = PrepGeom{0, 0, 0, 0, 0}) {
  ProfScope ps(ctx, KK_PREP);
  const uintptr_t al = reinterpret_cast<uintptr_t>(img) % (4 * sizeof(TI)) |
                       reinterpret_cast<uintptr_t>(msk) % (4 * sizeof(TM)) |
                       reinterpret_cast<uintptr_t>(tc) % 16 | reinterpret_cast<uintptr_t>(cf) % 16;
  const bool vec = al == 0 && (pg.chunks <= 1 || pg.nx % 4 == 0);
  const int64_t n4 = vec ? n / 4 : 0;
  if (n4 > 0) {
    // one 16-byte piece per thread (the kernel's loop runs once): on MI355X a stream moves
    // faster as many short-lived workgroups than as a grid-stride loop of a few thousand
    // (scripts/experiments/stream_probe.hip: fill 6.9 vs 5.0 TB/s, copy 6.4 vs 4.9)
    const unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 0x7fffffff);
    hipLaunchKernelGGL((prep_kernel_vec4<TI, TM>), dim3(blocks), dim3(256), 0, ctx->stream, img,
                       msk, tc, cf, n4, pg);
    IFE_HIP(ctx, hipGetLastError());
  }
  if (n4 * 4 < n) {
    const int64_t rest = n - n4 * 4;
    const unsigned blocks = (unsigned)std::min<int64_t>((rest + 255) / 256, 8192);
    hipLaunchKernelGGL((prep_kernel_scalar<TI, TM>), dim3(blocks), dim3(256), 0, ctx->stream,
                       img, msk, tc, cf, n4 * 4, n, pg);
    IFE_HIP(ctx, hipGetLastError());
  }
  return IFE_OK;
}

// SmoothingRecursiveGaussianImageFilter for a group of up to IFE_MAX_SLOTS scales at
// once: Z pass from the shared sources (image*certainty, certainty), then X, then Y, each
// axis as ONE launch over all (scale, field) jobs.  Scale k of the group ends in slot k:
// numerator in fld[k][0], denominator in fld[k][2].
int smooth_group(ife_ctx *ctx, const float *src_num, const float *src_den,
                 const ife_volume_desc *v, const double *sigmas, int nscales) {
  const int nf = src_den ? 2 : 1;
  const float *in[IIR_MAX_JOBS];
  float *out[IIR_MAX_JOBS];
  double sg[IIR_MAX_JOBS];
  int nj = 0;
  for (int k = 0; k < nscales; ++k)
    for (int f = 0; f < nf; ++f) sg[nj++] = sigmas[k];
  int rc = ensure_slots(ctx, v, nscales);
  if (rc) return rc;
  auto fill = [&](int src_idx, int dst_idx, bool from_source) {
    int j = 0;
    for (int k = 0; k < nscales; ++k)
      for (int f = 0; f < nf; ++f, ++j) {
        in[j] = from_source ? (f == 0 ? src_num : src_den)
                            : (const float *)ctx->fld[k][2 * f + src_idx].p;
        out[j] = (float *)ctx->fld[k][2 * f + dst_idx].p;
      }
  };
  fill(0, 0, true);                                    // Z: sources -> ping
  rc = launch_iir(ctx, v, 2, nj, in, out, sg);
  fill(0, 1, false);                                   // X: ping -> pong
  if (!rc) rc = launch_iir(ctx, v, 0, nj, in, out, sg);
  if (nf == 2 && ctx->fused_divide && ctx->iir_ckpt == 2) {
    // Y: numerator and denominator in sibling waves, the quotient S goes to the numerator's slot
    const float *n1[IIR_MAX_JOBS], *d1[IIR_MAX_JOBS];
    float *o1[IIR_MAX_JOBS];
    double s1[IIR_MAX_JOBS];
    for (int k = 0; k < nscales; ++k) {
      n1[k] = (const float *)ctx->fld[k][1].p;
      d1[k] = (const float *)ctx->fld[k][3].p;
      o1[k] = (float *)ctx->fld[k][0].p;
      s1[k] = sigmas[k];
    }
    if (!rc) rc = launch_iir(ctx, v, 1, nscales, n1, o1, s1, 1, nullptr, d1);
    return rc;
  }
  fill(1, 0, false);                                   // Y: pong -> ping
  if (!rc) rc = launch_iir(ctx, v, 1, nj, in, out, sg);
  return rc;
}
// After smooth_group: is slot k's fld[k][0] already the quotient S (denominator folded in)?
bool slots_hold_quotient(const ife_ctx *ctx, bool has_den) {
  return has_den && ctx->fused_divide && ctx->iir_ckpt == 2;
}

template <int MODE, typename VAL, typename TM>
int launch_features(ife_ctx *ctx, VAL val, const TM *mask, float *out,
                    const ife_volume_desc *v, int layout, int halo_lo = 0, int halo_hi = 0,
                    const uint32_t *seg_base = nullptr, int64_t col_stride = 0,
                    int64_t col_offset = 0) {
  if constexpr (std::is_same<VAL, ValSmooth>::value) {
    // no denominator (certainty identically one): the single-field source, a kernel without
    // the run-time choice (measured 1.39 -> 1.23 ms per launch at 512^3)
    if (val.den == nullptr)
      return launch_features<MODE>(ctx, ValS{val.num}, mask, out, v, layout, halo_lo, halo_hi,
                                   seg_base, col_stride, col_offset);
  }
  FeatGeom g;
  g.seg_base = seg_base;
  g.col_offset = col_offset;
  g.nx = (int)v->nx; g.ny = (int)v->ny; g.nz = (int)v->nz;
  g.zchunk = ctx->zchunk;
  g.plane = v->nx * v->ny;
  g.nvox = MODE == FEAT_SAMPLES8 ? col_stride : g.plane * v->nz;
  g.zoff = halo_lo ? 1 : 0;
  g.zc_hi = (int)v->nz + g.zoff + (halo_hi ? 1 : 0) - 1;
  const DerivCoef dc = deriv_coeffs(v, ctx->dscale_mode);
  g.gx = (int)((v->nx + FT_TX - 1) / FT_TX);
  g.gy = (int)((v->ny + FT_TY - 1) / FT_TY);
  g.gz = (int)((v->nz + g.zchunk - 1) / g.zchunk);
  const int64_t nblocks = (int64_t)g.gx * g.gy * g.gz;
  if (nblocks > 0x7fffffff) return fail(ctx, IFE_E_SIZE, "volume too large for the feature kernel grid");
  {  // vector stores: a misaligned pointer would fault on the device, so it is refused here
    constexpr int nout = FeatNOut<MODE>::value;
    const uintptr_t need = layout == IFE_PLANAR ? 4 : (nout == 8 ? 16 : nout == 6 ? 8 : 4);
    if (reinterpret_cast<uintptr_t>(out) % need)
      return fail(ctx, IFE_E_ARG, "output pointer must be aligned to %d bytes for this layout", (int)need);
  }
  dim3 grid((unsigned)nblocks, 1, 1);
  ProfScope ps(ctx, KK_FEATURES);
  const bool unit = v->sx == 1.0 && v->sy == 1.0 && v->sz == 1.0;
  const int planar = layout == IFE_PLANAR ? 1 : 0;
  constexpr bool has_eig = MODE == FEAT_FEATURES8 || MODE == FEAT_EIG6 || MODE == FEAT_SAMPLES8;
  // ring form (feature_kernels.hpp): one float field, narrow mask, dword-aligned mask planes,
  // in-plane byte offsets that fit 32 bits
  constexpr bool ring_types = (std::is_same<VAL, ValS>::value || std::is_same<VAL, ValRaw<float>>::value) &&
                              sizeof(TM) <= 2;
  bool ring = false;
  const float *ring_src = nullptr;
  if constexpr (ring_types) {
    if constexpr (std::is_same<VAL, ValS>::value) ring_src = val.s;
    else ring_src = val.img;
    const int64_t row_bytes = v->nx * (int64_t)sizeof(TM), plane_bytes = g.plane * (int64_t)sizeof(TM);
    ring = ctx->feat_ring && g.plane < ((int64_t)1 << 30) && reinterpret_cast<uintptr_t>(ring_src) % 4 == 0 &&
           (mask == nullptr ||
            (row_bytes % 4 == 0 && plane_bytes % 4 == 0 && reinterpret_cast<uintptr_t>(mask) % 4 == 0));
  }
#define IFE_LAUNCH_FEAT2(UNIT_, TRIG_, PL_)                                                    \
  do {                                                                                         \
    if constexpr (ring_types) {                                                                \
      if (ring) {                                                                              \
        hipLaunchKernelGGL((features_ring_kernel<MODE, UNIT_, TRIG_, PL_, TM>), grid,          \
                           dim3(FT_THREADS), 0, ctx->stream, ring_src, mask, out, g, dc);      \
        break;                                                                                 \
      }                                                                                        \
    }                                                                                          \
    hipLaunchKernelGGL((features_kernel<MODE, UNIT_, TRIG_, PL_, VAL, TM>), grid,              \
                       dim3(FT_THREADS), 0, ctx->stream, val, mask, out, g, dc);               \
  } while (0)
#define IFE_LAUNCH_FEAT(UNIT_, TRIG_)                                                  \
  do {                                                                                 \
    if constexpr (MODE == FEAT_SAMPLES8) IFE_LAUNCH_FEAT2(UNIT_, TRIG_, true);         \
    else if (planar) IFE_LAUNCH_FEAT2(UNIT_, TRIG_, true);                             \
    else IFE_LAUNCH_FEAT2(UNIT_, TRIG_, false);                                        \
  } while (0)
  if (has_eig && ctx->trig_mode == 1) {
    if constexpr (has_eig) {
      if (unit) IFE_LAUNCH_FEAT(true, 1);
      else IFE_LAUNCH_FEAT(false, 1);
    }
  } else if (has_eig && ctx->trig_mode == 2) {
    if constexpr (has_eig) {
      if (unit) IFE_LAUNCH_FEAT(true, 2);
      else IFE_LAUNCH_FEAT(false, 2);
    }
  } else {
    if (unit) IFE_LAUNCH_FEAT(true, 0);
    else IFE_LAUNCH_FEAT(false, 0);
  }
#undef IFE_LAUNCH_FEAT2
#undef IFE_LAUNCH_FEAT
  IFE_HIP(ctx, hipGetLastError());
  return IFE_OK;
}

size_t dtype_size(int dt) {
  switch (dt) {
    case IFE_F32: return 4;
    case IFE_I16: return 2;
    case IFE_U8: return 1;
    case IFE_U16: return 2;
  }
  return 0;
}

int check_layout_mem(ife_ctx *ctx, int layout, int mem) {
  if (layout != IFE_INTERLEAVED && layout != IFE_PLANAR)
    return fail(ctx, IFE_E_ARG, "bad layout %d", layout);
  if (mem != IFE_MEM_HOST && mem != IFE_MEM_DEVICE) return fail(ctx, IFE_E_ARG, "bad mem %d", mem);
  return IFE_OK;
}

// Stage a host input on the device (HOST mode) or pass the device pointer through.
int stage_in(ife_ctx *ctx, int mem, const void *src, size_t bytes, DevBuf &buf, const void **dev) {
  if (src == nullptr) { *dev = nullptr; return IFE_OK; }
  if (mem == IFE_MEM_DEVICE) { *dev = src; return IFE_OK; }
  int rc = ensure(ctx, buf, bytes);
  if (rc) return rc;
  IFE_HIP(ctx, hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  *dev = buf.p;
  return IFE_OK;
}

int stage_out_begin(ife_ctx *ctx, int mem, void *dst, size_t bytes, void **dev) {
  if (mem == IFE_MEM_DEVICE) { *dev = dst; return IFE_OK; }
  int rc = ensure(ctx, ctx->st_out, bytes);
  if (rc) return rc;
  *dev = ctx->st_out.p;
  return IFE_OK;
}

int stage_out_end(ife_ctx *ctx, int mem, void *dst, size_t bytes) {
  if (mem == IFE_MEM_DEVICE) return IFE_OK;
  IFE_HIP(ctx, hipMemcpyAsync(dst, ctx->st_out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  IFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IFE_OK;
}

int bind(ife_ctx *ctx) {
  if (!ctx) return IFE_E_ARG;
  IFE_HIP(ctx, hipSetDevice(ctx->device));
  // hipGetLastError() after a launch must report THAT launch: drop whatever an earlier call of
  // this thread (ours or another library's) left pending
  (void)hipGetLastError();
  return IFE_OK;
}

}  // namespace

// Row f1: instead of a feature volume per scale, the features of the sampled voxels go
// straight into eight sample columns per scale (stats_capi.inc).
struct SampleSink {
  const uint8_t *code;      // per voxel: bit 0 = sample here, bit 1 = label non-zero
  const uint32_t *seg_base; // exclusive scan of the per-row-segment sample counts
  float *columns;           // column (scale*8 + k) at columns + (scale*8 + k) * stride
  int64_t stride, offset;   // elements between columns, samples already in them
};

template <typename TI, typename TM>
static int emphysema_typed(ife_ctx *ctx, const TI *img, const TM *msk, const ife_volume_desc *vol,
                           const float *sigmas, int n_sigmas, float *dout, int layout,
                           const SampleSink *sink = nullptr) {
  const size_t n = (size_t)(vol->nx * vol->ny * vol->nz);
  float *tc = (float *)ctx->pre[0].p, *cf = (float *)ctx->pre[1].p;
  // Cast + Multiply once for all scales (the reference redoes them per scale, a9).
  // mask == NULL: certainty == 1 everywhere, image*1 is the image and G(1) is exactly
  // 1.0f at every voxel (DESIGN.md "all-ones certainty"), so the denominator is skipped.
  const float *src_num;
  if (msk == nullptr && std::is_same<TI, float>::value) {
    src_num = reinterpret_cast<const float *>(img);
  } else {
    int rc = launch_prep<TI, TM>(ctx, img, msk, tc, msk ? cf : nullptr, (int64_t)n);
    if (rc) return rc;
    src_num = tc;
  }
  for (int s0 = 0; s0 < n_sigmas; s0 += IFE_MAX_SLOTS) {
    const int ns = std::min(IFE_MAX_SLOTS, n_sigmas - s0);
    double sg[IFE_MAX_SLOTS];
    for (int k = 0; k < ns; ++k) sg[k] = (double)sigmas[s0 + k];
    int rc = smooth_group(ctx, src_num, msk ? cf : nullptr, vol, sg, ns);
    for (int k = 0; k < ns && !rc; ++k) {
      const bool q = slots_hold_quotient(ctx, msk != nullptr);
      const ValSmooth vs{(const float *)ctx->fld[k][0].p, msk && !q ? (const float *)ctx->fld[k][2].p : nullptr};
      if (sink)
        rc = launch_features<FEAT_SAMPLES8>(ctx, vs, sink->code,
                                            sink->columns + (size_t)(s0 + k) * IFE_NUM_FEATURES * sink->stride,
                                            vol, IFE_PLANAR, 0, 0, sink->seg_base, sink->stride, sink->offset);
      else
        rc = launch_features<FEAT_FEATURES8>(ctx, vs, msk, dout + (size_t)(s0 + k) * n * IFE_NUM_FEATURES, vol,
                                             layout);
      if (!rc && ctx->scale_events) {
        hipEvent_t e;
        IFE_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->scale_events->push_back(e);
        IFE_HIP(ctx, hipEventRecord(e, ctx->stream));
      }
    }
    if (rc) return rc;
  }
  return IFE_OK;
}

// =====================================================================================
extern "C" {

int ife_abi_version(void) { return IFE_ABI_VERSION; }

int ife_ctx_create(int device, ife_ctx **out) {
  if (!out) return fail(nullptr, IFE_E_ARG, "ctx out pointer is null");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, IFE_E_HIP, "no HIP device available (%s); this library has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= count)
    return fail(nullptr, IFE_E_ARG, "device %d out of range (0..%d)", device, count - 1);
  hipDeviceProp_t prop;
  IFE_HIP(nullptr, hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, IFE_E_HIP, "device %d is %s; this library is built for gfx950 only",
                device, prop.gcnArchName);
  IFE_HIP(nullptr, hipSetDevice(device));
  ife_ctx *c = new (std::nothrow) ife_ctx();
  if (!c) return fail(nullptr, IFE_E_NOMEM, "host allocation failed");
  c->device = device;
  // IFE_TRIG_MODE in the environment sets the initial IFE_OPT_TRIG_MODE, for callers that
  // cannot reach ife_ctx_set_option (the drop-in tools); set_option still overrides it.
  if (const char *e = getenv("IFE_TRIG_MODE"))
    if ((e[0] == '0' || e[0] == '1' || e[0] == '2') && e[1] == 0) c->trig_mode = e[0] - '0';
  *out = c;
  return IFE_OK;
}

void ife_ctx_destroy(ife_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto e : ctx->sc_done) (void)hipEventDestroy(e);
  if (ctx->sc_copy) (void)hipStreamDestroy(ctx->sc_copy);
  std::vector<DevBuf *> bufs = {&ctx->pre[0], &ctx->pre[1], &ctx->st_img, &ctx->st_mask,
                                &ctx->st_aux, &ctx->st_out, &ctx->sc_out};
  for (auto &sl : ctx->fld)
    for (auto &b : sl) bufs.push_back(&b);
  for (auto &b : ctx->ck_y) bufs.push_back(&b);
  for (auto &b : ctx->ck_x) bufs.push_back(&b);
  for (DevBuf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (auto &r : ctx->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  delete ctx;
}

const char *ife_last_error(const ife_ctx *ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int ife_ctx_set_stream(ife_ctx *ctx, void *hip_stream) {
  if (!ctx) return IFE_E_ARG;
  ctx->stream = (hipStream_t)hip_stream;
  return IFE_OK;
}

int ife_ctx_set_option(ife_ctx *ctx, int option, int value) {
  if (!ctx) return IFE_E_ARG;
  switch (option) {
    case IFE_OPT_TRIG_MODE:
      if (value < 0 || value > 2) return fail(ctx, IFE_E_ARG, "trig mode must be 0, 1 or 2");
      ctx->trig_mode = value;
      return IFE_OK;
    case IFE_OPT_DSCALE_MODE:
      if (value != 0 && value != 1) return fail(ctx, IFE_E_ARG, "dscale mode must be 0 or 1");
      ctx->dscale_mode = value;
      return IFE_OK;
    case IFE_OPT_PROFILE:
      ctx->profile = value ? 1 : 0;
      return IFE_OK;
    case IFE_OPT_ZCHUNK:
      if (value < 1) return fail(ctx, IFE_E_ARG, "zchunk must be >= 1");
      ctx->zchunk = value;
      return IFE_OK;
    case IFE_OPT_IIR_FMA:
      ctx->iir_fma = value ? 1 : 0;
      return IFE_OK;
    case IFE_OPT_IIR_CKPT:
      if (value != 1 && value != 2) return fail(ctx, IFE_E_ARG, "iir checkpoint stride must be 1 or 2");
      ctx->iir_ckpt = value;
      return IFE_OK;
    case IFE_OPT_CONST_LINES:
      ctx->const_lines = value ? 1 : 0;
      return IFE_OK;
    case IFE_OPT_FUSED_DIVIDE:
      ctx->fused_divide = value ? 1 : 0;
      return IFE_OK;
    case IFE_OPT_FEAT_RING:
      ctx->feat_ring = value ? 1 : 0;
      return IFE_OK;
    case IFE_OPT_IIR_BLOCK:
      if (value != 0 && value != 8 && value != 10 && value != 12 && value != 16)
        return fail(ctx, IFE_E_ARG, "iir block must be 0 (per axis), 8, 10, 12 (strided axes only; x keeps 16) or 16");
      ctx->iir_block = value;
      return IFE_OK;
  }
  return fail(ctx, IFE_E_ARG, "unknown option %d", option);
}

int ife_ctx_reserve(ife_ctx *ctx, const ife_volume_desc *vol) {
  int rc = bind(ctx);
  if (rc) return rc;
  rc = check_vol(ctx, vol, false);
  if (rc) return rc;
  const size_t nb = (size_t)(vol->nx * vol->ny * vol->nz) * sizeof(float);
  rc = ensure_slots(ctx, vol, IFE_MAX_SLOTS);
  for (int i = 0; i < 2 && !rc; ++i) rc = ensure(ctx, ctx->pre[i], nb);
  if (!rc) rc = ensure_ck(ctx, vol, 2 * IFE_MAX_SLOTS);
  return rc;
}

int ife_ctx_synchronize(ife_ctx *ctx) {
  int rc = bind(ctx);
  if (rc) return rc;
  IFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return IFE_OK;
}

// ---- a1 / a2 ------------------------------------------------------------------------
static int eig_batch(ife_ctx *ctx, const float *A6, int64_t n, float *outv, int mem, int nout) {
  int rc = bind(ctx);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!A6 || !outv))) return fail(ctx, IFE_E_ARG, "null pointer");
  if (mem != IFE_MEM_HOST && mem != IFE_MEM_DEVICE) return fail(ctx, IFE_E_ARG, "bad mem");
  if (n == 0) return IFE_OK;
  const void *dA;
  void *dO;
  rc = stage_in(ctx, mem, A6, (size_t)n * 6 * 4, ctx->st_img, &dA);
  if (!rc) rc = stage_out_begin(ctx, mem, outv, (size_t)n * nout * 4, &dO);
  if (rc) return rc;
  {
    ProfScope ps(ctx, KK_EIG_BATCH);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (nout == 3)
      hipLaunchKernelGGL((eig_batch_kernel<3>), dim3(blocks), dim3(256), 0, ctx->stream,
                         (const float *)dA, (float *)dO, n, ctx->trig_mode);
    else
      hipLaunchKernelGGL((eig_batch_kernel<6>), dim3(blocks), dim3(256), 0, ctx->stream,
                         (const float *)dA, (float *)dO, n, ctx->trig_mode);
    IFE_HIP(ctx, hipGetLastError());
  }
  return stage_out_end(ctx, mem, outv, (size_t)n * nout * 4);
}

int ife_eigenvalues(ife_ctx *ctx, const float *A6, int64_t n, float *ev3, int mem) {
  return eig_batch(ctx, A6, n, ev3, mem, 3);
}
int ife_eigenvalue_features(ife_ctx *ctx, const float *A6, int64_t n, float *f6, int mem) {
  return eig_batch(ctx, A6, n, f6, mem, 6);
}

// ---- a3 -----------------------------------------------------------------------------
int ife_hessian3d(ife_ctx *ctx, const float *image, const ife_volume_desc *vol, float *out6,
                  int layout, int mem) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, vol, false))) return rc;
  if ((rc = check_layout_mem(ctx, layout, mem))) return rc;
  if (!image || !out6) return fail(ctx, IFE_E_ARG, "null pointer");
  const size_t n = (size_t)(vol->nx * vol->ny * vol->nz);
  const void *dI;
  void *dO;
  if ((rc = stage_in(ctx, mem, image, n * 4, ctx->st_img, &dI))) return rc;
  if ((rc = stage_out_begin(ctx, mem, out6, n * 24, &dO))) return rc;
  rc = launch_features<FEAT_HESSIAN6>(ctx, ValRaw<float>{(const float *)dI},
                                      (const uint8_t *)nullptr, (float *)dO, vol, layout);
  if (rc) return rc;
  return stage_out_end(ctx, mem, out6, n * 24);
}

int ife_gradient_magnitude(ife_ctx *ctx, const float *image, const ife_volume_desc *vol,
                           float *out, int mem) {
  return ife_fd_gradient_features(ctx, image, nullptr, vol, out, mem);
}

// ---- a4 -----------------------------------------------------------------------------
int ife_normalized_gaussian_convolution(ife_ctx *ctx, const float *image,
                                        const float *certainty, const ife_volume_desc *vol,
                                        double sigma, float *out, int mem) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, vol, true))) return rc;
  if (mem != IFE_MEM_HOST && mem != IFE_MEM_DEVICE) return fail(ctx, IFE_E_ARG, "bad mem");
  if (!image || !certainty || !out) return fail(ctx, IFE_E_ARG, "null pointer");
  if (!(sigma > 0.0)) return fail(ctx, IFE_E_ARG, "sigma must be positive");
  if ((rc = ife_ctx_reserve(ctx, vol))) return rc;
  const size_t n = (size_t)(vol->nx * vol->ny * vol->nz);
  const void *dI, *dC;
  void *dO;
  if ((rc = stage_in(ctx, mem, image, n * 4, ctx->st_img, &dI))) return rc;
  if ((rc = stage_in(ctx, mem, certainty, n * 4, ctx->st_aux, &dC))) return rc;
  if ((rc = stage_out_begin(ctx, mem, out, n * 4, &dO))) return rc;
  float *tc = (float *)ctx->pre[0].p;
  rc = launch_prep<float, float>(ctx, (const float *)dI, (const float *)dC, tc, nullptr,
                                 (int64_t)n);
  const double sg = sigma;
  if (!rc) rc = smooth_group(ctx, tc, (const float *)dC, vol, &sg, 1);
  if (rc) return rc;
  const float *num = (const float *)ctx->fld[0][0].p, *den = (const float *)ctx->fld[0][2].p;
  if (slots_hold_quotient(ctx, true)) {  // the last axis pass has divided already
    IFE_HIP(ctx, hipMemcpyAsync(dO, num, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    ProfScope ps(ctx, KK_DIVIDE);
    hipLaunchKernelGGL(divide_kernel, dim3(2048), dim3(256), 0, ctx->stream, num, den,
                       (float *)dO, (int64_t)n);
    IFE_HIP(ctx, hipGetLastError());
  }
  return stage_out_end(ctx, mem, out, n * 4);
}

// ---- f4: differential normalized convolution ------------------------------------------
int ife_differential_normalized_convolution(ife_ctx *ctx, const float *image,
                                            const float *certainty, const ife_volume_desc *vol,
                                            double sigma, int axis, float *out, int mem) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, vol, true))) return rc;
  if (mem != IFE_MEM_HOST && mem != IFE_MEM_DEVICE) return fail(ctx, IFE_E_ARG, "bad mem");
  if (!image || !certainty || !out) return fail(ctx, IFE_E_ARG, "null pointer");
  if (!(sigma > 0.0)) return fail(ctx, IFE_E_ARG, "sigma must be positive");
  if (axis < 0 || axis > 2) return fail(ctx, IFE_E_ARG, "axis must be 0 (x), 1 (y) or 2 (z)");
  if ((rc = ensure_slots(ctx, vol, 2))) return rc;
  const size_t n = (size_t)(vol->nx * vol->ny * vol->nz);
  const void *dI, *dC;
  void *dO;
  if ((rc = stage_in(ctx, mem, image, n * 4, ctx->st_img, &dI))) return rc;
  if ((rc = stage_in(ctx, mem, certainty, n * 4, ctx->st_aux, &dC))) return rc;
  if ((rc = stage_out_begin(ctx, mem, out, n * 4, &dO))) return rc;
  if ((rc = ensure(ctx, ctx->pre[0], n * sizeof(float)))) return rc;
  float *tc = (float *)ctx->pre[0].p;
  if ((rc = launch_prep<float, float>(ctx, (const float *)dI, (const float *)dC, tc, nullptr, (int64_t)n)))
    return rc;
  // four fields through the three axis passes in one launch each: a*cT, a*c, a_x*cT, a_x*c
  // (slot 0: the 0th-order pair, slot 1: the pair differentiated along `axis`)
  const float *in[4];
  float *o[4];
  const double sg[4] = {sigma, sigma, sigma, sigma};
  const int pass_axis[3] = {2, 0, 1};  // SmoothingRecursiveGaussianImageFilter: z, x, y
  for (int p = 0; p < 3 && !rc; ++p) {
    const int a = pass_axis[p];
    const int ord[4] = {0, 0, a == axis ? 1 : 0, a == axis ? 1 : 0};
    for (int j = 0; j < 4; ++j) {
      const int slot = j / 2, f = j % 2;
      const int src = (p % 2 == 0) ? 1 : 0, dst = (p % 2 == 0) ? 0 : 1;  // ping-pong: ->0, ->1, ->0
      in[j] = p == 0 ? (f == 0 ? tc : (const float *)dC) : (const float *)ctx->fld[slot][2 * f + src].p;
      o[j] = (float *)ctx->fld[slot][2 * f + dst].p;
    }
    rc = launch_iir(ctx, vol, a, 4, in, o, sg, 1, ord);
  }
  if (rc) return rc;
  {
    ProfScope ps(ctx, KK_DIVIDE);
    hipLaunchKernelGGL(diffconv_kernel, dim3(2048), dim3(256), 0, ctx->stream,
                       (const float *)ctx->fld[0][0].p, (const float *)ctx->fld[0][2].p,
                       (const float *)ctx->fld[1][0].p, (const float *)ctx->fld[1][2].p,
                       (float)(1.0 / (axis == 0 ? vol->sx : axis == 1 ? vol->sy : vol->sz)), (float *)dO,
                       (int64_t)n);
    IFE_HIP(ctx, hipGetLastError());
  }
  return stage_out_end(ctx, mem, out, n * 4);
}

// ---- a5 + a9 ------------------------------------------------------------------------
int ife_emphysema_features(ife_ctx *ctx, const void *image, int image_dtype, const void *mask,
                           int mask_dtype, const ife_volume_desc *vol, const float *sigmas,
                           int n_sigmas, float *out, int layout, int mem) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, vol, true))) return rc;
  if ((rc = check_layout_mem(ctx, layout, mem))) return rc;
  if (!image || !out || !sigmas) return fail(ctx, IFE_E_ARG, "null pointer");
  if (n_sigmas < 1) return fail(ctx, IFE_E_ARG, "at least one sigma is required");
  for (int s = 0; s < n_sigmas; ++s)
    if (!(sigmas[s] > 0.0f)) return fail(ctx, IFE_E_ARG, "sigma[%d] must be positive", s);
  if (image_dtype != IFE_F32 && image_dtype != IFE_I16)
    return fail(ctx, IFE_E_ARG, "image dtype must be IFE_F32 or IFE_I16");
  if (mask && mask_dtype != IFE_U8 && mask_dtype != IFE_U16)
    return fail(ctx, IFE_E_ARG, "mask dtype must be IFE_U8 or IFE_U16");
  if ((rc = ife_ctx_reserve(ctx, vol))) return rc;
  const size_t n = (size_t)(vol->nx * vol->ny * vol->nz);
  const size_t out_bytes = n * IFE_NUM_FEATURES * 4 * (size_t)n_sigmas;
  const void *dI, *dM;
  void *dO;
  if ((rc = stage_in(ctx, mem, image, n * dtype_size(image_dtype), ctx->st_img, &dI))) return rc;
  if ((rc = stage_in(ctx, mem, mask, n * (mask ? dtype_size(mask_dtype) : 0), ctx->st_mask, &dM)))
    return rc;
  if ((rc = stage_out_begin(ctx, mem, out, out_bytes, &dO))) return rc;
  float *dout = (float *)dO;
  const bool u16 = mask && mask_dtype == IFE_U16;
  if (image_dtype == IFE_F32) {
    rc = u16 ? emphysema_typed(ctx, (const float *)dI, (const uint16_t *)dM, vol, sigmas, n_sigmas,
                               dout, layout)
             : emphysema_typed(ctx, (const float *)dI, (const uint8_t *)dM, vol, sigmas, n_sigmas,
                               dout, layout);
  } else {
    rc = u16 ? emphysema_typed(ctx, (const int16_t *)dI, (const uint16_t *)dM, vol, sigmas,
                               n_sigmas, dout, layout)
             : emphysema_typed(ctx, (const int16_t *)dI, (const uint8_t *)dM, vol, sigmas,
                               n_sigmas, dout, layout);
  }
  if (rc) return rc;
  return stage_out_end(ctx, mem, out, out_bytes);
}

// ---- a9, streaming form -------------------------------------------------------------
int ife_emphysema_features_end(ife_ctx *ctx) {
  int rc = bind(ctx);
  if (rc) return rc;
  IFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->sc_copy) IFE_HIP(ctx, hipStreamSynchronize(ctx->sc_copy));
  for (auto e : ctx->sc_done) (void)hipEventDestroy(e);
  ctx->sc_done.clear();
  if (ctx->sc_out.p) {
    IFE_HIP(ctx, hipFree(ctx->sc_out.p));
    ctx->sc_out.p = nullptr;
    ctx->sc_out.cap = 0;
  }
  ctx->sc_scale_bytes = 0;
  return IFE_OK;
}

int ife_emphysema_features_begin(ife_ctx *ctx, const void *image, int image_dtype, const void *mask,
                                 int mask_dtype, const ife_volume_desc *vol, const float *sigmas,
                                 int n_sigmas, int layout) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, vol, true))) return rc;
  if ((rc = check_layout_mem(ctx, layout, IFE_MEM_HOST))) return rc;
  if (!image || !sigmas) return fail(ctx, IFE_E_ARG, "null pointer");
  if (n_sigmas < 1) return fail(ctx, IFE_E_ARG, "at least one sigma is required");
  for (int s = 0; s < n_sigmas; ++s)
    if (!(sigmas[s] > 0.0f)) return fail(ctx, IFE_E_ARG, "sigma[%d] must be positive", s);
  if (image_dtype != IFE_F32 && image_dtype != IFE_I16)
    return fail(ctx, IFE_E_ARG, "image dtype must be IFE_F32 or IFE_I16");
  if (mask && mask_dtype != IFE_U8 && mask_dtype != IFE_U16)
    return fail(ctx, IFE_E_ARG, "mask dtype must be IFE_U8 or IFE_U16");
  if ((rc = ife_emphysema_features_end(ctx))) return rc;  // drop an earlier, unfinished run
  if ((rc = ife_ctx_reserve(ctx, vol))) return rc;
  const size_t n = (size_t)(vol->nx * vol->ny * vol->nz);
  ctx->sc_scale_bytes = n * IFE_NUM_FEATURES * sizeof(float);
  if ((rc = ensure(ctx, ctx->sc_out, ctx->sc_scale_bytes * (size_t)n_sigmas))) return rc;
  if (!ctx->sc_copy) IFE_HIP(ctx, hipStreamCreateWithFlags(&ctx->sc_copy, hipStreamNonBlocking));
  const void *dI, *dM;
  if ((rc = stage_in(ctx, IFE_MEM_HOST, image, n * dtype_size(image_dtype), ctx->st_img, &dI))) return rc;
  if ((rc = stage_in(ctx, IFE_MEM_HOST, mask, n * (mask ? dtype_size(mask_dtype) : 0), ctx->st_mask, &dM)))
    return rc;
  float *dout = (float *)ctx->sc_out.p;
  const bool u16 = mask && mask_dtype == IFE_U16;
  ctx->scale_events = &ctx->sc_done;
  if (image_dtype == IFE_F32)
    rc = u16 ? emphysema_typed(ctx, (const float *)dI, (const uint16_t *)dM, vol, sigmas, n_sigmas, dout, layout)
             : emphysema_typed(ctx, (const float *)dI, (const uint8_t *)dM, vol, sigmas, n_sigmas, dout, layout);
  else
    rc = u16 ? emphysema_typed(ctx, (const int16_t *)dI, (const uint16_t *)dM, vol, sigmas, n_sigmas, dout, layout)
             : emphysema_typed(ctx, (const int16_t *)dI, (const uint8_t *)dM, vol, sigmas, n_sigmas, dout, layout);
  ctx->scale_events = nullptr;
  if (rc) (void)ife_emphysema_features_end(ctx);
  return rc;
}

int ife_emphysema_features_fetch(ife_ctx *ctx, int scale, float *out) {
  int rc = bind(ctx);
  if (rc) return rc;
  if (!out) return fail(ctx, IFE_E_ARG, "null pointer");
  if (scale < 0 || scale >= (int)ctx->sc_done.size())
    return fail(ctx, IFE_E_STATE, "scale %d was not started by ife_emphysema_features_begin (%d scales)",
                scale, (int)ctx->sc_done.size());
  IFE_HIP(ctx, hipStreamWaitEvent(ctx->sc_copy, ctx->sc_done[scale], 0));
  IFE_HIP(ctx, hipMemcpyAsync(out, (const char *)ctx->sc_out.p + (size_t)scale * ctx->sc_scale_bytes,
                              ctx->sc_scale_bytes, hipMemcpyDeviceToHost, ctx->sc_copy));
  IFE_HIP(ctx, hipStreamSynchronize(ctx->sc_copy));
  return IFE_OK;
}

// ---- a6 -----------------------------------------------------------------------------
int ife_fd_hessian_features(ife_ctx *ctx, const void *image, int image_dtype, const void *mask,
                            int mask_dtype, const ife_volume_desc *vol, float *out6, int layout,
                            int mem) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, vol, false))) return rc;
  if ((rc = check_layout_mem(ctx, layout, mem))) return rc;
  if (!image || !out6) return fail(ctx, IFE_E_ARG, "null pointer");
  if (image_dtype != IFE_F32 && image_dtype != IFE_I16)
    return fail(ctx, IFE_E_ARG, "image dtype must be IFE_F32 or IFE_I16");
  if (mask && mask_dtype != IFE_U8 && mask_dtype != IFE_U16)
    return fail(ctx, IFE_E_ARG, "mask dtype must be IFE_U8 or IFE_U16");
  const size_t n = (size_t)(vol->nx * vol->ny * vol->nz);
  const void *dI, *dM;
  void *dO;
  if ((rc = stage_in(ctx, mem, image, n * dtype_size(image_dtype), ctx->st_img, &dI))) return rc;
  if ((rc = stage_in(ctx, mem, mask, n * (mask ? dtype_size(mask_dtype) : 0), ctx->st_mask, &dM)))
    return rc;
  if ((rc = stage_out_begin(ctx, mem, out6, n * 24, &dO))) return rc;
  const bool u16 = mask && mask_dtype == IFE_U16;
  if (image_dtype == IFE_F32) {
    rc = u16 ? launch_features<FEAT_EIG6>(ctx, ValRaw<float>{(const float *)dI},
                                          (const uint16_t *)dM, (float *)dO, vol, layout)
             : launch_features<FEAT_EIG6>(ctx, ValRaw<float>{(const float *)dI},
                                          (const uint8_t *)dM, (float *)dO, vol, layout);
  } else {
    rc = u16 ? launch_features<FEAT_EIG6>(ctx, ValRaw<int16_t>{(const int16_t *)dI},
                                          (const uint16_t *)dM, (float *)dO, vol, layout)
             : launch_features<FEAT_EIG6>(ctx, ValRaw<int16_t>{(const int16_t *)dI},
                                          (const uint8_t *)dM, (float *)dO, vol, layout);
  }
  if (rc) return rc;
  return stage_out_end(ctx, mem, out6, n * 24);
}

// ---- a7 -----------------------------------------------------------------------------
int ife_fd_gradient_features(ife_ctx *ctx, const float *image, const float *mask,
                             const ife_volume_desc *vol, float *out, int mem) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, vol, false))) return rc;
  if (mem != IFE_MEM_HOST && mem != IFE_MEM_DEVICE) return fail(ctx, IFE_E_ARG, "bad mem");
  if (!image || !out) return fail(ctx, IFE_E_ARG, "null pointer");
  const size_t n = (size_t)(vol->nx * vol->ny * vol->nz);
  const void *dI, *dM;
  void *dO;
  if ((rc = stage_in(ctx, mem, image, n * 4, ctx->st_img, &dI))) return rc;
  if ((rc = stage_in(ctx, mem, mask, mask ? n * 4 : 0, ctx->st_aux, &dM))) return rc;
  if ((rc = stage_out_begin(ctx, mem, out, n * 4, &dO))) return rc;
  rc = launch_features<FEAT_GRADMAG>(ctx, ValRaw<float>{(const float *)dI}, (const float *)dM,
                                     (float *)dO, vol, IFE_PLANAR);
  if (rc) return rc;
  return stage_out_end(ctx, mem, out, n * 4);
}

// ---- a8 -----------------------------------------------------------------------------
int ife_mask_image_f64(ife_ctx *ctx, const double *image, const double *mask, double outside,
                       int64_t n, double *out, int mem) {
  int rc = bind(ctx);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!image || !mask || !out))) return fail(ctx, IFE_E_ARG, "null pointer");
  if (mem != IFE_MEM_HOST && mem != IFE_MEM_DEVICE) return fail(ctx, IFE_E_ARG, "bad mem");
  if (n == 0) return IFE_OK;
  const void *dI, *dM;
  void *dO;
  if ((rc = stage_in(ctx, mem, image, (size_t)n * 8, ctx->st_img, &dI))) return rc;
  if ((rc = stage_in(ctx, mem, mask, (size_t)n * 8, ctx->st_aux, &dM))) return rc;
  if ((rc = stage_out_begin(ctx, mem, out, (size_t)n * 8, &dO))) return rc;
  {
    ProfScope ps(ctx, KK_MASK);
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(mask_f64_kernel, dim3(blocks), dim3(256), 0, ctx->stream,
                       (const double *)dI, (const double *)dM, outside, (double *)dO, n);
    IFE_HIP(ctx, hipGetLastError());
  }
  return stage_out_end(ctx, mem, out, (size_t)n * 8);
}

// ---- stage entry points (device pointers only; Z-slab orchestration) --------------------
int ife_stage_prepare(ife_ctx *ctx, const void *image, int image_dtype, const void *mask,
                      int mask_dtype, const ife_volume_desc *slab, int y_chunks, float *tc,
                      float *cf) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, slab, false))) return rc;
  if (!image || !tc) return fail(ctx, IFE_E_ARG, "null pointer");
  if (image_dtype != IFE_F32 && image_dtype != IFE_I16)
    return fail(ctx, IFE_E_ARG, "image dtype must be IFE_F32 or IFE_I16");
  if (mask && mask_dtype != IFE_U8 && mask_dtype != IFE_U16)
    return fail(ctx, IFE_E_ARG, "mask dtype must be IFE_U8 or IFE_U16");
  if (y_chunks < 1 || slab->ny % y_chunks)
    return fail(ctx, IFE_E_SIZE, "ny must be a multiple of the number of Y chunks");
  const int64_t n = slab->nx * slab->ny * slab->nz;
  const PrepGeom pg{slab->nx, slab->ny, slab->nz, y_chunks, slab->ny / y_chunks};
  const bool u16 = mask && mask_dtype == IFE_U16;
  if (image_dtype == IFE_F32)
    return u16 ? launch_prep(ctx, (const float *)image, (const uint16_t *)mask, tc, cf, n, pg)
               : launch_prep(ctx, (const float *)image, (const uint8_t *)mask, tc, cf, n, pg);
  return u16 ? launch_prep(ctx, (const int16_t *)image, (const uint16_t *)mask, tc, cf, n, pg)
             : launch_prep(ctx, (const int16_t *)image, (const uint8_t *)mask, tc, cf, n, pg);
}

int ife_stage_recursive_gaussian(ife_ctx *ctx, const float *in, float *out,
                                 const ife_volume_desc *vol, int axis, double sigma) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, vol, false))) return rc;
  if (!in || !out) return fail(ctx, IFE_E_ARG, "null pointer");
  if (in == out) return fail(ctx, IFE_E_ARG, "the axis pass is not in place");
  if (axis < 0 || axis > 2) return fail(ctx, IFE_E_ARG, "axis must be 0 (x), 1 (y) or 2 (z)");
  if (!(sigma > 0.0)) return fail(ctx, IFE_E_ARG, "sigma must be positive");
  const int64_t len = axis == 0 ? vol->nx : axis == 1 ? vol->ny : vol->nz;
  if (len < 4)
    return fail(ctx, IFE_E_SIZE, "the recursive Gaussian needs at least 4 voxels along axis %d",
                axis);
  const float *ins[1] = {in};
  float *outs[1] = {out};
  return launch_iir(ctx, vol, axis, 1, ins, outs, &sigma);
}

int ife_stage_recursive_gaussian_batch(ife_ctx *ctx, int njobs, const float *const *in,
                                       float *const *out, const ife_volume_desc *vol, int axis,
                                       const double *sigmas, int in_y_chunks) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, vol, false))) return rc;
  if (!in || !out || !sigmas) return fail(ctx, IFE_E_ARG, "null pointer");
  if (axis < 0 || axis > 2) return fail(ctx, IFE_E_ARG, "axis must be 0 (x), 1 (y) or 2 (z)");
  for (int j = 0; j < njobs; ++j)
    if (!(sigmas[j] > 0.0)) return fail(ctx, IFE_E_ARG, "sigma must be positive");
  const int64_t len = axis == 0 ? vol->nx : axis == 1 ? vol->ny : vol->nz;
  if (len < 4)
    return fail(ctx, IFE_E_SIZE, "the recursive Gaussian needs at least 4 voxels along axis %d",
                axis);
  return launch_iir(ctx, vol, axis, njobs, in, out, sigmas, in_y_chunks);
}

size_t ife_stage_z_ck_bytes(const ife_volume_desc *slab) {
  if (!slab || slab->nx <= 0 || slab->ny <= 0 || slab->nz <= 0) return 0;
  return zslab_ck_bytes(slab);
}

int ife_stage_z_sweep(ife_ctx *ctx, int direction, int njobs, const float *const *in,
                      const ife_volume_desc *slab, int64_t line0, int64_t nlines,
                      const double *sigmas, int has_neighbour, const void *state_in,
                      void *state_out, void *const *ck) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, slab, false))) return rc;
  if (direction != 0 && direction != 1) return fail(ctx, IFE_E_ARG, "direction must be 0 (causal) or 1 (anticausal)");
  return launch_zslab(ctx, direction, njobs, in, nullptr, slab, line0, nlines, sigmas,
                      direction == 0 ? has_neighbour : 0, direction == 1 ? has_neighbour : 0, state_in,
                      state_out, ck);
}

int ife_stage_z_combine(ife_ctx *ctx, int njobs, const float *const *in, float *const *out,
                        const ife_volume_desc *slab, int64_t line0, int64_t nlines,
                        const double *sigmas, int has_lo, int has_hi, void *const *ck) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, slab, false))) return rc;
  return launch_zslab(ctx, 2, njobs, in, out, slab, line0, nlines, sigmas, has_lo, has_hi, nullptr,
                      nullptr, ck);
}

int ife_stage_z_fused(ife_ctx *ctx, int direction, int njobs, const float *const *in,
                      float *const *out, const ife_volume_desc *slab, int64_t line0, int64_t nlines,
                      const double *sigmas, int has_lo, int has_hi, const void *state_in,
                      void *state_out, void *const *ck) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, slab, false))) return rc;
  if (direction != 0 && direction != 1) return fail(ctx, IFE_E_ARG, "direction must be 0 (causal) or 1 (anticausal)");
  return launch_zslab(ctx, 3 + direction, njobs, in, out, slab, line0, nlines, sigmas, has_lo, has_hi,
                      state_in, state_out, ck);
}

int ife_stage_recursive_gaussian_quotient(ife_ctx *ctx, int njobs, const float *const *num,
                                          const float *const *den, float *const *out,
                                          const ife_volume_desc *vol, int axis, const double *sigmas) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, vol, false))) return rc;
  if (!num || !den || !out || !sigmas) return fail(ctx, IFE_E_ARG, "null pointer");
  if (axis != 1 && axis != 2) return fail(ctx, IFE_E_ARG, "the quotient form runs on the strided axes: 1 (y) or 2 (z)");
  for (int j = 0; j < njobs; ++j)
    if (!(sigmas[j] > 0.0)) return fail(ctx, IFE_E_ARG, "sigma must be positive");
  if ((axis == 1 ? vol->ny : vol->nz) < 4)
    return fail(ctx, IFE_E_SIZE, "the recursive Gaussian needs at least 4 voxels along axis %d", axis);
  return launch_iir(ctx, vol, axis, njobs, num, out, sigmas, 1, nullptr, den);
}

int ife_stage_features(ife_ctx *ctx, const float *num, const float *den, const void *mask,
                       int mask_dtype, const ife_volume_desc *slab, int halo_lo, int halo_hi,
                       float *out, int layout) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = check_vol(ctx, slab, false))) return rc;
  if ((rc = check_layout_mem(ctx, layout, IFE_MEM_DEVICE))) return rc;
  if (!num || !out) return fail(ctx, IFE_E_ARG, "null pointer");
  if (mask && mask_dtype != IFE_U8 && mask_dtype != IFE_U16)
    return fail(ctx, IFE_E_ARG, "mask dtype must be IFE_U8 or IFE_U16");
  if (mask && mask_dtype == IFE_U16)
    return launch_features<FEAT_FEATURES8>(ctx, ValSmooth{num, den}, (const uint16_t *)mask, out,
                                           slab, layout, halo_lo, halo_hi);
  return launch_features<FEAT_FEATURES8>(ctx, ValSmooth{num, den}, (const uint8_t *)mask, out,
                                         slab, layout, halo_lo, halo_hi);
}

// ---- measurement ----------------------------------------------------------------------
int ife_get_kernel_times(ife_ctx *ctx, ife_kernel_time *entries, int max_entries) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = drain_profile(ctx))) return rc;
  int k = 0;
  for (int i = 0; i < KK_COUNT && k < max_entries; ++i) {
    if (ctx->acc_n[i] == 0) continue;
    memset(&entries[k], 0, sizeof entries[k]);
    strncpy(entries[k].name, kKindNames[i], sizeof entries[k].name - 1);
    entries[k].launches = ctx->acc_n[i];
    entries[k].total_ms = ctx->acc_ms[i];
    ++k;
  }
  return k;
}

// The box's own streaming rates, measured with this library's access shape (16 B per lane,
// grid-stride, hipEvents on the context's stream): what the roofline fractions in bench.py are
// read against beside the 8 TB/s peak.
__global__ __launch_bounds__(256) void stream_fill_kernel(float4 *__restrict__ dst, int64_t n4, float v) {
  const float4 x = make_float4(v, v, v, v);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    __builtin_nontemporal_store(f32x4{x.x, x.y, x.z, x.w}, reinterpret_cast<f32x4 *>(dst) + i);
}
__global__ __launch_bounds__(256) void stream_copy_kernel(const float4 *__restrict__ src,
                                                          float4 *__restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src) + i),
                                reinterpret_cast<f32x4 *>(dst) + i);
}
int ife_measure_stream(ife_ctx *ctx, int mode, void *dst, const void *src, size_t bytes, int reps,
                       double *ms_per_pass) {
  int rc = bind(ctx);
  if (rc) return rc;
  if (!dst || !ms_per_pass || (mode == 1 && !src) || (mode != 0 && mode != 1) || reps < 1)
    return fail(ctx, IFE_E_ARG, "bad arguments");
  if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 || bytes % 16 || bytes == 0)
    return fail(ctx, IFE_E_ARG, "buffers and size must be multiples of 16 bytes");
  const int64_t n4 = (int64_t)(bytes / 16);
  // one piece per thread: the fastest shape on this chip (see launch_prep)
  const unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 0x7fffffff);
  hipEvent_t a, b;
  IFE_HIP(ctx, hipEventCreate(&a));
  IFE_HIP(ctx, hipEventCreate(&b));
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL(stream_fill_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (float4 *)dst, n4, 1.0f);
    else hipLaunchKernelGGL(stream_copy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const float4 *)src, (float4 *)dst, n4);
  };
  launch();  // warm
  (void)hipEventRecord(a, ctx->stream);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b, ctx->stream);
  hipError_t e = hipEventSynchronize(b);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  if (e != hipSuccess) return fail(ctx, IFE_E_HIP, "stream measurement: %s", hipGetErrorString(e));
  *ms_per_pass = (double)ms / reps;
  return IFE_OK;
}

int ife_reset_kernel_times(ife_ctx *ctx) {
  int rc = bind(ctx);
  if (rc) return rc;
  if ((rc = drain_profile(ctx))) return rc;
  for (int i = 0; i < KK_COUNT; ++i) { ctx->acc_ms[i] = 0; ctx->acc_n[i] = 0; }
  return IFE_OK;
}

}  // extern "C"

#include "stats_capi.inc"
#include "multi_capi.inc"
