/*
 * ife_hip.h -- C-ABI of the MI355X (gfx950) per-voxel Hessian feature engine.
 *
 * This is the drop-in boundary for the hot path of orting/image-feature-extraction:
 * the reference runs the path as ITK filter objects inside one process
 * (headers under include/ife/Filters and include/ife/Numerics); here every stage that the
 * reference exposes as a filter / functor / tool body has one extern "C" entry
 * point taking plain pointers and sizes.  The C++ host classes under
 * image-feature-extraction_amd/host/ (same class and method names as the
 * reference) and the tools forward to these entry points; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - volumes are x-fastest: index = x + nx*(y + ny*z)  (itk::Image buffer order)
 *   - vector outputs: IFE_INTERLEAVED [voxel*ncomp + c] (itk::VectorImage order) or
 *     IFE_PLANAR [c*nvox + voxel]
 *   - IFE_MEM_HOST: pointers are host memory, the call copies in/out and blocks;
 *     IFE_MEM_DEVICE: pointers are device (HBM) memory on the context's device, the
 *     call enqueues on the context's stream and returns without synchronising
 *   - every call returns 0 or a negative ife_status; ife_last_error() gives text
 *   - an ife_ctx is single-owner (one host thread at a time); different contexts
 *     may run concurrently
 *   - there is no CPU fallback: every entry point fails with IFE_E_HIP when no
 *     gfx950 device is usable
 *
 * All "reference" citations are paths relative to the reference repository root.
 */
#ifndef IFE_HIP_H
#define IFE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IFE_ABI_VERSION 1

typedef struct ife_ctx ife_ctx;

typedef enum {
  IFE_OK = 0,
  IFE_E_ARG = -1,    /* null pointer, bad enum, bad sigma/spacing */
  IFE_E_SIZE = -2,   /* an axis shorter than 4 where the recursive Gaussian runs
                        (ITK throws for ln < 4), or a non-positive size */
  IFE_E_HIP = -3,    /* HIP runtime error (text in ife_last_error) */
  IFE_E_NOMEM = -4,  /* device or host allocation failed */
  IFE_E_STATE = -5   /* call sequence error (slab API) */
} ife_status;

typedef enum { IFE_F32 = 0, IFE_I16 = 1, IFE_U8 = 2, IFE_U16 = 3 } ife_dtype;
typedef enum { IFE_INTERLEAVED = 0, IFE_PLANAR = 1 } ife_layout;
typedef enum { IFE_MEM_HOST = 0, IFE_MEM_DEVICE = 1 } ife_mem;
/* IFE_MEM_DEVICE pointers must be aligned to their element size, interleaved 8-component
 * outputs to 16 bytes and 6-component ones to 8 (vector stores); anything else is refused
 * with IFE_E_ARG.  hipMalloc'ed buffers and whole voxels / planes inside them qualify. */

/* Options for ife_ctx_set_option */
typedef enum {
  /* How the solver's unqualified sqrt/acos/cos (Symmetric3x3EigenvalueSolver.h:88,115,
   * 119-120) are evaluated.  0: in double, the binding the reference gets when only <cmath>
   * is visible (bit-identical to a libm build except at about one voxel in 10^8);
   * 1: float overloads (<math.h> visible), correctly rounded;
   * 2 (default): q, p, B and r as in 0/1, acos and cos by float polynomials: eigenvalues
   * within 1e-6 |lambda_1| of mode 0 (measured maximum in DESIGN.md; the contract of this
   * library is 1e-5), the spread the reference itself has between contexts 0 and 1, at a
   * fifth of the instructions.  Two eigenvalues whose magnitudes tie within that error may
   * come out in the other order.  The environment variable IFE_TRIG_MODE=0|1|2 sets the
   * initial value at ife_ctx_create (for the tools, which have no flag for it). */
  IFE_OPT_TRIG_MODE = 1,
  /* 0 (default): DerivativeImageFilter scales by 1/spacing once whatever the order
   * (upstream ITK behaviour); 1: 1/spacing^order.  Same for unit spacing. */
  IFE_OPT_DSCALE_MODE = 2,
  /* 1: record a hipEvent pair around every kernel launch so that
   * ife_get_kernel_times can report per-kernel device time.  Default 0. */
  IFE_OPT_PROFILE = 3,
  /* planes per workgroup march of the feature kernel (default 64) */
  IFE_OPT_ZCHUNK = 4,
  /* samples per register block of the recursive-Gaussian kernels: 0 (default: chosen per
   * axis -- z 12, y 12, x 16), or 8, 10, 12, 16 for both strided axes
   * (x keeps 16 unless 8 is asked for).  Never changes results. */
  IFE_OPT_IIR_BLOCK = 5,
  /* register blocks per checkpoint of the strided (z, y) line kernel: 2 (default, measured
   * faster) or 1 */
  IFE_OPT_IIR_CKPT = 6,
  /* 1: fuse each multiply-add of the recursive Gaussian (half the double operations).  Not
   * the reference's arithmetic: outputs differ by one float ulp at about one voxel in 10^8,
   * which the second differences can amplify past the 1e-5 bar there.  Default 0. */
  IFE_OPT_IIR_FMA = 7,
  /* 1 (default): the last axis pass of the normalized convolution runs numerator and
   * denominator in sibling waves and stores only their quotient (the Div functor,
   * NormalizedGaussianConvolutionImageFilter.hxx:57-61); 0: two float fields and a division
   * in the consumer.  Same results bit for bit. */
  IFE_OPT_FUSED_DIVIDE = 8,
  /* 1 (default): a line of the recursive Gaussian whose samples all have one bit pattern --
   * +0, -0 (T * 0 where T < 0) or 1.0f -- is answered with the constant the filter makes of
   * it, where the host has established that constant by running this sigma and line length
   * through the kernels' arithmetic (decided per wave of 64 adjacent lines): the exterior of a
   * mask costs a read and a write.  Same results bit for bit, signs of zero included; 0:
   * every line is filtered. */
  IFE_OPT_CONST_LINES = 9,
  /* 1 (default): where the feature kernel reads ONE float field (the smoothed value after a
   * quotient-storing last pass, or a raw float image) and the mask is 1 or 2 bytes wide, its
   * planes go from memory straight into an LDS ring several planes ahead of the arithmetic
   * (no staging registers, counted waits); 0: the register-staged form everywhere.  Same
   * arithmetic function, same results bit for bit. */
  IFE_OPT_FEAT_RING = 10
} ife_option;

typedef struct {
  int64_t nx, ny, nz; /* size in voxels */
  double sx, sy, sz;  /* spacing (physical units per voxel), > 0 */
} ife_volume_desc;

/* number of feature components, ImageToEmphysemaFeaturesFilter.h:62 (numFeatures) */
#define IFE_NUM_FEATURES 8
/* component order of ife_emphysema_features, ExtractFeatures.cxx:126-130 */
#define IFE_FEATURE_NAMES                                                              \
  {"GaussianBlur", "GradientMagnitude", "Eigenvalue1", "Eigenvalue2", "Eigenvalue3",   \
   "LaplacianOfGaussian", "GaussianCurvature", "FrobeniusNorm"}

/* ---- context ----------------------------------------------------------------- */

int ife_abi_version(void);
/* Creates a context on HIP device `device`.  Replaces nothing in the reference
 * (which has no device); it owns the workspace the ITK mini-pipeline would hold as
 * intermediate images (ImageToEmphysemaFeaturesFilter.h:74-117). */
int ife_ctx_create(int device, ife_ctx **ctx);
void ife_ctx_destroy(ife_ctx *ctx);
/* Last error text of this context (or of ife_ctx_create when ctx is NULL).
 * Stands in for itk::ExceptionObject::what() (ExtractFeatures.cxx:145-152). */
const char *ife_last_error(const ife_ctx *ctx);
/* Borrow a hipStream_t for all later launches (NULL = the default stream). */
int ife_ctx_set_stream(ife_ctx *ctx, void *hip_stream);
int ife_ctx_set_option(ife_ctx *ctx, int option, int value);
/* Grow the workspace for volumes of this size now (so that later DEVICE-mode calls
 * allocate nothing). */
int ife_ctx_reserve(ife_ctx *ctx, const ife_volume_desc *vol);
/* Block until everything enqueued by this context has finished. */
int ife_ctx_synchronize(ife_ctx *ctx);

/* ---- a1 / a2: per-voxel numerics ----------------------------------------------- */

/* Symmetric3x3EigenvalueSolver<float>::operator()
 * (include/ife/Numerics/Symmetric3x3EigenvalueSolver.h:33-132) on n matrices.
 * A6: n x (xx,xy,xz,yy,yz,zz); ev3: n x 3, |ev0| >= |ev1| >= |ev2|. */
int ife_eigenvalues(ife_ctx *ctx, const float *A6, int64_t n, float *ev3, int mem);
/* EigenvalueFeaturesFunctor<float>::operator()
 * (include/ife/Numerics/EigenvalueFeaturesFunctor.h:20-31): f6 = n x
 * (ev0, ev1, ev2, sum, product, sqrt(sum of squares)). */
int ife_eigenvalue_features(ife_ctx *ctx, const float *A6, int64_t n, float *f6, int mem);

/* ---- a3: Hessian3DImageFilter --------------------------------------------------- */

/* itk::Hessian3DImageFilter<Image<float,3>,VectorImage<float,3>>: SetInput + Update
 * + GetOutput (include/ife/Filters/Hessian3DImageFilter.h:23-28,
 * Hessian3DImageFilter.hxx:13-60,80-97).  out6: 6 comps xx,xy,xz,yy,yz,zz. */
int ife_hessian3d(ife_ctx *ctx, const float *image, const ife_volume_desc *vol, float *out6,
                  int layout, int mem);

/* itk::GradientMagnitudeImageFilter as wired at
 * ImageToEmphysemaFeaturesFilter.hxx:27-28 / FiniteDifference_GradientFeatures.cxx:104-106 */
int ife_gradient_magnitude(ife_ctx *ctx, const float *image, const ife_volume_desc *vol,
                           float *out, int mem);

/* ---- a4: NormalizedGaussianConvolutionImageFilter ------------------------------- */

/* SetInputImage + SetInputCertainty + SetSigma + Update
 * (include/ife/Filters/NormalizedGaussianConvolutionImageFilter.h:86-93,
 * NormalizedGaussianConvolutionImageFilter.hxx:40-63):
 * out = G_sigma(image*certainty) / G_sigma(certainty), G = ITK recursive Gaussian
 * run Z, X, Y; zero denominator -> FLT_MAX.  Every axis must be >= 4 voxels.
 * sigma is double here (ScalarRealType of the Gaussian filter, .h:75,92-93; the
 * MaskedNormalizedConvolution tool parses doubles, :117) and float in
 * ife_emphysema_features (ScalarRealType = PixelType there, .h:41,58-59). */
int ife_normalized_gaussian_convolution(ife_ctx *ctx, const float *image,
                                        const float *certainty, const ife_volume_desc *vol,
                                        double sigma, float *out, int mem);

/* ---- f4: differential normalized convolution ---------------------------------------- */

/* The derivative form the reference sketches but does not implement
 * (include/ife/Filters/NormalizedGaussianConvolutionImageFilter.h:28-44):
 *   out = ({a_x*cT}{a*c} - {a_x*c}{a*cT}) / {a*c}^2  / spacing[axis]
 * a = the recursive Gaussian of ife_normalized_gaussian_convolution (axes z, x, y), a_x = the
 * same with ITK's FirstOrder recursive Gaussian along `axis` (0 = x, 1 = y, 2 = z; pixel units,
 * hence the division by the spacing).  Float arithmetic as written; {a*c}^2 == 0 -> FLT_MAX. */
int ife_differential_normalized_convolution(ife_ctx *ctx, const float *image,
                                            const float *certainty, const ife_volume_desc *vol,
                                            double sigma, int axis, float *out, int mem);

/* ---- a5 + a9: ImageToEmphysemaFeaturesFilter, one execution per scale ------------ */

/* SetInputImage + SetInputMask + for each sigma {SetSigma; Update; GetOutput}
 * (include/ife/Filters/ImageToEmphysemaFeaturesFilter.h:44-62,
 * ImageToEmphysemaFeaturesFilter.hxx:15-55,99-121; scale loop
 * tools/ExtractFeatures.cxx:132-154).
 * image: IFE_F32 or IFE_I16 (converted exactly to float on load);
 * mask: IFE_U8 or IFE_U16, value used as certainty weight and tested != 0 for the
 * final masking; NULL means all ones.
 * out: n_sigmas consecutive 8-component volumes (scale-major), each in `layout`. */
int ife_emphysema_features(ife_ctx *ctx, const void *image, int image_dtype, const void *mask,
                           int mask_dtype, const ife_volume_desc *vol, const float *sigmas,
                           int n_sigmas, float *out, int layout, int mem);

/* The scale loop of tools/ExtractFeatures.cxx:132-154 as a stream: _begin uploads image and
 * mask ONCE (host pointers), runs Cast + Multiply once and enqueues every scale, leaving the
 * outputs in device memory owned by the context; _fetch(scale) blocks until that scale has
 * finished and copies its 8-component volume (`layout` of _begin) to host memory on a copy
 * stream of its own -- so a caller can write scale k to disk while the device is still busy
 * with the later scales, and never holds more than one scale in host memory; _end frees the
 * device copy (also done by the next _begin and by ife_ctx_destroy).  IFE_E_STATE when a
 * scale is fetched that _begin did not start. */
int ife_emphysema_features_begin(ife_ctx *ctx, const void *image, int image_dtype,
                                 const void *mask, int mask_dtype, const ife_volume_desc *vol,
                                 const float *sigmas, int n_sigmas, int layout);
int ife_emphysema_features_fetch(ife_ctx *ctx, int scale, float *out);
int ife_emphysema_features_end(ife_ctx *ctx);

/* ---- a6 / a7 / a8: tool bodies ---------------------------------------------------- */

/* Body of tools/FiniteDifference_HessianFeatures.cxx:126-229 (un-smoothed Hessian,
 * eigen features, mask==0 -> six zeros), with Hessian3DImageFilter.hxx:34-37 as the
 * normative derivative wiring (the tool's :155 direction slip is not reproduced).
 * out6 comps: eig1, eig2, eig3, LoG, Curvature, Frobenius.  mask may be NULL. */
int ife_fd_hessian_features(ife_ctx *ctx, const void *image, int image_dtype,
                            const void *mask, int mask_dtype, const ife_volume_desc *vol,
                            float *out6, int layout, int mem);
/* Body of tools/FiniteDifference_GradientFeatures.cxx:104-113: the mask is a float
 * image there. */
int ife_fd_gradient_features(ife_ctx *ctx, const float *image, const float *mask,
                             const ife_volume_desc *vol, float *out, int mem);
/* Body of tools/MaskedImageFilter.cxx:75-93 (double pixels, -v outside value). */
int ife_mask_image_f64(ife_ctx *ctx, const double *image, const double *mask, double outside,
                       int64_t n, double *out, int mem);

/* ---- stage entry points: Z-slab decomposition across GPUs ----------------------------
 *
 * The reference runs the path in one address space (it has no
 * distributed code).  A multi-GPU host cuts the volume into Z-slabs, one per device, and
 * interposes two neighbour exchanges between these stages (image-feature-extraction_amd/
 * slab.py, csrc/multi_capi.inc, DESIGN.md "Multi-GPU"): the state of the Z recursion (the only
 * one that crosses slabs) handed from slab to slab, 32 bytes per line and job, and a
 * one-plane halo exchange in front of the stencil.  All pointers are device memory; calls
 * enqueue on the context's stream and do not synchronise. */

/* CastImageFilter + MultiplyImageFilter (ImageToEmphysemaFeaturesFilter.hxx:21,110;
 * NormalizedGaussianConvolutionImageFilter.hxx:48-49) on a slab: tc = float(image) *
 * float(mask), cf = float(mask).  mask NULL: tc = float(image), cf untouched (may be NULL).
 * y_chunks = 1: outputs in the slab's own order.  y_chunks = W > 1: outputs in the order an
 * all-to-all sends from, [W][nz][ny/W][nx] (chunk h = the rows of rank h's Y-slab), so the
 * host needs no packing pass. */
int ife_stage_prepare(ife_ctx *ctx, const void *image, int image_dtype, const void *mask,
                      int mask_dtype, const ife_volume_desc *slab, int y_chunks, float *tc,
                      float *cf);
/* One axis of itk::SmoothingRecursiveGaussianImageFilter (the reference reaches it at
 * NormalizedGaussianConvolutionImageFilter.hxx:51-55); axis 0 = x, 1 = y, 2 = z; the
 * caller keeps ITK's order z, x, y.  Not in place. */
int ife_stage_recursive_gaussian(ife_ctx *ctx, const float *in, float *out,
                                 const ife_volume_desc *vol, int axis, double sigma);
/* The same over njobs (<= 8) float volumes of one geometry in a single launch, each with
 * its own sigma: numerator and denominator of several scales.  A slab host needs this to
 * keep the device full (a 64-plane slab has too few lines for one job per launch).
 * in_y_chunks = W > 1 (axis 0 only): every input is the Y-chunked image of the slab as the
 * all-to-all back from the Z pass leaves it, [W][nz][ny/W][nx]; the outputs are plain
 * slabs.  Needs (ny/W) % 64 == 0.  in_y_chunks = 1: plain inputs. */
int ife_stage_recursive_gaussian_batch(ife_ctx *ctx, int njobs, const float *const *in,
                                       float *const *out, const ife_volume_desc *vol, int axis,
                                       const double *sigmas, int in_y_chunks);
/* The Z pass of a Z-slab with the recursion state handed across the slab boundaries (the
 * reference has no counterpart: its RecursiveGaussianImageFilter sees whole lines,
 * NormalizedGaussianConvolutionImageFilter.hxx:51-55).  A slab holds planes [z0, z1) of every
 * Z line; `in` are njobs (<= 8) float slabs [nz][ny][nx] with one sigma each.  Three calls
 * (or two: ife_stage_z_fused below replaces the later sweep and the combine):
 *   ife_stage_z_sweep(direction 0): causal recursion upwards.  has_neighbour = a slab below
 *     exists and state_in holds its outgoing record; else ITK's start-of-line rule applies.
 *     Leaves the causal checkpoints in ck[job] and the state at z1 in state_out.
 *   ife_stage_z_sweep(direction 1): anticausal recursion downwards, state_in from the slab
 *     above (or ITK's end-of-line rule); anticausal checkpoints, state at z0 in state_out.
 *   ife_stage_z_combine: rebuilds both recursions of every block pair from the checkpoints
 *     and writes float(causal + anticausal) to out[job].
 * Stitched over the slabs this is bit for bit the sequential recursion of the whole line.
 * Lines [line0, line0 + nlines) of the nx*ny lines are processed (x-fastest line index), so
 * that a host can pipeline groups of lines across devices.  A state buffer holds, per job,
 * 4*nlines doubles y[k][line] (IFE_Z_STATE_BYTES * nlines bytes; 8-byte aligned): the last
 * four outputs of the recursion, nearest first.  The INPUT samples a state refers to do not
 * travel with it: where a neighbour exists, `in[job]` must be preceded by the neighbour's last
 * 3 planes (has_lo / direction 0) and followed by its first 4 planes (has_hi / direction 1) --
 * the slab's input is cut from the volume with that overlap (IFE_Z_OVERLAP_LO / _HI planes),
 * which every device can do from its own copy of the raw data, and `in[job]` points at the
 * slab's own plane 0 inside it.  ck[job] points at ife_stage_z_ck_bytes(slab) bytes of device
 * memory that must survive from the sweeps to the combine.  Every slab needs at least 4
 * planes. */
#define IFE_Z_STATE_BYTES 32
#define IFE_Z_OVERLAP_LO 3
#define IFE_Z_OVERLAP_HI 4
size_t ife_stage_z_ck_bytes(const ife_volume_desc *slab);
int ife_stage_z_sweep(ife_ctx *ctx, int direction, int njobs, const float *const *in,
                      const ife_volume_desc *slab, int64_t line0, int64_t nlines,
                      const double *sigmas, int has_neighbour, const void *state_in,
                      void *state_out, void *const *ck);
int ife_stage_z_combine(ife_ctx *ctx, int njobs, const float *const *in, float *const *out,
                        const ife_volume_desc *slab, int64_t line0, int64_t nlines,
                        const double *sigmas, int has_lo, int has_hi, void *const *ck);
/* Sweep of `direction` and combine in one: for the direction whose state reaches a slab LAST.
 * The recursion of `direction` runs through the slab from state_in (or ITK's border rule where
 * that side has no neighbour: has_lo for direction 0, has_hi for direction 1), the other
 * direction's values come from the checkpoints its ife_stage_z_sweep left in ck[job]; writes
 * float(causal + anticausal) to out[job] and the carried state to state_out.  Three recursion
 * steps per sample for the slab (one lean sweep + this) instead of four (two sweeps + combine),
 * the same operations on every sample: same bits. */
int ife_stage_z_fused(ife_ctx *ctx, int direction, int njobs, const float *const *in,
                      float *const *out, const ife_volume_desc *slab, int64_t line0, int64_t nlines,
                      const double *sigmas, int has_lo, int has_hi, const void *state_in,
                      void *state_out, void *const *ck);
/* The last axis pass of the normalized convolution in its quotient form
 * (NormalizedGaussianConvolutionImageFilter.hxx:51-61: the two smoothings and the Div functor):
 * job j filters num[j] and den[j] along `axis` (1 = y or 2 = z) with sigmas[j] and stores
 * numerator / denominator (denominator == 0: the functor's max()) in out[j].  njobs <= 4. */
int ife_stage_recursive_gaussian_quotient(ife_ctx *ctx, int njobs, const float *const *num,
                                          const float *const *den, float *const *out,
                                          const ife_volume_desc *vol, int axis, const double *sigmas);

/* Everything after the smoothing (ImageToEmphysemaFeaturesFilter.hxx:27-54 plus the
 * Divide of NormalizedGaussianConvolutionImageFilter.hxx:57-61) on a slab of slab->nz
 * planes.  num/den hold halo_lo + slab->nz + halo_hi planes: with halo_lo (halo_hi) = 1
 * the first (last) plane is the neighbouring slab's boundary plane, with 0 the boundary
 * is replicated as at the end of the volume.  den NULL: certainty == 1.  mask/out cover
 * the slab's own planes only. */
int ife_stage_features(ife_ctx *ctx, const float *num, const float *den, const void *mask,
                       int mask_dtype, const ife_volume_desc *slab, int halo_lo, int halo_hi,
                       float *out, int layout);

/* ---- several devices in one process --------------------------------------------------
 *
 * The same Z-slab decomposition driven from C++ by ONE host thread: device r of the list
 * owns the r-th range of planes (nz / n planes each, remainder on the first ones; every
 * slab needs 4), states and stencil planes travel by peer copies over xGMI, every dependency
 * is a HIP event.  Host pointers in, host pointers out; blocks until the whole output is
 * there.  Same results bit for bit as ife_emphysema_features on one device.  A device may be
 * listed more than once (how a one-GPU box tests n > 1). */
typedef struct ife_multi ife_multi;
int ife_multi_create(const int *devices, int n_devices, ife_multi **out);
void ife_multi_destroy(ife_multi *m);
const char *ife_multi_last_error(const ife_multi *m);
/* ife_ctx_set_option on every context of the engine */
int ife_multi_set_option(ife_multi *m, int option, int value);
int ife_multi_emphysema_features(ife_multi *m, const void *image, int image_dtype,
                                 const void *mask, int mask_dtype, const ife_volume_desc *vol,
                                 const float *sigmas, int n_sigmas, float *out, int layout);
/* The scale loop of tools/ExtractFeatures.cxx:132-154 over several devices, one scale at a
 * time to the host (the multi-device form of ife_emphysema_features_begin / _fetch / _end):
 * _begin uploads every slab once, runs Cast + Multiply once and enqueues every scale on every
 * device without waiting; _fetch(k) blocks until scale k of every slab has been copied into
 * `out` (that scale's whole volume: nx*ny*nz*8 floats of host memory) on a stream of its own,
 * while the later scales keep computing; _end drains.  IFE_E_STATE out of sequence. */
int ife_multi_emphysema_features_begin(ife_multi *m, const void *image, int image_dtype,
                                       const void *mask, int mask_dtype, const ife_volume_desc *vol,
                                       const float *sigmas, int n_sigmas, int layout);
int ife_multi_emphysema_features_fetch(ife_multi *m, int scale, float *out);
int ife_multi_emphysema_features_end(ife_multi *m);

/* ---- rows f1 / f2: sample columns, equalizing histogram edges, dense histograms ------- *
 * The immediate consumer of the feature volume.  The samples never
 * leave HBM; only the nbins-1 edges per column come back.                                 */

/* std::sort of one sample column
 * (tools/DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures.cxx:284).  Ascending;
 * -0 sorts before +0 (std::sort leaves their order unspecified).  in == out is allowed. */
int ife_sort_f32(ife_ctx *ctx, const float *in, int64_t n, float *out, int mem);

/* determineEdgesForEqualizedHistogram(first, last, d_first, nBins),
 * include/ife/Statistics/DetermineEdgesForEqualizedHistogram.h:21-137.  sorted[0..n) must
 * be ascending; writes nbins-1 edges.  IFE_E_ARG when n < nbins (std::out_of_range there,
 * :36-38); IFE_E_STATE when the walk would pass the last sample (an assert there, :74). */
int ife_equalized_edges_f32(ife_ctx *ctx, const float *sorted, int64_t n, int nbins,
                            float *edges, int mem);
/* the same on doubles (the reference's own test instantiates the template on double,
 * test/DetermineEdgesForEqualizedHistogramTest.cxx:11) */
int ife_equalized_edges_f64(ife_ctx *ctx, const double *sorted, int64_t n, int nbins,
                            double *edges, int mem);

/* DenseHistogram<float>: insert every value, getCounts
 * (include/ife/Statistics/DenseHistogram.h:29-64).  Bins (-inf,e0], (e0,e1], ...,
 * (e_last,inf); counts holds n_edges+1 entries; 1 <= n_edges <= 1024. */
int ife_dense_histogram_f32(ife_ctx *ctx, const float *edges, int n_edges, const float *values,
                            int64_t n, uint32_t *counts, int mem);

/* Row f2: the bag rows of tools/MakeBag.cxx:405-472 for ONE feature volume.  rois holds
 * n_rois boxes {x, y, z, sx, sy, sz} (index and size in voxels, as ROIReader.hxx:28-47 reads
 * them); for every box and component c the component values of the box's voxels with
 * mask != 0 are binned by edges[c*n_edges ..] (DenseHistogram bins), into
 * counts[(roi*ncomp + c)*(n_edges+1) + bin].  IFE_E_ARG when a box leaves the volume
 * (RegionOfInterestImageFilter throws there).  ncomp*(2*n_edges+1) <= 8192. */
int ife_roi_histograms(ife_ctx *ctx, const float *features, int layout, int ncomp,
                       const void *mask, int mask_dtype, const ife_volume_desc *vol,
                       const int64_t *rois, int n_rois, const float *edges, int n_edges,
                       uint32_t *counts, int mem);
/* One image of MakeBag: labels clamped to {0,1} (:236-244), a5 at every scale with the
 * features left in HBM, then the rows above per scale.  rois, edges ([n_sigmas*8][n_edges],
 * the rows of the histogram specification) and counts
 * ([n_rois][n_sigmas*8][n_edges+1]) are HOST arrays; mem describes image and mask. */
int ife_bag_image(ife_ctx *ctx, const void *image, int image_dtype, const void *mask,
                  int mask_dtype, const ife_volume_desc *vol, const float *sigmas, int n_sigmas,
                  const int64_t *rois, int n_rois, const float *edges, int n_edges,
                  uint32_t *counts, int mem);

/* The tool's `samples( scales.size() * numFeatures )` (:165-166): one growing column per
 * (scale, feature), kept in device memory.  Belongs to the context it was created on. */
typedef struct ife_samples ife_samples;
int ife_samples_create(ife_ctx *ctx, int n_columns, ife_samples **out);
void ife_samples_destroy(ife_samples *s);
int ife_samples_count(const ife_samples *s, int column, int64_t *n);
int ife_samples_clear(ife_samples *s);

/* Append the ncomp feature values of the accepted voxels of one feature volume to columns
 * [first_column, first_column+ncomp).  indices == NULL: every voxel whose mask value
 * equals one of foreground[0..n_foreground) (the nSamples == 0 branch, :221-236;
 * 1 <= n_foreground <= 8).  indices != NULL: the listed voxels, x-fastest linear index,
 * repeats allowed (the sampled branch, :238-263: the host draws the positions).  Samples
 * are appended in raster order / list order, as the reference pushes them. */
int ife_samples_add_features(ife_ctx *ctx, ife_samples *s, int first_column,
                             const float *features, int layout, int ncomp, const void *mask,
                             int mask_dtype, int64_t nvox, const uint32_t *foreground,
                             int n_foreground, const int64_t *indices, int64_t n_indices,
                             int mem);

/* One image of the tool's loop (:176-265): the labels are clamped to {0,1} for the filter
 * (ClampImageFilter, :147-152), a5 runs at every scale, and the samples of scale i go to
 * columns [8i, 8i+8) -- in the all-foreground branch straight from the feature kernel (no
 * feature volume is stored), in the sampled branch from a feature volume left in HBM.  The samples object must have
 * 8*n_sigmas columns.  indices, when given, holds n_sigmas * n_indices_per_scale voxel
 * indices, scale-major. */
int ife_samples_add_image(ife_ctx *ctx, ife_samples *s, const void *image, int image_dtype,
                          const void *mask, int mask_dtype, const ife_volume_desc *vol,
                          const float *sigmas, int n_sigmas, const uint32_t *foreground,
                          int n_foreground, const int64_t *indices,
                          int64_t n_indices_per_scale, int mem);

/* :282-289: sort every column and write its equalizing edges; edges is a HOST array of
 * n_columns * (nbins-1) floats, one row per column in column order. */
int ife_samples_sort(ife_ctx *ctx, ife_samples *s);
int ife_samples_equalized_edges(ife_ctx *ctx, ife_samples *s, int nbins, float *edges);
/* copy one column (count values) to a HOST array */
int ife_samples_read_column(ife_ctx *ctx, const ife_samples *s, int column, float *out,
                            int64_t capacity);

/* ---- measurement ------------------------------------------------------------------- */

#define IFE_MAX_KERNEL_KINDS 16
typedef struct {
  char name[48];
  int64_t launches;
  double total_ms; /* sum of hipEvent elapsed times */
} ife_kernel_time;
/* With IFE_OPT_PROFILE=1: per-kernel-kind device time accumulated since the last
 * ife_reset_kernel_times.  Synchronises the stream.  Returns the number of entries
 * written (<= max_entries) or a negative status. */
int ife_get_kernel_times(ife_ctx *ctx, ife_kernel_time *entries, int max_entries);
int ife_reset_kernel_times(ife_ctx *ctx);
/* The box's streaming rate with 16-byte accesses per lane (no reference counterpart; bench.py
 * prints it beside the 8 TB/s peak): mode 0 writes `bytes` to dst (a fill), mode 1 copies
 * `bytes` from src to dst; DEVICE pointers, 16-byte aligned.  One warm pass, then `reps`
 * timed passes between hipEvents on the context's stream; *ms_per_pass is their average. */
int ife_measure_stream(ife_ctx *ctx, int mode, void *dst, const void *src, size_t bytes, int reps,
                       double *ms_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* IFE_HIP_H */
