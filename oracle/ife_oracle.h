/*
 * ife_oracle.h -- CPU restatement of the per-voxel Hessian feature path of
 * orting/image-feature-extraction.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libife_hip.so) never links, loads or calls
 * anything in oracle/.
 *
 * Parity status
 *   - eigen solver / feature functor (a1, a2): PINNED by the reference's own
 *     seven known-answer cases, test/Symmetric3x3EigenvalueSolverTest.cxx:48-90
 *     (committed as tests/golden/eigen_kat.json).
 *   - everything wired from ITK classes (a3..a9: recursive Gaussian, divide,
 *     derivative operators, gradient magnitude, mask): PARITY UNPINNED.  ITK is
 *     a third-party dependency of the reference (find_package(ITK) with no
 *     version, CMakeLists.txt:14-16) that is absent from /root/reference and from
 *     this image; the reference holds no test or fixture at that boundary.  The
 *     restatement follows the published ITK 4.x algorithms named at each function.
 *
 * Layout: volumes are x-fastest (index = x + nx*(y + ny*z)), as itk::Image
 * buffers.  Vector outputs are interleaved [voxel*ncomp + c] as itk::VectorImage.
 */
#ifndef IFE_ORACLE_H
#define IFE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* trig_mode for the float solver (SURVEY TL;DR item 5): unqualified sqrt/acos/cos
 * in Symmetric3x3EigenvalueSolver.h:88,115,119-120 bind to the double C functions
 * when only <cmath> is visible (mode 0) and to the float overloads when <math.h>
 * is visible too (mode 1). */
enum { IFE_OR_TRIG_CMATH = 0, IFE_OR_TRIG_MATH_H = 1 };

/* ITK DerivativeImageFilter scales the operator by 1/spacing once, whatever the
 * order (mode 0, upstream itkDerivativeImageFilter.hxx); mode 1 scales by
 * 1/spacing^order.  Identical for unit spacing. */
enum { IFE_OR_DSCALE_ITK = 0, IFE_OR_DSCALE_POW = 1 };

typedef struct {
  int64_t nx, ny, nz;
  double sx, sy, sz;
} ife_or_dims;

typedef struct {
  double N0, N1, N2, N3;
  double D1, D2, D3, D4;
  double M1, M2, M3, M4;
  double BN1, BN2, BN3, BN4;
  double BM1, BM2, BM3, BM4;
} ife_or_gauss_coeffs;

void ife_or_set_threads(int n);
int ife_or_get_threads(void);

/* a1: Symmetric3x3EigenvalueSolver<T>::operator() */
void ife_or_eig3_f64(const double A[6], double ev[3]);
void ife_or_eig3_f32(const float A[6], float ev[3], int trig_mode);
/* a2: EigenvalueFeaturesFunctor<T>::operator() */
void ife_or_eigfeat_f64(const double A[6], double f[6]);
void ife_or_eigfeat_f32(const float A[6], float f[6], int trig_mode);
void ife_or_eig3_batch_f32(const float *A6, int64_t n, float *ev3, int trig_mode);
void ife_or_eigfeat_batch_f32(const float *A6, int64_t n, float *f6, int trig_mode);
void ife_or_eig3_batch_f64(const double *A6, int64_t n, double *ev3);

/* a4 pieces: itk::RecursiveGaussianImageFilter (ZeroOrder) */
int ife_or_gauss_coeffs_zero_order(double sigma, double spacing, ife_or_gauss_coeffs *c);
/* orders 0, 1, 2 of itk::RecursiveGaussianImageFilter::SetUp (row f4) */
int ife_or_gauss_coeffs_order(double sigma, double spacing, int order, ife_or_gauss_coeffs *c);
int ife_or_recursive_gaussian_axis_order(const float *in, float *out, const ife_or_dims *d,
                                         int axis, double sigma, int order);
int ife_or_differential_normalized_convolution(const float *image, const float *certainty,
                                               float *out, const ife_or_dims *d, double sigma,
                                               int axis);
void ife_or_iir_line(const double *data, double *outs, double *scratch, int64_t ln,
                     const ife_or_gauss_coeffs *c);
int ife_or_recursive_gaussian_axis(const float *in, float *out, const ife_or_dims *d,
                                   int axis, double sigma);
int ife_or_smoothing_recursive_gaussian(const float *in, float *out, const ife_or_dims *d,
                                        double sigma);
/* a4: NormalizedGaussianConvolutionImageFilter */
int ife_or_normalized_gaussian_convolution(const float *image, const float *certainty,
                                           float *out, const ife_or_dims *d, double sigma);

/* a3 pieces: itk::DerivativeImageFilter */
int ife_or_derivative(const float *in, float *out, const ife_or_dims *d, int order,
                      int direction, int dscale_mode);
/* a3: Hessian3DImageFilter, out interleaved 6 comps xx,xy,xz,yy,yz,zz */
int ife_or_hessian3d(const float *in, float *out6, const ife_or_dims *d, int dscale_mode);
/* itk::GradientMagnitudeImageFilter */
int ife_or_gradient_magnitude(const float *in, float *out, const ife_or_dims *d);

/* a5: ImageToEmphysemaFeaturesFilter, one sigma; out interleaved 8 comps */
int ife_or_emphysema_features(const float *image, const uint8_t *mask, float *out8,
                              const ife_or_dims *d, double sigma, int trig_mode,
                              int dscale_mode);
/* a6: FiniteDifference_HessianFeatures body (normative a3 o a2 o mask); out interleaved 6 */
int ife_or_fd_hessian_features(const float *image, const uint8_t *mask, float *out6,
                               const ife_or_dims *d, int trig_mode, int dscale_mode);
/* a7: FiniteDifference_GradientFeatures body */
int ife_or_fd_gradient_features(const float *image, const float *mask, float *out,
                                const ife_or_dims *d);
/* a8: MaskedImageFilter tool body (double pixels) */
void ife_or_mask_image_f64(const double *image, const double *mask, double outside,
                           double *out, int64_t n);

/* ---- rows f1/f2 (histogram edges and dense histograms).  PINNED: the reference's own
 * headers for these compile here (oracle/_ref, `make -C oracle _ref`) and its tests hold
 * known answers (test/DetermineEdgesForEqualizedHistogramTest.cxx:30-72,
 * test/DenseHistogramTest.cxx:10-55); tests/test_oracle_stats.py checks the restatement
 * against both and against tests/golden/stats_*.json generated from oracle/_ref. ---- */

/* f1: determineEdgesForEqualizedHistogram, include/ife/Statistics/
 * DetermineEdgesForEqualizedHistogram.h:21-137.  sorted[0..n) ascending; writes nbins-1
 * edges.  Returns 0; 1 when n < nbins (std::out_of_range there, :36-38); 3 when the walk
 * would step to or past the end (assert at :74; undefined behaviour in a release build
 * of the reference, an error here). */
int ife_or_equalized_edges_f32(const float *sorted, int64_t n, int64_t nbins, float *edges);
int ife_or_equalized_edges_f64(const double *sorted, int64_t n, int64_t nbins, double *edges);
/* std::sort of one sample column (tools/DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures.cxx:284) */
void ife_or_sort_f32(float *v, int64_t n);
/* the all-foreground sample gather of the same tool (:221-236): for every voxel whose mask
 * value equals one of fg[0..nfg), append its ncomp interleaved feature values to the
 * ncomp columns; columns[c] must hold the number of such voxels (returned). */
int64_t ife_or_gather_foreground(const float *features, int ncomp, const uint8_t *mask,
                                 int64_t nvox, const uint32_t *fg, int nfg, float **columns);
/* f2: DenseHistogram<float> (include/ife/Statistics/DenseHistogram.h:29-66): bins
 * (-inf,e0], (e0,e1], ..., (e_last, inf); counts and freqs hold nedges+1 entries. */
void ife_or_dense_histogram_f32(const float *edges, int64_t nedges, const float *values,
                                int64_t n, uint32_t *counts, float *freqs);

/* f2: the bag rows of tools/MakeBag.cxx:405-472 for one feature volume: for every box
 * rois[r] = {x, y, z, sx, sy, sz} and component c, insert the component value of every
 * voxel of the box whose mask is non-zero (raster order, :437-447) into a DenseHistogram
 * with edges[c*nedges ..]; counts[(r*ncomp + c)*(nedges+1) + bin], freqs likewise
 * (getFrequencies, :455-462; 0/0 = NaN for a box without mask voxels).  Returns 0, or 4 when
 * a box leaves the volume (RegionOfInterestImageFilter throws there). */
int ife_or_roi_histograms(const float *features, int ncomp, const uint8_t *mask,
                          const ife_or_dims *d, const int64_t *rois, int nrois,
                          const float *edges, int64_t nedges, uint32_t *counts, float *freqs);

#ifdef __cplusplus
}
#endif
#endif
