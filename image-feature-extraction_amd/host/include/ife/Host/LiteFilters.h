// LiteFilters.h -- the remaining ITK filters the hot-path tools wire around the feature
// filter, reduced to what those tools use (eager, no pipeline): ClampImageFilter
// (ExtractFeatures.cxx:99-104), VectorIndexSelectionCastImageFilter (:116-119),
// MaskImageFilter (MaskedImageFilter.cxx:85-93, FiniteDifference_GradientFeatures.cxx:108-113),
// GradientMagnitudeImageFilter (FiniteDifference_GradientFeatures.cxx:104-106).
#ifndef IFE_HOST_LITEFILTERS_H
#define IFE_HOST_LITEFILTERS_H

#include <algorithm>
#include <type_traits>

#include "ife/Host/Engine.h"

namespace itk {

template <typename TIn, typename TOut>
class ClampImageFilter {
 public:
  typedef ClampImageFilter Self;
  ifeNewMacro(Self);
  void InPlaceOn() {}
  void SetBounds(typename TOut::PixelType lo, typename TOut::PixelType hi) { lo_ = lo; hi_ = hi; }
  void SetInput(TIn *in) { in_ = in; }
  void Update() {
    if (!in_) throw ExceptionObject("Input is required", "ClampImageFilter");
    if (out_.IsNull()) out_ = TOut::New();
    out_->CopyInformation(in_);
    out_->Allocate();
    const size_t n = (size_t)in_->GetLargestPossibleRegion().GetNumberOfPixels();
    const typename TIn::PixelType *s = in_->GetBufferPointer();
    typename TOut::PixelType *d = out_->GetBufferPointer();
    for (size_t i = 0; i < n; ++i) {
      const double v = (double)s[i];
      d[i] = v < (double)lo_ ? lo_ : (v > (double)hi_ ? hi_ : (typename TOut::PixelType)s[i]);
    }
  }
  TOut *GetOutput() { Update(); return out_.GetPointer(); }

 private:
  TIn *in_ = nullptr;
  typename TOut::PixelType lo_ = 0, hi_ = 0;
  typename TOut::Pointer out_;
};

template <typename TVectorImage, typename TImage>
class VectorIndexSelectionCastImageFilter {
 public:
  typedef VectorIndexSelectionCastImageFilter Self;
  ifeNewMacro(Self);
  void SetInput(const TVectorImage *in) { in_ = in; }
  void SetIndex(unsigned int i) { index_ = i; }
  void Update() {
    if (!in_) throw ExceptionObject("Input is required", "VectorIndexSelectionCastImageFilter");
    const unsigned int nc = in_->GetNumberOfComponentsPerPixel();
    if (index_ >= nc) throw ExceptionObject("Selected index is out of range", "VectorIndexSelectionCastImageFilter");
    if (out_.IsNull()) out_ = TImage::New();
    out_->CopyInformation(in_);
    out_->Allocate();
    const size_t n = (size_t)in_->GetLargestPossibleRegion().GetNumberOfPixels();
    const typename TVectorImage::InternalPixelType *s = in_->GetBufferPointer();
    typename TImage::PixelType *d = out_->GetBufferPointer();
    for (size_t i = 0; i < n; ++i) d[i] = (typename TImage::PixelType)s[i * nc + index_];
  }
  TImage *GetOutput() { return out_.IsNull() ? (out_ = TImage::New()).GetPointer() : out_.GetPointer(); }
  // give the output image away (the next Update allocates a new one): for writers that
  // outlive this filter's next execution
  typename TImage::Pointer DetachOutput() {
    typename TImage::Pointer p = out_;
    out_ = typename TImage::Pointer();
    return p;
  }

 private:
  const TVectorImage *in_ = nullptr;
  unsigned int index_ = 0;
  typename TImage::Pointer out_;
};

// mask != 0 ? image : outside.  Double images run on the device (ife_mask_image_f64, the
// MaskedImageFilter tool); other pixel types are a host loop (plumbing, not hot path).
template <typename TImage, typename TMask, typename TOut = TImage>
class MaskImageFilter {
 public:
  typedef MaskImageFilter Self;
  ifeNewMacro(Self);
  void SetInput1(const TImage *a) { a_ = a; }
  void SetInput(const TImage *a) { a_ = a; }
  void SetInput2(const TMask *m) { m_ = m; }
  void SetMaskImage(const TMask *m) { m_ = m; }
  void SetOutsideValue(typename TOut::PixelType v) { outside_ = v; }
  void Update() {
    if (!a_ || !m_) throw ExceptionObject("Image and mask are required", "MaskImageFilter");
    ife::host::same_size(*a_, *m_, "MaskImageFilter");
    if (out_.IsNull()) out_ = TOut::New();
    out_->CopyInformation(a_);
    out_->Allocate();
    const size_t n = (size_t)a_->GetLargestPossibleRegion().GetNumberOfPixels();
    run(n, std::integral_constant<bool, std::is_same<typename TImage::PixelType, double>::value &&
                                            std::is_same<typename TMask::PixelType, double>::value &&
                                            std::is_same<typename TOut::PixelType, double>::value>());
  }
  TOut *GetOutput() { return out_.IsNull() ? (out_ = TOut::New()).GetPointer() : out_.GetPointer(); }

 private:
  void run(size_t n, std::true_type) {
    ife::host::Engine &e = ife::host::Engine::Instance();
    e.check(ife_mask_image_f64(e.ctx(), a_->GetBufferPointer(), m_->GetBufferPointer(), outside_,
                               (int64_t)n, out_->GetBufferPointer(), IFE_MEM_HOST),
            "MaskImageFilter");
  }
  void run(size_t n, std::false_type) {
    const typename TImage::PixelType *s = a_->GetBufferPointer();
    const typename TMask::PixelType *m = m_->GetBufferPointer();
    typename TOut::PixelType *d = out_->GetBufferPointer();
    for (size_t i = 0; i < n; ++i) d[i] = m[i] != 0 ? (typename TOut::PixelType)s[i] : outside_;
  }
  const TImage *a_ = nullptr;
  const TMask *m_ = nullptr;
  typename TOut::PixelType outside_ = 0;
  typename TOut::Pointer out_;
};

template <typename TIn, typename TOut>
class GradientMagnitudeImageFilter {
 public:
  typedef GradientMagnitudeImageFilter Self;
  ifeNewMacro(Self);
  void SetInput(const TIn *in) { in_ = in; }
  void Update() {
    if (!in_) throw ExceptionObject("Input is required", "GradientMagnitudeImageFilter");
    ife::host::Engine &e = ife::host::Engine::Instance();
    const ife_volume_desc d = ife::host::describe(*in_);
    if (out_.IsNull()) out_ = TOut::New();
    out_->CopyInformation(in_);
    out_->Allocate();
    e.check(ife_gradient_magnitude(e.ctx(), in_->GetBufferPointer(), &d, out_->GetBufferPointer(),
                                   IFE_MEM_HOST),
            "GradientMagnitudeImageFilter");
  }
  TOut *GetOutput() { return out_.IsNull() ? (out_ = TOut::New()).GetPointer() : out_.GetPointer(); }

 private:
  const TIn *in_ = nullptr;
  typename TOut::Pointer out_;
};

}  // namespace itk

#endif
