// NormalizedGaussianConvolutionImageFilter.h -- host mirror of
// include/ife/Filters/NormalizedGaussianConvolutionImageFilter.h:86-93 (.hxx:40-63):
// U = G_sigma(T*c) / G_sigma(c), forwarded to ife_normalized_gaussian_convolution.
// One addition: SetDerivativeDirection(0|1|2) selects the differential form the reference
// sketches in its header comment (.h:28-44), d/dx_axis of U with the derivative taken on the
// Gaussian (ife_differential_normalized_convolution); -1 (default) is the reference's filter.
#ifndef NormalizedGaussianConvolutionImageFilter_h
#define NormalizedGaussianConvolutionImageFilter_h

#include "ife/Host/Engine.h"

namespace itk {

template <typename TImage>
class NormalizedGaussianConvolutionImageFilter {
 public:
  typedef NormalizedGaussianConvolutionImageFilter Self;
  typedef TImage ImageType;
  typedef double ScalarRealType;  // the Gaussian filter's ScalarRealType, .h:75
  ifeNewMacro(Self);
  void SetInputImage(const TImage *image) { image_ = image; dirty_ = true; }
  void SetInputCertainty(const TImage *c) { cert_ = c; dirty_ = true; }
  void SetSigma(ScalarRealType s) { if (s != sigma_) { sigma_ = s; dirty_ = true; } }
  ScalarRealType GetSigma() const { return sigma_; }
  void SetDerivativeDirection(int axis) { if (axis != daxis_) { daxis_ = axis; dirty_ = true; } }
  int GetDerivativeDirection() const { return daxis_; }
  void Update() {
    if (!dirty_ && out_.IsNotNull()) return;
    if (!image_ || !cert_)
      throw ExceptionObject("Input image and certainty are required",
                            "NormalizedGaussianConvolutionImageFilter");
    ife::host::same_size(*image_, *cert_, "NormalizedGaussianConvolutionImageFilter");
    ife::host::Engine &e = ife::host::Engine::Instance();
    const ife_volume_desc d = ife::host::describe(*image_);
    if (out_.IsNull()) out_ = TImage::New();
    out_->CopyInformation(image_);
    out_->Allocate();
    if (daxis_ >= 0)
      e.check(ife_differential_normalized_convolution(e.ctx(), image_->GetBufferPointer(),
                                                      cert_->GetBufferPointer(), &d, sigma_, daxis_,
                                                      out_->GetBufferPointer(), IFE_MEM_HOST),
              "NormalizedGaussianConvolutionImageFilter");
    else
      e.check(ife_normalized_gaussian_convolution(e.ctx(), image_->GetBufferPointer(),
                                                  cert_->GetBufferPointer(), &d, sigma_,
                                                  out_->GetBufferPointer(), IFE_MEM_HOST),
              "NormalizedGaussianConvolutionImageFilter");
    dirty_ = false;
  }
  void UpdateLargestPossibleRegion() { Update(); }
  TImage *GetOutput() {
    if (out_.IsNull()) out_ = TImage::New();
    return out_.GetPointer();
  }

 private:
  const TImage *image_ = nullptr, *cert_ = nullptr;
  ScalarRealType sigma_ = 1.0;  // .hxx:18
  int daxis_ = -1;
  bool dirty_ = true;
  typename TImage::Pointer out_;
};

}  // namespace itk

#endif
