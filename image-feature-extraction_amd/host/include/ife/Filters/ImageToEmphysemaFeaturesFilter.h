// ImageToEmphysemaFeaturesFilter.h -- host mirror of the reference's composite filter
// (include/ife/Filters/ImageToEmphysemaFeaturesFilter.h:44-64, .hxx:15-55,99-121): same
// class name, template parameters and methods, forwarding to ife_emphysema_features.
//
//   typedef itk::ImageToEmphysemaFeaturesFilter<ImageType, MaskType, VectorImageType> F;
//   F::Pointer f = F::New();
//   f->SetInputImage(image); f->SetInputMask(mask);
//   for (auto s : scales) { f->SetSigma(s); f->UpdateLargestPossibleRegion(); f->GetOutput(); }
//
// As with ITK's MTime logic, Update() re-executes only after an input or sigma changed, so
// the eight Update() calls per scale of tools/ExtractFeatures.cxx:135-143 cost one
// execution.
#ifndef __ImageToEmphysemaFeaturesFilter_h
#define __ImageToEmphysemaFeaturesFilter_h

#include "ife/Host/Engine.h"

namespace itk {

template <typename TInputImage, typename TInputMask, typename TOutputImage>
class ImageToEmphysemaFeaturesFilter {
 public:
  typedef ImageToEmphysemaFeaturesFilter Self;
  typedef TInputImage InputImageType;
  typedef TInputMask InputMaskType;
  typedef TOutputImage OutputImageType;
  typedef typename InputImageType::PixelType PixelType;
  typedef PixelType ScalarRealType;  // .h:41
  ifeNewMacro(Self);

  void SetInputImage(const InputImageType *image) { image_ = image; dirty_ = true; }
  void SetInputMask(const InputMaskType *mask) { mask_ = mask; dirty_ = true; }
  void SetSigma(ScalarRealType s) { if (s != sigma_) { sigma_ = s; dirty_ = true; } }
  ScalarRealType GetSigma() const { return sigma_; }
  static const size_t numFeatures = 8;  // .h:62

  void Modified() { dirty_ = true; }
  void Update() {
    if (!dirty_ && out_.IsNotNull()) return;
    if (!image_ || !mask_)
      throw ExceptionObject("Input image and mask are required", "ImageToEmphysemaFeaturesFilter");
    ife::host::same_size(*image_, *mask_, "ImageToEmphysemaFeaturesFilter");
    ife::host::Engine &e = ife::host::Engine::Instance();
    const ife_volume_desc d = ife::host::describe(*image_);
    if (out_.IsNull()) out_ = OutputImageType::New();
    out_->CopyInformation(image_);
    out_->SetNumberOfComponentsPerPixel(numFeatures);  // .hxx:83-90
    out_->Allocate();
    const float sig = (float)sigma_;
    e.check(ife_emphysema_features(
                e.ctx(), image_->GetBufferPointer(), ife::host::ImageDType<PixelType>::value,
                mask_->GetBufferPointer(),
                ife::host::MaskDType<typename InputMaskType::PixelType>::value, &d, &sig, 1,
                out_->GetBufferPointer(), IFE_INTERLEAVED, IFE_MEM_HOST),
            "ImageToEmphysemaFeaturesFilter");
    dirty_ = false;
  }
  void UpdateLargestPossibleRegion() { Update(); }
  OutputImageType *GetOutput() {
    if (out_.IsNull()) out_ = OutputImageType::New();
    return out_.GetPointer();
  }

 private:
  const InputImageType *image_ = nullptr;
  const InputMaskType *mask_ = nullptr;
  ScalarRealType sigma_ = 1.0;  // .hxx:18
  bool dirty_ = true;
  typename OutputImageType::Pointer out_;
};

}  // namespace itk

#endif
