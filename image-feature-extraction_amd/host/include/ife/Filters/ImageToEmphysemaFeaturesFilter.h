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
//
// One addition to the reference's interface: SetScales(all sigmas the caller is about to
// visit).  With that hint the first Update() uploads image and mask once, runs Cast +
// Multiply once and enqueues every scale on the device (ife_emphysema_features_begin); each
// Update() then only fetches its scale, so writing scale k overlaps the device work of the
// later ones.  Without the hint every scale is one self-contained call, as in the reference.
// The same holds with IFE_DEVICES (several devices, Z-slabs): one upload of the slabs and one
// prepass for all announced scales (ife_multi_emphysema_features_begin), a fetch per scale.
#ifndef __ImageToEmphysemaFeaturesFilter_h
#define __ImageToEmphysemaFeaturesFilter_h

#include <vector>

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

  void SetInputImage(const InputImageType *image) { EndStream(); image_ = image; dirty_ = true; }
  void SetInputMask(const InputMaskType *mask) { EndStream(); mask_ = mask; dirty_ = true; }
  ~ImageToEmphysemaFeaturesFilter() { EndStream(); }
  void SetSigma(ScalarRealType s) { if (s != sigma_) { sigma_ = s; dirty_ = true; } }
  void SetScales(const std::vector<ScalarRealType> &scales) {
    EndStream();
    scales_.assign(scales.begin(), scales.end());
    dirty_ = true;
  }
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
    int which = -1;
    for (size_t k = 0; k < scales_.size(); ++k)
      if (scales_[k] == (float)sigma_) { which = (int)k; break; }
    if (ife_multi *multi = e.multi()) {  // IFE_DEVICES: Z-slabs over several devices
      const int idt = ife::host::ImageDType<PixelType>::value;
      const int mdt = ife::host::MaskDType<typename InputMaskType::PixelType>::value;
      if (which >= 0) {  // the announced schedule: one upload and prepass, every scale enqueued, fetched per scale
        if (!streaming_) {
          e.check_multi(ife_multi_emphysema_features_begin(multi, image_->GetBufferPointer(), idt,
                                                           mask_->GetBufferPointer(), mdt, &d, scales_.data(),
                                                           (int)scales_.size(), IFE_INTERLEAVED),
                        "ImageToEmphysemaFeaturesFilter");
          streaming_ = true;
        }
        e.check_multi(ife_multi_emphysema_features_fetch(multi, which, out_->GetBufferPointer()),
                      "ImageToEmphysemaFeaturesFilter");
      } else {
        const float sig1 = (float)sigma_;
        e.check_multi(ife_multi_emphysema_features(multi, image_->GetBufferPointer(), idt, mask_->GetBufferPointer(),
                                                   mdt, &d, &sig1, 1, out_->GetBufferPointer(), IFE_INTERLEAVED),
                      "ImageToEmphysemaFeaturesFilter");
      }
      dirty_ = false;
      return;
    }
    if (which >= 0) {  // the announced schedule: everything is started once, fetched per scale
      if (!streaming_) {
        e.check(ife_emphysema_features_begin(
                    e.ctx(), image_->GetBufferPointer(), ife::host::ImageDType<PixelType>::value,
                    mask_->GetBufferPointer(),
                    ife::host::MaskDType<typename InputMaskType::PixelType>::value, &d,
                    scales_.data(), (int)scales_.size(), IFE_INTERLEAVED),
                "ImageToEmphysemaFeaturesFilter");
        streaming_ = true;
      }
      e.check(ife_emphysema_features_fetch(e.ctx(), which, out_->GetBufferPointer()),
              "ImageToEmphysemaFeaturesFilter");
      dirty_ = false;
      return;
    }
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
  void EndStream() {
    if (streaming_) {
      ife::host::Engine &e = ife::host::Engine::Instance();
      if (ife_multi *multi = e.multi()) (void)ife_multi_emphysema_features_end(multi);
      else (void)ife_emphysema_features_end(e.ctx());
      streaming_ = false;
    }
  }
  std::vector<float> scales_;
  bool streaming_ = false;
  const InputImageType *image_ = nullptr;
  const InputMaskType *mask_ = nullptr;
  ScalarRealType sigma_ = 1.0;  // .hxx:18
  bool dirty_ = true;
  typename OutputImageType::Pointer out_;
};

}  // namespace itk

#endif
