// Hessian3DImageFilter.h -- host mirror of include/ife/Filters/Hessian3DImageFilter.h
// (SetInput / Update / GetOutput, :23-28,48; wiring .hxx:13-60): six components
// xx,xy,xz,yy,yz,zz per voxel, forwarded to ife_hessian3d.
#ifndef __Hessian3DImageFilter_h
#define __Hessian3DImageFilter_h

#include "ife/Host/Engine.h"

namespace itk {

template <typename TInputImage, typename TOutputImage = VectorImage<typename TInputImage::PixelType, 3> >
class Hessian3DImageFilter {
 public:
  typedef Hessian3DImageFilter Self;
  typedef TInputImage InputImageType;
  typedef TOutputImage OutputImageType;
  ifeNewMacro(Self);
  void SetInput(const InputImageType *image) { in_ = image; dirty_ = true; }
  void Update() {
    if (!dirty_ && out_.IsNotNull()) return;
    if (!in_) throw ExceptionObject("Input is required", "Hessian3DImageFilter");
    ife::host::Engine &e = ife::host::Engine::Instance();
    const ife_volume_desc d = ife::host::describe(*in_);
    if (out_.IsNull()) out_ = OutputImageType::New();
    out_->CopyInformation(in_);
    out_->SetNumberOfComponentsPerPixel(6);  // .hxx:66-73
    out_->Allocate();
    e.check(ife_hessian3d(e.ctx(), in_->GetBufferPointer(), &d, out_->GetBufferPointer(),
                          IFE_INTERLEAVED, IFE_MEM_HOST),
            "Hessian3DImageFilter");
    dirty_ = false;
  }
  void UpdateLargestPossibleRegion() { Update(); }
  OutputImageType *GetOutput() {
    if (out_.IsNull()) out_ = OutputImageType::New();
    return out_.GetPointer();
  }

 private:
  const InputImageType *in_ = nullptr;
  bool dirty_ = true;
  typename OutputImageType::Pointer out_;
};

}  // namespace itk

#endif
