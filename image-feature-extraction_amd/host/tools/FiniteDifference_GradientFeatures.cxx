// FiniteDifference_GradientFeatures -- masked gradient magnitude; flags and output name of
// the reference's tools/FiniteDifference_GradientFeatures.cxx (:31-68 flags -i -m -o -p,
// default prefix "gradient_"; :121-125 name <outdir>/<prefix>GradientMagnitude.nii.gz).
// As there, the mask is read as a float image (:101-102).
#include <iostream>

#include "tclap/CmdLine.h"

#include "ife/Host/Engine.h"
#include "ife/Host/ImageIO.h"
#include "ife/Util/Path.h"

const std::string VERSION("0.1");

int main(int argc, char *argv[]) {
  TCLAP::CmdLine cmd("Calculate gradient based features.", ' ', VERSION);
  TCLAP::ValueArg<std::string> imageArg("i", "image", "Path to image.", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> maskArg("m", "mask", "Path to mask. Must match image dimensions.", true,
                                       "", "path", cmd);
  TCLAP::ValueArg<std::string> outDirArg("o", "outdir", "Path to output directory", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> prefixArg("p", "prefix", "Prefix to use for output filenames", false,
                                         "gradient_", "string", cmd);
  try {
    cmd.parse(argc, argv);
  } catch (TCLAP::ArgException &e) {
    std::cerr << "Error : " << e.error() << " for arg " << e.argId() << std::endl;
    return EXIT_FAILURE;
  }
  const std::string imagePath(imageArg.getValue()), maskPath(maskArg.getValue());
  const std::string outDirPath(outDirArg.getValue()), prefix(prefixArg.getValue());
  const char *ft = std::getenv("IFE_OUT_FILE_TYPE");
  const std::string OUT_FILE_TYPE(ft ? ft : ".nii.gz");
  typedef itk::Image<float, 3> ImageType;
  const std::string baseFileName = Path::join(outDirPath, prefix);
  try {
    itk::ImageFileReader<ImageType>::Pointer imageReader = itk::ImageFileReader<ImageType>::New();
    imageReader->SetFileName(imagePath);
    itk::ImageFileReader<ImageType>::Pointer maskReader = itk::ImageFileReader<ImageType>::New();
    maskReader->SetFileName(maskPath);
    ImageType *image = imageReader->GetOutput();
    ImageType *mask = maskReader->GetOutput();
    ife::host::same_size(*image, *mask, "FiniteDifference_GradientFeatures");
    ImageType::Pointer out = ImageType::New();
    out->CopyInformation(image);
    out->Allocate();
    ife::host::Engine &e = ife::host::Engine::Instance();
    const ife_volume_desc d = ife::host::describe(*image);
    e.check(ife_fd_gradient_features(e.ctx(), image->GetBufferPointer(), mask->GetBufferPointer(), &d,
                                     out->GetBufferPointer(), IFE_MEM_HOST),
            "FiniteDifference_GradientFeatures");
    itk::ImageFileWriter<ImageType>::Pointer writer = itk::ImageFileWriter<ImageType>::New();
    writer->SetInput(out);
    writer->SetFileName(baseFileName + "GradientMagnitude" + OUT_FILE_TYPE);
    writer->Update();
  } catch (itk::ExceptionObject &e) {
    std::cerr << "Failed to process." << std::endl
              << "Image: " << imagePath << std::endl
              << "Mask: " << maskPath << std::endl
              << "Base file name: " << baseFileName << std::endl
              << "ExceptionObject: " << e << std::endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
