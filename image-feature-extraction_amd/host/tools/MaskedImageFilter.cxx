// MaskedImageFilter -- mask != 0 ? image : outside value, double pixels; flags of the
// reference's tools/MaskedImageFilter.cxx (:23-51: -i -m -o, -v/--outside-value default 0).
#include <iostream>

#include "tclap/CmdLine.h"

#include "ife/Host/ImageIO.h"
#include "ife/Host/LiteFilters.h"

const std::string VERSION("0.1");

int main(int argc, char *argv[]) {
  TCLAP::CmdLine cmd("Mask an image.", ' ', VERSION);
  TCLAP::ValueArg<std::string> imageArg("i", "image", "Path to image.", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> maskArg("m", "mask", "Path to mask.", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> outArg("o", "out", "Output path", true, "", "path", cmd);
  TCLAP::ValueArg<double> outsideArg("v", "outside-value", "Value to use outside mask", false, 0, "double",
                                     cmd);
  try {
    cmd.parse(argc, argv);
  } catch (TCLAP::ArgException &e) {
    std::cerr << "Error : " << e.error() << " for arg " << e.argId() << std::endl;
    return EXIT_FAILURE;
  }
  const std::string imagePath(imageArg.getValue()), maskPath(maskArg.getValue()), outPath(outArg.getValue());
  const double outsideValue(outsideArg.getValue());
  typedef itk::Image<double, 3> ImageType;
  try {
    itk::ImageFileReader<ImageType>::Pointer imageReader = itk::ImageFileReader<ImageType>::New();
    imageReader->SetFileName(imagePath);
    itk::ImageFileReader<ImageType>::Pointer maskReader = itk::ImageFileReader<ImageType>::New();
    maskReader->SetFileName(maskPath);
    typedef itk::MaskImageFilter<ImageType, ImageType, ImageType> MaskFilterType;
    MaskFilterType::Pointer maskFilter = MaskFilterType::New();
    maskFilter->SetOutsideValue(outsideValue);
    maskFilter->SetInput1(imageReader->GetOutput());
    maskFilter->SetInput2(maskReader->GetOutput());
    maskFilter->Update();
    itk::ImageFileWriter<ImageType>::Pointer writer = itk::ImageFileWriter<ImageType>::New();
    writer->SetInput(maskFilter->GetOutput());
    writer->SetFileName(outPath);
    writer->Update();
  } catch (itk::ExceptionObject &e) {
    std::cerr << "Failed to process." << std::endl
              << "Image: " << imagePath << std::endl
              << "Mask: " << maskPath << std::endl
              << "Out: " << outPath << std::endl
              << "ExceptionObject: " << e << std::endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
