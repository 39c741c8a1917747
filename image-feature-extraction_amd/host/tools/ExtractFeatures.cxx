// ExtractFeatures -- multi-scale driver of the feature filter; same flags, flow and output
// names as the reference's tools/ExtractFeatures.cxx (:26-63 flags, :126-143 naming:
// <out>_scale_<std::to_string(scale)><FeatureName>.nii.gz).  The arithmetic runs on the
// MI355X through libife_hip.so; the eight files of a scale are compressed and written on
// worker threads while the device computes the next scale (ife/Host/AsyncWriter.h).
#include <iostream>

#include "tclap/CmdLine.h"

#include "ife/Filters/ImageToEmphysemaFeaturesFilter.h"
#include "ife/Host/AsyncWriter.h"
#include "ife/Host/ImageIO.h"
#include "ife/Host/LiteFilters.h"
#include "ife/Util/Path.h"

const std::string VERSION("0.1");

static std::string outFileType() {
  const char *e = std::getenv("IFE_OUT_FILE_TYPE");  // ".nii.gz" (reference), ".nii" or ".mhd"
  return e ? std::string(e) : std::string(".nii.gz");
}

int main(int argc, char *argv[]) {
  TCLAP::CmdLine cmd("Create a bag of instances samples from an image.", ' ', VERSION);
  TCLAP::ValueArg<std::string> imageArg("i", "image", "Path to image.", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> maskArg("m", "mask", "Path to mask.", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> outArg("o", "out", "Base output path", true, "", "path", cmd);
  TCLAP::MultiArg<float> scalesArg("s", "scale", "Scales for the Gauss applicability function", true,
                                   "double", cmd);
  try {
    cmd.parse(argc, argv);
  } catch (TCLAP::ArgException &e) {
    std::cerr << "Error : " << e.error() << " for arg " << e.argId() << std::endl;
    return EXIT_FAILURE;
  }
  const std::string imagePath(imageArg.getValue());
  const std::string maskPath(maskArg.getValue());
  const std::string outBasePath(outArg.getValue());
  const std::vector<float> scales(scalesArg.getValue());

  typedef float PixelType;
  typedef unsigned char MaskPixelType;
  const unsigned int Dimension = 3;
  typedef itk::Image<PixelType, Dimension> ImageType;
  typedef itk::Image<MaskPixelType, Dimension> MaskType;
  typedef itk::VectorImage<PixelType, Dimension> VectorImageType;

  typedef itk::ImageFileReader<ImageType> ReaderType;
  ReaderType::Pointer reader = ReaderType::New();
  reader->SetFileName(imagePath);
  typedef itk::ImageFileReader<MaskType> MaskReaderType;
  MaskReaderType::Pointer maskReader = MaskReaderType::New();
  maskReader->SetFileName(maskPath);

  typedef itk::ClampImageFilter<MaskType, MaskType> ClampFilterType;
  ClampFilterType::Pointer clampFilter = ClampFilterType::New();
  clampFilter->InPlaceOn();
  clampFilter->SetBounds(0, 1);

  typedef itk::ImageToEmphysemaFeaturesFilter<ImageType, MaskType, VectorImageType> FeatureFilterType;
  FeatureFilterType::Pointer featureFilter = FeatureFilterType::New();
  typedef itk::VectorIndexSelectionCastImageFilter<VectorImageType, ImageType> IndexSelectionType;
  IndexSelectionType::Pointer indexSelectionFilter = IndexSelectionType::New();
  ife::host::AsyncWriter<ImageType> writers;

  std::vector<std::string> featureNames{"GaussianBlur", "GradientMagnitude", "Eigenvalue1",
                                        "Eigenvalue2",  "Eigenvalue3",       "LaplacianOfGaussian",
                                        "GaussianCurvature", "FrobeniusNorm"};
  std::string outPath;
  try {
    clampFilter->SetInput(maskReader->GetOutput());
    featureFilter->SetInputImage(reader->GetOutput());
    featureFilter->SetInputMask(clampFilter->GetOutput());
    indexSelectionFilter->SetInput(featureFilter->GetOutput());
    featureFilter->SetScales(scales);  // all scales start on the device at the first Update()
    for (auto scale : scales) {
      featureFilter->SetSigma(scale);
      for (unsigned int i = 0; i < featureNames.size(); ++i) {
        indexSelectionFilter->SetIndex(i);
        outPath = outBasePath + "_scale_" + std::to_string(scale) + featureNames[i] + outFileType();
        featureFilter->UpdateLargestPossibleRegion();
        indexSelectionFilter->Update();
        // hand the selected component to the writers; the filter makes a new image next time
        writers.Submit(indexSelectionFilter->DetachOutput(), outPath);
      }
    }
    writers.Wait();
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
