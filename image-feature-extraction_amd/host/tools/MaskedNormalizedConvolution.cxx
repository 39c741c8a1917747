// MaskedNormalizedConvolution -- normalized Gaussian convolution at several scales; flags
// and output names of the reference's tools/MaskedNormalizedConvolution.cxx (:45-102 flags
// -i -c -s... -o -p -m, default prefix "normconv_"; :187-188 name
// <outdir>/<prefix>scale_<std::to_string(scale)>.nii.gz; scales are doubles, :117).
#include <iostream>

#include "tclap/CmdLine.h"

#include "ife/Filters/NormalizedGaussianConvolutionImageFilter.h"
#include "ife/Host/ImageIO.h"
#include "ife/Host/LiteFilters.h"
#include "ife/Util/Path.h"

const std::string VERSION("0.1");

int main(int argc, char *argv[]) {
  TCLAP::CmdLine cmd("Perform normalized convolution of an image with a Gaussian.", ' ', VERSION);
  TCLAP::ValueArg<std::string> imageArg("i", "image", "Path to image.", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> certaintyArg("c", "certainty",
                                            "Path to certainty image. Must match image dimensions.", true,
                                            "", "path", cmd);
  TCLAP::MultiArg<double> scalesArg("s", "scale", "Scales for the Gauss applicability function", true,
                                    "double", cmd);
  TCLAP::ValueArg<std::string> outDirArg("o", "outdir", "Path to output directory", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> prefixArg("p", "prefix", "Prefix to use for output filenames", false,
                                         "normconv_", "string", cmd);
  TCLAP::ValueArg<bool> maskOutputArg("m", "maskoutput", "Mask the output after convolution.", false, false,
                                      "boolean", cmd);
  try {
    cmd.parse(argc, argv);
  } catch (TCLAP::ArgException &e) {
    std::cerr << "Error : " << e.error() << " for arg " << e.argId() << std::endl;
    return EXIT_FAILURE;
  }
  const std::string imagePath(imageArg.getValue()), certaintyPath(certaintyArg.getValue());
  const std::string outDirPath(outDirArg.getValue()), prefix(prefixArg.getValue());
  std::vector<double> scales(scalesArg.getValue());
  const bool maskOutput(maskOutputArg.getValue());
  const char *ft = std::getenv("IFE_OUT_FILE_TYPE");
  const std::string OUT_FILE_TYPE(ft ? ft : ".nii.gz");

  typedef itk::Image<float, 3> ImageType;
  const std::string baseFileName = Path::join(outDirPath, prefix);
  double current = 0;
  try {
    itk::ImageFileReader<ImageType>::Pointer imageReader = itk::ImageFileReader<ImageType>::New();
    imageReader->SetFileName(imagePath);
    itk::ImageFileReader<ImageType>::Pointer certaintyReader = itk::ImageFileReader<ImageType>::New();
    certaintyReader->SetFileName(certaintyPath);
    typedef itk::NormalizedGaussianConvolutionImageFilter<ImageType> FilterType;
    FilterType::Pointer normConvFilter = FilterType::New();
    normConvFilter->SetInputImage(imageReader->GetOutput());
    normConvFilter->SetInputCertainty(certaintyReader->GetOutput());
    typedef itk::MaskImageFilter<ImageType, ImageType, ImageType> MaskFilterType;
    MaskFilterType::Pointer maskFilter = MaskFilterType::New();
    itk::ImageFileWriter<ImageType>::Pointer writer = itk::ImageFileWriter<ImageType>::New();
    for (auto scale : scales) {
      current = scale;
      std::cout << "Processing scale " << scale << std::endl;
      normConvFilter->SetSigma(scale);
      normConvFilter->Update();
      if (maskOutput) {
        maskFilter->SetInput1(normConvFilter->GetOutput());
        maskFilter->SetInput2(certaintyReader->GetOutput());
        maskFilter->Update();
        writer->SetInput(maskFilter->GetOutput());
      } else {
        writer->SetInput(normConvFilter->GetOutput());
      }
      writer->SetFileName(baseFileName + "scale_" + std::to_string(scale) + OUT_FILE_TYPE);
      writer->Update();
    }
  } catch (itk::ExceptionObject &e) {
    std::cerr << "Failed to process." << std::endl
              << "Image: " << imagePath << std::endl
              << "Certainty: " << certaintyPath << std::endl
              << "Scale: " << current << std::endl
              << "Base file name: " << baseFileName << std::endl
              << "ExceptionObject: " << e << std::endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
