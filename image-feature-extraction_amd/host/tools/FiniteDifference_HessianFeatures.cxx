// FiniteDifference_HessianFeatures -- un-smoothed Hessian + eigenvalue features; flags and
// output names of the reference's tools/FiniteDifference_HessianFeatures.cxx (:45-82 flags
// -i -m -o -p, default prefix "hessian_"; :253-264 names eig1, eig2, eig3, LoG, Curvature,
// Frobenius).  The reference tool is dead code (tools/CMakeLists.txt:32) with a direction
// slip at :155; this one computes the normative a3 o a2 o mask on the MI355X.
#include <iostream>

#include "tclap/CmdLine.h"

#include "ife/Host/Engine.h"
#include "ife/Host/ImageIO.h"
#include "ife/Util/Path.h"

const std::string VERSION("0.1");

int main(int argc, char *argv[]) {
  TCLAP::CmdLine cmd("Calculate Hessian based features.", ' ', VERSION);
  TCLAP::ValueArg<std::string> imageArg("i", "image", "Path to image.", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> maskArg("m", "mask", "Path to mask. Must match image dimensions.", true,
                                       "", "path", cmd);
  TCLAP::ValueArg<std::string> outDirArg("o", "outdir", "Path to output directory", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> prefixArg("p", "prefix", "Prefix to use for output filenames", false,
                                         "hessian_", "string", cmd);
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
  typedef itk::Image<unsigned char, 3> MaskType;
  const std::string baseFileName = Path::join(outDirPath, prefix);
  try {
    itk::ImageFileReader<ImageType>::Pointer imageReader = itk::ImageFileReader<ImageType>::New();
    imageReader->SetFileName(imagePath);
    itk::ImageFileReader<MaskType>::Pointer maskReader = itk::ImageFileReader<MaskType>::New();
    maskReader->SetFileName(maskPath);
    ImageType *image = imageReader->GetOutput();
    MaskType *mask = maskReader->GetOutput();
    ife::host::same_size(*image, *mask, "FiniteDifference_HessianFeatures");

    const size_t n = (size_t)image->GetLargestPossibleRegion().GetNumberOfPixels();
    std::vector<float> planar(n * 6);
    ife::host::Engine &e = ife::host::Engine::Instance();
    const ife_volume_desc d = ife::host::describe(*image);
    e.check(ife_fd_hessian_features(e.ctx(), image->GetBufferPointer(), IFE_F32,
                                    mask->GetBufferPointer(), IFE_U8, &d, planar.data(), IFE_PLANAR,
                                    IFE_MEM_HOST),
            "FiniteDifference_HessianFeatures");

    std::vector<std::string> featureNames{"eig1", "eig2", "eig3", "LoG", "Curvature", "Frobenius"};
    ImageType::Pointer comp = ImageType::New();
    comp->CopyInformation(image);
    comp->Allocate();
    itk::ImageFileWriter<ImageType>::Pointer writer = itk::ImageFileWriter<ImageType>::New();
    writer->SetInput(comp);
    for (unsigned int i = 0; i < featureNames.size(); ++i) {
      std::copy(planar.begin() + (std::ptrdiff_t)(i * n), planar.begin() + (std::ptrdiff_t)((i + 1) * n),
                comp->GetBufferPointer());
      writer->SetFileName(baseFileName + featureNames.at(i) + OUT_FILE_TYPE);
      writer->Update();
    }
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
