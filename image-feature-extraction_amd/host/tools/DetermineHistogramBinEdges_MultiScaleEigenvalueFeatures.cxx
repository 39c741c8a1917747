// DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures -- equal-frequency histogram bin
// edges of the eight multi-scale features over a population of image/mask pairs; flags,
// output file and exit codes of the reference's tool of the same name
// (tools/DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures.cxx:33-296):
//
//   -i list.csv   lines "image,mask"          -o edges.txt
//   -b bins       -S samples per image (0 = every foreground voxel)
//   -s scale ...  -f foreground label ...
//
// Per image the features of all scales are computed and sampled on the device; the sample
// columns stay in HBM across images (ife_samples_*), and only the (bins-1) edges of each of
// the scales*8 columns come back.  The sampled branch draws its voxel positions on the host
// (uniform over the volume, with replacement, kept when the label is a foreground value,
// until -S positions are kept; a fresh draw per scale as the reference's random iterator
// gives).  The reference seeds from std::random_device; IFE_SEED fixes the seed here.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <random>
#include <string>
#include <vector>

#include "tclap/CmdLine.h"

#include "ife/Filters/ImageToEmphysemaFeaturesFilter.h"
#include "ife/Host/ImageIO.h"
#include "ife/IO/IO.h"
#include "ife/Statistics/DetermineEdgesForEqualizedHistogram.h"

const std::string VERSION("0.1");

namespace {
struct SamplesHandle {  // ife_samples with scope-bound lifetime
  ife_samples *s = nullptr;
  ~SamplesHandle() { ife_samples_destroy(s); }
};
}  // namespace

int main(int argc, char *argv[]) {
  TCLAP::CmdLine cmd("Determine bin edges for histograms.", ' ', VERSION);
  TCLAP::ValueArg<std::string> imageArg("i", "infile", "Path to image/mask list.", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> outArg("o", "outfile", "Path to output file", true, "", "path", cmd);
  TCLAP::ValueArg<unsigned int> nBinsArg("b", "bins", "Number of bins to use", true, 41, "unsigned int", cmd);
  TCLAP::ValueArg<unsigned int> nSamplesArg("S", "samples", "Number of samples to use from each (0 = all)", true,
                                            0, "unsigned int", cmd);
  TCLAP::MultiArg<float> scalesArg("s", "scale", "Scales for the Gauss applicability function", true, "double",
                                   cmd);
  TCLAP::MultiArg<unsigned int> foregroundValueArg("f", "foreground", "Voxel value of foreground in mask", true,
                                                   "unsigned int", cmd);
  try {
    cmd.parse(argc, argv);
  } catch (TCLAP::ArgException &e) {
    std::cerr << "Error : " << e.error() << " for arg " << e.argId() << std::endl;
    return EXIT_FAILURE;
  }
  const std::string infilePath(imageArg.getValue()), outfilePath(outArg.getValue());
  const unsigned int nBins(nBinsArg.getValue()), nSamples(nSamplesArg.getValue());
  const std::vector<float> scales(scalesArg.getValue());
  const std::vector<unsigned int> foregroundValues(foregroundValueArg.getValue());

  typedef float PixelType;
  typedef itk::Image<PixelType, 3> ImageType;
  typedef unsigned char MaskPixelType;
  typedef itk::Image<MaskPixelType, 3> MaskType;
  typedef itk::VectorImage<PixelType, 3> VectorImageType;
  const size_t numFeatures = itk::ImageToEmphysemaFeaturesFilter<ImageType, MaskType, VectorImageType>::numFeatures;

  std::vector<StringPair> imageMaskPairList;
  try {
    imageMaskPairList = readPairList(infilePath);
  } catch (...) {
    std::cerr << "Could not read image/mask list" << std::endl;
    return EXIT_FAILURE;
  }

  std::mt19937_64 gen;
  if (const char *seed = std::getenv("IFE_SEED")) gen.seed(std::strtoull(seed, nullptr, 10));
  else gen.seed(std::random_device()());

  ife::host::Engine &engine = ife::host::Engine::Instance();
  SamplesHandle samples;
  std::vector<float> edges((size_t)scales.size() * numFeatures * (nBins > 0 ? nBins - 1 : 0));
  try {
    engine.check(ife_samples_create(engine.ctx(), (int)(scales.size() * numFeatures), &samples.s),
                 "ife_samples_create");
    for (const StringPair &imageMaskPair : imageMaskPairList) {
      std::cout << "Processing " << std::endl
                << "Image: '" << imageMaskPair.first << "'" << std::endl
                << "Mask: '" << imageMaskPair.second << "'" << std::endl;
      itk::ImageFileReader<ImageType>::Pointer imageReader = itk::ImageFileReader<ImageType>::New();
      itk::ImageFileReader<MaskType>::Pointer maskReader = itk::ImageFileReader<MaskType>::New();
      imageReader->SetFileName(imageMaskPair.first);
      maskReader->SetFileName(imageMaskPair.second);
      const ImageType *image;
      const MaskType *mask;
      try {
        image = imageReader->GetOutput();
        mask = maskReader->GetOutput();
        ife::host::same_size(*image, *mask, "DetermineHistogramBinEdges");
      } catch (itk::ExceptionObject &e) {
        std::cerr << "Failed to Update mask reader." << std::endl
                  << "Image: '" << imageMaskPair.first << "'" << std::endl
                  << "Mask: '" << imageMaskPair.second << "'" << std::endl
                  << "ExceptionObject: " << e << std::endl;
        return EXIT_FAILURE;
      }
      const ife_volume_desc d = ife::host::describe(*image);
      const int64_t nvox = d.nx * d.ny * d.nz;
      const MaskPixelType *labels = mask->GetBufferPointer();

      std::vector<int64_t> positions;  // scale-major, nSamples per scale
      if (nSamples > 0) {
        bool any = false;
        for (int64_t v = 0; v < nvox && !any; ++v)
          for (unsigned int acceptV : foregroundValues) any = any || labels[v] == acceptV;
        if (!any) {  // the reference would draw forever
          std::cerr << "No foreground voxel in mask '" << imageMaskPair.second << "'" << std::endl;
          return EXIT_FAILURE;
        }
        std::uniform_int_distribution<int64_t> pick(0, nvox - 1);
        positions.reserve((size_t)nSamples * scales.size());
        for (size_t i = 0; i < scales.size(); ++i)
          for (unsigned int nSampled = 0; nSampled < nSamples;) {
            const int64_t v = pick(gen);
            for (unsigned int acceptV : foregroundValues)
              if (labels[v] == acceptV) {
                positions.push_back(v);
                ++nSampled;
                break;
              }
          }
      }
      try {
        engine.check(ife_samples_add_image(engine.ctx(), samples.s, image->GetBufferPointer(), IFE_F32, labels,
                                           IFE_U8, &d, scales.data(), (int)scales.size(),
                                           foregroundValues.data(), (int)foregroundValues.size(),
                                           positions.empty() ? nullptr : positions.data(), (int64_t)nSamples,
                                           IFE_MEM_HOST),
                     "ImageToEmphysemaFeaturesFilter");
      } catch (itk::ExceptionObject &e) {
        std::cerr << "Failed to Update feature filter." << std::endl
                  << "Image: '" << imageMaskPair.first << "'" << std::endl
                  << "Mask: '" << imageMaskPair.second << "'" << std::endl
                  << "ExceptionObject: " << e << std::endl;
        return EXIT_FAILURE;
      }
    }
    engine.check(ife_samples_equalized_edges(engine.ctx(), samples.s, (int)nBins, edges.data()),
                 "determineEdgesForEqualizedHistogram");
  } catch (itk::ExceptionObject &e) {
    std::cerr << "Failed to determine the edges." << std::endl << "ExceptionObject: " << e << std::endl;
    return EXIT_FAILURE;
  }

  std::ofstream out(outfilePath.c_str());
  out << "# Features: GaussianBlur GradientMagnitude Eigenvalue1 Eigenvalue2 Eigenvalue3 LaplacianOfGaussian "
         "GaussianCurvature FrobeniusNorm\n"
      << "# Scales: ";
  for (size_t i = 0; i < scales.size(); ++i) out << scales[i] << (i + 1 < scales.size() ? ' ' : '\n');
  if (!out.good()) {
    std::cerr << "Error writing edges header to file." << std::endl << "Out path: " << outfilePath << std::endl;
    return EXIT_FAILURE;
  }
  const size_t perRow = nBins > 0 ? nBins - 1 : 0;
  for (size_t i = 0; i < scales.size() * numFeatures; ++i) {
    writeSequenceAsText(out, edges.begin() + i * perRow, edges.begin() + (i + 1) * perRow);
    out << std::endl;
    if (!out.good()) {
      std::cerr << "Error writing to edges to file." << std::endl << "Out path: " << outfilePath << std::endl;
      return EXIT_FAILURE;
    }
  }
  return EXIT_SUCCESS;
}
