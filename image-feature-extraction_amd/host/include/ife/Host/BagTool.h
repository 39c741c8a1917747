// BagTool.h -- the body shared by the three bag tools of the reference, which differ only
// in where the regions come from and in what is binned:
//   MakeBag              tools/MakeBag.cxx              regions from a file or sampled on the mask; 8 features x scales
//   MakeBagDense         tools/MakeBagDense.cxx:239-250 one region per mask voxel (DenseROIGenerator); 8 features x scales
//   MakeBagOnlyIntensity tools/MakeBagOnlyIntensity.cxx regions as MakeBag; the image intensity itself, ONE histogram
// Flags, files and exit codes are the reference's (MakeBag.cxx:31-199 arguments, :283-316
// .ROIInfo, :341-392 histogram specification, :405-472 the bag, :475-486 .bag; the variants
// drop the flags they do not read: MakeBagDense has no -r/-R/-n, MakeBagOnlyIntensity no -s).
//
// The features of all scales stay in HBM and the per-region histograms are counted there
// (ife_bag_image; ife_roi_histograms on the image for the intensity variant); only the
// counts come back.  Frequencies are formed as DenseHistogram::getFrequencies does
// (DenseHistogram.h:55-60).
#ifndef IFE_HOST_BAGTOOL_H
#define IFE_HOST_BAGTOOL_H

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "tclap/CmdLine.h"

#include "ife/Filters/ImageToEmphysemaFeaturesFilter.h"
#include "ife/Host/ImageIO.h"
#include "ife/IO/IO.h"
#include "ife/IO/ROIReader.h"
#include "ife/ROI/DenseROIGenerator.h"
#include "ife/ROI/RegionOfInterestGenerator.h"
#include "ife/Util/Path.h"

namespace ife {
namespace host {

enum BagVariant { BAG_SAMPLED = 0, BAG_DENSE = 1, BAG_ONLY_INTENSITY = 2 };

inline int bag_main(int argc, char *argv[], BagVariant variant, const char *toolName) {
  const std::string VERSION("0.1");
  const bool dense = variant == BAG_DENSE, intensity = variant == BAG_ONLY_INTENSITY;
  typedef float PixelType;
  typedef unsigned short MaskPixelType;

  TCLAP::CmdLine cmd("Create a bag of instances samples from an image.", ' ', VERSION);
  TCLAP::ValueArg<std::string> imageArg("i", "image", "Path to image.", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> maskArg("m", "mask", "Path to mask.", true, "", "path", cmd);
  TCLAP::ValueArg<std::string> histArg("H", "histogram-spec", "Path to histogram specification.", true, "",
                                       "path", cmd);
  TCLAP::ValueArg<std::string> outDirArg("o", "outdir", "Path to output directory", true, "", "path", cmd);
  // a flag a variant does not read is not registered, exactly as in the reference
  std::unique_ptr<TCLAP::MultiArg<float> > scalesArg;
  if (!intensity)
    scalesArg.reset(new TCLAP::MultiArg<float>("s", "scale", "Scales for the Gauss applicability function", true,
                                               "double", cmd));
  std::unique_ptr<TCLAP::ValueArg<std::string> > roiArg;
  std::unique_ptr<TCLAP::ValueArg<bool> > roiHasHeaderArg;
  if (!dense) {
    roiArg.reset(new TCLAP::ValueArg<std::string>("r", "roi-file",
                                                  "Path to ROI file. If given the ROIs in this file will be used,"
                                                  "otherwise ROIs will be generated.",
                                                  false, "", "path", cmd));
    roiHasHeaderArg.reset(new TCLAP::ValueArg<bool>("R", "roi-file-has-header",
                                                    "Flag indicating if the ROI file has a header", false, true,
                                                    "boolean", cmd));
  }
  TCLAP::ValueArg<std::string> roiMaskArg("M", "roi-mask",
                                          "Path to ROI mask file. If ROIs are generated an optional mask "
                                          "controlling the ROI generation can be used. If not given then "
                                          "the image mask will be used.",
                                          false, "", "path", cmd);
  TCLAP::ValueArg<MaskPixelType> roiMaskValueArg("v", "roi-mask-value",
                                                 "Value in the ROI mask that should be used for inclusion.", false, 1,
                                                 "MaskPixelType", cmd);
  std::unique_ptr<TCLAP::ValueArg<size_t> > numROIsArg;
  if (!dense)
    numROIsArg.reset(new TCLAP::ValueArg<size_t>("n", "num-rois", "Number of ROIs to sample", false, 50, "N>=2", cmd));
  TCLAP::ValueArg<size_t> roiSizeXArg("x", "roi-size-x", "Size of ROI in x dimension", false, 41, "N>=1", cmd);
  TCLAP::ValueArg<size_t> roiSizeYArg("y", "roi-size-y", "Size of ROI in y dimension", false, 41, "N>=1", cmd);
  TCLAP::ValueArg<size_t> roiSizeZArg("z", "roi-size-z", "Size of ROI in z dimension", false, 41, "N>=1", cmd);
  TCLAP::ValueArg<std::string> prefixArg("p", "prefix", "Prefix to use for output filenames", false, "", "string",
                                         cmd);
  try {
    cmd.parse(argc, argv);
  } catch (TCLAP::ArgException &e) {
    std::cerr << "Error : " << e.error() << " for arg " << e.argId() << std::endl;
    return EXIT_FAILURE;
  }
  const std::string imagePath(imageArg.getValue()), maskPath(maskArg.getValue()), histPath(histArg.getValue());
  const std::string outDirPath(outDirArg.getValue()), roiPath(roiArg ? roiArg->getValue() : std::string());
  const std::string roiMaskPath(roiMaskArg.getValue());
  const std::vector<float> scales(scalesArg ? scalesArg->getValue() : std::vector<float>());
  const bool roiHasHeader(roiHasHeaderArg ? roiHasHeaderArg->getValue() : true);
  const MaskPixelType roiMaskValue(roiMaskValueArg.getValue());
  const size_t numROIs(numROIsArg ? numROIsArg->getValue() : 0);
  const std::string prefix(prefixArg.getValue());
  const size_t numFeatures = 8;

  typedef itk::Image<PixelType, 3> ImageType;
  typedef itk::Image<MaskPixelType, 3> MaskImageType;
  typedef ImageType::SizeType SizeType;
  typedef ImageType::RegionType RegionType;

  itk::ImageFileReader<ImageType>::Pointer imageReader = itk::ImageFileReader<ImageType>::New();
  imageReader->SetFileName(imagePath);
  itk::ImageFileReader<MaskImageType>::Pointer maskReader = itk::ImageFileReader<MaskImageType>::New();
  maskReader->SetFileName(maskPath);

  // regions: from the file, or generated on the (binary) mask / the thresholded ROI mask
  std::vector<RegionType> rois;
  const ImageType *image = nullptr;
  const MaskImageType *mask = nullptr;
  try {
    image = imageReader->GetOutput();
    mask = maskReader->GetOutput();
    ife::host::same_size(*image, *mask, toolName);
  } catch (itk::ExceptionObject &e) {
    std::cerr << "Failed to read the inputs." << std::endl
              << "Image: " << imagePath << std::endl
              << "Mask: " << maskPath << std::endl
              << "ExceptionObject: " << e << std::endl;
    return EXIT_FAILURE;
  }
  if (roiPath.empty()) {
    try {
      MaskImageType::Pointer roiMask = MaskImageType::New();
      if (!roiMaskPath.empty()) {
        std::cout << "Using ROI mask." << std::endl;
        itk::ImageFileReader<MaskImageType>::Pointer roiMaskReader = itk::ImageFileReader<MaskImageType>::New();
        roiMaskReader->SetFileName(roiMaskPath);
        const MaskImageType *rm = roiMaskReader->GetOutput();
        ife::host::same_size(*rm, *mask, toolName);
        roiMask->CopyInformation(rm);
        roiMask->Allocate();
        for (uint64_t v = 0; v < rm->GetLargestPossibleRegion().GetNumberOfPixels(); ++v)
          roiMask->GetBufferPointer()[v] = rm->GetBufferPointer()[v] == roiMaskValue ? 1 : 0;
      }
      const MaskImageType *genMask = roiMaskPath.empty() ? mask : roiMask.GetPointer();
      SizeType roiSize;
      roiSize[0] = roiSizeXArg.getValue();
      roiSize[1] = roiSizeYArg.getValue();
      roiSize[2] = roiSizeZArg.getValue();
      if (dense) rois = itk::DenseROIGenerator<MaskImageType>(genMask).generate(roiSize);
      else rois = itk::RegionOfInterestGenerator<MaskImageType>(genMask).generate(numROIs, roiSize);
      const std::string roiOutPath(Path::join(outDirPath, prefix + ".ROIInfo"));
      std::ofstream out(roiOutPath.c_str());
      for (const RegionType &roi : rois) out << roi.GetIndex() << roi.GetSize() << '\n';
      if (!out.good()) {
        std::cerr << "Error writing ROI info file" << std::endl;
        return EXIT_FAILURE;
      }
    } catch (itk::ExceptionObject &e) {
      std::cerr << "Failed to generate ROIs." << std::endl << "ExceptionObject: " << e << std::endl;
      return EXIT_FAILURE;
    }
  } else {
    try {
      rois = ROIReader<RegionType>::read(roiPath, roiHasHeader);
      std::cout << "Got " << rois.size() << " rois." << std::endl;
    } catch (std::exception &e) {
      std::cerr << "Error reading ROIs" << std::endl
                << "roiPath: " << roiPath << std::endl
                << "exception: " << e.what() << std::endl;
      return EXIT_FAILURE;
    }
  }

  // histogram specification: one row of edges per (scale, feature); '#' lines skipped, an
  // empty line ends it
  std::vector<float> edges;
  size_t histSize = 0, numHistograms = 0;
  std::ifstream isHist(histPath.c_str());
  if (!isHist.good()) {
    std::cerr << "Could not read histogram file '" << histPath << "'" << std::endl;
    return EXIT_FAILURE;
  }
  while (isHist.good()) {
    std::string line;
    std::getline(isHist, line);
    if (line.empty()) {
      std::cout << "Empty line. Breaking" << std::endl;
      break;
    }
    if (line[0] == '#') {
      std::cout << "Skipping a line" << std::endl;
      continue;
    }
    std::stringstream ss(line);
    std::vector<PixelType> row;
    readTextSequence<PixelType, char>(ss, std::back_inserter(row));
    if (row.empty()) {
      std::cerr << "A histogram needs at least one edge" << std::endl;
      return EXIT_FAILURE;
    }
    ++numHistograms;
    if (histSize == 0) {
      histSize = row.size() + 1;
    } else if (histSize != row.size() + 1) {
      std::cerr << "Histograms must have the same bin count" << std::endl
                << "Expected " << histSize << " Got " << row.size() << std::endl
                << "Number of histograms " << numHistograms << std::endl;
      return EXIT_FAILURE;
    }
    edges.insert(edges.end(), row.begin(), row.end());
  }
  if (intensity) {
    if (numHistograms != 1) {  // MakeBagOnlyIntensity.cxx:326-330
      std::cerr << "[ERROR] Expected exactly one histogram in histogram specification. Got " << numHistograms
                << std::endl;
      return EXIT_FAILURE;
    }
  } else if (numHistograms != numFeatures * scales.size()) {
    std::cerr << "Number of histograms must match number of features times number of scales" << std::endl
              << "Number of histograms = " << numHistograms << std::endl
              << "Number of features*scales = " << numFeatures * scales.size() << std::endl;
    return EXIT_FAILURE;
  }
  const size_t totalBins = histSize * numHistograms;

  // the bag: one row per region, histSize columns per (scale, feature)
  std::vector<int64_t> boxes;
  for (const RegionType &roi : rois)
    for (int k = 0; k < 6; ++k) boxes.push_back(k < 3 ? roi.GetIndex()[k] : (int64_t)roi.GetSize()[k - 3]);
  std::vector<uint32_t> counts(rois.size() * totalBins);
  for (size_t i = 0; i < scales.size(); ++i) std::cout << "Processing scale " << scales[i] << std::endl;
  try {
    ife::host::Engine &engine = ife::host::Engine::Instance();
    const ife_volume_desc d = ife::host::describe(*image);
    if (rois.size() > 0x7fffffff) throw itk::ExceptionObject("more than 2^31-1 regions", toolName);
    if (!rois.empty() && intensity)  // the image itself is the one-component "feature volume"
      engine.check(ife_roi_histograms(engine.ctx(), image->GetBufferPointer(), IFE_INTERLEAVED, 1,
                                      mask->GetBufferPointer(), IFE_U16, &d, boxes.data(), (int)rois.size(),
                                      edges.data(), (int)(histSize - 1), counts.data(), IFE_MEM_HOST),
                   toolName);
    else if (!rois.empty())
      engine.check(ife_bag_image(engine.ctx(), image->GetBufferPointer(), IFE_F32, mask->GetBufferPointer(), IFE_U16,
                                 &d, scales.data(), (int)scales.size(), boxes.data(), (int)rois.size(), edges.data(),
                                 (int)(histSize - 1), counts.data(), IFE_MEM_HOST),
                   toolName);
  } catch (itk::ExceptionObject &e) {
    std::cerr << "Failed to update featureFilter." << std::endl << "ExceptionObject: " << e << std::endl;
    return EXIT_FAILURE;
  }
  const std::string outPath(Path::join(outDirPath, prefix + ".bag"));
  std::ofstream out(outPath.c_str());
  for (size_t r = 0; r < rois.size(); ++r) {
    for (size_t h = 0; h < numHistograms; ++h) {
      const uint32_t *c = &counts[(r * numHistograms + h) * histSize];
      int sum = 0;  // getFrequencies sums into an int, then divides floats
      for (size_t b = 0; b < histSize; ++b) sum = (int)(sum + c[b]);
      for (size_t b = 0; b < histSize; ++b) {
        out << (PixelType)c[b] / (PixelType)sum;
        if (h * histSize + b + 1 < totalBins) out << ",";
      }
    }
    out << '\n';
  }
  if (!out.good()) {
    std::cerr << "Error writing histogram to file" << std::endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}

}  // namespace host
}  // namespace ife

#endif
