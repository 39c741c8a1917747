// RegionOfInterestGenerator.h -- random boxes centred on mask voxels, with the interface of
// the reference's itk::RegionOfInterestGenerator<TMask> (include/ife/ROI/
// RegionOfInterestGenerator.h:9-31, sampling rule of .hxx:22-62): positions are drawn
// uniformly over the volume with replacement; a position is kept when the mask is non-zero
// there and the box [pos - size/2, pos - size/2 + size) lies inside the image; until
// numberOfROIs boxes are kept.  Host code (a handful of boxes per image).  The reference
// seeds ITK's generator from the clock; IFE_SEED fixes the seed here.
#ifndef __RegionOfInterestGenerator_h
#define __RegionOfInterestGenerator_h

#include <cstdlib>
#include <random>
#include <vector>

#include "ife/Host/Image.h"

namespace itk {

template <typename TMask>
class RegionOfInterestGenerator {
 public:
  typedef TMask MaskType;
  typedef const MaskType *MaskPointer;
  typedef typename MaskType::IndexType IndexType;
  typedef typename MaskType::SizeType SizeType;
  typedef typename MaskType::RegionType RegionType;

  explicit RegionOfInterestGenerator(MaskPointer mask) : m_Mask(mask) {}
  void setMask(MaskPointer mask) { m_Mask = mask; }

  std::vector<RegionType> generate(size_t numberOfROIs, SizeType size) {
    const RegionType imageRegion = m_Mask->GetLargestPossibleRegion();
    const SizeType &n = imageRegion.GetSize();
    const typename MaskType::PixelType *m = m_Mask->GetBufferPointer();
    // a position that can be kept must exist, or the draw below would never end
    bool possible = false;
    for (uint64_t z = 0; z < n[2] && !possible; ++z)
      for (uint64_t y = 0; y < n[1] && !possible; ++y)
        for (uint64_t x = 0; x < n[0] && !possible; ++x)
          possible = m[x + n[0] * (y + n[1] * z)] != 0 && imageRegion.IsInside(box(x, y, z, size));
    if (!possible && numberOfROIs > 0)
      throw ExceptionObject("no mask voxel admits a region of the requested size inside the image",
                            "RegionOfInterestGenerator");
    std::mt19937_64 gen;
    if (const char *seed = std::getenv("IFE_SEED")) gen.seed(std::strtoull(seed, nullptr, 10));
    else gen.seed(std::random_device()());
    std::uniform_int_distribution<uint64_t> pick(0, imageRegion.GetNumberOfPixels() - 1);
    std::vector<RegionType> rois;
    rois.reserve(numberOfROIs);
    while (rois.size() < numberOfROIs) {
      const uint64_t v = pick(gen);
      if (m[v] == 0) continue;
      const RegionType roi = box(v % n[0], (v / n[0]) % n[1], v / (n[0] * n[1]), size);
      if (imageRegion.IsInside(roi)) rois.push_back(roi);
    }
    return rois;
  }

 private:
  static RegionType box(uint64_t x, uint64_t y, uint64_t z, const SizeType &size) {
    IndexType start;
    start[0] = (int64_t)x - (int64_t)(size[0] / 2);
    start[1] = (int64_t)y - (int64_t)(size[1] / 2);
    start[2] = (int64_t)z - (int64_t)(size[2] / 2);
    return RegionType(start, size);
  }
  MaskPointer m_Mask;
};

}  // namespace itk

#endif
