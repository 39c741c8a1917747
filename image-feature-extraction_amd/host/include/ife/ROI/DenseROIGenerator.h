// DenseROIGenerator.h -- one box per mask voxel, with the interface of the reference's
// itk::DenseROIGenerator<TMask> (include/ife/ROI/DenseROIGenerator.h:8-29, rule of
// DenseROIGenerator.hxx:24-46): the mask is walked in raster order (x fastest); every
// non-zero voxel whose box [pos - size/2, pos - size/2 + size) lies inside the image yields
// that box.  Host code: the list is what the device kernel is then handed.
#ifndef __DenseROIGenerator_h
#define __DenseROIGenerator_h

#include <vector>

#include "ife/Host/Image.h"

namespace itk {

template <typename TMask>
class DenseROIGenerator {
 public:
  typedef TMask MaskType;
  typedef const MaskType *MaskPointer;
  typedef typename MaskType::IndexType IndexType;
  typedef typename MaskType::SizeType SizeType;
  typedef typename MaskType::RegionType RegionType;

  explicit DenseROIGenerator(MaskPointer mask) : m_Mask(mask) {}
  void setMask(MaskPointer mask) { m_Mask = mask; }

  std::vector<RegionType> generate(SizeType size) {
    const RegionType imageRegion = m_Mask->GetLargestPossibleRegion();
    const SizeType &n = imageRegion.GetSize();
    const typename MaskType::PixelType *m = m_Mask->GetBufferPointer();
    std::vector<RegionType> rois;
    for (uint64_t z = 0; z < n[2]; ++z)
      for (uint64_t y = 0; y < n[1]; ++y)
        for (uint64_t x = 0; x < n[0]; ++x) {
          if (m[x + n[0] * (y + n[1] * z)] == 0) continue;
          IndexType start;
          start[0] = (int64_t)x - (int64_t)(size[0] / 2);
          start[1] = (int64_t)y - (int64_t)(size[1] / 2);
          start[2] = (int64_t)z - (int64_t)(size[2] / 2);
          const RegionType roi(start, size);
          if (imageRegion.IsInside(roi)) rois.push_back(roi);
        }
    return rois;
  }

 private:
  MaskPointer m_Mask;
};

}  // namespace itk

#endif
