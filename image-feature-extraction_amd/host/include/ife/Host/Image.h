// Image.h -- the few itk:: types the hot-path callers of the reference touch, without ITK.
//
// The reference's tools hold images as itk::Image<T,3> / itk::VectorImage<T,3> behind
// itk::SmartPointer (e.g. tools/ExtractFeatures.cxx:81-96).  ITK is not available in
// this build, so the host mirror provides these names with the members those callers
// use; buffers are x-fastest like itk::Image, vector images interleaved like
// itk::VectorImage.  When real ITK is present the filters are used through the adapter
// shown in INTEGRATION.md instead and this header is not needed.
#ifndef IFE_HOST_IMAGE_H
#define IFE_HOST_IMAGE_H

#include <array>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace itk {

// itk::ExceptionObject: what the tools catch in main (ExtractFeatures.cxx:145-152)
class ExceptionObject : public std::runtime_error {
 public:
  explicit ExceptionObject(const std::string &what, const std::string &where = "")
      : std::runtime_error(where.empty() ? what : where + ": " + what) {}
  const char *GetDescription() const { return what(); }
};
inline std::ostream &operator<<(std::ostream &os, const ExceptionObject &e) {
  return os << "itk::ExceptionObject: " << e.what();
}

template <typename T>
class SmartPointer {
 public:
  SmartPointer() = default;
  SmartPointer(std::nullptr_t) {}
  explicit SmartPointer(T *p) : p_(p) {}
  SmartPointer(const std::shared_ptr<T> &p) : p_(p) {}
  template <typename U>
  SmartPointer(const SmartPointer<U> &o) : p_(o.shared()) {}
  T *operator->() const { return p_.get(); }
  T &operator*() const { return *p_; }
  T *GetPointer() const { return p_.get(); }
  bool IsNull() const { return !p_; }
  bool IsNotNull() const { return bool(p_); }
  operator T *() const { return p_.get(); }
  const std::shared_ptr<T> &shared() const { return p_; }

 private:
  std::shared_ptr<T> p_;
};

#define ifeNewMacro(Self)                              \
  typedef ::itk::SmartPointer<Self> Pointer;           \
  typedef ::itk::SmartPointer<const Self> ConstPointer; \
  static Pointer New() { return Pointer(std::make_shared<Self>()); }

struct Size3 {
  std::array<uint64_t, 3> v{{0, 0, 0}};
  uint64_t &operator[](size_t i) { return v[i]; }
  const uint64_t &operator[](size_t i) const { return v[i]; }
};
inline std::ostream &operator<<(std::ostream &os, const Size3 &s) {  // as itk::Size prints
  return os << "[" << s[0] << ", " << s[1] << ", " << s[2] << "]";
}
struct Index3 {
  std::array<int64_t, 3> v{{0, 0, 0}};
  int64_t &operator[](size_t i) { return v[i]; }
  const int64_t &operator[](size_t i) const { return v[i]; }
};
inline std::ostream &operator<<(std::ostream &os, const Index3 &s) {  // as itk::Index prints
  return os << "[" << s[0] << ", " << s[1] << ", " << s[2] << "]";
}
struct Spacing3 {
  std::array<double, 3> v{{1.0, 1.0, 1.0}};
  double &operator[](size_t i) { return v[i]; }
  const double &operator[](size_t i) const { return v[i]; }
};
struct Region3 {  // itk::ImageRegion<3>: index + size
  typedef Size3 SizeType;
  typedef Index3 IndexType;
  Index3 index;
  Size3 size;
  Region3() = default;
  Region3(const Index3 &i, const Size3 &s) : index(i), size(s) {}
  const Size3 &GetSize() const { return size; }
  const Index3 &GetIndex() const { return index; }
  uint64_t GetNumberOfPixels() const { return size[0] * size[1] * size[2]; }
  bool IsInside(const Region3 &r) const {  // r entirely inside this region
    for (size_t d = 0; d < 3; ++d)
      if (r.index[d] < index[d] || r.index[d] + (int64_t)r.size[d] > index[d] + (int64_t)size[d]) return false;
    return true;
  }
};
inline std::ostream &operator<<(std::ostream &os, const Region3 &r) {
  return os << "ImageRegion Index: " << r.index << " Size: " << r.size;
}

// World geometry of the file an image was read from, carried through untouched and written
// back as it was: the arithmetic of the hot path uses the spacing only (the reference ignores
// direction cosines too, Hessian3DImageFilter.hxx), but an output volume has to overlay its
// source in any NIfTI / MetaImage aware consumer.  NIfTI: qform/sform codes, quaternion,
// offsets, qfac and the three srow vectors; MetaImage: the TransformMatrix and
// CenterOfRotation lines.  What ITK's own writer would derive from its direction matrix for a
// file produced by OTHER software is not reproduced ("parity unpinned": no ITK here).
struct FileGeometry {
  bool has_nifti = false;
  int16_t qform_code = 0, sform_code = 0;
  float qfac = 1.0f, quatern[3] = {0, 0, 0}, qoffset[3] = {0, 0, 0};
  float srow[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  char xyzt_units = 2;
  bool has_mhd = false;
  std::string mhd_transform, mhd_center, mhd_orientation;
};

class ImageBase3 {
 public:
  typedef Size3 SizeType;
  typedef Index3 IndexType;
  typedef Region3 RegionType;
  virtual ~ImageBase3() {}
  void SetRegions(const Size3 &s) { region_.size = s; }
  const Region3 &GetLargestPossibleRegion() const { return region_; }
  const Region3 &GetBufferedRegion() const { return region_; }
  void SetSpacing(const Spacing3 &s) { spacing_ = s; }
  const Spacing3 &GetSpacing() const { return spacing_; }
  void SetOrigin(const Spacing3 &o) { origin_ = o; }
  const Spacing3 &GetOrigin() const { return origin_; }
  void SetFileGeometry(const FileGeometry &g) { geom_ = g; }
  const FileGeometry &GetFileGeometry() const { return geom_; }
  void CopyInformation(const ImageBase3 *o) {
    region_ = o->region_;
    spacing_ = o->spacing_;
    origin_ = o->origin_;
    geom_ = o->geom_;
  }

 protected:
  Region3 region_;
  Spacing3 spacing_, origin_;
  FileGeometry geom_;
};

template <typename TPixel, unsigned int VDim = 3>
class Image : public ImageBase3 {
  static_assert(VDim == 3, "the hot path is three-dimensional");

 public:
  typedef Image Self;
  typedef TPixel PixelType;
  static const unsigned int ImageDimension = 3;
  ifeNewMacro(Self);
  void Allocate() { buf_.assign(region_.GetNumberOfPixels(), TPixel()); }
  TPixel *GetBufferPointer() { return buf_.data(); }
  const TPixel *GetBufferPointer() const { return buf_.data(); }
  unsigned int GetNumberOfComponentsPerPixel() const { return 1; }
  std::vector<TPixel> &Buffer() { return buf_; }

 private:
  std::vector<TPixel> buf_;
};

template <typename TPixel, unsigned int VDim = 3>
class VectorImage : public ImageBase3 {
  static_assert(VDim == 3, "the hot path is three-dimensional");

 public:
  typedef VectorImage Self;
  typedef TPixel InternalPixelType;
  typedef TPixel PixelType;  // component type (the reference uses it as such: .h:38,94)
  static const unsigned int ImageDimension = 3;
  ifeNewMacro(Self);
  void SetNumberOfComponentsPerPixel(unsigned int n) { ncomp_ = n; }
  unsigned int GetNumberOfComponentsPerPixel() const { return ncomp_; }
  void Allocate() { buf_.assign(region_.GetNumberOfPixels() * ncomp_, TPixel()); }
  TPixel *GetBufferPointer() { return buf_.data(); }  // [voxel*ncomp + c]
  const TPixel *GetBufferPointer() const { return buf_.data(); }

 private:
  unsigned int ncomp_ = 1;
  std::vector<TPixel> buf_;
};

}  // namespace itk

#endif
