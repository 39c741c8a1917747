// ImageIO.h -- itk::ImageFileReader / itk::ImageFileWriter for the two on-disk formats
// this build supports without ITK: NIfTI-1 single file (.nii, and .nii.gz through zlib;
// the reference writes .nii.gz, ExtractFeatures.cxx:15) and MetaImage (.mhd + .raw).
// As in ITK, the reader converts the stored component type to the requested pixel type.
#ifndef IFE_HOST_IMAGEIO_H
#define IFE_HOST_IMAGEIO_H

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "ife/Host/Image.h"

namespace itk {
namespace io_detail {

inline bool ends_with(const std::string &s, const std::string &suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

struct RawVolume {
  Size3 size;
  Spacing3 spacing, origin;
  FileGeometry geom;
  int nifti_type = 16;  // NIfTI datatype code of `bytes`
  double slope = 0.0, inter = 0.0;
  std::vector<unsigned char> bytes;
};

inline size_t nifti_type_size(int t) {
  switch (t) {
    case 2: return 1;    // uint8
    case 256: return 1;  // int8
    case 4: return 2;    // int16
    case 512: return 2;  // uint16
    case 8: return 4;    // int32
    case 768: return 4;  // uint32
    case 16: return 4;   // float32
    case 64: return 8;   // float64
  }
  return 0;
}

template <typename T> struct NiftiCode;
template <> struct NiftiCode<unsigned char> { enum { value = 2 }; };
template <> struct NiftiCode<short> { enum { value = 4 }; };
template <> struct NiftiCode<unsigned short> { enum { value = 512 }; };
template <> struct NiftiCode<float> { enum { value = 16 }; };
template <> struct NiftiCode<double> { enum { value = 64 }; };

#pragma pack(push, 1)
struct Nifti1Header {
  int32_t sizeof_hdr;
  char unused1[35];
  char dim_info;
  int16_t dim[8];
  float intent_p1, intent_p2, intent_p3;
  int16_t intent_code, datatype, bitpix, slice_start;
  float pixdim[8];
  float vox_offset, scl_slope, scl_inter;
  int16_t slice_end;
  char slice_code, xyzt_units;
  float cal_max, cal_min, slice_duration, toffset;
  int32_t glmax, glmin;
  char descrip[80], aux_file[24];
  int16_t qform_code, sform_code;
  float quatern_b, quatern_c, quatern_d, qoffset_x, qoffset_y, qoffset_z;
  float srow_x[4], srow_y[4], srow_z[4];
  char intent_name[16], magic[4];
};
#pragma pack(pop)
static_assert(sizeof(Nifti1Header) == 348, "NIfTI-1 header is 348 bytes");

inline void read_all(const std::string &path, bool gz, std::vector<unsigned char> &out) {
  if (gz) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw ExceptionObject("cannot open " + path, "ImageFileReader");
    unsigned char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof buf)) > 0) out.insert(out.end(), buf, buf + n);
    const bool bad = n < 0;
    gzclose(f);
    if (bad) throw ExceptionObject("error while inflating " + path, "ImageFileReader");
  } else {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw ExceptionObject("cannot open " + path, "ImageFileReader");
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    out.resize((size_t)n);
    f.read(reinterpret_cast<char *>(out.data()), n);
    if (!f) throw ExceptionObject("short read on " + path, "ImageFileReader");
  }
}

inline RawVolume read_nifti(const std::string &path) {
  std::vector<unsigned char> all;
  read_all(path, ends_with(path, ".gz"), all);
  if (all.size() < 352) throw ExceptionObject(path + " is too short for NIfTI-1", "ImageFileReader");
  Nifti1Header h;
  std::memcpy(&h, all.data(), sizeof h);
  if (h.sizeof_hdr != 348 || std::strncmp(h.magic, "n+1", 3) != 0)
    throw ExceptionObject(path + " is not a little-endian single-file NIfTI-1", "ImageFileReader");
  if (h.dim[0] < 3 || h.dim[0] > 4 || (h.dim[0] == 4 && h.dim[4] > 1))
    throw ExceptionObject(path + " is not a 3-D scalar volume", "ImageFileReader");
  RawVolume v;
  for (int a = 0; a < 3; ++a) {
    if (h.dim[1 + a] < 1) throw ExceptionObject(path + ": non-positive dimension in the header", "ImageFileReader");
    v.size[a] = (uint64_t)h.dim[1 + a];
    v.spacing[a] = (h.pixdim[1 + a] > 0 && std::isfinite(h.pixdim[1 + a])) ? (double)h.pixdim[1 + a] : 1.0;
  }
  v.origin[0] = h.qoffset_x; v.origin[1] = h.qoffset_y; v.origin[2] = h.qoffset_z;
  v.geom.has_nifti = true;
  v.geom.qform_code = h.qform_code; v.geom.sform_code = h.sform_code;
  v.geom.qfac = h.pixdim[0];
  v.geom.quatern[0] = h.quatern_b; v.geom.quatern[1] = h.quatern_c; v.geom.quatern[2] = h.quatern_d;
  v.geom.qoffset[0] = h.qoffset_x; v.geom.qoffset[1] = h.qoffset_y; v.geom.qoffset[2] = h.qoffset_z;
  std::memcpy(v.geom.srow[0], h.srow_x, sizeof h.srow_x);
  std::memcpy(v.geom.srow[1], h.srow_y, sizeof h.srow_y);
  std::memcpy(v.geom.srow[2], h.srow_z, sizeof h.srow_z);
  v.geom.xyzt_units = h.xyzt_units;
  v.nifti_type = h.datatype;
  v.slope = h.scl_slope;
  v.inter = h.scl_inter;
  const size_t es = nifti_type_size(h.datatype);
  if (!es) throw ExceptionObject(path + ": unsupported NIfTI datatype", "ImageFileReader");
  // dim[] are int16 (<= 32767 each), so the byte count fits 64 bits; vox_offset is a float
  // from an untrusted file: it must be finite and inside the file before it becomes a size_t
  if (!std::isfinite(h.vox_offset) || h.vox_offset < 352.0f || (double)h.vox_offset > (double)all.size())
    throw ExceptionObject(path + ": bad vox_offset in the header", "ImageFileReader");
  const size_t off = (size_t)h.vox_offset, nb = (size_t)(v.size[0] * v.size[1] * v.size[2]) * es;
  if (all.size() - off < nb)
    throw ExceptionObject(path + ": truncated voxel data", "ImageFileReader");
  v.bytes.assign(all.begin() + (std::ptrdiff_t)off, all.begin() + (std::ptrdiff_t)(off + nb));
  return v;
}

inline RawVolume read_mhd(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw ExceptionObject("cannot open " + path, "ImageFileReader");
  RawVolume v;
  std::string line, datafile, etype;
  int ndims = 3;
  while (std::getline(f, line)) {
    const size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    std::string key = line.substr(0, eq), val = line.substr(eq + 1);
    auto trim = [](std::string &s) {
      const char *ws = " \t\r\n";
      s.erase(0, s.find_first_not_of(ws));
      s.erase(s.find_last_not_of(ws) + 1);
    };
    trim(key); trim(val);
    std::istringstream is(val);
    if (key == "NDims") is >> ndims;
    else if (key == "DimSize") is >> v.size[0] >> v.size[1] >> v.size[2];
    else if (key == "ElementSpacing" || key == "ElementSize") is >> v.spacing[0] >> v.spacing[1] >> v.spacing[2];
    else if (key == "Offset" || key == "Position") is >> v.origin[0] >> v.origin[1] >> v.origin[2];
    else if (key == "TransformMatrix" || key == "Rotation" || key == "Orientation") { v.geom.has_mhd = true; v.geom.mhd_transform = val; }
    else if (key == "CenterOfRotation") { v.geom.has_mhd = true; v.geom.mhd_center = val; }
    else if (key == "AnatomicalOrientation") { v.geom.has_mhd = true; v.geom.mhd_orientation = val; }
    else if (key == "ElementType") etype = val;
    else if (key == "ElementDataFile") datafile = val;
    else if (key == "ElementByteOrderMSB" || key == "BinaryDataByteOrderMSB") {
      if (val == "True" || val == "true")
        throw ExceptionObject(path + ": big-endian data is not supported", "ImageFileReader");
    } else if (key == "CompressedData" && (val == "True" || val == "true")) {
      throw ExceptionObject(path + ": compressed MetaImage is not supported", "ImageFileReader");
    }
  }
  if (ndims != 3) throw ExceptionObject(path + " is not 3-D", "ImageFileReader");
  for (int a = 0; a < 3; ++a) {
    if (v.size[a] < 1 || v.size[a] > ((uint64_t)1 << 31))
      throw ExceptionObject(path + ": missing or absurd DimSize", "ImageFileReader");
    if (!(v.spacing[a] > 0.0) || !std::isfinite(v.spacing[a])) v.spacing[a] = 1.0;
  }
  if (v.size[0] * v.size[1] > ((uint64_t)1 << 62) / v.size[2] / 8)
    throw ExceptionObject(path + ": volume too large", "ImageFileReader");
  if (etype == "MET_FLOAT") v.nifti_type = 16;
  else if (etype == "MET_DOUBLE") v.nifti_type = 64;
  else if (etype == "MET_UCHAR") v.nifti_type = 2;
  else if (etype == "MET_CHAR") v.nifti_type = 256;
  else if (etype == "MET_SHORT") v.nifti_type = 4;
  else if (etype == "MET_USHORT") v.nifti_type = 512;
  else if (etype == "MET_INT") v.nifti_type = 8;
  else if (etype == "MET_UINT") v.nifti_type = 768;
  else throw ExceptionObject(path + ": unsupported ElementType " + etype, "ImageFileReader");
  if (datafile.empty() || datafile == "LOCAL")
    throw ExceptionObject(path + ": ElementDataFile must name a raw file", "ImageFileReader");
  const size_t slash = path.find_last_of('/');
  const std::string raw = (slash == std::string::npos || datafile[0] == '/')
                              ? datafile : path.substr(0, slash + 1) + datafile;
  std::vector<unsigned char> all;
  read_all(raw, false, all);
  const size_t nb = (size_t)(v.size[0] * v.size[1] * v.size[2]) * nifti_type_size(v.nifti_type);
  if (all.size() < nb) throw ExceptionObject(raw + ": truncated voxel data", "ImageFileReader");
  all.resize(nb);
  v.bytes.swap(all);
  return v;
}

template <typename TOut, typename TIn>
inline void convert(const unsigned char *src, size_t n, double slope, double inter, TOut *dst) {
  const TIn *s = reinterpret_cast<const TIn *>(src);
  const bool rescale = slope != 0.0 && (slope != 1.0 || inter != 0.0);
  for (size_t i = 0; i < n; ++i) {
    TIn x;
    std::memcpy(&x, s + i, sizeof x);
    dst[i] = rescale ? static_cast<TOut>((double)x * slope + inter) : static_cast<TOut>(x);
  }
}

template <typename TOut>
inline void convert_any(const RawVolume &v, TOut *dst) {
  const size_t n = (size_t)(v.size[0] * v.size[1] * v.size[2]);
  const unsigned char *p = v.bytes.data();
  switch (v.nifti_type) {
    case 2: convert<TOut, unsigned char>(p, n, v.slope, v.inter, dst); break;
    case 256: convert<TOut, signed char>(p, n, v.slope, v.inter, dst); break;
    case 4: convert<TOut, short>(p, n, v.slope, v.inter, dst); break;
    case 512: convert<TOut, unsigned short>(p, n, v.slope, v.inter, dst); break;
    case 8: convert<TOut, int32_t>(p, n, v.slope, v.inter, dst); break;
    case 768: convert<TOut, uint32_t>(p, n, v.slope, v.inter, dst); break;
    case 16: convert<TOut, float>(p, n, v.slope, v.inter, dst); break;
    case 64: convert<TOut, double>(p, n, v.slope, v.inter, dst); break;
    default: throw ExceptionObject("unsupported component type", "ImageFileReader");
  }
}

template <typename T>
inline void write_nifti(const std::string &path, const ImageBase3 &info, const T *data) {
  Nifti1Header h;
  std::memset(&h, 0, sizeof h);
  h.sizeof_hdr = 348;
  const Size3 &sz = info.GetLargestPossibleRegion().GetSize();
  h.dim[0] = 3;
  for (int a = 0; a < 3; ++a) {
    if (sz[a] > 32767)
      throw ExceptionObject("an axis longer than 32767 voxels does not fit NIfTI-1 (use .mhd): " + path,
                            "ImageFileWriter");
    h.dim[1 + a] = (int16_t)sz[a];
    h.pixdim[1 + a] = (float)info.GetSpacing()[a];
  }
  for (int a = 4; a < 8; ++a) h.dim[a] = 1;
  h.pixdim[0] = 1.0f;
  h.datatype = (int16_t)NiftiCode<T>::value;
  h.bitpix = (int16_t)(8 * sizeof(T));
  h.vox_offset = 352.0f;
  h.scl_slope = 1.0f;
  h.xyzt_units = 2;  // mm
  h.qform_code = 1;
  h.sform_code = 1;
  h.quatern_b = h.quatern_c = h.quatern_d = 0.0f;
  h.qoffset_x = (float)info.GetOrigin()[0];
  h.qoffset_y = (float)info.GetOrigin()[1];
  h.qoffset_z = (float)info.GetOrigin()[2];
  h.srow_x[0] = h.pixdim[1]; h.srow_x[3] = h.qoffset_x;
  h.srow_y[1] = h.pixdim[2]; h.srow_y[3] = h.qoffset_y;
  h.srow_z[2] = h.pixdim[3]; h.srow_z[3] = h.qoffset_z;
  const FileGeometry &fg = info.GetFileGeometry();
  if (fg.has_nifti) {  // the world geometry of the source file, as it was
    h.qform_code = fg.qform_code; h.sform_code = fg.sform_code;
    h.pixdim[0] = fg.qfac;
    h.quatern_b = fg.quatern[0]; h.quatern_c = fg.quatern[1]; h.quatern_d = fg.quatern[2];
    h.qoffset_x = fg.qoffset[0]; h.qoffset_y = fg.qoffset[1]; h.qoffset_z = fg.qoffset[2];
    std::memcpy(h.srow_x, fg.srow[0], sizeof h.srow_x);
    std::memcpy(h.srow_y, fg.srow[1], sizeof h.srow_y);
    std::memcpy(h.srow_z, fg.srow[2], sizeof h.srow_z);
    h.xyzt_units = fg.xyzt_units;
  }
  std::memcpy(h.magic, "n+1", 4);
  const unsigned char ext[4] = {0, 0, 0, 0};
  const size_t nb = (size_t)(sz[0] * sz[1] * sz[2]) * sizeof(T);
  if (ends_with(path, ".gz")) {
    gzFile f = gzopen(path.c_str(), "wb1");
    if (!f) throw ExceptionObject("cannot create " + path, "ImageFileWriter");
    bool ok = gzwrite(f, &h, sizeof h) == (int)sizeof h && gzwrite(f, ext, 4) == 4;
    const char *p = reinterpret_cast<const char *>(data);
    size_t left = nb;
    while (ok && left) {
      const unsigned chunk = (unsigned)std::min<size_t>(left, 1u << 30);
      ok = gzwrite(f, p, chunk) == (int)chunk;
      p += chunk;
      left -= chunk;
    }
    ok = gzclose(f) == Z_OK && ok;
    if (!ok) throw ExceptionObject("write failed: " + path, "ImageFileWriter");
  } else {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw ExceptionObject("cannot create " + path, "ImageFileWriter");
    f.write(reinterpret_cast<const char *>(&h), sizeof h);
    f.write(reinterpret_cast<const char *>(ext), 4);
    f.write(reinterpret_cast<const char *>(data), (std::streamsize)nb);
    if (!f) throw ExceptionObject("write failed: " + path, "ImageFileWriter");
  }
}

template <typename T>
inline void write_mhd(const std::string &path, const ImageBase3 &info, const T *data) {
  const char *et = NiftiCode<T>::value == 16 ? "MET_FLOAT" : NiftiCode<T>::value == 64 ? "MET_DOUBLE"
                   : NiftiCode<T>::value == 2 ? "MET_UCHAR" : NiftiCode<T>::value == 4 ? "MET_SHORT"
                   : "MET_USHORT";
  const Size3 &sz = info.GetLargestPossibleRegion().GetSize();
  std::string raw = path.substr(0, path.size() - 4) + ".raw";
  const size_t slash = raw.find_last_of('/');
  std::ofstream h(path);
  if (!h) throw ExceptionObject("cannot create " + path, "ImageFileWriter");
  h.precision(17);
  h << "ObjectType = Image\nNDims = 3\nBinaryData = True\nBinaryDataByteOrderMSB = False\n"
    << "CompressedData = False\n";
  const FileGeometry &fg = info.GetFileGeometry();
  if (fg.has_mhd) {
    if (!fg.mhd_transform.empty()) h << "TransformMatrix = " << fg.mhd_transform << "\n";
    if (!fg.mhd_center.empty()) h << "CenterOfRotation = " << fg.mhd_center << "\n";
    if (!fg.mhd_orientation.empty()) h << "AnatomicalOrientation = " << fg.mhd_orientation << "\n";
  }
  h << "Offset = " << info.GetOrigin()[0] << " " << info.GetOrigin()[1] << " " << info.GetOrigin()[2] << "\n"
    << "ElementSpacing = " << info.GetSpacing()[0] << " " << info.GetSpacing()[1] << " " << info.GetSpacing()[2] << "\n"
    << "DimSize = " << sz[0] << " " << sz[1] << " " << sz[2] << "\n"
    << "ElementType = " << et << "\n"
    << "ElementDataFile = " << (slash == std::string::npos ? raw : raw.substr(slash + 1)) << "\n";
  std::ofstream f(raw, std::ios::binary);
  if (!f) throw ExceptionObject("cannot create " + raw, "ImageFileWriter");
  f.write(reinterpret_cast<const char *>(data), (std::streamsize)((size_t)(sz[0] * sz[1] * sz[2]) * sizeof(T)));
  if (!f || !h) throw ExceptionObject("write failed: " + path, "ImageFileWriter");
}

}  // namespace io_detail

template <typename TImage>
class ImageFileReader {
 public:
  typedef ImageFileReader Self;
  ifeNewMacro(Self);
  void SetFileName(const std::string &f) { file_ = f; out_ = nullptr; }
  const std::string &GetFileName() const { return file_; }
  void Update() {
    if (out_.IsNotNull()) return;
    using namespace io_detail;
    RawVolume v;
    if (ends_with(file_, ".nii") || ends_with(file_, ".nii.gz")) v = read_nifti(file_);
    else if (ends_with(file_, ".mhd")) v = read_mhd(file_);
    else throw ExceptionObject("unsupported file type (use .nii, .nii.gz or .mhd): " + file_,
                               "ImageFileReader");
    typename TImage::Pointer img = TImage::New();
    img->SetRegions(v.size);
    img->SetSpacing(v.spacing);
    img->SetOrigin(v.origin);
    img->SetFileGeometry(v.geom);
    img->Allocate();
    convert_any<typename TImage::PixelType>(v, img->GetBufferPointer());
    out_ = img;
  }
  TImage *GetOutput() {
    Update();
    return out_.GetPointer();
  }

 private:
  std::string file_;
  typename TImage::Pointer out_;
};

template <typename TImage>
class ImageFileWriter {
 public:
  typedef ImageFileWriter Self;
  ifeNewMacro(Self);
  void SetFileName(const std::string &f) { file_ = f; }
  void SetInput(const TImage *img) { in_ = img; }
  void Update() {
    using namespace io_detail;
    if (!in_) throw ExceptionObject("no input", "ImageFileWriter");
    if (ends_with(file_, ".nii") || ends_with(file_, ".nii.gz"))
      write_nifti(file_, *in_, in_->GetBufferPointer());
    else if (ends_with(file_, ".mhd"))
      write_mhd(file_, *in_, in_->GetBufferPointer());
    else
      throw ExceptionObject("unsupported file type (use .nii, .nii.gz or .mhd): " + file_,
                            "ImageFileWriter");
  }

 private:
  std::string file_;
  const TImage *in_ = nullptr;
};

}  // namespace itk

#endif
