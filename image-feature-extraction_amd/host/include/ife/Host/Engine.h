// Engine.h -- process-wide ife_ctx shared by the host filter classes, and the
// translation of C-ABI status codes into itk::ExceptionObject, the error convention the
// reference's tools are written against (catch in main, print, EXIT_FAILURE:
// tools/ExtractFeatures.cxx:141-152).
#ifndef IFE_HOST_ENGINE_H
#define IFE_HOST_ENGINE_H

#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "ife/Host/Image.h"
#include "ife_hip.h"

namespace ife {
namespace host {

class Engine {
 public:
  // One context per process on device IFE_DEVICE (default 0); created on first use.
  static Engine &Instance() {
    static Engine e;
    return e;
  }
  ife_ctx *ctx() {
    if (!ctx_) {
      const char *d = std::getenv("IFE_DEVICE");
      const int rc = ife_ctx_create(d ? std::atoi(d) : 0, &ctx_);
      if (rc != IFE_OK) throw itk::ExceptionObject(ife_last_error(nullptr), "ife::host::Engine");
      if (const char *t = std::getenv("IFE_TRIG_MODE")) ife_ctx_set_option(ctx_, IFE_OPT_TRIG_MODE, std::atoi(t));
      if (const char *t = std::getenv("IFE_DSCALE_MODE")) ife_ctx_set_option(ctx_, IFE_OPT_DSCALE_MODE, std::atoi(t));
    }
    return ctx_;
  }
  void check(int rc, const char *where) {
    if (rc != IFE_OK) throw itk::ExceptionObject(ife_last_error(ctx_), where);
  }
  // IFE_DEVICES=0,1,2,3: the feature filter cuts the volume into Z-slabs over these devices
  // (ife_multi_*, one host thread, peer copies).  Unset or one entry: nullptr.
  ife_multi *multi() {
    if (!multi_tried_) {
      multi_tried_ = true;
      std::vector<int> devs;
      if (const char *d = std::getenv("IFE_DEVICES")) {
        std::stringstream ss(d);
        std::string tok;
        while (std::getline(ss, tok, ','))
          if (!tok.empty()) devs.push_back(std::atoi(tok.c_str()));
      }
      if (devs.size() > 1) {
        if (ife_multi_create(devs.data(), (int)devs.size(), &multi_) != IFE_OK)
          throw itk::ExceptionObject(ife_multi_last_error(nullptr), "ife::host::Engine");
        if (const char *t = std::getenv("IFE_TRIG_MODE")) ife_multi_set_option(multi_, IFE_OPT_TRIG_MODE, std::atoi(t));
        if (const char *t = std::getenv("IFE_DSCALE_MODE")) ife_multi_set_option(multi_, IFE_OPT_DSCALE_MODE, std::atoi(t));
      }
    }
    return multi_;
  }
  void check_multi(int rc, const char *where) {
    if (rc != IFE_OK) throw itk::ExceptionObject(ife_multi_last_error(multi_), where);
  }
  ~Engine() {
    if (multi_) ife_multi_destroy(multi_);
    if (ctx_) ife_ctx_destroy(ctx_);
  }

 private:
  Engine() = default;
  ife_ctx *ctx_ = nullptr;
  ife_multi *multi_ = nullptr;
  bool multi_tried_ = false;
};

inline ife_volume_desc describe(const itk::ImageBase3 &img) {
  ife_volume_desc d;
  const itk::Size3 &s = img.GetLargestPossibleRegion().GetSize();
  d.nx = (int64_t)s[0]; d.ny = (int64_t)s[1]; d.nz = (int64_t)s[2];
  d.sx = img.GetSpacing()[0]; d.sy = img.GetSpacing()[1]; d.sz = img.GetSpacing()[2];
  return d;
}

inline void same_size(const itk::ImageBase3 &a, const itk::ImageBase3 &b, const char *where) {
  const itk::Size3 &x = a.GetLargestPossibleRegion().GetSize(), &y = b.GetLargestPossibleRegion().GetSize();
  if (x[0] != y[0] || x[1] != y[1] || x[2] != y[2])
    throw itk::ExceptionObject("inputs do not occupy the same physical space (size mismatch)", where);
}

template <typename T> struct ImageDType;
template <> struct ImageDType<float> { static const int value = IFE_F32; };
template <> struct ImageDType<short> { static const int value = IFE_I16; };
template <typename T> struct MaskDType;
template <> struct MaskDType<unsigned char> { static const int value = IFE_U8; };
template <> struct MaskDType<unsigned short> { static const int value = IFE_U16; };

}  // namespace host
}  // namespace ife

#endif
