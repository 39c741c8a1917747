// Path.h -- output naming helper with the semantics of the reference's
// include/ife/Util/Path.h:7-21: trailing separators of p1 and leading separators of p2
// are dropped and exactly one '/' is put between them (an all-separator p1 becomes "/").
#ifndef IFE_HOST_PATH_H
#define IFE_HOST_PATH_H

#include <string>

namespace Path {
inline std::string join(const std::string &a, const std::string &b) {
  const char sep = '/';
  std::string out = a;
  while (!out.empty() && out.back() == sep) out.pop_back();
  out.push_back(sep);
  size_t k = 0;
  while (k < b.size() && b[k] == sep) ++k;
  out.append(b, k, std::string::npos);
  return out;
}
}  // namespace Path

#endif
