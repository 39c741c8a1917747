// DetermineEdgesForEqualizedHistogram.h -- host mirror of the reference's function template
// of the same name (include/ife/Statistics/DetermineEdgesForEqualizedHistogram.h:21-26):
//
//   determineEdgesForEqualizedHistogram(first, last, d_first, nBins);   // [first,last) sorted
//
// The walk runs on the device (ife_equalized_edges_f32 / _f64).  Errors follow the
// reference: std::out_of_range when there are more bins than samples (:36-38),
// std::logic_error when last precedes first (:31-33); where the reference asserts (:74,
// the walk would step past the last sample) this throws std::out_of_range as well.
#ifndef __DetermineEdgesForEqualizedHistogram_h
#define __DetermineEdgesForEqualizedHistogram_h

#include <algorithm>
#include <iterator>
#include <stdexcept>
#include <vector>

#include "ife/Host/Engine.h"

namespace ife {
namespace host {
inline int equalized_edges(ife_ctx *c, const float *v, int64_t n, int nb, float *e) {
  return ife_equalized_edges_f32(c, v, n, nb, e, IFE_MEM_HOST);
}
inline int equalized_edges(ife_ctx *c, const double *v, int64_t n, int nb, double *e) {
  return ife_equalized_edges_f64(c, v, n, nb, e, IFE_MEM_HOST);
}
}  // namespace host
}  // namespace ife

template <typename InputIt, typename OutputIt>
void determineEdgesForEqualizedHistogram(InputIt first, InputIt last, OutputIt d_first, size_t nBins) {
  typedef typename std::iterator_traits<InputIt>::value_type T;
  const typename std::iterator_traits<InputIt>::difference_type n = std::distance(first, last);
  if (n < 0) throw std::logic_error("Iterator first must come before iterator last");
  if ((size_t)n < nBins)
    throw std::out_of_range("Too many bins. Number of bins must be less or equal to number of samples");
  if (nBins < 2) return;
  const std::vector<T> samples(first, last);
  std::vector<T> edges(nBins - 1);
  ife::host::Engine &e = ife::host::Engine::Instance();
  const int rc = ife::host::equalized_edges(e.ctx(), samples.data(), (int64_t)samples.size(), (int)nBins,
                                            edges.data());
  if (rc == IFE_E_ARG || rc == IFE_E_STATE) throw std::out_of_range(ife_last_error(e.ctx()));
  e.check(rc, "determineEdgesForEqualizedHistogram");
  std::copy(edges.begin(), edges.end(), d_first);
}

#endif
