// ref_stats_driver.cxx -- C entry points around the reference's own ITK-free statistics
// and IO headers, compiled where they lie (-I/root/reference/include) into
// oracle/_ref/libife_ref_stats.so by `make -C oracle _ref`.  TEST INFRASTRUCTURE ONLY.
//
// What is reference code here: everything behind the #include "ife/..." lines and the
// two reference translation units the Makefile compiles beside this file
// (src/Util/String.cxx, src/IO/IO.cxx).  What is ours: this driver, which only marshals
// plain arrays in and out.  The standard headers below are included first because the
// reference headers use assert, std::abs and std::out_of_range without including them.
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "ife/IO/IO.h"
#include "ife/Statistics/DenseHistogram.h"
#include "ife/Statistics/DetermineEdgesForEqualizedHistogram.h"

namespace {

template <typename T>
int edges(const T *sorted, size_t n, size_t nbins, T *out) {
  try {
    determineEdgesForEqualizedHistogram(sorted, sorted + n, out, nbins);
  } catch (const std::out_of_range &) {
    return 1;
  } catch (const std::logic_error &) {
    return 2;
  }
  return 0;
}

int copy_out(const std::string &s, char *buf, size_t cap) {
  if (s.size() + 1 > cap) return -1;
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}

}  // namespace

extern "C" {

// determineEdgesForEqualizedHistogram (DetermineEdgesForEqualizedHistogram.h:21-137);
// out must hold nbins-1 values.  0 ok, 1 std::out_of_range, 2 std::logic_error.
int ife_ref_edges_f32(const float *sorted, size_t n, size_t nbins, float *out) {
  return edges(sorted, n, nbins, out);
}
int ife_ref_edges_f64(const double *sorted, size_t n, size_t nbins, double *out) {
  return edges(sorted, n, nbins, out);
}

// DenseHistogram<float>: insert every value, then getCounts / getFrequencies
// (DenseHistogram.h:29-66).  counts and freqs hold nedges+1 entries.
int ife_ref_dense_histogram_f32(const float *edge, size_t nedges, const float *values, size_t n,
                                unsigned int *counts, float *freqs) {
  DenseHistogram<float> h(edge, edge + nedges);
  for (size_t i = 0; i < n; ++i) h.insert(values[i]);
  const std::vector<unsigned int> c = h.getCounts();
  const std::vector<float> f = h.getFrequencies();
  for (size_t i = 0; i < c.size(); ++i) {
    counts[i] = c[i];
    freqs[i] = f[i];
  }
  return (int)c.size();
}

// writeSequenceAsText (IO.h:24-41) on floats, the format of the edge file rows
int ife_ref_write_sequence_f32(const float *v, size_t n, char sep, char *buf, size_t cap) {
  std::ostringstream os;
  writeSequenceAsText(os, v, v + n, sep);
  return copy_out(os.str(), buf, cap);
}

// readPairList (src/IO/IO.cxx:20-41): pairs come back as "first\tsecond\n" lines;
// -2 when the reference throws (a line without separator).
int ife_ref_read_pair_list(const char *path, char sep, char *buf, size_t cap) {
  try {
    const std::vector<StringPair> pairs = readPairList(path, sep);
    std::string s;
    for (const auto &p : pairs) s += p.first + "\t" + p.second + "\n";
    return copy_out(s, buf, cap);
  } catch (const std::invalid_argument &) {
    return -2;
  }
}

}  // extern "C"
