// DenseHistogram.h -- host mirror of the reference's DenseHistogram<NumType>
// (include/ife/Statistics/DenseHistogram.h:12-84): bins (-inf,e0], (e0,e1], ..., (e_last,inf).
// insert() queues the value; the binning of everything queued runs on the device
// (ife_dense_histogram_f32) when counts or frequencies are asked for.
#ifndef __DenseHistogram_h
#define __DenseHistogram_h

#include <initializer_list>
#include <iostream>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "ife/Host/Engine.h"
#include "ife/IO/IO.h"

template <typename NumType>
class DenseHistogram;
template <typename T>
std::ostream &operator<<(std::ostream &, DenseHistogram<T> &);

template <>
class DenseHistogram<float> {
 public:
  typedef float value_type;
  typedef std::vector<value_type>::size_type size_type;

  template <typename InputIt>
  DenseHistogram(InputIt begin, InputIt end) : m_Edges(begin, end), m_Counts(m_Edges.size() + 1) { check(); }
  DenseHistogram(std::initializer_list<value_type> edges) : m_Edges(edges), m_Counts(edges.size() + 1) {
    check();
  }

  void insert(value_type value) { m_Pending.push_back(value); }
  template <typename InputIt>
  void insert(InputIt begin, InputIt end) { m_Pending.insert(m_Pending.end(), begin, end); }

  std::vector<value_type> getFrequencies() {
    flush();
    // the reference sums into an int (the literal 0 of std::accumulate, :56), then divides floats
    const value_type sum = (value_type)std::accumulate(m_Counts.begin(), m_Counts.end(), 0);
    std::vector<value_type> f(m_Counts.size());
    for (size_type i = 0; i < f.size(); ++i) f[i] = (value_type)m_Counts[i] / sum;
    return f;
  }
  std::vector<unsigned int> getCounts() {
    flush();
    return m_Counts;
  }
  void resetCounts() {
    m_Pending.clear();
    std::fill(m_Counts.begin(), m_Counts.end(), 0u);
  }
  std::size_t getNumberOfBins() const { return m_Counts.size(); }

 private:
  void check() const {
    if (m_Edges.empty()) throw std::invalid_argument("DenseHistogram needs at least one edge");
  }
  void flush() {
    if (m_Pending.empty()) return;
    ife::host::Engine &e = ife::host::Engine::Instance();
    std::vector<unsigned int> c(m_Counts.size());
    e.check(ife_dense_histogram_f32(e.ctx(), m_Edges.data(), (int)m_Edges.size(), m_Pending.data(),
                                    (int64_t)m_Pending.size(), c.data(), IFE_MEM_HOST),
            "DenseHistogram");
    for (size_type i = 0; i < c.size(); ++i) m_Counts[i] += c[i];
    m_Pending.clear();
  }
  std::vector<value_type> m_Edges;
  std::vector<unsigned int> m_Counts;
  std::vector<value_type> m_Pending;
};

// "c0,c1,...": the row format of the .bag files (DenseHistogram.h:80-84)
template <typename T>
std::ostream &operator<<(std::ostream &os, DenseHistogram<T> &hist) {
  const std::vector<unsigned int> c = hist.getCounts();
  return writeSequenceAsText(os, c.begin(), c.end());
}

#endif
