// Symmetric3x3EigenvalueSolver.h -- host mirror of the reference functor
// (include/ife/Numerics/Symmetric3x3EigenvalueSolver.h:10-33): operator() on one 6-vector
// (xx,xy,xz,yy,yz,zz) returns the three eigenvalues by decreasing magnitude.  The device
// does the arithmetic (ife_eigenvalues); a per-voxel call is only sensible for tests, so
// Apply() takes a whole batch.  TRealType = float is the instantiation the tools use.
#ifndef __Symmetric3x3EigenvalueSolver_h
#define __Symmetric3x3EigenvalueSolver_h

#include <cassert>
#include <type_traits>
#include <vector>

#include "ife/Host/Engine.h"

namespace itk {
template <typename T>
class VariableLengthVector {
 public:
  VariableLengthVector() {}
  explicit VariableLengthVector(unsigned int n) : v_(n) {}
  VariableLengthVector(const T *p, unsigned int n) : v_(p, p + n) {}
  unsigned int Size() const { return (unsigned int)v_.size(); }
  unsigned int GetSize() const { return Size(); }
  T &operator[](unsigned int i) { return v_[i]; }
  const T &operator[](unsigned int i) const { return v_[i]; }
  const T *GetDataPointer() const { return v_.data(); }
  T *GetDataPointer() { return v_.data(); }

 private:
  std::vector<T> v_;
};
}  // namespace itk

template <typename TRealType>
struct Symmetric3x3EigenvalueSolver {
  static_assert(std::is_same<TRealType, float>::value,
                "the device path implements TRealType = float (ImageToEmphysemaFeaturesFilter.h:94)");
  typedef TRealType RealType;
  typedef itk::VariableLengthVector<RealType> InputType;
  typedef itk::VariableLengthVector<RealType> OutputType;
  bool operator!=(const Symmetric3x3EigenvalueSolver &) const { return false; }
  bool operator==(const Symmetric3x3EigenvalueSolver &o) const { return !(*this != o); }

  OutputType operator()(const InputType &A) const {
    assert(A.Size() == 6);
    OutputType ev(3);
    Apply(A.GetDataPointer(), 1, ev.GetDataPointer());
    return ev;
  }
  // n matrices, A6[n][6] -> ev[n][3]
  static void Apply(const RealType *A6, int64_t n, RealType *ev) {
    ife::host::Engine &e = ife::host::Engine::Instance();
    e.check(ife_eigenvalues(e.ctx(), A6, n, ev, IFE_MEM_HOST), "Symmetric3x3EigenvalueSolver");
  }
};

#endif
