// EigenvalueFeaturesFunctor.h -- host mirror of
// include/ife/Numerics/EigenvalueFeaturesFunctor.h:10-31: (ev0, ev1, ev2, sum, product,
// Frobenius norm) of the magnitude-sorted eigenvalues.
#ifndef __EigenvalueFeaturesFunctor_h
#define __EigenvalueFeaturesFunctor_h

#include "ife/Numerics/Symmetric3x3EigenvalueSolver.h"

template <typename TRealType>
struct EigenvalueFeaturesFunctor : public Symmetric3x3EigenvalueSolver<TRealType> {
  typedef Symmetric3x3EigenvalueSolver<TRealType> SuperClass;
  typedef typename SuperClass::InputType InputType;
  typedef typename SuperClass::OutputType OutputType;
  OutputType operator()(const InputType &A) const {
    assert(A.Size() == 6);
    OutputType f(6);
    Apply(A.GetDataPointer(), 1, f.GetDataPointer());
    return f;
  }
  static void Apply(const TRealType *A6, int64_t n, TRealType *f6) {
    ife::host::Engine &e = ife::host::Engine::Instance();
    e.check(ife_eigenvalue_features(e.ctx(), A6, n, f6, IFE_MEM_HOST), "EigenvalueFeaturesFunctor");
  }
};

#endif
