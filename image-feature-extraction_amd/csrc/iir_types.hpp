// iir_types.hpp -- host-visible types of the recursive-Gaussian line kernels
// (shared by both builds of the kernels: exact and fused-multiply-add, iir_kernels.inc).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ife {

struct IirCoef {
  double N0, N1, N2, N3;
  double D1, D2, D3, D4;
  double M1, M2, M3, M4;
  double BN1, BN2, BN3, BN4;
  double BM1, BM2, BM3, BM4;
};

struct IirGeom {
  int64_t n;        // samples per line
  int64_t nlines;   // number of lines
  int64_t sstride;  // element stride between consecutive samples of a line
  int64_t inner;    // strided: lines per contiguous row (line L -> base (L%inner)+(L/inner)*outer)
  int64_t outer;    // strided: element stride between rows of lines; contig: line pitch
  // contiguous-axis kernel only: the INPUT may be the Y-chunked image of a Z-slab as an
  // all-to-all leaves it, [in_w][nz][in_group][nx] (chunk h holds rows h*in_group ..), while
  // the output is the plain [nz][ny][nx] slab.  in_w <= 1: plain input.  Needs 64 | in_group
  // so that the 64 lines of a wave stay contiguous.
  int64_t in_w, in_group, in_nz;
  // launch shape: workgroups are numbered job-fastest (see iir_block_of)
  int32_t njobs, ngroups;
};

// Workgroup -> (line group, job).  The jobs of one line group are dispatched side by side
// and, ids lin, lin+8, ... sharing an XCD under round-robin dispatch, onto the same XCD: the
// first axis pass runs three scales over the SAME input, and the second and third reader
// then find it in that XCD's L2 or in the Infinity Cache instead of HBM.
struct IirBlock {
  uint32_t group, job;
  bool live;
};
__device__ __forceinline__ IirBlock iir_block_of(const IirGeom &g, uint32_t lin) {
  const uint32_t per = 8u * (uint32_t)g.njobs;
  const uint32_t blk = lin / per, r = lin % per;
  IirBlock b;
  b.job = r / 8u;
  b.group = blk * 8u + (r % 8u);
  b.live = b.group < (uint32_t)g.ngroups;
  return b;
}

constexpr int IIR_MAX_JOBS = 8;
struct IirJob {
  const float *in;
  float *out;
  double *ck_y;  // [npairs][4][nlines]: y[i-1..i-4] at the start of every second block
  float *ck_x;   // [npairs][3][nlines]: x[i-1..i-3], contiguous-axis kernel only
  IirCoef c;
};
struct IirJobs {
  IirJob j[IIR_MAX_JOBS];
};

}  // namespace ife
