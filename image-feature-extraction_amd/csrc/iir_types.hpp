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
  // paired form (last pass of the normalized convolution): `in` is the numerator, `in2` the
  // denominator with its own checkpoint area, `out` receives numerator / denominator
  const float *in2;
  double *ck_y2;
  // constant lines (IFE_OPT_CONST_LINES): a line made of one bit pattern is answered with the
  // constant the host found this filter to make of it (ife_capi.hip const_line_flags): bit 0:
  // +0 -> +0, bit 1: 1.0f -> 1.0f, bit 2: -0 -> -0, bit 3: -0 -> +0; 0 = always filter
  uint32_t const_lines;
  IirCoef c;
};
struct IirJobs {
  IirJob j[IIR_MAX_JOBS];
};

// ---- Z pass of a Z-slab (multi-GPU): the recursion state crosses the slab boundary -------
// A slab holds planes [z0, z1) of every Z line.  The causal recursion enters from the slab
// below with the state y[z0-1..z0-4] and leaves with y[z1-1..z1-4]; the anticausal one enters
// from above with y[z1..z1+3] and leaves with y[z0..z0+3].  A record is those 4 doubles per
// line, struct-of-arrays y[k][line] (k = 0: nearest sample): 32 B per line and job.  The x
// history that belongs to a state (x[z0-1..z0-3], x[z1..z1+3]) does not travel: the INPUT of
// a slab with a neighbour carries that neighbour's adjacent planes (3 below `in`, 4 above
// its last plane), which every rank can produce from its own copy of the raw data.  Every
// pair of register blocks has one causal checkpoint (the y values in front of its first
// sample) and one anticausal checkpoint (the y values in front of its last sample, coming
// from above); their x history too is re-read from the samples around the pair.  The combine
// kernel rebuilds both recursions of a pair from them -- the same sequential arithmetic as the
// single-device kernel, sample for sample.
struct ZSlabJob {
  const float *in;         // plane 0 of the slab; planes -3..-1 / n..n+3 readable where a neighbour exists
  float *out;              // combine only
  double *cy, *ay;         // causal / anticausal checkpoints: the four y values, [npairs][4][ck_stride]
  const double *sin_y;     // incoming state of the sweep, [4][nlines]
  double *sout_y;          // outgoing state of the sweep, [4][nlines]
  IirCoef c;
};
struct ZSlabJobs {
  ZSlabJob j[IIR_MAX_JOBS];
};
struct ZSlabGeom {
  int64_t n;          // planes of the slab
  int64_t nlines;     // lines of this launch (pointers are already offset to the first one)
  int64_t sstride;    // elements between planes
  int64_t ck_stride;  // lines of the checkpoint arrays
  int32_t has_lo, has_hi;  // a slab below / above exists (else: ITK's border form)
  int32_t njobs, ngroups;
};

}  // namespace ife
