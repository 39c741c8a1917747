// stats_kernels.hpp -- device side of row f1: the sample columns
// of tools/DetermineHistogramBinEdges_MultiScaleEigenvalueFeatures.cxx stay in HBM.
//
//   gather   :221-263  features of the foreground voxels appended to one column per
//                      (scale, feature)
//   sort     :284      std::sort of every column -> least-significant-digit radix sort,
//                      8-bit digits, all columns of a batch in one launch (blockIdx.y)
//   edges    :285-288  determineEdgesForEqualizedHistogram
//                      (include/ife/Statistics/DetermineEdgesForEqualizedHistogram.h:21-137)
//   histogram          DenseHistogram<float>::insert (include/ife/Statistics/
//                      DenseHistogram.h:47-53), row f2's inner loop
//
// Everything here is integer / compare work on 4-byte keys: HBM-bound, no MFMA.  Per
// radix pass a key is read twice (tile histogram, scatter) and written once: 12 B/key/pass,
// 48 B/key for the four passes against 8 B/key algorithmic (read once, write once).
#ifndef IFE_STATS_KERNELS_HPP
#define IFE_STATS_KERNELS_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ife {

constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / 64;
#ifndef IFE_SORT_KPT
#define IFE_SORT_KPT 16
#endif
#ifndef IFE_SORT_SUB
#define IFE_SORT_SUB 4
#endif
constexpr int SORT_KPT = IFE_SORT_KPT;                    // keys per thread and sub-tile
constexpr int SORT_SUB = IFE_SORT_SUB;                    // sub-tiles a workgroup walks through
constexpr int SORT_SUBTILE = SORT_THREADS * SORT_KPT;     // keys ranked in LDS at once
constexpr int SORT_TILE = SORT_SUBTILE * SORT_SUB;        // keys per workgroup
constexpr int SORT_WAVE_KEYS = SORT_SUBTILE / SORT_WAVES; // contiguous chunk of one wave
constexpr int SORT_SEGS = 16;                             // tile segments of the offset scan

struct SortGeom {
  int64_t n;        // keys per column
  int64_t stride;   // elements between columns (same for both ping-pong buffers)
  int ntiles;       // ceil(n / SORT_TILE)
  int seg_tiles;    // ceil(ntiles / SORT_SEGS)
};

// float bits -> unsigned key with the same order (-inf < ... < -0 < +0 < ... < +inf)
__device__ __forceinline__ uint32_t f32_to_key(uint32_t u) {
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint32_t key_to_f32(uint32_t k) {
  return (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// exclusive scan of one value per thread over a 256-thread workgroup; tmp holds 4 words
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *tmp, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < SORT_WAVES; ++w) {
    const uint32_t t = tmp[w];
    if (w < wave) base += t;
    sum += t;
  }
  __syncthreads();
  if (total) *total = sum;
  return base + inc - v;
}

// ---- pass part 1: digit counts of every tile ---------------------------------------------
// table[(col*ntiles + tile)*256 + digit].  The LDS counters are 32 copies per digit, one
// per bank, so that a wave's 64 adds meet at most two to an address whatever the digit
// distribution (the high bytes of float keys are nearly constant).
template <bool FIRST>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ table, SortGeom g, int shift) {
  __shared__ uint32_t cnt[256][32];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x, col = blockIdx.y;
#pragma unroll
  for (int c = 0; c < 32; ++c) cnt[tid][(c + tid) & 31] = 0;
  __syncthreads();
  const int64_t base = (int64_t)tile * SORT_TILE;
  const uint32_t *src = in + (int64_t)col * g.stride;
  const int copy = tid & 31;
  for (int sub = 0; sub < SORT_SUB; ++sub) {
    // all loads of a sub-tile first, then the adds: the loads stay in flight together
    const int64_t sb = base + (int64_t)sub * SORT_SUBTILE;
    if (sb >= g.n) break;
    uint32_t key[SORT_KPT];
#pragma unroll
    for (int r = 0; r < SORT_KPT; ++r) {
      const int64_t i = sb + r * SORT_THREADS + tid;
      key[r] = i < g.n ? src[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < SORT_KPT; ++r) {
      const int64_t i = sb + r * SORT_THREADS + tid;
      const uint32_t k = FIRST ? f32_to_key(key[r]) : key[r];
      if (i < g.n) atomicAdd(&cnt[(k >> shift) & 255u][copy], 1u);
    }
  }
  __syncthreads();
  uint32_t total = 0;
#pragma unroll
  for (int c = 0; c < 32; ++c) total += cnt[tid][(c + tid) & 31];
  table[((int64_t)col * g.ntiles + tile) * 256 + tid] = total;
}

// ---- pass part 2: tile counts -> global offsets, in two launches ---------------------------
// (a) per segment of tiles, the digit sums: segsum[(col*SORT_SEGS + seg)*256 + digit]
__global__ __launch_bounds__(SORT_THREADS) void sort_segsum_kernel(
    const uint32_t *__restrict__ table, uint32_t *__restrict__ segsum, SortGeom g) {
  const int tid = threadIdx.x, seg = blockIdx.x, col = blockIdx.y;
  const int t0 = seg * g.seg_tiles, t1 = min(t0 + g.seg_tiles, g.ntiles);
  const uint32_t *row = table + (int64_t)col * g.ntiles * 256 + tid;
  uint32_t sum = 0;
#pragma unroll 8
  for (int t = t0; t < t1; ++t) sum += row[(int64_t)t * 256];
  segsum[(col * SORT_SEGS + seg) * 256 + tid] = sum;
}
// (b) exclusive offsets: keys of smaller digits, then of earlier tiles of the same digit
__global__ __launch_bounds__(SORT_THREADS) void sort_scan_kernel(
    uint32_t *__restrict__ table, const uint32_t *__restrict__ segsum, SortGeom g) {
  __shared__ uint32_t tmp[SORT_WAVES];
  const int tid = threadIdx.x, seg = blockIdx.x, col = blockIdx.y;
  uint32_t digit_total = 0, before = 0;
#pragma unroll
  for (int s = 0; s < SORT_SEGS; ++s) {
    const uint32_t v = segsum[(col * SORT_SEGS + s) * 256 + tid];
    digit_total += v;
    if (s < seg) before += v;
  }
  uint32_t run = block_excl_scan_256(digit_total, tmp, nullptr) + before;
  const int t0 = seg * g.seg_tiles, t1 = min(t0 + g.seg_tiles, g.ntiles);
  uint32_t *row = table + (int64_t)col * g.ntiles * 256 + tid;
#pragma unroll 4
  for (int t = t0; t < t1; ++t) {
    const uint32_t c = row[(int64_t)t * 256];
    row[(int64_t)t * 256] = run;
    run += c;
  }
}

// ---- pass part 3: stable scatter ------------------------------------------------------------
// A workgroup walks its tile sub-tile by sub-tile.  Inside a sub-tile each wave owns a
// contiguous chunk, so (wave, round, lane) order is key order.  Ranks inside a wave come
// from ballots (lanes with the same digit), ranks across waves from the per-wave digit
// counts; the sub-tile is then laid out digit by digit in LDS so that the global stores of
// one digit are consecutive addresses.
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    const uint32_t *__restrict__ table, SortGeom g, int shift) {
  __shared__ uint32_t cnt[SORT_WAVES][256];
  __shared__ uint32_t sub_excl[256];
  __shared__ uint32_t gbase[256];
  __shared__ uint32_t tmp[SORT_WAVES];
  __shared__ uint32_t sorted[SORT_SUBTILE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile = blockIdx.x, col = blockIdx.y;
  const uint32_t *src = in + (int64_t)col * g.stride;
  uint32_t *dst = out + (int64_t)col * g.stride;
  uint32_t gofs = table[((int64_t)col * g.ntiles + tile) * 256 + tid];  // thread = digit
  // LDS pointer kept in its address space (a generic pointer would turn these into flat ops)
  typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;
  lds_u32 *wcnt = (lds_u32 *)&cnt[wave][0];

  for (int sub = 0; sub < SORT_SUB; ++sub) {
    const int64_t base = (int64_t)tile * SORT_TILE + (int64_t)sub * SORT_SUBTILE;
    if (base >= g.n) break;
    const int sub_n = (int)min((int64_t)SORT_SUBTILE, g.n - base);
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) cnt[w][tid] = 0;
    uint32_t key[SORT_KPT];
    uint32_t rank[SORT_KPT];
#pragma unroll
    for (int r = 0; r < SORT_KPT; ++r) {
      const int j = wave * SORT_WAVE_KEYS + r * 64 + lane;
      uint32_t k = j < sub_n ? src[base + j] : 0xffffffffu;
      if (FIRST && j < sub_n) k = f32_to_key(k);
      key[r] = k;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_KPT; ++r) {
      const int j = wave * SORT_WAVE_KEYS + r * 64 + lane;
      const bool valid = j < sub_n;
      const uint32_t d = (key[r] >> shift) & 255u;
      uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int32_t sx = (int32_t)(key[r] << (31 - b - shift));  // digit bit b in the sign
        const uint64_t bal = __builtin_amdgcn_ballot_w64(sx < 0);
        const uint32_t same = (uint32_t)(sx >> 31);  // ~0 where the bit is set
        const uint32_t lo = ~((uint32_t)bal ^ same), hi = ~((uint32_t)(bal >> 32) ^ same);
        peers &= ((uint64_t)hi << 32) | lo;  // lanes whose bit equals this lane's
      }
      const uint32_t below = lanes_below(peers);
      const uint32_t seen = wcnt[d];
      rank[r] = seen + below;
      __builtin_amdgcn_wave_barrier();
      if (valid && below == 0) wcnt[d] = seen + (uint32_t)__popcll(peers);
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {  // thread = digit: offsets of the waves inside the digit, of the digit inside the sub-tile
      uint32_t run = 0;
#pragma unroll
      for (int w = 0; w < SORT_WAVES; ++w) {
        const uint32_t c = cnt[w][tid];
        cnt[w][tid] = run;
        run += c;
      }
      const uint32_t ex = block_excl_scan_256(run, tmp, nullptr);
      sub_excl[tid] = ex;
      gbase[tid] = gofs - ex;
      gofs += run;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_KPT; ++r) {
      const int j = wave * SORT_WAVE_KEYS + r * 64 + lane;
      if (j < sub_n) {
        const uint32_t d = (key[r] >> shift) & 255u;
        sorted[sub_excl[d] + cnt[wave][d] + rank[r]] = key[r];
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < SORT_KPT; ++r) {
      const int j = r * SORT_THREADS + tid;
      if (j < sub_n) {
        const uint32_t k = sorted[j];
        const uint32_t d = (k >> shift) & 255u;
        dst[gbase[d] + (uint32_t)j] = LAST ? key_to_f32(k) : k;
      }
    }
    __syncthreads();
  }
}

// ---- gather --------------------------------------------------------------------------------
constexpr int GATHER_MAX_FG = 8;
struct GatherArgs {
  int64_t nvox;
  int64_t feat_comp_stride;   // PLANAR: nvox, INTERLEAVED: 1
  int64_t feat_vox_stride;    // PLANAR: 1,    INTERLEAVED: ncomp
  int64_t col_stride;         // elements between sample columns
  int64_t col_offset;         // samples already in these columns
  int ncomp;
  int nfg;
  uint32_t fg[GATHER_MAX_FG];
};

template <typename TM>
__device__ __forceinline__ bool is_foreground(const TM *mask, int64_t i, const GatherArgs &a) {
  const uint32_t m = (uint32_t)mask[i];
  bool hit = false;
  for (int k = 0; k < a.nfg; ++k) hit = hit || (m == a.fg[k]);
  return hit;
}

// Foreground compaction in raster order (tools/...MultiScaleEigenvalueFeatures.cxx:221-236)
// in three launches: per-chunk counts, their exclusive scan, the gather itself.
constexpr int GATHER_CHUNK = 4096;  // voxels per workgroup: 4 waves x 16 rounds x 64 lanes

template <typename TM>
__global__ __launch_bounds__(256) void count_foreground_kernel(const TM *__restrict__ mask,
                                                               GatherArgs a,
                                                               uint32_t *__restrict__ chunk_counts) {
  __shared__ uint32_t wsum[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t base = (int64_t)blockIdx.x * GATHER_CHUNK + wave * (GATHER_CHUNK / 4);
  uint32_t c = 0;
#pragma unroll 4
  for (int r = 0; r < GATHER_CHUNK / 256; ++r) {
    const int64_t i = base + r * 64 + lane;
    c += (i < a.nvox && is_foreground(mask, i, a)) ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if (lane == 0) wsum[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) chunk_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the chunk counts in place (one workgroup); total[0] = all foreground voxels
__global__ __launch_bounds__(SORT_THREADS) void scan_chunks_kernel(uint32_t *__restrict__ chunk_counts,
                                                                  int nchunks,
                                                                  unsigned long long *total) {
  __shared__ uint32_t tmp[SORT_WAVES];
  const int tid = threadIdx.x;
  const int per = (nchunks + SORT_THREADS - 1) / SORT_THREADS;
  const int lo = min(tid * per, nchunks), hi = min(lo + per, nchunks);
  uint32_t sum = 0;
  for (int t = lo; t < hi; ++t) sum += chunk_counts[t];
  uint32_t all = 0;
  uint32_t run = block_excl_scan_256(sum, tmp, &all);
  for (int t = lo; t < hi; ++t) {
    const uint32_t c = chunk_counts[t];
    chunk_counts[t] = run;
    run += c;
  }
  if (tid == 0) *total = all;
}

// append the ncomp feature values of every foreground voxel to ncomp columns, raster order
template <typename TM>
__global__ __launch_bounds__(256) void gather_foreground_kernel(const float *__restrict__ feat,
                                                                const TM *__restrict__ mask,
                                                                float *__restrict__ columns,
                                                                GatherArgs a,
                                                                const uint32_t *__restrict__ chunk_offsets) {
  __shared__ uint32_t wsum[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t base = (int64_t)blockIdx.x * GATHER_CHUNK + wave * (GATHER_CHUNK / 4);
  uint32_t flags = 0, wave_total = 0;
#pragma unroll 4
  for (int r = 0; r < GATHER_CHUNK / 256; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool fgv = i < a.nvox && is_foreground(mask, i, a);
    flags |= (fgv ? 1u : 0u) << r;
    wave_total += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fgv));
  }
  if (lane == 0) wsum[wave] = wave_total;
  __syncthreads();
  int64_t run = a.col_offset + chunk_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) run += wsum[w];
  for (int r = 0; r < GATHER_CHUNK / 256; ++r) {
    const bool fgv = (flags >> r) & 1u;
    const uint64_t m = __builtin_amdgcn_ballot_w64(fgv);
    if (fgv) {
      const int64_t i = base + r * 64 + lane;
      const int64_t pos = run + lanes_below(m);
      for (int c = 0; c < a.ncomp; ++c)
        columns[c * a.col_stride + pos] = feat[c * a.feat_comp_stride + i * a.feat_vox_stride];
    }
    run += __popcll(m);
  }
}

// append the feature values at listed voxels (the sampled branch, :238-263: the host
// draws the positions)
__global__ __launch_bounds__(256) void gather_indexed_kernel(const float *__restrict__ feat,
                                                             const int64_t *__restrict__ idx,
                                                             int64_t nidx,
                                                             float *__restrict__ columns,
                                                             GatherArgs a) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nidx;
       s += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx[s];
    for (int c = 0; c < a.ncomp; ++c)
      columns[c * a.col_stride + a.col_offset + s] =
          feat[c * a.feat_comp_stride + i * a.feat_vox_stride];
  }
}

// Fused sampling (FEAT_SAMPLES8): per voxel a code -- bit 0 the label is a foreground value
// (the voxel is sampled, tool :224-232), bit 1 the label is non-zero (its features are not
// masked to zero) -- and per 64-voxel row segment the number of sampled voxels.
// One wave per row segment s = bx + gx*row (x in [64 bx, min(nx, 64 bx + 64)) of row (y, z)):
// writes the codes and the segment's sample count.
template <typename TM>
__global__ __launch_bounds__(256) void sample_code_kernel(const TM *__restrict__ labels,
                                                          uint8_t *__restrict__ code,
                                                          uint32_t *__restrict__ counts, int nx,
                                                          int gx, int64_t nseg, GatherArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t s = wave0; s < nseg; s += nwaves) {
    const int64_t row = s / gx;
    const int x = (int)(s % gx) * 64 + lane;
    bool on = false;
    if (x < nx) {
      const int64_t i = row * nx + x;
      on = is_foreground(labels, i, a);
      code[i] = (uint8_t)((on ? 1 : 0) | (labels[i] != 0 ? 2 : 0));
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(on);
    if (lane == 0) counts[s] = (uint32_t)__popcll(m);
  }
}

// Exclusive scan of n counts in place, three launches: sums of chunks of SCAN_CHUNK entries,
// scan_chunks_kernel over those sums (total in total[0]), then the scan inside every chunk.
constexpr int SCAN_CHUNK = 4096;  // 256 threads x 16 entries
__global__ __launch_bounds__(SORT_THREADS) void chunk_sum_kernel(const uint32_t *__restrict__ v,
                                                                int64_t n,
                                                                uint32_t *__restrict__ sums) {
  __shared__ uint32_t tmp[SORT_WAVES];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x * 16;
  uint32_t t = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += base + k < n ? v[base + k] : 0u;
  uint32_t all = 0;
  (void)block_excl_scan_256(t, tmp, &all);
  if (threadIdx.x == 0) sums[blockIdx.x] = all;
}
__global__ __launch_bounds__(SORT_THREADS) void chunk_scan_kernel(uint32_t *__restrict__ v, int64_t n,
                                                                 const uint32_t *__restrict__ bases) {
  __shared__ uint32_t tmp[SORT_WAVES];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x * 16;
  uint32_t e[16], t = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    e[k] = base + k < n ? v[base + k] : 0u;
    t += e[k];
  }
  uint32_t run = bases[blockIdx.x] + block_excl_scan_256(t, tmp, nullptr);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (base + k < n) v[base + k] = run;
    run += e[k];
  }
}

// ClampImageFilter bounds (0, 1) of the tool (:147-152): labels -> {0, 1}
template <typename TM>
__global__ __launch_bounds__(256) void clamp01_kernel(const TM *__restrict__ in,
                                                      uint8_t *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[i] != 0 ? 1 : 0;
}

// ---- edges ----------------------------------------------------------------------------------
// determineEdgesForEqualizedHistogram, DetermineEdgesForEqualizedHistogram.h:21-137, one
// column per workgroup; the walk is sequential (each edge depends on the surplus / deficit
// the previous one left), so lane 0 walks and the binary searches are its dependent loads.
// status[col]: 0 ok, 1 fewer samples than bins (:36-38), 3 walk past the end (assert :74).
template <typename T>
__global__ __launch_bounds__(64) void equalized_edges_kernel(const T *__restrict__ cols,
                                                             int64_t col_stride, int64_t n,
                                                             int nbins_i, T *__restrict__ edges,
                                                             int *__restrict__ status) {
  if (threadIdx.x != 0) return;
  const int col = blockIdx.x;
  const T *v = cols + (int64_t)col * col_stride;
  T *e = edges + (int64_t)col * (nbins_i - 1);
  const int64_t nbins = nbins_i;
  if (n < nbins) { status[col] = 1; return; }
  const int64_t per_bin = n / nbins;
  int64_t surplus = n - per_bin * nbins, deficit = 0, nedge = 0, it = 0;
  while (nedge + 1 < nbins) {
    int64_t index = per_bin;
    if (surplus) {
      int64_t take = surplus / (nbins - nedge);
      if (take == 0) take = 1;
      index += take;
      surplus -= take;
    } else if (deficit) {
      int64_t take = deficit / (nbins - nedge);
      if (take == 0) take = 1;
      index -= take;
      deficit -= take;
    }
    if (!(n - it > index)) { status[col] = 3; return; }
    it += index;
    const T x = v[it];
    int64_t lo = 0, hi = it;
    while (lo < hi) {
      const int64_t mid = lo + (hi - lo) / 2;
      if (v[mid] < x) lo = mid + 1; else hi = mid;
    }
    const int64_t lb = lo;
    if (lb != it) {
      lo = it;
      hi = n;
      while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (!(x < v[mid])) lo = mid + 1; else hi = mid;
      }
      const int64_t ub = lo;
      if (ub == n) {
        it = lb;
      } else {
        const int64_t lbdist = it - lb, ubdist = ub - it;
        if (lbdist < ubdist || (lbdist == ubdist && deficit)) {
          it = lb;
          if (lbdist > deficit) { surplus = lbdist - deficit; deficit = 0; }
          else deficit -= lbdist;
        } else {
          it = ub;
          if (ubdist > surplus) { deficit = ubdist - surplus; surplus = 0; }
          else surplus -= ubdist;
        }
      }
    }
    e[nedge++] = v[it];
  }
  status[col] = 0;
}

// ---- dense histogram (row f2) --------------------------------------------------------------
// DenseHistogram<float>::insert (DenseHistogram.h:47-53): bin = lower_bound(edges, value),
// i.e. (-inf,e0], (e0,e1], ..., (e_last,inf).  Edges live in LDS, counts in LDS per
// workgroup, merged with one global atomic per bin and workgroup.
constexpr int HIST_MAX_EDGES = 1024;
__global__ __launch_bounds__(256) void dense_histogram_kernel(const float *__restrict__ edges,
                                                              int nedges,
                                                              const float *__restrict__ values,
                                                              int64_t n,
                                                              unsigned int *__restrict__ counts) {
  __shared__ float e[HIST_MAX_EDGES];
  __shared__ unsigned int c[HIST_MAX_EDGES + 1];
  for (int i = threadIdx.x; i < nedges; i += blockDim.x) e[i] = edges[i];
  for (int i = threadIdx.x; i <= nedges; i += blockDim.x) c[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float x = values[i];
    int lo = 0, hi = nedges;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (e[mid] < x) lo = mid + 1; else hi = mid;
    }
    atomicAdd(&c[lo], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= nedges; i += blockDim.x)
    if (c[i]) atomicAdd(&counts[i], c[i]);
}

// ---- per-ROI histograms: the bag rows of tools/MakeBag.cxx:405-472 ---------------------------
// grid (n_rois, ROI_SPLIT): each workgroup bins a z-range of one box for all components.
// Edges [ncomp][nedges] and counters [ncomp][nedges+1] live in LDS; counts are merged with
// global atomics (the host zeroes them first).
constexpr int ROI_SPLIT = 8;
constexpr int ROI_MAX_LDS_WORDS = 8192;  // ncomp * (2*nedges + 1) must fit
struct RoiArgs {
  int64_t nx, ny, nz;
  int64_t feat_comp_stride, feat_vox_stride;
  int ncomp, nedges;
};
template <typename TM>
__global__ __launch_bounds__(256) void roi_histogram_kernel(const float *__restrict__ feat,
                                                            const TM *__restrict__ mask,
                                                            const int64_t *__restrict__ rois,
                                                            const float *__restrict__ edges,
                                                            unsigned int *__restrict__ counts,
                                                            RoiArgs a) {
  __shared__ float lds[ROI_MAX_LDS_WORDS];
  const int nb = a.nedges + 1;
  float *e = lds;
  unsigned int *c = reinterpret_cast<unsigned int *>(lds + a.ncomp * a.nedges);
  for (int i = threadIdx.x; i < a.ncomp * a.nedges; i += blockDim.x) e[i] = edges[i];
  for (int i = threadIdx.x; i < a.ncomp * nb; i += blockDim.x) c[i] = 0;
  __syncthreads();
  const int64_t *q = rois + 6 * (int64_t)blockIdx.x;
  const int64_t x0 = q[0], y0 = q[1], z0 = q[2], sx = q[3], sy = q[4], sz = q[5];
  const int64_t zper = (sz + ROI_SPLIT - 1) / ROI_SPLIT;
  const int64_t za = min((int64_t)blockIdx.y * zper, sz), zb = min(za + zper, sz);
  const int64_t n = sx * sy * (zb - za);
  for (int64_t t = threadIdx.x; t < n; t += blockDim.x) {
    const int64_t x = t % sx, y = (t / sx) % sy, z = za + t / (sx * sy);
    const int64_t i = (x0 + x) + a.nx * ((y0 + y) + a.ny * (z0 + z));
    if (mask[i] == 0) continue;
    for (int k = 0; k < a.ncomp; ++k) {
      const float v = feat[k * a.feat_comp_stride + i * a.feat_vox_stride];
      const float *ek = e + k * a.nedges;
      int lo = 0, hi = a.nedges;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ek[mid] < v) lo = mid + 1; else hi = mid;
      }
      atomicAdd(&c[k * nb + lo], 1u);
    }
  }
  __syncthreads();
  unsigned int *out = counts + (int64_t)blockIdx.x * a.ncomp * nb;
  for (int i = threadIdx.x; i < a.ncomp * nb; i += blockDim.x)
    if (c[i]) atomicAdd(&out[i], c[i]);
}

}  // namespace ife
#endif
