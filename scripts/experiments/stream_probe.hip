// Throw-away probe: fill / copy rates of a 4-GiB buffer for several kernel shapes
// (hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip; run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void fill_gs(f32x4 *d, long n4) {
  const f32x4 x = {1.f, 1.f, 1.f, 1.f};
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if (NT) __builtin_nontemporal_store(x, d + i); else d[i] = x;
  }
}
// contiguous chunk per block, 4 stores per thread per iteration
template <bool NT>
__global__ __launch_bounds__(256) void fill_chunk(f32x4 *d, long n4) {
  const f32x4 x = {1.f, 1.f, 1.f, 1.f};
  const long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long b0 = (long)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
  for (long i = b0 + threadIdx.x; i < b1; i += 1024) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < b1) { if (NT) __builtin_nontemporal_store(x, d + i + k * 256); else d[i + k * 256] = x; }
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void copy_gs(const f32x4 *s, f32x4 *d, long n4) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i); else d[i] = s[i];
  }
}
template <typename F> double timeit(F f, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  const size_t bytes = (size_t)4 << 30; const long n4 = bytes / 16;
  f32x4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  for (int blocks : {1024, 2048, 8192, 65536, (int)(n4 / 256)}) {
    double t;
    t = timeit([&] { hipLaunchKernelGGL(fill_gs<true>, dim3(blocks), dim3(256), 0, 0, a, n4); });
    printf("fill grid-stride nt   blocks %7d: %.1f GB/s\n", blocks, bytes / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(fill_gs<false>, dim3(blocks), dim3(256), 0, 0, a, n4); });
    printf("fill grid-stride      blocks %7d: %.1f GB/s\n", blocks, bytes / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(fill_chunk<true>, dim3(blocks), dim3(256), 0, 0, a, n4); });
    printf("fill chunked nt       blocks %7d: %.1f GB/s\n", blocks, bytes / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(fill_chunk<false>, dim3(blocks), dim3(256), 0, 0, a, n4); });
    printf("fill chunked          blocks %7d: %.1f GB/s\n", blocks, bytes / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(copy_gs<true>, dim3(blocks), dim3(256), 0, 0, a, b, n4); });
    printf("copy grid-stride nt   blocks %7d: %.1f GB/s (read+write)\n", blocks, 2.0 * bytes / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(copy_gs<false>, dim3(blocks), dim3(256), 0, 0, a, b, n4); });
    printf("copy grid-stride      blocks %7d: %.1f GB/s (read+write)\n", blocks, 2.0 * bytes / t / 1e6);
  }
  // 1 GiB like the torch fill the round-2 profile happened to contain
  double t = timeit([&] { hipLaunchKernelGGL(fill_gs<false>, dim3(n4 / 4 / 256), dim3(256), 0, 0, a, n4 / 4); });
  printf("fill 1 GiB, one element per thread: %.1f GB/s\n", bytes / 4 / t / 1e6);
  return 0;
}
