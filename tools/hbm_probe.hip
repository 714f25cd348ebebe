// hbm_probe.hip -- what this MI355X's HBM delivers to the SIMPLEST kernels with the band kernel's traffic shape:
// a read-only stream (the frames: 16 B per lane, non-temporal, each byte once, a few bytes out) and a 1:1 copy.
// bench.py times both with HIP events and reports them as `roofline.achievable` next to the 8 TB/s spec figure
// ("a measured device-copy ceiling as the achievable line too").  Not part of libuchirp.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 8;  // 8 x 16 B in flight per lane

__device__ __forceinline__ v4u ld_nt(const v4u* p) { return __builtin_nontemporal_load(p); }

__global__ __launch_bounds__(kThreads) void read_kernel(const v4u* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
  const size_t tile = (size_t)kThreads * kUnroll;
  v4u acc = {0u, 0u, 0u, 0u};
  for (size_t base = (size_t)blockIdx.x * tile; base + tile <= n16; base += (size_t)gridDim.x * tile) {
    v4u v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) v[u] = ld_nt(src + base + (size_t)u * kThreads + threadIdx.x);
#pragma unroll
    for (int u = 0; u < kUnroll; u++) acc ^= v[u];
  }
  const unsigned r = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (r == 0x9e3779b9u) sink[blockIdx.x] = r;  // keeps the loads alive; (almost) never stores
}

__global__ __launch_bounds__(kThreads) void copy_kernel(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n16) {
  const size_t tile = (size_t)kThreads * kUnroll;
  for (size_t base = (size_t)blockIdx.x * tile; base + tile <= n16; base += (size_t)gridDim.x * tile) {
    v4u v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) v[u] = ld_nt(src + base + (size_t)u * kThreads + threadIdx.x);
#pragma unroll
    for (int u = 0; u < kUnroll; u++) __builtin_nontemporal_store(v[u], dst + base + (size_t)u * kThreads + threadIdx.x);
  }
}

// the same copy with a given number of 16-byte loads in flight per lane and threads per block (what a kernel with
// that footprint can expect: e.g. sinc5_kernel = one 1024-thread block per CU, 2 loads in flight per lane)
template <int UNROLL, int THREADS>
__global__ __launch_bounds__(THREADS) void copy_kernel_t(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n16) {
  const size_t tile = (size_t)THREADS * UNROLL;
  for (size_t base = (size_t)blockIdx.x * tile; base + tile <= n16; base += (size_t)gridDim.x * tile) {
    v4u v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = ld_nt(src + base + (size_t)u * THREADS + threadIdx.x);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) __builtin_nontemporal_store(v[u], dst + base + (size_t)u * THREADS + threadIdx.x);
  }
}

// sinc5_kernel's access shape without its arithmetic: wave w of the grid takes tiles w, w + W, ...; a tile is a
// 1024-byte load at byte 1008 * tile (16-byte aligned, NOT cache-line aligned, overlapping the next tile by 16 bytes)
// and a 1008-byte store at the same offset (lane 0 masked).  ALIGNED = 1: the same traffic with 1024-byte tiles.
template <int ALIGNED>
__global__ __launch_bounds__(1024) void copy_kernel_tiles(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n16) {
  const int lane = threadIdx.x & 63;
  const size_t step = ALIGNED ? 64 : 63;  // 16-byte units per tile
  const size_t tiles = (n16 - 64) / step;
  const size_t w0 = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6), W = (size_t)gridDim.x * 16;
  size_t t = w0;
  v4u q0 = {0u, 0u, 0u, 0u}, q1 = q0;
  if (t < tiles) q0 = ld_nt(src + t * step + lane);
  if (t + W < tiles) q1 = ld_nt(src + (t + W) * step + lane);
  for (; t < tiles; t += W) {
    const v4u v = q0;
    q0 = q1;
    if (t + 2 * W < tiles) q1 = ld_nt(src + (t + 2 * W) * step + lane);
    if (ALIGNED || lane > 0) __builtin_nontemporal_store(v, dst + t * step + lane - (ALIGNED ? 0 : 1));
  }
}

// aligned 1024-byte tiles, but every wave takes RUNS of `run` consecutive tiles (run r of the grid = tiles r*run ..),
// runs dealt round robin over the waves: what a kernel sees whose waves carry state from one tile into the next
__global__ __launch_bounds__(1024) void copy_kernel_runs(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n16, int run) {
  const int lane = threadIdx.x & 63;
  const size_t tiles = n16 / 64;
  const size_t w0 = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6), W = (size_t)gridDim.x * 16;
  const size_t nruns = tiles / run;
  for (size_t r = w0; r < nruns; r += W) {
    size_t t = r * run;
    v4u q0 = ld_nt(src + t * 64 + lane), q1 = ld_nt(src + (t + 1) * 64 + lane);
    for (int i = 0; i < run; i++, t++) {
      const v4u v = q0;
      q0 = q1;
      if (i + 2 < run) q1 = ld_nt(src + (t + 2) * 64 + lane);
      __builtin_nontemporal_store(v, dst + t * 64 + lane);
    }
  }
}

}  // namespace

extern "C" {

int hbm_probe_copy_runs(const void* src, void* dst, size_t bytes, int blocks, int run, void* stream) {
  hipLaunchKernelGGL(copy_kernel_runs, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, (const v4u*)src, (v4u*)dst, bytes / 16, run);
  return (int)hipGetLastError();
}

// shape 0: sinc5_kernel's tiles (1008-byte stride), 1: the same with 1024-byte tiles
int hbm_probe_copy_tiles(const void* src, void* dst, size_t bytes, int blocks, int aligned, void* stream) {
  if (aligned) hipLaunchKernelGGL((copy_kernel_tiles<1>), dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, (const v4u*)src, (v4u*)dst, bytes / 16);
  else hipLaunchKernelGGL((copy_kernel_tiles<0>), dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, (const v4u*)src, (v4u*)dst, bytes / 16);
  return (int)hipGetLastError();
}


// shape: 0 = 1024 threads x 2 in flight, 1 = 1024 x 4, 2 = 1024 x 8, 3 = 256 x 2
int hbm_probe_copy_shape(const void* src, void* dst, size_t bytes, int blocks, int shape, void* stream) {
  const v4u* s = (const v4u*)src;
  v4u* d = (v4u*)dst;
  hipStream_t st = (hipStream_t)stream;
  switch (shape) {
    case 0: hipLaunchKernelGGL((copy_kernel_t<2, 1024>), dim3((unsigned)blocks), dim3(1024), 0, st, s, d, bytes / 16); break;
    case 1: hipLaunchKernelGGL((copy_kernel_t<4, 1024>), dim3((unsigned)blocks), dim3(1024), 0, st, s, d, bytes / 16); break;
    case 2: hipLaunchKernelGGL((copy_kernel_t<8, 1024>), dim3((unsigned)blocks), dim3(1024), 0, st, s, d, bytes / 16); break;
    default: hipLaunchKernelGGL((copy_kernel_t<2, 256>), dim3((unsigned)blocks), dim3(256), 0, st, s, d, bytes / 16); break;
  }
  return (int)hipGetLastError();
}


// bytes: a multiple of 32 KiB is read in full (a tail shorter than one tile is skipped); returns hipError_t
int hbm_probe_read(const void* src, size_t bytes, void* sink, int blocks, void* stream) {
  hipLaunchKernelGGL(read_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, (const v4u*)src,
                     bytes / 16, (unsigned*)sink);
  return (int)hipGetLastError();
}

int hbm_probe_copy(const void* src, void* dst, size_t bytes, int blocks, void* stream) {
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, (const v4u*)src,
                     (v4u*)dst, bytes / 16);
  return (int)hipGetLastError();
}

}
