// onewave_shape3.hip -- the one-wave-per-frame SHAPE (tools/onewave_shape.hip) at THREE waves per SIMD: 12 waves per CU in one
// workgroup, the window*chirp table in LDS instead of 64 VGPRs (168 VGPRs per wave), and only 6 tiles for the 12 waves
// (waves w and w + 6 use tile w % 6 WITHOUT any synchronisation: the data race is irrelevant to the timing, and a real
// kernel's tile semaphore could only cost more) -- an OPTIMISTIC bound for that design.  Per wave and frame: 32 coalesced
// dword loads of the next frame, 32 table reads (ds_read_b64), the same butterfly-shaped placeholder arithmetic as
// onewave_shape.hip on random data, one exchange through the tile, pass-2 twiddles from LDS in four batches of 8.
// NOT a transform: timing only.
// build: hipcc --offload-arch=gfx950 -O3 -o onewave_shape3 onewave_shape3.hip ; run: ./onewave_shape3 [frames_log2=20] [iters=20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int kWaves = 12;                      // per workgroup = per CU: 3 per SIMD
constexpr int kTiles = 6;
constexpr int kTile = 65 * 32 * 2;              // floats: T[l][r] at l + 65 r (complex)
constexpr int kTw = 64 * 32 * 2;                // shared twiddle table
constexpr int kTab = 64 * 32 * 2;               // shared window*chirp table
constexpr int kLdsFloats = kTw + kTab + kTiles * kTile;

__global__ __launch_bounds__(64 * kWaves, 1) void shape(const float* __restrict__ frames, unsigned n_frames, unsigned char* sym) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* tile = lds + kTw + kTab + (wave % kTiles) * kTile;
  for (int e = threadIdx.x; e < kTw + kTab; e += blockDim.x) lds[e] = 0.5f + 1.0f / (1 + e);
  __syncthreads();
  const unsigned nwaves = gridDim.x * kWaves, w0 = blockIdx.x * kWaves + wave;
  const unsigned ngroups = (n_frames + 31) / 32;
  v2 x[32];
  float raw[32];
  const v2 m2 = {-2.0f, -2.0f}, half = {0.03125f, 0.03125f};
  unsigned grp = w0;
  if (grp >= ngroups) return;
  unsigned f = grp * 32;
  auto load = [&](unsigned fr) {
    const float* p = frames + (size_t)fr * 2048 + lane;
#pragma unroll
    for (int m = 0; m < 32; m++) raw[m] = __builtin_nontemporal_load(p + 64 * m);
  };
  load(f);
  const unsigned wr = (unsigned)(size_t)(tile - lds) * 4 + lane * 8;                        // + 520 r
  const unsigned rd = (unsigned)(size_t)(tile - lds) * 4 + ((lane >> 5) + 65 * (lane & 31)) * 8;  // + 16 u
  const unsigned twa = lane * 8;                                                            // + 512 r
  const unsigned taba = kTw * 4 + lane * 8;                                                 // + 512 m
  float accum = 0.f;
#define STAGE(S)                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 32; i++)                                                                   \
    if ((i & (S)) == 0) asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_fma_f32 %1, %1, %2, %0" : "+v"(x[i]), "+v"(x[i ^ (S)]) : "v"(m2))
#define SCALE() _Pragma("unroll") for (int i = 0; i < 32; i += 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(half))
  for (;;) {
    unsigned fn = f + 1;
    if ((fn & 31) == 0 || fn >= n_frames) { grp += nwaves; fn = grp * 32; }
    const bool more = grp < ngroups && fn < n_frames;
    // table multiply, the table out of LDS in four batches of 8
#pragma unroll
    for (int b = 0; b < 4; b++) {
      v2 t8[8];
#pragma unroll
      for (int m = 0; m < 8; m++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t8[m]) : "v"(taba), "n"((8 * b + m) * 512) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int m = 0; m < 8; m++) x[8 * b + m] = t8[m] * raw[8 * b + m];
    }
    if (more) load(fn);
    STAGE(1); STAGE(2); STAGE(4); STAGE(8); STAGE(16); STAGE(1);   // 192: the radix-32
#pragma unroll
    for (int r = 0; r < 32; r++) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wr), "v"(x[r]), "n"(r * 520) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 32; u++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[u]) : "v"(rd), "n"(u * 16) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < 4; b++) {
      v2 t8[8];
#pragma unroll
      for (int m = 0; m < 8; m++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t8[m]) : "v"(twa), "n"((8 * b + m) * 512) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int m = 0; m < 8; m++)
        if (8 * b + m) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_fma_f32 %0, %0, %1, %0" : "+v"(x[8 * b + m]) : "v"(t8[m]));
    }
    STAGE(2); STAGE(4); STAGE(8); STAGE(16); STAGE(1);   // 160: the pruned radix-32
    SCALE();                                             // 11
#pragma unroll
    for (int u = 0; u < 10; u++) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[u].x), "+v"(x[u + 10].y));
    STAGE(2);                                            // 32: combine, Hermitian split
#pragma unroll
    for (int u = 0; u < 5; u++) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wr), "v"(x[u]), "n"(u * 512) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 5; u++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[u]) : "v"(wr), "n"(u * 512) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAGE(4); STAGE(8); STAGE(16);                       // 96: magnitudes, window search, bookkeeping
    accum += x[0].x + x[17].y;
    if ((f & 31) == 31 || !more) {
      const unsigned f0 = f & ~31u;
      if (lane < 32 && f0 + lane < n_frames) sym[f0 + lane] = (unsigned char)(accum > 1e30f);
    }
    if (!more) break;
    f = fn;
  }
}

__global__ void fill_noise(float* d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + (unsigned)(i >> 32) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    d[i] = ((int)(h & 0xffff) - 32768) * 0.25f;
  }
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 20, iters = argc > 2 ? atoi(argv[2]) : 20;
  const size_t nf = (size_t)1 << lg;
  float* d;
  unsigned char* s;
  if (hipMalloc(&d, nf * 2048 * 4) != hipSuccess || hipMalloc(&s, nf) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipLaunchKernelGGL(fill_noise, dim3(4096), dim3(256), 0, 0, d, nf * 2048);
  const size_t ldsb = kLdsFloats * 4;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(shape), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) {
    printf("LDS opt-in failed (%zu bytes)\n", ldsb);
    return 1;
  }
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int i = 0; i < 60; i++) hipLaunchKernelGGL(shape, dim3(256), dim3(64 * kWaves), ldsb, 0, d, (unsigned)nf, s);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<float> ts;
  for (int i = 0; i < iters; i++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(shape, dim3(256), dim3(64 * kWaves), ldsb, 0, d, (unsigned)nf, s);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("one-wave-per-frame shape at 3 waves/SIMD: %zu frames, LDS %zu B per CU, median %.4f ms per launch = %.4g frames/s (err %s)\n", nf,
         ldsb, ts[ts.size() / 2], nf / (ts[ts.size() / 2] * 1e-3), hipGetErrorString(hipGetLastError()));
  return 0;
}
