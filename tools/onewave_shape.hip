// onewave_shape.hip -- the SHAPE of a one-wave-per-frame band kernel, with real HBM traffic but placeholder arithmetic:
// does the structure (32 x (2 x 32): one LDS exchange of 16.6 KiB per frame, no barrier, 8 independent waves per CU) run
// faster than the two-waves-per-frame kernel (~2.0 ms per 2^20 frames)?  Per wave and frame: 32 coalesced dword loads
// of the NEXT frame (a whole frame time ahead), ~600 dependent-chain packed FMAs in the blocks of the real algorithm,
// 32 ds_write_b64 + 32 + 31 ds_read_b64 with the real (padded, conflict-free) addresses, 10 v_permlane32_swap,
// 5 + 5 LDS accesses of the mirror exchange, one byte stored per frame by a finaliser lane.  NOT a transform: timing only.
// The placeholder arithmetic is butterfly-shaped (a += b; b = a - 2 b: one packed add + one packed FMA, in place, over
// random data with a resident 32-entry complex table) so that operands toggle as a transform's do: the clock the chip
// holds depends on that (zeros: 2.38 GHz, noise: 1.9 GHz under the real kernel).
// build: hipcc --offload-arch=gfx950 -O3 -o onewave_shape onewave_shape.hip ; run: ./onewave_shape [frames_log2=20] [iters=20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int kWaves = 8;                       // per workgroup = per CU
constexpr int kTile = 65 * 32 * 2;              // floats: T[l][r] at l + 65 r (complex)
constexpr int kTw = 64 * 32 * 2;                // shared twiddle table
constexpr int kLdsFloats = kTw + kWaves * kTile + kWaves * 64;

__global__ __launch_bounds__(64 * kWaves, 1) void shape(const float* __restrict__ frames, unsigned n_frames, unsigned char* sym,
                                                       unsigned* ctr) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* tile = lds + kTw + wave * kTile;
  for (int e = threadIdx.x; e < kTw; e += blockDim.x) lds[e] = 1.0f / (1 + e);
  __syncthreads();
  const unsigned nwaves = gridDim.x * kWaves, w0 = blockIdx.x * kWaves + wave;
  // static round robin over groups of 32 frames (the probe has no dynamic hand-out)
  const unsigned ngroups = (n_frames + 31) / 32;
  v2 x[32];
  float raw[32];
  v2 tab[32];  // resident table, as the real kernel's window * chirp entries (64 VGPRs)
#pragma unroll
  for (int m = 0; m < 32; m++) tab[m] = (v2){0.5f + 0.001f * ((lane * 7 + m * 13) & 63), 0.3f - 0.002f * ((lane * 5 + m * 11) & 63)};
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
  float accum = 0.f;
// one butterfly stage over the 32 registers with partner stride S: 16 x (packed add + packed FMA) = 32 instructions
#define STAGE(S)                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 32; i++)                                                                   \
    if ((i & (S)) == 0) asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_fma_f32 %1, %1, %2, %0" : "+v"(x[i]), "+v"(x[i ^ (S)]) : "v"(m2))
#define SCALE() _Pragma("unroll") for (int i = 0; i < 32; i += 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(half))
  for (;;) {
    unsigned fn = f + 1;
    if ((fn & 31) == 0 || fn >= n_frames) { grp += nwaves; fn = grp * 32; }
    const bool more = grp < ngroups && fn < n_frames;
#pragma unroll
    for (int m = 0; m < 32; m++) x[m] = tab[m] * raw[m];  // 32 (table multiply)
    if (more) load(fn);
    STAGE(1); STAGE(2); STAGE(4); STAGE(8); STAGE(16); STAGE(1);   // 192: the radix-32
#pragma unroll
    for (int r = 0; r < 32; r++) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wr), "v"(x[r]), "n"(r * 520) : "memory");
    v2 tw[31];
#pragma unroll
    for (int r = 1; r < 32; r++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(tw[r - 1]) : "v"(twa), "n"(r * 512) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 32; u++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[u]) : "v"(rd), "n"(u * 16) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 1; r < 32; r++) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_fma_f32 %0, %0, %1, %0" : "+v"(x[r]) : "v"(tw[r - 1]));
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
    if ((f & 31) == 31 || !more) {  // one lane per frame of the finished group stores a byte (the finaliser's store)
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
    d[i] = ((int)(h & 0xffff) - 32768) * 0.25f;   // uniform in +-8192: sigma ~ the bench's -10 dB noise (3162) x 1.5
  }
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 20, iters = argc > 2 ? atoi(argv[2]) : 20;
  const size_t nf = (size_t)1 << lg;
  float* d;
  unsigned char* s;
  if (hipMalloc(&d, nf * 2048 * 4) != hipSuccess || hipMalloc(&s, nf) != hipSuccess) { printf("alloc failed\n"); return 1; }
  if (argc > 3 && atoi(argv[3]) == 0) (void)hipMemset(d, 0, nf * 2048 * 4);
  else hipLaunchKernelGGL(fill_noise, dim3(4096), dim3(256), 0, 0, d, nf * 2048);
  const size_t ldsb = kLdsFloats * 4;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(shape), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) {
    printf("LDS opt-in failed (%zu bytes)\n", ldsb);
    return 1;
  }
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int i = 0; i < 60; i++) hipLaunchKernelGGL(shape, dim3(256), dim3(64 * kWaves), ldsb, 0, d, (unsigned)nf, s, nullptr);
  (void)hipDeviceSynchronize();
  std::vector<float> ts;
  for (int i = 0; i < iters; i++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(shape, dim3(256), dim3(64 * kWaves), ldsb, 0, d, (unsigned)nf, s, nullptr);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("one-wave-per-frame shape: %zu frames, LDS %zu B per CU, median %.4f ms per launch = %.4g frames/s (err %s)\n", nf, ldsb,
         ts[ts.size() / 2], nf / (ts[ts.size() / 2] * 1e-3), hipGetErrorString(hipGetLastError()));
  return 0;
}
