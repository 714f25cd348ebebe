// onewave_probe.hip -- would ONE wave per 2048-point frame (32 x (2 x 32): a single LDS exchange of 16 KiB, the last
// radix-2 across half-waves by v_permlane32_swap) beat the two-waves-per-frame band kernel?  LDS capacity allows 8 such
// frames per CU (2 waves/SIMD).  A frame-shaped body without HBM, per wave and frame:
//   230 packed VALU (table multiply folded, radix-32), 32 ds_write_b64 (the exchange), 32 + 31 ds_read_b64 (exchange back,
//   pass-2 twiddles from a shared table), 62 + 172 VALU (twiddle products, radix-32 pruned to 10 outputs),
//   10 v_permlane32_swap + 15 VALU (half-wave combine), 5 ds_write_b64 + 5 ds_read_b64 (mirror bins), 110 VALU (Hermitian
//   split, magnitudes, window search, bookkeeping)  = ~600 VALU, 105 LDS instructions, 16 KiB written, ~33 KiB read.
// Printed: clocks per wave-frame, and per frame and CU = that / (4 SIMDs x waves per SIMD), beside the same body without LDS
// and beside the two-waves-per-frame body of tools/lds_valu_probe.hip (829 per frame and CU at 3 waves/SIMD there).
// build: hipcc --offload-arch=gfx950 -O3 -o onewave_probe onewave_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2 __attribute__((ext_vector_type(2)));

template <int LDS>
__global__ __launch_bounds__(1024) void body(int iters, unsigned long long* out, float* sink) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane(wave * 16384);  // bytes: one 16 KiB tile per wave
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f;
  v2 x[32], w = {b, a};
#pragma unroll
  for (int u = 0; u < 32; u++) x[u] = (v2){a + u, b - u};
  for (int e = threadIdx.x; e < (int)(blockDim.x >> 6) * 4096; e += blockDim.x) lds[e] = (float)e;
  __syncthreads();
  const unsigned a8 = base + lane * 8;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#define VALU_BLOCK(N) _Pragma("unroll") for (int k = 0; k < (N); k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x[k & 31]) : "v"(w))
#define WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  for (int i = 0; i < iters; i++) {
    VALU_BLOCK(230);
    if (LDS) {
#pragma unroll
      for (int u = 0; u < 32; u++) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(x[u]), "n"((u & 31) * 512) : "memory");
      WAIT();
#pragma unroll
      for (int u = 0; u < 32; u++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[u]) : "v"(a8), "n"((u & 31) * 512) : "memory");
      v2 tw[8];
#pragma unroll
      for (int u = 0; u < 31; u++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(tw[u & 7]) : "v"(a8), "n"((u & 31) * 512) : "memory");
      WAIT();
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] += tw[u];
    }
    VALU_BLOCK(62);
    VALU_BLOCK(172);
    // half-wave combine: 10 swaps (plain VALU) + 15 packed
#pragma unroll
    for (int u = 0; u < 10; u++) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[u].x), "+v"(x[u + 10].y));
    VALU_BLOCK(15);
    if (LDS) {
#pragma unroll
      for (int u = 0; u < 5; u++) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(x[u]), "n"(u * 512) : "memory");
      WAIT();
#pragma unroll
      for (int u = 0; u < 5; u++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[u]) : "v"(a8), "n"(u * 512) : "memory");
      WAIT();
    }
    VALU_BLOCK(110);
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int u = 0; u < 32; u++) s += x[u].x + x[u].y;
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if (lane == 0) {
    out[2 * (blockIdx.x * (blockDim.x >> 6) + wave)] = c1 - c0;
    out[2 * (blockIdx.x * (blockDim.x >> 6) + wave) + 1] = t1 - t0;
  }
}

static unsigned long long* d_out;
static float* d_sink;

template <int LDS>
static void row(int wps, int iters, const char* what) {
  const int waves = 4 * wps, nwg = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(body<LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, waves * 16384);
  for (int rep = 0; rep < 3; rep++)
    hipLaunchKernelGGL((body<LDS>), dim3(nwg), dim3(64 * waves), waves * 16384, 0, iters, d_out, d_sink);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%d waves/SIMD  %s: launch failed (LDS)\n", wps, what); (void)hipGetLastError(); return; }
  static unsigned long long h[2 * 256 * 16];
  hipMemcpy(h, d_out, sizeof(unsigned long long) * 2 * nwg * waves, hipMemcpyDeviceToHost);
  double c = 0, t = 0, cmax = 0;
  for (int g = 0; g < nwg; g++) {
    unsigned long long m = 0;
    for (int w = 0; w < waves; w++) {
      const int i = g * waves + w;
      c += (double)h[2 * i];
      t += (double)h[2 * i + 1];
      if (h[2 * i] > m) m = h[2 * i];
    }
    cmax += (double)m;
  }
  const double tw = cmax / nwg / iters;
  printf("%d waves/SIMD  one wave per frame, %-28s %8.1f clocks per wave-frame = %6.1f per frame and CU  [%.2f GHz]\n", wps, what, tw,
         tw / (4.0 * wps), c / t * 0.1);
}

int main(int argc, char** argv) {
  hipMalloc(&d_out, sizeof(unsigned long long) * 2 * 256 * 16);
  hipMalloc(&d_sink, 4096 * 4);
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  for (int wps = 1; wps <= 2; wps++) {
    row<0>(wps, iters, "no LDS (~600 VALU)");
    row<1>(wps, iters, "with its LDS traffic");
  }
  return 0;
}
