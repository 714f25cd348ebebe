// mfma_valu_probe.hip -- do f32 MFMAs (v_mfma_f32_16x16x4_f32) and packed f32 VALU (v_pk_fma_f32) of two waves that
// share a SIMD overlap?  One 512-thread workgroup per CU = two waves per SIMD (waves 0-3 and 4-7 pair up on SIMDs
// 0-3).  Mode bits: what waves 0-3 run, what waves 4-7 run (0 = idle, 1 = MFMA loop, 2 = packed-FMA loop), each `iters`
// iterations of 16 instructions.  Prints ms per launch for: M|idle, V|idle, M|M, V|V, M|V.
// If the pipes were independent, M|V would take max(M|idle, V|idle); if they share the multipliers, about the sum.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void probe(int mode_lo, int mode_hi, int iters, float* sink) {
  const int wave = threadIdx.x >> 6;
  const int mode = wave < 4 ? mode_lo : mode_hi;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f;
  if (mode == 1) {
    v4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
      }
    }
    const v4 s = c0 + c1 + c2 + c3;
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[threadIdx.x] = s.x;
  } else if (mode == 2) {
    v2 x[8], w = {b, a};
#pragma unroll
    for (int u = 0; u < 8; u++) x[u] = (v2){a + u, b - u};
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int r = 0; r < 2; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x[u]) : "v"(w));
      }
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) s += x[u].x + x[u].y;
    if (s == 12345.678f) sink[threadIdx.x] = s;
  }
}

static float run(int lo, int hi, int iters, float* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, lo, hi, iters, sink);
  hipEventRecord(a, 0);
  for (int w = 0; w < 10; w++) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, lo, hi, iters, sink);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 10;
}

int main() {
  float* sink;
  hipMalloc(&sink, 4096);
  const int iters = 20000;  // x 16 instructions per wave
  const char* names[3] = {"idle", "MFMA", "VALU"};
  const int cases[5][2] = {{1, 0}, {2, 0}, {1, 1}, {2, 2}, {1, 2}};
  for (auto& c : cases) {
    const float ms = run(c[0], c[1], iters, sink);
    const double cyc_per_inst = ms * 1e-3 * 2.4e9 / (iters * 16.0);
    printf("waves 0-3: %-4s | waves 4-7: %-4s : %8.3f ms  (%.1f clocks of 2.4 GHz per instruction of one wave)\n", names[c[0]],
           names[c[1]], ms, cyc_per_inst);
  }
  return 0;
}
