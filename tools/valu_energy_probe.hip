// valu_energy_probe.hip -- joules per wave-instruction, by instruction: every kernel of the library runs at the socket power
// cap (profiles/r03_power_knock.txt), so the energy of an instruction is what it costs.  All SIMDs of the chip issue ONE
// instruction type back to back (32 independent register pairs per wave, random operands, 3 waves per SIMD as the band
// kernel) for a fixed number of iterations; the caller samples socket power meanwhile (tools/valu_energy.sh).
//   0 v_pk_fma_f32   1 v_pk_add_f32   2 v_pk_mul_f32   3 v_fma_f32   4 v_add_f32   5 v_mov_b32 (VGPR to VGPR)
//   6 ds_write_b64 + ds_read_b64 (one pair per "instruction", conflict-free lane-linear addresses)   7 s_nop (issue only)
//   8 v_pk_fma_f32 with two of its three sources in SGPRs
// build: hipcc --offload-arch=gfx950 -O3 -o valu_energy_probe valu_energy_probe.hip ; run: ./valu_energy_probe <type> [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2 __attribute__((ext_vector_type(2)));

template <int TY>
__global__ __launch_bounds__(192, 1) void spin(int iters, const float* seed, float* sink) {
  __shared__ float lds[192 * 2 * 4];
  v2 x[32];
  const v2 c = {seed[threadIdx.x & 63], seed[64 + (threadIdx.x & 63)]};
#pragma unroll
  for (int i = 0; i < 32; i++) x[i] = (v2){seed[(threadIdx.x * 7 + i * 13) & 1023], seed[(threadIdx.x * 5 + i * 11 + 1) & 1023]};
  const v2 k = {0.999f, -1.001f};   // keeps the values bounded and toggling
  const unsigned a = threadIdx.x * 8;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 32; i++) {
        if (TY == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(k), "v"(x[(i + 1) & 31]));
        if (TY == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 7) & 31]));
        if (TY == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(k));
        if (TY == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(k.x), "v"(x[(i + 1) & 31].y));
        if (TY == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i].x) : "v"(x[(i + 7) & 31].y));
        if (TY == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i].x) : "v"(x[(i + 7) & 31].y));
        if (TY == 6) asm volatile("ds_write_b64 %1, %0\n\tds_read_b64 %0, %1 offset:1536" : "+v"(x[i]) : "v"(a) : "memory");
        if (TY == 7) asm volatile("s_nop 0");
        if (TY == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "s"(k));
      }
      if (TY == 1 || TY == 4) {  // additions alone drift: pull the values back now and then (1 in 33 instructions)
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[r]) : "v"((v2){0.03125f, 0.03125f}));
      }
    }
    if (TY == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) s += x[i].x + x[i].y;
  if (s == 1234.5f) sink[threadIdx.x] = s + c.x + lds[threadIdx.x];
}

template <int TY>
static void run(int iters, const float* seed, float* sink) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  // 4 workgroups of 3 waves per CU = 3 waves per SIMD
  hipLaunchKernelGGL((spin<TY>), dim3(1024), dim3(192), 0, 0, iters / 50, seed, sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a, 0);
  hipLaunchKernelGGL((spin<TY>), dim3(1024), dim3(192), 0, 0, iters, seed, sink);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  const double insts = (double)iters * 256.0 * 1024 * 3;   // wave-instructions issued by the whole chip
  printf("type %d: %.3e wave-instructions in %.1f ms = %.4g per s (%s)\n", TY, insts, ms, insts / (ms * 1e-3), hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  const int ty = argc > 1 ? atoi(argv[1]) : 0;
  const int iters = argc > 2 ? atoi(argv[2]) : 2000000;
  float h[1024];
  unsigned s = 99;
  for (int i = 0; i < 1024; i++) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 13)); }
  float *seed, *sink;
  (void)hipMalloc(&seed, sizeof(h));
  (void)hipMalloc(&sink, 4096);
  (void)hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  switch (ty) {
    case 0: run<0>(iters, seed, sink); break;
    case 1: run<1>(iters, seed, sink); break;
    case 2: run<2>(iters, seed, sink); break;
    case 3: run<3>(iters, seed, sink); break;
    case 4: run<4>(iters, seed, sink); break;
    case 5: run<5>(iters, seed, sink); break;
    case 6: run<6>(iters / 4, seed, sink); break;
    case 7: run<7>(iters, seed, sink); break;
    default: run<8>(iters, seed, sink); break;
  }
  return 0;
}
