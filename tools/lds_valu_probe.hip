// lds_valu_probe.hip -- what does an LDS instruction cost a SIMD that is also busy with packed-f32 VALU work?
// One workgroup per CU, W waves per SIMD; every wave runs `iters` times a body of A v_pk_fma_f32 and B LDS
// instructions of one kind (conflict-free, lane-linear addresses, its own 4 KiB of LDS), s_waitcnt lgkmcnt(0) once per
// body.  Printed: core clocks (s_memtime) per body and SIMD = wave clocks / W, for the VALU part alone, the LDS part
// alone and both together.  If both = max(alone, alone) the LDS traffic is free beside the arithmetic; if both = sum,
// every LDS instruction takes issue time from the VALU.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_valu_probe lds_valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));

enum { K_NONE, K_RD64, K_WR64, K_WR128, K_RD128, K_RD2ST64_B32, K_WR_ADDTID, K_RD_ADDTID, K_WR32, K_RD32, K_RD2_B64, K_KINDS };
static const char* kNames[K_KINDS] = {"none", "ds_read_b64", "ds_write_b64", "ds_write_b128", "ds_read_b128",
                                      "ds_read2st64_b32", "ds_write_addtid_b32", "ds_read_addtid_b32", "ds_write_b32",
                                      "ds_read_b32", "ds_read2_b64"};

template <int KIND, int A, int B>
__global__ __launch_bounds__(1024) void probe(int iters, unsigned long long* out, float* sink) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane(wave * 4096);  // bytes
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f;
  v2 x[8], w = {b, a};
#pragma unroll
  for (int u = 0; u < 8; u++) x[u] = (v2){a + u, b - u};
  v2 r2[8];
  v4 r4[8];
  float r1[8];
#pragma unroll
  for (int u = 0; u < 8; u++) { r2[u] = x[u]; r4[u] = (v4){a, b, a, b}; r1[u] = a; }
  for (int e = threadIdx.x; e < (int)(blockDim.x >> 6) * 1024; e += blockDim.x) lds[e] = (float)e;
  __syncthreads();
  const unsigned a8 = base + lane * 8, a16 = base + lane * 16, a4 = base + lane * 4;
  constexpr int per = B > 0 ? A / B : A;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    if (KIND == K_WR_ADDTID || KIND == K_RD_ADDTID) asm volatile("s_mov_b32 m0, %0" ::"s"(base) : "memory");
#pragma unroll
    for (int u = 0; u < (B > 0 ? B : 1); u++) {
#pragma unroll
      for (int k = 0; k < per; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x[(u * per + k) & 7]) : "v"(w));
      if (B == 0) break;
      switch (KIND) {
        case K_RD64: asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2[u & 7]) : "v"(a8), "n"((u & 7) * 512) : "memory"); break;
        case K_WR64: asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(r2[u & 7]), "n"((u & 7) * 512) : "memory"); break;
        case K_WR128: asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a16), "v"(r4[u & 7]), "n"((u & 3) * 1024) : "memory"); break;
        case K_RD128: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4[u & 7]) : "v"(a16), "n"((u & 3) * 1024) : "memory"); break;
        case K_RD2ST64_B32: asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r2[u & 7]) : "v"(a4), "n"(u & 7), "n"((u & 7) + 8) : "memory"); break;
        case K_WR_ADDTID: asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(r1[u & 7]), "n"((u & 7) * 256) : "memory"); break;
        case K_RD_ADDTID: asm volatile("ds_read_addtid_b32 %0 offset:%1" : "=v"(r1[u & 7]) : "n"((u & 7) * 256) : "memory"); break;
        case K_WR32: asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a4), "v"(r1[u & 7]), "n"((u & 7) * 256) : "memory"); break;
        case K_RD32: asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r1[u & 7]) : "v"(a4), "n"((u & 7) * 256) : "memory"); break;
        case K_RD2_B64: asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r4[u & 7]) : "v"(a8), "n"((u & 3) * 2), "n"((u & 3) * 2 + 64) : "memory"); break;
        default: break;
      }
    }
    if (B > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int u = 0; u < 8; u++) s += x[u].x + x[u].y + r2[u].x + r2[u].y + r4[u].x + r4[u].y + r4[u].z + r4[u].w + r1[u];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if (lane == 0) {
    out[2 * (blockIdx.x * (blockDim.x >> 6) + wave)] = c1 - c0;
    out[2 * (blockIdx.x * (blockDim.x >> 6) + wave) + 1] = t1 - t0;
  }
}


// A frame-shaped body (what one wave of the band kernel does per frame: ~304 packed VALU instructions in four blocks,
// exchange-1 stores, 16 exchange-1 reads, exchange-2 stores, 24 pruned-pass reads), with the exchange written
//   FORM 0: as the kernel has it -- 8 ds_write_b128, 16 ds_read_b64, 16 ds_write_b64, 24 ds_read_b64
//   FORM 1: split re / im planes -- 32 ds_write_addtid_b32, 16 ds_read2_b32, 32 ds_write_addtid_b32, 24 ds_read2_b32
//   FORM 2: as 0 with exchange 1 as 16 ds_write_b64
//   FORM 3: no LDS at all
template <int FORM>
__global__ __launch_bounds__(1024) void frame_probe(int iters, unsigned long long* out, float* sink) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane(wave * 4096);  // bytes
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f;
  v2 x[16], w = {b, a};
#pragma unroll
  for (int u = 0; u < 16; u++) x[u] = (v2){a + u, b - u};
  for (int e = threadIdx.x; e < (int)(blockDim.x >> 6) * 1024; e += blockDim.x) lds[e] = (float)e;
  __syncthreads();
  const unsigned a8 = base + lane * 8, a16 = base + lane * 16, a4 = base + lane * 4;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#define VALU_BLOCK(N) _Pragma("unroll") for (int k = 0; k < (N); k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x[k & 15]) : "v"(w))
#define WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  for (int i = 0; i < iters; i++) {
    if (FORM == 1) asm volatile("s_mov_b32 m0, %0" ::"s"(base) : "memory");
    VALU_BLOCK(76);
    // exchange 1: stores
    if (FORM == 0) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        v4 q = {x[2 * u].x, x[2 * u].y, x[2 * u + 1].x, x[2 * u + 1].y};
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a16), "v"(q), "n"((u & 3) * 1024) : "memory");
      }
    } else if (FORM == 1) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(x[u].x), "n"((u & 7) * 512) : "memory");
        asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(x[u].y), "n"((u & 7) * 512 + 256) : "memory");
      }
    } else if (FORM == 2) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(x[u]), "n"((u & 7) * 512) : "memory");
    }
    if (FORM != 3) WAIT();
    // exchange 1: reads
    if (FORM == 0 || FORM == 2) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[u]) : "v"(a8), "n"((u & 7) * 512) : "memory");
    } else if (FORM == 1) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(x[u]) : "v"(a4), "n"((u & 7) * 8), "n"((u & 7) * 8 + 128) : "memory");
    }
    if (FORM != 3) WAIT();
    VALU_BLOCK(76);
    VALU_BLOCK(76);
    if (FORM == 0 || FORM == 2) {
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(x[u]), "n"((u & 7) * 512) : "memory");
    } else if (FORM == 1) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(x[u].x), "n"((u & 7) * 512) : "memory");
        asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(x[u].y), "n"((u & 7) * 512 + 256) : "memory");
      }
    }
    if (FORM != 3) WAIT();
    if (FORM == 0 || FORM == 2) {
#pragma unroll
      for (int u = 0; u < 24; u++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[u & 15]) : "v"(a8), "n"((u & 7) * 512) : "memory");
    } else if (FORM == 1) {
#pragma unroll
      for (int u = 0; u < 24; u++) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(x[u & 15]) : "v"(a4), "n"((u & 7) * 16), "n"((u & 7) * 16 + 128) : "memory");
    }
    if (FORM != 3) WAIT();
    VALU_BLOCK(76);
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int u = 0; u < 16; u++) s += x[u].x + x[u].y;
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if (lane == 0) {
    out[2 * (blockIdx.x * (blockDim.x >> 6) + wave)] = c1 - c0;
    out[2 * (blockIdx.x * (blockDim.x >> 6) + wave) + 1] = t1 - t0;
  }
}

static unsigned long long* d_out;
static float* d_sink;

template <int KIND, int A, int B>
static double run(int wps, int iters, double* ghz) {
  const int waves = 4 * wps, nwg = 256;
  for (int rep = 0; rep < 3; rep++)
    hipLaunchKernelGGL((probe<KIND, A, B>), dim3(nwg), dim3(64 * waves), waves * 4096, 0, iters, d_out, d_sink);
  hipDeviceSynchronize();
  static unsigned long long h[2 * 256 * 16];
  hipMemcpy(h, d_out, sizeof(unsigned long long) * 2 * nwg * waves, hipMemcpyDeviceToHost);
  // all waves of a workgroup start together behind the barrier: the workgroup's time is its SLOWEST wave's
  // (the arbiter favours the oldest wave, so the mean would flatter the SIMD)
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
  *ghz = c / t * 0.1;
  return cmax / nwg / iters / wps;  // clocks per body and SIMD
}

static double reduce(int nwg, int waves, int iters, int wps, double* ghz) {
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
  *ghz = c / t * 0.1;
  return cmax / nwg / iters / wps;
}

template <int FORM>
static void frame_row(int wps, int iters, const char* what) {
  const int waves = 4 * wps, nwg = 256;
  for (int rep = 0; rep < 3; rep++)
    hipLaunchKernelGGL((frame_probe<FORM>), dim3(nwg), dim3(64 * waves), waves * 4096, 0, iters, d_out, d_sink);
  hipDeviceSynchronize();
  double ghz;
  const double c = reduce(nwg, waves, iters, wps, &ghz);
  printf("%d waves/SIMD  frame body, %-58s %7.1f clocks per wave-frame and SIMD = %6.1f per frame and CU  [%.2f GHz]\n", wps, what, c,
         c * 2 / 4, ghz);
}

template <int KIND>
static void row(int wps, int iters) {
  double g0, g1, g2;
  const double l = run<KIND, 0, 8>(wps, iters, &g0);
  const double v = run<K_NONE, 32, 0>(wps, iters, &g1);
  const double b = run<KIND, 32, 8>(wps, iters, &g2);
  printf("%d waves/SIMD  %-20s 8 LDS alone %6.1f (%.1f per instr.)   32 pk_fma alone %6.1f   both %6.1f   -> +%.1f per LDS instr. beside the VALU  [%.2f GHz]\n",
         wps, kNames[KIND], l, l / 8, v, b, (b - v) / 8, g2);
}

int main(int argc, char** argv) {
  hipMalloc(&d_out, sizeof(unsigned long long) * 2 * 256 * 16);
  hipMalloc(&d_sink, 4096 * 4);
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  for (int wps = 2; wps <= 4; wps++) {
    frame_row<3>(wps, iters / 10, "no LDS (304 v_pk_fma_f32)");
    frame_row<0>(wps, iters / 10, "b128 / b64 stores, b64 reads (the kernel's exchange)");
    frame_row<2>(wps, iters / 10, "b64 stores only, b64 reads");
    frame_row<1>(wps, iters / 10, "re / im planes: ds_write_addtid_b32, ds_read2_b32");
  }
  for (int wps = 1; wps <= 4; wps++) {
    if (wps == 2) continue;
    row<K_RD64>(wps, iters);
    row<K_WR64>(wps, iters);
    row<K_WR128>(wps, iters);
    row<K_RD128>(wps, iters);
    row<K_RD2ST64_B32>(wps, iters);
    row<K_RD2_B64>(wps, iters);
    row<K_WR_ADDTID>(wps, iters);
    row<K_RD_ADDTID>(wps, iters);
    row<K_WR32>(wps, iters);
    row<K_RD32>(wps, iters);
  }
  return 0;
}
