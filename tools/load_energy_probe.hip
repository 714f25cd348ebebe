// load_energy_probe.hip -- what does a byte cost on its way from HBM into a register, by load instruction?
// The headline kernel runs at the socket power cap and its memory path is 53 % of the energy of a frame
// (profiles/r03_power_knock.txt), so joules per byte matter as much as bytes per second.  Persistent workgroups stream a
// buffer (default 8 GiB) for a fixed number of passes with one load flavour; the caller samples socket power meanwhile
// (tools/load_energy.sh).  Flavours: 0 dword nt (the band kernel's frame loads: 256 B per wave instruction)
//                                    1 dword default policy      2 dwordx4 nt (1 KiB per wave instruction)
//                                    3 dwordx4 default policy    4 dwordx2 nt
//                                    5 dword nt through LDS-DMA (global_load_lds_dword) + ds_read_b32
// Every flavour keeps 16 (dword), 8 (x2) or 4 (x4) loads = 64 B per lane in flight per wave and folds the data into one
// XOR (so that nothing is dead code and no arithmetic energy is spent).
// build: hipcc --offload-arch=gfx950 -O3 -o load_energy_probe load_energy_probe.hip
// run:   ./load_energy_probe <flavour> [GiB=8] [passes=400]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <int F>
__global__ __launch_bounds__(128, 3) void stream_read(const char* buf, size_t bytes, unsigned* sink) {
  __shared__ unsigned stage[2 * 2048];
  const int j = threadIdx.x;
  const size_t chunk = 8192;  // one "frame" per workgroup and step: 128 threads x 64 B
  const size_t nchunks = bytes / chunk;
  unsigned acc = 0;
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const __amdgpu_buffer_rsrc_t r = rsrc(buf + c * chunk, (unsigned)chunk);
    if (F == 0 || F == 1) {
      unsigned x[16];
#pragma unroll
      for (int t = 0; t < 16; t++) x[t] = __builtin_amdgcn_raw_buffer_load_b32(r, j * 4, 512 * t, F == 0 ? 2 : 0);
#pragma unroll
      for (int t = 0; t < 16; t++) acc ^= x[t];
    } else if (F == 2 || F == 3) {
      v4u x[4];
#pragma unroll
      for (int t = 0; t < 4; t++) x[t] = __builtin_amdgcn_raw_buffer_load_b128(r, j * 16, 2048 * t, F == 2 ? 2 : 0);
#pragma unroll
      for (int t = 0; t < 4; t++) acc ^= x[t].x ^ x[t].y ^ x[t].z ^ x[t].w;
    } else if (F == 4) {
      v2u x[8];
#pragma unroll
      for (int t = 0; t < 8; t++) x[t] = __builtin_amdgcn_raw_buffer_load_b64(r, j * 8, 1024 * t, 2);
#pragma unroll
      for (int t = 0; t < 8; t++) acc ^= x[t].x ^ x[t].y;
    } else {
      // LDS-DMA: each wave's 16 dword loads land in its half of a 16 KiB staging area (M0 = LDS base of the instruction,
      // lane l at + 4 l), double buffered by step parity; then 16 ds_read_b32
      unsigned* st = stage + ((c / gridDim.x) & 1) * 2048 + (j >> 6) * 1024;
      const char* g = buf + c * chunk + (j >> 6) * 256 + (j & 63) * 4;   // lane's source; the LDS side is wave-uniform + 4 lane
#pragma unroll
      for (int t = 0; t < 16; t++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 512 * t),
                                         (__attribute__((address_space(3))) void*)(st + 64 * t), 4, 0, 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 16; t++) acc ^= st[64 * t + (j & 63)];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int F>
static void run(const char* d, size_t bytes, unsigned* sink, int passes) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL((stream_read<F>), dim3(1536), dim3(128), 0, 0, d, bytes, sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < passes; i++) hipLaunchKernelGGL((stream_read<F>), dim3(1536), dim3(128), 0, 0, d, bytes, sink);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  printf("flavour %d: %d passes over %.1f GiB in %.1f ms = %.3f TB/s (%s)\n", F, passes, bytes / 1073741824.0, ms,
         (double)bytes * passes / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  const int f = argc > 1 ? atoi(argv[1]) : 0;
  const size_t bytes = (size_t)(argc > 2 ? atof(argv[2]) : 8.0) * 1073741824ull;
  const int passes = argc > 3 ? atoi(argv[3]) : 400;
  char* d;
  unsigned* sink;
  if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(d, 0x5a, bytes);
  // random-looking contents (a memset pattern would not toggle the data paths)
  {
    unsigned* h = (unsigned*)malloc(1 << 26);
    unsigned s = 12345;
    for (size_t i = 0; i < (1u << 24); i++) { s = s * 1664525u + 1013904223u; h[i] = s; }
    for (size_t o = 0; o < bytes; o += (1u << 26)) (void)hipMemcpy(d + o, h, (1u << 26), hipMemcpyHostToDevice);
    free(h);
  }
  switch (f) {
    case 0: run<0>(d, bytes, sink, passes); break;
    case 1: run<1>(d, bytes, sink, passes); break;
    case 2: run<2>(d, bytes, sink, passes); break;
    case 3: run<3>(d, bytes, sink, passes); break;
    case 4: run<4>(d, bytes, sink, passes); break;
    default: run<5>(d, bytes, sink, passes); break;
  }
  return 0;
}
