// uc_dev.hpp -- device-side helpers shared by the gfx950 kernels: buffer loads, LDS accessors for
// complex pairs, DPP wave reductions, the fixed radix-16 constants.
#pragma once
#include <hip/hip_runtime.h>

#include "uc_pk.hpp"
#include "../../include/uchirp.h"

namespace uc {

constexpr float kSqrtHalfF = 0.70710678118654752440f;
constexpr float kCos8 = 0.92387953251128675613f;   // cos(pi/8)
constexpr float kSin8 = 0.38268343236508977173f;   // sin(pi/8)

typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Buffer loads (SGPR resource + ONE VGPR byte offset + SGPR/immediate offset): the strided loads of
// a thread share a single address register (global_load would keep a 64-bit VGPR pointer per 4 KiB
// of span, which hipcc hoists and then spills), and reads past `bytes` return 0 without a branch.
constexpr int kRsrcFlags = 0x00020000;  // gfx9 raw buffer, 32-bit data format
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, kRsrcFlags);
}
// *_stream: the same loads for data that is read exactly once (the frames): cache-policy hint UC_STREAM_CPOL
// (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef UC_STREAM_CPOL
#define UC_STREAM_CPOL 2
#endif
// raw 32-bit word; int32 words are converted when consumed (the ISR's (float) cast, receiver/Src/main.c:664)
__device__ __forceinline__ float buf_ld32(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_ld32_stream(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, UC_STREAM_CPOL));
}
__device__ __forceinline__ v2f buf_ld64(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const v2u w = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return mkv(__uint_as_float(w.x), __uint_as_float(w.y));
}
__device__ __forceinline__ v4u buf_ld128(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ v4u buf_ld128_stream(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, UC_STREAM_CPOL);
}
template <int DTYPE>
__device__ __forceinline__ v2f cvt_pair(v2f raw) {
  if (DTYPE == UC_DTYPE_I32) return mkv((float)__float_as_int(raw.x), (float)__float_as_int(raw.y));
  return raw;
}

// one complex value (re, im) at complex index cidx of an LDS tile
__device__ __forceinline__ v2f lds_ld(const float* lds, int cidx) {
  return *reinterpret_cast<const v2f*>(lds + 2 * cidx);
}
__device__ __forceinline__ void lds_st(float* lds, int cidx, v2f v) {
  *reinterpret_cast<v2f*>(lds + 2 * cidx) = v;
}

// ---- diagnostic build only (-DUC_CLOCKSTAMP, libuchirp_clock.so; tools/clock_probe.py) -----------------------------------
// ONE s_memtime / s_memrealtime stamp pair around a kernel's whole persistent loop: shader clock under this kernel =
// d(s_memtime) / d(s_memrealtime) x 100 MHz.  Four words per wave go to `dbg` (never read by a kernel, never part of
// an output): cycles (low 40 bits; bits 40..51 name the CU), 100 MHz ticks, absolute start and end ticks (start / end
// skew across the grid and, grouped by CU, when each CU ran out of work).
#ifdef UC_CLOCKSTAMP
#define UC_CLOCK_BEGIN()                                              \
  const unsigned long long clk0_ = __builtin_readcyclecounter();     \
  const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime()
#define UC_CLOCK_END(dbg, waves_per_wg)                                                                          \
  do {                                                                                                           \
    if ((threadIdx.x & 63) == 0 && (dbg)) {                                                                      \
      const unsigned long long rt1_ = __builtin_amdgcn_s_memrealtime();                                          \
      unsigned long long* d_ = (dbg) + ((size_t)blockIdx.x * (waves_per_wg) + (threadIdx.x >> 6)) * 4;            \
      /* bits 40..51: which CU ran the wave -- XCC_ID[3:0] (hwreg 20) and HW_ID's cu / sh / se fields [15:8] */   \
      const unsigned long long cu_ = ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u) << 8) | \
                                     ((__builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11))) & 255u);              \
      d_[0] = ((__builtin_readcyclecounter() - clk0_) & 0xffffffffffull) | (cu_ << 40);                          \
      d_[1] = rt1_ - rt0_;                                                                                       \
      d_[2] = rt0_;                                                                                              \
      d_[3] = rt1_;                                                                                              \
    }                                                                                                            \
  } while (0)
#else
#define UC_CLOCK_BEGIN() do { } while (0)
#define UC_CLOCK_END(dbg, waves_per_wg) do { } while (0)
#endif

// ---- dynamic hand-out counters (uc_api_core.cpp: take_work_counter) -----------------------------------
// ctr[0] = the next ticket, ctr[1] = workgroups that have left.  Every workgroup of a dynamically dealt launch calls
// this ONCE, from one thread, on its way out; the last one to leave puts both words back to zero.  A counter slot is
// therefore zero whenever no launch is using it: no memset in front of a launch, and none recorded into a captured
// graph (a memset NODE in front of a replayed kernel node was measured not to be seen by the kernel's device-scope
// atomics on short launches: profiles/r03_graph_probe.txt).
__device__ __forceinline__ void handout_leave(unsigned int* ctr) {
  // a ticket request this thread issued and never read (the ragged end of a batch) must have been performed before
  // the count below can reach its final value: wait for every outstanding memory operation of the thread
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  const unsigned left = atomicAdd(ctr + 1, 1u);
  if (left == gridDim.x - 1u) {
    atomicExch(ctr + 1, 0u);
    atomicExch(ctr, 0u);
  }
}

// ---- wave-wide reductions without LDS ----------------------------------------
// Six DPP steps (row_ror 1/2/4/8 make every lane of a 16-lane row hold the row
// result, row_bcast:15 / :31 fold the rows into lane 63) and one v_readlane.
// hipcc does not see inside asm, so each step carries the 2 wait states a DPP
// read of a just-written VGPR needs (s_nop 1).
#define UC_DPP_REDUCE(OP, v)                                                                       \
  do {                                                                                             \
    asm("s_nop 1\n\t" OP " %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(v));            \
    asm("s_nop 1\n\t" OP " %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf" : "+v"(v));            \
    asm("s_nop 1\n\t" OP " %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(v));            \
    asm("s_nop 1\n\t" OP " %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v));            \
    asm("s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));         \
    asm("s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));         \
  } while (0)

__device__ __forceinline__ float wave_max_f32(float v) {
  UC_DPP_REDUCE("v_max_f32_dpp", v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_u32(int v) {
  UC_DPP_REDUCE("v_min_u32_dpp", v);
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i32(int v) {
  UC_DPP_REDUCE("v_max_i32_dpp", v);
  return __builtin_amdgcn_readlane(v, 63);
}
// IEEE maxNum without the canonicalising moves fmaxf() drags in
__device__ __forceinline__ float max_f32(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

}  // namespace uc
