// uc_xform.hpp -- the 2048-point FFT x H x IFFT core shared by compress_kernel (uc_full_kernel.hip) and
// stream_kernel (uc_stream_kernel.hip): chirp compression as
//   arm_rfft_fast_f32 -> arm_cmplx_mult_cmplx_f32(., H) -> inverse arm_rfft_fast_f32
//   (experiments/chirp_compression_time_domain/Src/chirp.c:78-83)
// on ONE complex transform per 2-wave workgroup.  Forward 16 x 16 x 8 Stockham, the spectrum is
// multiplied by H/N in registers and -- because a Stockham first pass consumes exactly the
// stride-256 octets the forward last pass produced -- the inverse 8 x 16 x 16 starts in the same
// registers.  Inverse butterflies reuse the forward ones: IDFT_R[k] = DFT_R[-k mod R], conjugated
// twiddles.  Every exchange is write, ONE barrier (the caller's), read: the passes ping-pong between
// two LDS tiles `src` / `dst` of 2048 complex values each.
//
// Call sequence (a __syncthreads() between consecutive calls):
//   [pass 1 of the caller: registers -> xf_store1(A)] | xf_fwd2(A -> B) | xf_fwd3_h_invA(B -> A)
//   | xf_invB(A -> B) | xf_invC(B -> registers)
#pragma once
#include "uc_dev.hpp"
#include "uc_kernels.hpp"

// wave priority, as in the band kernel: low while a wave issues a burst of LDS stores, raised otherwise (UC_XF_PRIO_OFF: the
// A/B switch of round 6)
#ifdef UC_XF_PRIO_OFF
#define UC_XF_PRIO(n) do { } while (0)
#else
#define UC_XF_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif

namespace uc {

constexpr int kXfThreads = 128;

// LDS addresses (complex units) of thread j
struct XfAddr {
  int s1;          // exchange 1: element 16 j + t lives at 16 j + (t ^ s1)
  int wr1;
  int rd1e, rd1o;  // exchange-1 read of element j + 128 t: even / odd t
  int wr2;         // exchange 2 write: + 16 t
  int rdA;         // inverse exchange A read: + 128 t
  int wrBe, wrBo;  // inverse exchange B write: + 8 t, even / odd t
  int rdBe, rdBo;  // inverse exchange B read: + 128 t, even / odd t
};

__device__ __forceinline__ XfAddr xf_addresses(int j) {
  XfAddr a;
  a.s1 = j & 15;
  a.wr1 = 16 * j;
  a.rd1e = (j & ~15) + ((j & 15) ^ (j >> 4));
  a.rd1o = (j & ~15) + ((j & 15) ^ (j >> 4) ^ 8);
  a.wr2 = (j >> 4) * 256 + (j & 15);
  a.rdA = j ^ ((j >> 4) & 7);
  // element (j>>3)*128 + (j&7) + 8 t, swizzled by flipping bit 3 for odd j>>3, i.e. t -> t ^ ((j>>3)&1)
  a.wrBe = (j >> 3) * 128 + (j & 7) + 8 * ((j >> 3) & 1);
  a.wrBo = (j >> 3) * 128 + (j & 7) - 8 * ((j >> 3) & 1);
  a.rdBe = j;
  a.rdBo = j ^ 8;
  return a;
}

// the two small twiddle tables the loop reads from LDS (vector loads return in order: nothing but
// the input stream may be loaded from memory inside the loop)
constexpr int kXfTw2Floats = 2 * 256;  // W_256^(t k), t < 16, k < 16: forward pass 2
constexpr int kXfTwBFloats = 2 * 128;  // W_128^(t k), t < 16, k < 8: inverse pass B
__device__ __forceinline__ void xf_fill_twiddle_tables(float* tw2t, float* twBt, __amdgpu_buffer_rsrc_t rs_tw, int j) {
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int e = j + kXfThreads * r;  // t = e >> 4, k = e & 15
    lds_st(tw2t, e, buf_ld64(rs_tw, ((8 * (e >> 4) * (e & 15)) & (kN - 1)) * 8, 0));
  }
  lds_st(twBt, j, buf_ld64(rs_tw, ((16 * (j >> 3) * (j & 7)) & (kN - 1)) * 8, 0));  // t = j >> 3, k = j & 7
}

// forward pass-1 outputs (pk_dft16 slot order) -> exchange 1
__device__ __forceinline__ void xf_store1(float* dst, const XfAddr& a, int s1v, const v2f (&v)[16]) {
  UC_XF_PRIO(0);
#pragma unroll
  for (int t = 0; t < 16; t++) lds_st(dst, a.wr1 + (t ^ s1v), v[pk_slot16(t)]);
  UC_XF_PRIO(2);
}

// forward pass 2: radix-16, twiddles W_256^(t k), k = j & 15
__device__ __forceinline__ void xf_fwd2(const float* src, float* dst, const float* tw2t, const XfAddr& a, int j, v2f K,
                                        v2f H) {
  v2f v[16];
#pragma unroll
  for (int t = 0; t < 16; t++) v[t] = lds_ld(src, ((t & 1) ? a.rd1o : a.rd1e) + 128 * t);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 1; t < 16; t++) v[t] = pk_cmul(v[t], lds_ld(tw2t, 16 * t + (j & 15)));
  pk_dft16(v, K, H);
  UC_XF_PRIO(0);
#pragma unroll
  for (int t = 0; t < 16; t++) lds_st(dst, a.wr2 + 16 * t, v[pk_slot16(t)]);
  UC_XF_PRIO(2);
}

// pass-3 twiddles W_2048^(t (j + 128 h)), t = 1..7, from the three resident powers t3a/b/c = W_2048^j, ^2j, ^4j
__device__ __forceinline__ void xf_twiddles3(v2f (&w)[8], int h, v2f t3a, v2f t3b, v2f t3c, v2f K, v2f H) {
  if (h == 0) {
    w[1] = t3a; w[2] = t3b; w[4] = t3c;
  } else {  // W_2048^(t (j+128)) = W_2048^(t j) W_16^t
    w[1] = pk_mul_w1(t3a, K); w[2] = pk_mul_w2(t3b, H); w[4] = pk_mul_mj(t3c);
  }
  w[3] = pk_cmul(w[1], w[2]);
  w[5] = pk_cmul(w[1], w[4]);
  w[6] = pk_cmul(w[2], w[4]);
  w[7] = pk_cmul(w[3], w[4]);
}

// pass-C twiddles W_2048^(t j), t = 1..15, as products of the three resident powers (at most three factors deep)
__device__ __forceinline__ void xf_twiddlesC(v2f (&w)[16], v2f t3a, v2f t3b, v2f t3c) {
  w[1] = t3a; w[2] = t3b; w[4] = t3c; w[8] = pk_cmul(t3c, t3c);
  w[3] = pk_cmul(w[1], w[2]);
  w[5] = pk_cmul(w[1], w[4]);
  w[6] = pk_cmul(w[2], w[4]);
  w[9] = pk_cmul(w[1], w[8]);
  w[10] = pk_cmul(w[2], w[8]);
  w[12] = pk_cmul(w[4], w[8]);
  w[7] = pk_cmul(w[3], w[4]);
  w[11] = pk_cmul(w[3], w[8]);
  w[13] = pk_cmul(w[5], w[8]);
  w[14] = pk_cmul(w[6], w[8]);
  w[15] = pk_cmul(w[7], w[8]);
}

// forward pass 3 (full radix-8) x H/N, inverse pass A (radix-8, no twiddles).
// butterfly b = j (h = 0) and b = j + 128 (h = 1): X[b + 256 t], t = 0..7; hres[h][t] = H[b + 256 t] / N.
// RESIDENT: the twiddles come from w3 (kept in registers by the caller), else they are derived per call.
template <bool RESIDENT>
__device__ __forceinline__ void xf_fwd3_h_invA(const float* src, float* dst, const v2f (&hres)[2][8], const v2f (&w3)[2][8],
                                               v2f t3a, v2f t3b, v2f t3c, int j, v2f K, v2f H) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int b = j + kXfThreads * h;
    v2f u[8];
#pragma unroll
    for (int t = 0; t < 8; t++) u[t] = lds_ld(src, b + 256 * t);
    __builtin_amdgcn_sched_barrier(0);
    v2f w[8];
    if (RESIDENT) {
#pragma unroll
      for (int t = 1; t < 8; t++) w[t] = w3[h][t];
    } else {
      xf_twiddles3(w, h, t3a, t3b, t3c, K, H);
    }
#pragma unroll
    for (int t = 1; t < 8; t++) u[t] = pk_cmul(u[t], w[t]);
    pk_dft8(u, H);
    // spectrum bin k = b + 256 t sits in u[slot8(t)]: multiply by H[k] / N
#pragma unroll
    for (int t = 0; t < 8; t++) u[pk_slot8(t)] = pk_cmul(u[pk_slot8(t)], hres[h][t]);
    // inverse radix-8, Ns = 1 (no twiddles): inputs in natural t order
    v2f g[8];
#pragma unroll
    for (int t = 0; t < 8; t++) g[t] = u[pk_slot8(t)];
    pk_dft8(g, H);
    // IDFT8[t] = DFT8[(8 - t) & 7]; inverse exchange A: element 8 b + t, swizzled phys = o ^ ((o >> 4) & 7)
    UC_XF_PRIO(0);
#pragma unroll
    for (int t = 0; t < 8; t++) lds_st(dst, 8 * b + (t ^ ((b >> 1) & 7)), g[pk_slot8((8 - t) & 7)]);
    UC_XF_PRIO(2);
  }
}

// inverse pass B: radix-16, Ns = 8, conj twiddles W_128^(t k), k = j & 7
__device__ __forceinline__ void xf_invB(const float* src, float* dst, const float* twBt, const XfAddr& a, int j, v2f K,
                                        v2f H) {
  v2f v[16];
#pragma unroll
  for (int t = 0; t < 16; t++) v[t] = lds_ld(src, a.rdA + 128 * t);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 1; t < 16; t++) v[t] = pk_cmulc(v[t], lds_ld(twBt, 8 * t + (j & 7)));
  pk_dft16(v, K, H);
  // output t of the inverse = forward output (16 - t) & 15
  UC_XF_PRIO(0);
#pragma unroll
  for (int t = 0; t < 16; t++) lds_st(dst, ((t & 1) ? a.wrBo : a.wrBe) + 8 * t, v[pk_slot16((16 - t) & 15)]);
  UC_XF_PRIO(2);
}

// inverse pass C: radix-16, Ns = 128, conj twiddles W_2048^(t j); y[t] = output sample j + 128 t
template <bool RESIDENT>
__device__ __forceinline__ void xf_invC(const float* src, v2f (&y)[16], const XfAddr& a, const v2f (&wC)[16], v2f t3a,
                                        v2f t3b, v2f t3c, v2f K, v2f H) {
  v2f v[16];
#pragma unroll
  for (int t = 0; t < 16; t++) v[t] = lds_ld(src, ((t & 1) ? a.rdBo : a.rdBe) + 128 * t);
  __builtin_amdgcn_sched_barrier(0);
  {
    v2f w[16];
    if (RESIDENT) {
#pragma unroll
      for (int t = 1; t < 16; t++) w[t] = wC[t];
    } else {
      xf_twiddlesC(w, t3a, t3b, t3c);
    }
#pragma unroll
    for (int t = 1; t < 16; t++) v[t] = pk_cmulc(v[t], w[t]);
  }
  pk_dft16(v, K, H);
#pragma unroll
  for (int t = 0; t < 16; t++) y[t] = v[pk_slot16((16 - t) & 15)];
}

}  // namespace uc
