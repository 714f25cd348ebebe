// uc_pk.hpp -- packed-fp32 complex arithmetic for gfx950 (v_pk_*_f32, VOP3P).
//
// A complex number is one 64-bit VGPR pair (re = low dword, im = high dword).
// hipcc (ROCm 7.2) does not fold the "swap halves + negate one" operand of a
// complex product or of a multiplication by -j into the op_sel / neg modifiers
// of v_pk_mul/fma/add_f32: it emits v_xor + v_mov + v_pk_* (3-4 instructions).
// The forms below are the 1-2 instruction encodings, written out by hand:
//   D.lo = S0[op_sel[0]]    * S1[op_sel[1]]    (+ S2[op_sel[2]])
//   D.hi = S0[op_sel_hi[0]] * S1[op_sel_hi[1]] (+ S2[op_sel_hi[2]])
// with neg_lo / neg_hi negating the operands of the low / high result.
// Packed fp32 VALU has no software-visible hazards (dependencies interlock),
// so no wait states are needed inside the asm statements.
#pragma once
#include <hip/hip_runtime.h>

namespace uc {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f mkv(float re, float im) { v2f r; r.x = re; r.y = im; return r; }

// a * w
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f w) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"
      : "=&v"(r) : "v"(a), "v"(w));
  return r;
}

// a * w with w a wave-uniform constant (SGPR pair)
__device__ __forceinline__ v2f pk_cmul_s(v2f a, v2f w) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"
      : "=&v"(r) : "v"(a), "s"(w));
  return r;
}

// a * conj(w)
__device__ __forceinline__ v2f pk_cmulc(v2f a, v2f w) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
      : "=&v"(r) : "v"(a), "v"(w));
  return r;
}

// acc + a * w
__device__ __forceinline__ v2f pk_cfma(v2f a, v2f w, v2f acc) {
  // in place on the accumulator: acc is both input and output
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"
      : "+v"(acc) : "v"(a), "v"(w));
  return acc;
}

// acc + a * conj(w)
__device__ __forceinline__ v2f pk_cfmac(v2f a, v2f w, v2f acc) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
      : "+v"(acc) : "v"(a), "v"(w));
  return acc;
}

// x + (-j) d  = (x.re + d.im, x.im - d.re)
__device__ __forceinline__ v2f pk_add_mj(v2f x, v2f d) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(d));
  return r;
}

// x - (-j) d  = x + j d = (x.re - d.im, x.im + d.re)
__device__ __forceinline__ v2f pk_sub_mj(v2f x, v2f d) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(x), "v"(d));
  return r;
}

// a + conj(b), a - conj(b)
__device__ __forceinline__ v2f pk_add_conj(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2f pk_sub_conj(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// (-j) a = (a.im, -a.re)
__device__ __forceinline__ v2f pk_mul_mj(v2f a) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a));
  return r;
}

// w * x.lo  /  w * x.hi  (real sample broadcast from one half of a register pair)
__device__ __forceinline__ v2f pk_scale_lo(v2f w, v2f x) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(w), "v"(x));
  return r;
}
__device__ __forceinline__ v2f pk_scale_hi(v2f w, v2f x) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(w), "v"(x));
  return r;
}

// w * b with the REAL factor b broadcast from the low half of an SGPR pair (two filter taps per pair, no (b, b) copies)
__device__ __forceinline__ v2f pk_mul_slo(v2f w, v2f bb) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(w), "s"(bb));
  return r;
}

// Two consecutive filter taps (k odd, k + 1) on eight independent accumulators as ONE asm block
// (hipcc puts an s_nop between dependent asm statements, so the blocks are made long):
//   acc[u] += w[u + 1] * hi(b0) + w[u] * lo(b1)      -- sample index falls as the tap index rises
__device__ __forceinline__ void pk_tap8x2(v2f (&acc)[8], const v2f* w, v2f b0, v2f b1) {
  asm("v_pk_fma_f32 %0, %9, %17, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %1, %10, %17, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %2, %11, %17, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %3, %12, %17, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %4, %13, %17, %4 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %5, %14, %17, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %6, %15, %17, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %7, %16, %17, %7 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %0, %8, %18, %0 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %1, %9, %18, %1 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %2, %10, %18, %2 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %3, %11, %18, %3 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %4, %12, %18, %4 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %5, %13, %18, %5 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %6, %14, %18, %6 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %7, %15, %18, %7 op_sel_hi:[1,0,1]"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
      : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "s"(b0), "s"(b1));
}

// ---- multiplications by the fixed radix-16 twiddles -------------------------
// K = (cos(pi/8), sin(pi/8)), H = (sqrt(1/2), sqrt(1/2)) live in two SGPR pairs ("s" operands:
// wave-uniform constants cost no VGPRs).

// a * W16^1 = a * (c, -s) = (a.re c + a.im s, a.im c - a.re s)
__device__ __forceinline__ v2f pk_mul_w1(v2f a, v2f K) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]"
      : "=&v"(r) : "v"(a), "s"(K));
  return r;
}
// a * W16^3 = a * (s, -c) = (a.re s + a.im c, a.im s - a.re c)
__device__ __forceinline__ v2f pk_mul_w3(v2f a, v2f K) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
      : "=&v"(r) : "v"(a), "s"(K));
  return r;
}
// a * W16^9 = -(a * W16^1)
__device__ __forceinline__ v2f pk_mul_w9(v2f a, v2f K) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
      : "=&v"(r) : "v"(a), "s"(K));
  return r;
}
// a * W16^2 = a * (h, -h) = h (a.re + a.im, a.im - a.re)
__device__ __forceinline__ v2f pk_mul_w2(v2f a, v2f H) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
      "v_pk_mul_f32 %0, %0, %2"
      : "=&v"(r) : "v"(a), "s"(H));
  return r;
}
// a * W16^6 = a * (-h, -h) = h (a.im - a.re, -(a.re + a.im))
__device__ __forceinline__ v2f pk_mul_w6(v2f a, v2f H) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]\n\t"
      "v_pk_mul_f32 %0, %0, %2"
      : "=&v"(r) : "v"(a), "s"(H));
  return r;
}

// forward 4-point DFT in place, natural order (8 packed instructions)
// One asm block: hipcc pads every dependent pair of separate asm statements with an
// s_nop, which the hardware does not need between packed-fp32 VALU instructions.
__device__ __forceinline__ void pk_dft4(v2f& x0, v2f& x1, v2f& x2, v2f& x3) {
  v2f a1, d;
  asm("v_pk_add_f32 %4, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"   // a1 = x0 - x2
      "v_pk_add_f32 %0, %0, %2\n\t"                              // a0 = x0 + x2   (in x0)
      "v_pk_add_f32 %5, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"   // d  = x1 - x3
      "v_pk_add_f32 %1, %1, %3\n\t"                              // a2 = x1 + x3   (in x1)
      "v_pk_add_f32 %2, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"   // X2 = a0 - a2
      "v_pk_add_f32 %0, %0, %1\n\t"                              // X0 = a0 + a2
      "v_pk_add_f32 %1, %4, %5 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"  // X1 = a1 + (-j) d
      "v_pk_add_f32 %3, %4, %5 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"        // X3 = a1 - (-j) d
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(a1), "=&v"(d));
}

// forward 4-point DFT of w_i * s_i, i = 0..3, where the REAL samples s_i are the low (HI = 0) or high
// (HI = 1) halves of the register pairs x_i: the table products are folded into the first butterfly
// additions as fused multiply-adds (2 multiplies + 4 FMAs + 4 additions instead of 4 + 8).
template <int HI>
__device__ __forceinline__ void pk_dft4_scaled(v2f& X0, v2f& X1, v2f& X2, v2f& X3, v2f x0, v2f x1, v2f x2, v2f x3, v2f w0,
                                               v2f w1, v2f w2, v2f w3) {
  v2f p0, p1, a1, d;
  if (HI) {
    asm("v_pk_mul_f32 %4, %12, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"                              // p0 = w0 s0
        "v_pk_mul_f32 %5, %13, %9 op_sel:[0,1] op_sel_hi:[1,1]\n\t"                              // p1 = w1 s1
        "v_pk_fma_f32 %6, %14, %10, %4 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"  // a1 = p0 - w2 s2
        "v_pk_fma_f32 %0, %14, %10, %4 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                     // a0 = p0 + w2 s2
        "v_pk_fma_f32 %7, %15, %11, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"  // d  = p1 - w3 s3
        "v_pk_fma_f32 %1, %15, %11, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                     // a2 = p1 + w3 s3
        "v_pk_add_f32 %2, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"                                   // X2 = a0 - a2
        "v_pk_add_f32 %0, %0, %1\n\t"                                                            // X0 = a0 + a2
        "v_pk_add_f32 %1, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"                   // X1 = a1 + (-j) d
        "v_pk_add_f32 %3, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"                        // X3 = a1 - (-j) d
        : "=&v"(X0), "=&v"(X1), "=&v"(X2), "=&v"(X3), "=&v"(p0), "=&v"(p1), "=&v"(a1), "=&v"(d)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
  } else {
    asm("v_pk_mul_f32 %4, %12, %8 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %5, %13, %9 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %6, %14, %10, %4 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %0, %14, %10, %4 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %7, %15, %11, %5 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %15, %11, %5 op_sel_hi:[1,0,1]\n\t"
        "v_pk_add_f32 %2, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_add_f32 %1, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %3, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
        : "=&v"(X0), "=&v"(X1), "=&v"(X2), "=&v"(X3), "=&v"(p0), "=&v"(p1), "=&v"(a1), "=&v"(d)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
  }
}

// second half of the 16-point DFT: the fixed twiddles and the four output butterflies
__device__ __forceinline__ void pk_dft16_finish(v2f (&v)[16], v2f K, v2f H);

// forward 16-point DFT as 4 x 4; X[t] ends up in v[4*(t&3) + (t>>2)]
__device__ __forceinline__ void pk_dft16(v2f (&v)[16], v2f K, v2f H) {
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) pk_dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
  pk_dft16_finish(v, K, H);
}

__device__ __forceinline__ void pk_dft16_finish(v2f (&v)[16], v2f K, v2f H) {
  v[5] = pk_mul_w1(v[5], K);
  v[6] = pk_mul_w2(v[6], H);
  v[7] = pk_mul_w3(v[7], K);
  v[9] = pk_mul_w2(v[9], H);
  v[10] = pk_mul_mj(v[10]);
  v[11] = pk_mul_w6(v[11], H);
  v[13] = pk_mul_w3(v[13], K);
  v[14] = pk_mul_w6(v[14], H);
  v[15] = pk_mul_w9(v[15], K);
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) pk_dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}

}  // namespace uc

namespace uc {

// forward 8-point DFT as 2 x 4; X[t] ends up in v[4*(t&1) + (t>>1)]
__device__ __forceinline__ void pk_dft8(v2f (&v)[8], v2f H) {
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) {
    const v2f a = v[n2], b = v[4 + n2];
    v[n2] = a + b;
    v[4 + n2] = a - b;
  }
  v[5] = pk_mul_w2(v[5], H);  // W8^1
  v[6] = pk_mul_mj(v[6]);     // W8^2 = -j
  v[7] = pk_mul_w6(v[7], H);  // W8^3
  pk_dft4(v[0], v[1], v[2], v[3]);
  pk_dft4(v[4], v[5], v[6], v[7]);
}
// The same 8-point DFT evaluated for the outputs X[0], X[1], X[6], X[7] only (the slots of the others hold garbage):
// a pass whose successor looks at the bins next to DC only needs the two lowest frequencies either side
// (25 instead of 29 packed instructions, and half the stores).
__device__ __forceinline__ void pk_dft4_03(v2f& x0, v2f& x1, v2f& x2, v2f& x3) {
  v2f a1, d;
  asm("v_pk_add_f32 %4, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"   // a1 = x0 - x2
      "v_pk_add_f32 %0, %0, %2\n\t"                              // a0 = x0 + x2   (in x0)
      "v_pk_add_f32 %5, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"   // d  = x1 - x3
      "v_pk_add_f32 %1, %1, %3\n\t"                              // a2 = x1 + x3   (in x1)
      "v_pk_add_f32 %0, %0, %1\n\t"                              // X0 = a0 + a2
      "v_pk_add_f32 %3, %4, %5 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"        // X3 = a1 - (-j) d
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(a1), "=&v"(d));
}
__device__ __forceinline__ void pk_dft8_0167(v2f (&v)[8], v2f H) {
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) {
    const v2f a = v[n2], b = v[4 + n2];
    v[n2] = a + b;
    v[4 + n2] = a - b;
  }
  v[5] = pk_mul_w2(v[5], H);  // W8^1
  v[6] = pk_mul_mj(v[6]);     // W8^2 = -j
  v[7] = pk_mul_w6(v[7], H);  // W8^3
  pk_dft4_03(v[0], v[1], v[2], v[3]);  // X[0] -> v[0], X[6] -> v[3]
  pk_dft4_03(v[4], v[5], v[6], v[7]);  // X[1] -> v[4], X[7] -> v[7]
}
__device__ __forceinline__ constexpr int pk_slot8(int t) { return 4 * (t & 1) + (t >> 1); }
__device__ __forceinline__ constexpr int pk_slot16(int t) { return 4 * (t & 3) + (t >> 2); }

}  // namespace uc
