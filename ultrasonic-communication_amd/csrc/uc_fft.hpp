// uc_fft.hpp -- register-resident DFT butterflies for the gfx950 kernels.
//
// One 2048-point frame is transformed as 16 x 16 x 8 (Stockham, forward sign
// e^{-j}): two radix-16 passes live entirely in VGPRs, the exchanges between
// passes go through LDS, and the last radix-8 pass is evaluated only for the
// bins the receiver's two windows look at (receiver/Src/main.c:205-208).
#pragma once
#include <hip/hip_runtime.h>

namespace uc {

struct cf {
  float re, im;
};

__device__ __forceinline__ cf mk(float re, float im) { cf r; r.re = re; r.im = im; return r; }
__device__ __forceinline__ cf operator+(cf a, cf b) { return mk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cf operator-(cf a, cf b) { return mk(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cf cmul(cf a, cf w) {
  return mk(a.re * w.re - a.im * w.im, a.re * w.im + a.im * w.re);
}
// a * conj(w)
__device__ __forceinline__ cf cmulc(cf a, cf w) {
  return mk(a.re * w.re + a.im * w.im, a.im * w.re - a.re * w.im);
}
__device__ __forceinline__ cf cconj(cf a) { return mk(a.re, -a.im); }
__device__ __forceinline__ cf mul_mj(cf a) { return mk(a.im, -a.re); }   // a * (-j)
__device__ __forceinline__ cf mul_pj(cf a) { return mk(-a.im, a.re); }   // a * (+j)
__device__ __forceinline__ cf scale(cf a, float s) { return mk(a.re * s, a.im * s); }
// acc += a * w
__device__ __forceinline__ cf cfma(cf a, cf w, cf acc) {
  return mk(fmaf(a.re, w.re, fmaf(-a.im, w.im, acc.re)), fmaf(a.re, w.im, fmaf(a.im, w.re, acc.im)));
}

constexpr float kSqrtHalf = 0.70710678118654752440f;
constexpr float kC8 = 0.92387953251128675613f;  // cos(pi/8)
constexpr float kS8 = 0.38268343236508977173f;  // sin(pi/8)

// forward 4-point DFT, natural order in and out
__device__ __forceinline__ void dft4(cf& x0, cf& x1, cf& x2, cf& x3) {
  cf a0 = x0 + x2, a1 = x0 - x2, a2 = x1 + x3, a3 = mul_mj(x1 - x3);
  x0 = a0 + a2;
  x2 = a0 - a2;
  x1 = a1 + a3;
  x3 = a1 - a3;
}

// inverse-sign 4-point DFT (e^{+j}), natural order
__device__ __forceinline__ void idft4(cf& x0, cf& x1, cf& x2, cf& x3) {
  cf a0 = x0 + x2, a1 = x0 - x2, a2 = x1 + x3, a3 = mul_pj(x1 - x3);
  x0 = a0 + a2;
  x2 = a0 - a2;
  x1 = a1 + a3;
  x3 = a1 - a3;
}

// Forward 16-point DFT as 4 x 4.  Input v[n] natural; on return X[t] sits in
// v[slot16(t)] with slot16(t) = 4*(t&3) + (t>>2).
__device__ __forceinline__ constexpr int slot16(int t) { return 4 * (t & 3) + (t >> 2); }

template <bool INV>
__device__ __forceinline__ void dft16(cf (&v)[16]) {
  // columns: DFT4 over n1 of v[4*n1 + n2]  -> v[4*k1 + n2]
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) {
    if (INV) idft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    else dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
  }
  // twiddles W16^(n2*k1)
  const cf w1 = mk(kC8, INV ? kS8 : -kS8);
  const cf w2 = mk(kSqrtHalf, INV ? kSqrtHalf : -kSqrtHalf);
  const cf w3 = mk(kS8, INV ? kC8 : -kC8);
  const cf w6 = mk(-kSqrtHalf, INV ? kSqrtHalf : -kSqrtHalf);
  const cf w9 = mk(-kC8, INV ? -kS8 : kS8);
  v[4 + 1] = cmul(v[4 + 1], w1);
  v[4 + 2] = cmul(v[4 + 2], w2);
  v[4 + 3] = cmul(v[4 + 3], w3);
  v[8 + 1] = cmul(v[8 + 1], w2);
  v[8 + 2] = INV ? mul_pj(v[8 + 2]) : mul_mj(v[8 + 2]);
  v[8 + 3] = cmul(v[8 + 3], w6);
  v[12 + 1] = cmul(v[12 + 1], w3);
  v[12 + 2] = cmul(v[12 + 2], w6);
  v[12 + 3] = cmul(v[12 + 3], w9);
  // rows: DFT4 over n2 of v[4*k1 + n2] -> X[k1 + 4*k2] in v[4*k1 + k2]
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) {
    if (INV) idft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    else dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
  }
}

// Forward / inverse 8-point DFT as 2 x 4.  On return X[t] sits in v[slot8(t)],
// slot8(t) = 4*(t&1) + (t>>1).
__device__ __forceinline__ constexpr int slot8(int t) { return 4 * (t & 1) + (t >> 1); }

template <bool INV>
__device__ __forceinline__ void dft8(cf (&v)[8]) {
  // n = 2*n1 + n2 (n1 in 0..3, n2 in 0..1), k = k1 + 4*k2 ... use 4 x 2:
  // step 1: DFT2 over n1' of v[4*n1' + n2'] (n = 4*n1' + n2', n1' in 0..1, n2' in 0..3)
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) {
    cf a = v[n2], b = v[4 + n2];
    v[n2] = a + b;       // k1 = 0
    v[4 + n2] = a - b;   // k1 = 1
  }
  // twiddles W8^(n2*k1), k1 = 1 row only
  const cf w1 = mk(kSqrtHalf, INV ? kSqrtHalf : -kSqrtHalf);
  const cf w3 = mk(-kSqrtHalf, INV ? kSqrtHalf : -kSqrtHalf);
  v[4 + 1] = cmul(v[4 + 1], w1);
  v[4 + 2] = INV ? mul_pj(v[4 + 2]) : mul_mj(v[4 + 2]);
  v[4 + 3] = cmul(v[4 + 3], w3);
  // step 2: DFT4 over n2 of v[4*k1 + n2] -> X[k1 + 2*k2] in v[4*k1 + k2]
  if (INV) {
    idft4(v[0], v[1], v[2], v[3]);
    idft4(v[4], v[5], v[6], v[7]);
  } else {
    dft4(v[0], v[1], v[2], v[3]);
    dft4(v[4], v[5], v[6], v[7]);
  }
}

}  // namespace uc
