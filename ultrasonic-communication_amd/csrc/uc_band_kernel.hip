// uc_band_kernel.hip -- the per-frame DSP of the receiver as ONE gfx950 kernel.
//
// Replaces, per frame (reference lines):
//   ISR cast            fifo_queue[..] = (float)buf[i]          receiver/Src/main.c:664
//   mult_ref_chirp      signal *= up/down_chirp                 receiver/Src/chirp.c:47-53
//   Hann                arm_mult_f32(signal, hann_window)       receiver/Src/main.c:171
//   arm_rfft_fast_f32 / arm_cfft_f32 (2048)                     receiver/Src/main.c:174,
//                                                               synchronization/Src/main.c:153
//   arm_cmplx_mag_f32 + 2 x arm_max_f32 over the two windows    receiver/Src/main.c:178,205-215
//   history fill, snr, symbol decision                          receiver/Src/main.c:220-229,518-531
//
// Design (MI355X): one 2-wave workgroup transforms one frame at a time and
// walks the batch persistently.  The 2048-point transform is 16 x 16 x 8
// Stockham: both radix-16 passes are register-resident, the two exchanges go
// through a 16 KiB LDS tile (XOR-swizzled so that every ds_write_b64 /
// ds_read_b64 lane group is conflict-free), and the last radix-8 pass is
// evaluated only for bins 0..bandwidth2 and their mirror images -- the only
// bins dsp() looks at.  RX_REAL: both real references ride in one complex FFT
// (re = x*up*hann, im = x*down*hann) and are separated by Hermitian symmetry
// inside the pruned last pass.  HBM traffic is the frame itself (8 KiB) plus
// one symbol byte; the window*chirp table (16 KiB) lives in VGPRs, twiddles too.
#include "uc_fft.hpp"
#include "uc_kernels.hpp"

namespace uc {

namespace {

constexpr int T = kBandThreads;  // 128
constexpr int kMagOff = 2 * kN;  // floats: two magnitude arrays of 256 after the data tile
constexpr int kResOff = kMagOff + 512;
constexpr int kLdsFloats = kResOff + 16;

template <int DTYPE>
__device__ __forceinline__ float ld_sample(const void* base, size_t idx) {
  if (DTYPE == UC_DTYPE_I32) return (float)(reinterpret_cast<const int32_t*>(base)[idx]);
  return reinterpret_cast<const float*>(base)[idx];
}

__device__ __forceinline__ cf lds_ld(const float* lds, int cidx) {
  const float2 v = *reinterpret_cast<const float2*>(lds + 2 * cidx);
  return mk(v.x, v.y);
}
__device__ __forceinline__ void lds_st(float* lds, int cidx, cf v) {
  *reinterpret_cast<float2*>(lds + 2 * cidx) = make_float2(v.re, v.im);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// idx2freq(): receiver/Src/main.c:154-160, 32-bit unsigned arithmetic as on the MCU
__device__ __forceinline__ int32_t idx2freq(uint32_t ifs, uint32_t idx) {
  if (idx < (uint32_t)kN / 2) return (int32_t)(ifs * idx / (uint32_t)kN);
  return (int32_t)((ifs * ((uint32_t)kN - idx) / (uint32_t)kN) * 0xFFFFFFFFu);
}

// One arm_max_f32 over window bins k in [lo, hi] of arr[], executed by one wave.
// prefer_small: ties resolve to the smallest k (right window, ascending index);
// otherwise to the largest k (left window: index n-k ascending = k descending).
__device__ __forceinline__ void wave_window_max(const float* arr, int lo, int hi, bool prefer_small,
                                                int lane, float& out_val, int& out_k) {
  float v[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int k = lane + 64 * s;
    v[s] = (k >= lo && k <= hi) ? arr[k] : -INFINITY;
  }
  const float m = wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
  const int first = prefer_small ? lo : hi;
  const float vfirst = arr[first];
  int k = first;
  if (vfirst != vfirst) {
    // arm_max_f32 starts from src[0]; a NaN there never loses a '<' compare
    out_val = vfirst;
    out_k = first;
    return;
  }
  if (prefer_small) {
    bool found = false;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const unsigned long long b = __ballot(v[s] == m);
      if (!found && b) { k = 64 * s + (__ffsll((long long)b) - 1); found = true; }
    }
  } else {
    bool found = false;
#pragma unroll
    for (int s = 3; s >= 0; s--) {
      const unsigned long long b = __ballot(v[s] == m);
      if (!found && b) { k = 64 * s + (63 - __clzll((long long)b)); found = true; }
    }
  }
  out_val = m;
  out_k = k;
}

struct Hist {
  float mag_max, mag_left, mag_right, snr;
  int32_t f, fl, fr;
};

// the tail of dsp(): receiver/Src/main.c:209-229
__device__ __forceinline__ Hist make_hist(float mr, int kr, float ml, int kl, float mm, uint32_t ifs,
                                          bool raw_idx, uint32_t bw2) {
  Hist h;
  h.mag_left = ml;
  h.mag_right = mr;
  uint32_t idx_r = (uint32_t)kr, idx_l = (uint32_t)kN - (uint32_t)kl, idx;
  if (ml > mr) { h.mag_max = ml; idx = idx_l; } else { h.mag_max = mr; idx = idx_r; }
  if (raw_idx) {
    // chirp_compression_freq_domain/Src/main.c:152-156: indices, left as bw*8 - local
    h.fr = kr;
    h.fl = (int32_t)(bw2 - (bw2 - (uint32_t)kl));
    h.f = (ml > mr) ? h.fl : h.fr;
  } else {
    h.f = idx2freq(ifs, idx);
    h.fl = idx2freq(ifs, idx_l);
    h.fr = idx2freq(ifs, idx_r);
  }
  h.snr = (h.mag_max - mm) / mm;
  return h;
}

__device__ __forceinline__ void store_hist(uc_stats* dst, const Hist& h, float mm) {
  float4 a, b;
  a.x = h.mag_max; a.y = h.mag_left; a.z = h.mag_right; a.w = __int_as_float(h.f);
  b.x = __int_as_float(h.fl); b.y = __int_as_float(h.fr); b.z = mm; b.w = h.snr;
  float4* d = reinterpret_cast<float4*>(dst);
  d[0] = a;
  d[1] = b;
}

template <int MODE, int DTYPE, int WAVES>
__global__ __launch_bounds__(T, WAVES) void band_kernel(const BandParams p) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  float* mag0 = lds + kMagOff;        // RX_REAL: |A[k]| (up)   CPLX: |Z[k]|
  float* mag1 = lds + kMagOff + 256;  // RX_REAL: |B[k]| (down) CPLX: |Z[n-k]|
  float* res = lds + kResOff;         // 4 tasks x (value, k)

  const int j = threadIdx.x;
  const int lane = j & 63;
  const int wave = j >> 6;
  const int bw2 = (int)p.bw2;

  // ---- per-thread constants, resident for the whole batch -----------------
  cf wtab[16];  // RX_REAL only: window*chirp table entries of this thread's samples
  if (MODE == kModeRxReal) {
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const float2 w = p.tab0[j + T * t];
      wtab[t] = mk(w.x, w.y);
    }
  }
  cf tw2[16];  // pass-2 twiddles W_256^(t*k), k = j & 15
#pragma unroll
  for (int t = 1; t < 16; t++) {
    const float2 w = p.tw[(8 * t * (j & 15)) & (kN - 1)];
    tw2[t] = mk(w.x, w.y);
  }
  // pass-3 twiddles W_2048^(t*j): only t = 1, 2, 4 stay resident, the other four
  // are one complex product away (keeps the kernel at 3 waves/SIMD)
  cf tw3_1, tw3_2, tw3_4;
  {
    const float2 w1 = p.tw[j & (kN - 1)], w2 = p.tw[(2 * j) & (kN - 1)], w4 = p.tw[(4 * j) & (kN - 1)];
    tw3_1 = mk(w1.x, w1.y);
    tw3_2 = mk(w2.x, w2.y);
    tw3_4 = mk(w4.x, w4.y);
  }

  // LDS addresses (complex units)
  const int s1 = j & 15;
  const int wr1 = 16 * j;                                        // + (t ^ s1)
  const int rd1e = (j & ~15) + ((j & 15) ^ (j >> 4));            // + 128 t, t even
  const int rd1o = (j & ~15) + ((j & 15) ^ (j >> 4) ^ 8);        // + 128 t, t odd
  const int wr2 = (j >> 4) * 256 + (j & 15);                     // + 16 t

  const size_t nfr = p.n_frames;
  const size_t gstep = gridDim.x;
  size_t f = blockIdx.x;
  if (f >= nfr) return;

  float xr[16];
  {
    const size_t base = f * p.stride;
#pragma unroll
    for (int t = 0; t < 16; t++) xr[t] = ld_sample<DTYPE>(p.frames, base + j + T * t);
  }

  size_t fprev = 0;
  bool have_prev = false;

  // finaliser: history[0], history[1], symbol of frame `ff` from res[]
  auto finalise = [&](size_t ff) {
    float mm_up = p.mag_mean_scalar, mm_dn = p.mag_mean_scalar;
    if (p.mag_mean) { mm_up = p.mag_mean[2 * ff]; mm_dn = p.mag_mean[2 * ff + 1]; }
    const Hist h0 = make_hist(res[0], __float_as_int(res[1]), res[2], __float_as_int(res[3]), mm_up,
                              p.ifs, p.single != 0, p.bw2);
    if (p.single) {
      if (p.stats) store_hist(p.stats + ff, h0, mm_up);
      if (p.symbols) p.symbols[ff] = (uint8_t)UC_SYM_NONE;
      return;
    }
    const Hist h1 = make_hist(res[4], __float_as_int(res[5]), res[6], __float_as_int(res[7]), mm_dn,
                              p.ifs, false, p.bw2);
    if (p.stats) {
      store_hist(p.stats + 2 * ff, h0, mm_up);
      store_hist(p.stats + 2 * ff + 1, h1, mm_dn);
    }
    if (p.symbols) {
      // receiver/Src/main.c:521-531
      uint8_t sym = (uint8_t)UC_SYM_NONE;
      if ((h0.snr >= p.snr_threshold) || (h1.snr >= p.snr_threshold))
        sym = (h1.snr > h0.snr) ? (uint8_t)UC_SYM_DOWN : (uint8_t)UC_SYM_UP;
      p.symbols[ff] = sym;
    }
  };

  for (; f < nfr; f += gstep) {
    // Opaque re-definitions: stop LICM from hoisting the 16 swizzled store
    // addresses and the derived pass-3 twiddles out of the frame loop, where they
    // would sit in (spilled) registers for the whole batch.
    int s1v = s1;
    cf t3a = tw3_1, t3b = tw3_2, t3c = tw3_4;
    asm volatile("" : "+v"(s1v), "+v"(t3a.re), "+v"(t3a.im), "+v"(t3b.re), "+v"(t3b.im), "+v"(t3c.re), "+v"(t3c.im));
    const size_t fnext = f + gstep;
    const bool has_next = fnext < nfr;
    constexpr int kRuns = (MODE == kModeCplx) ? 2 : 1;
#pragma unroll
    for (int run = 0; run < kRuns; run++) {
      cf v[16];
      // ---- pass 1: window*chirp multiply, radix-16, Ns = 1 ------------------
      if (MODE == kModeRxReal) {
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = mk(xr[t] * wtab[t].re, xr[t] * wtab[t].im);
      } else {
        const float2* tab = run == 0 ? p.tab0 : p.tab1;
#pragma unroll
        for (int t = 0; t < 16; t++) {
          const float2 w = tab[j + T * t];
          v[t] = mk(xr[t] * w.x, xr[t] * w.y);
        }
      }
      dft16<false>(v);
#pragma unroll
      for (int t = 0; t < 16; t++) lds_st(lds, wr1 + (t ^ s1v), v[slot16(t)]);
      __syncthreads();  // B1

      if (j == 0 && have_prev && run == 0) finalise(fprev);

      // ---- pass 2: radix-16, Ns = 16 ----------------------------------------
#pragma unroll
      for (int t = 0; t < 16; t++) v[t] = lds_ld(lds, ((t & 1) ? rd1o : rd1e) + 128 * t);
#pragma unroll
      for (int t = 1; t < 16; t++) v[t] = cmul(v[t], tw2[t]);
      dft16<false>(v);
      __syncthreads();  // B2: every pass-2 read is done before the tile is overwritten
#pragma unroll
      for (int t = 0; t < 16; t++) lds_st(lds, wr2 + 16 * t, v[slot16(t)]);
      __syncthreads();  // B3

      // prefetch the next frame while the pruned pass and the window search run
      if (run == kRuns - 1 && has_next) {
        const size_t base = fnext * p.stride;
#pragma unroll
        for (int t = 0; t < 16; t++) xr[t] = ld_sample<DTYPE>(p.frames, base + j + T * t);
      }

      // ---- pass 3: radix-8, Ns = 256, only bins i in [0, bw2] and n - i -----
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int i = j + T * r;
        if (i <= bw2) {
          cf w[8];
          if (r == 0) {
            w[1] = t3a;
            w[2] = t3b;
            w[4] = t3c;
            w[3] = cmul(t3a, t3b);
            w[5] = cmul(t3a, t3c);
            w[6] = cmul(t3b, t3c);
            w[7] = cmul(w[3], t3c);
          } else {
#pragma unroll
            for (int t = 1; t < 8; t++) {
              const float2 ww = p.tw[(t * i) & (kN - 1)];
              w[t] = mk(ww.x, ww.y);
            }
          }
          const int ib = (256 - i) & 255;
          cf a[8], b[8];
#pragma unroll
          for (int t = 0; t < 8; t++) {
            a[t] = lds_ld(lds, i + 256 * t);
            b[t] = lds_ld(lds, ib + 256 * t);
          }
          if (MODE == kModeRxReal) {
            // A[k] = (Z[k] + conj Z[n-k]) / 2,  B[k] = (Z[k] - conj Z[n-k]) / 2j
            cf sa = a[0] + cconj(b[0]);
            cf sb = a[0] - cconj(b[0]);
#pragma unroll
            for (int t = 1; t < 8; t++) {
              sa = cfma(a[t] + cconj(b[t]), w[t], sa);
              sb = cfma(a[t] - cconj(b[t]), w[t], sb);
            }
            float ma = 0.5f * sqrtf(sa.re * sa.re + sa.im * sa.im);
            float mb = 0.5f * sqrtf(sb.re * sb.re + sb.im * sb.im);
            if (i == 0) {
              // Q2: the packed RFFT stores Re X[n/2] in the imaginary slot of bin 0,
              // so the reference's mag[0] is hypot(X0, X[n/2]) (receiver/Src/main.c:178)
              const float a0 = 0.5f * sa.re;  // Re Z[0]  = X_up[0]
              const float b0 = 0.5f * sb.im;  // Im Z[0]  = X_down[0]   (sb = 2j*Im)
              cf zn = a[0];
#pragma unroll
              for (int t = 1; t < 8; t++) zn = (t & 1) ? (zn - a[t]) : (zn + a[t]);
              if (p.true_dc) {
                ma = fabsf(a0);
                mb = fabsf(b0);
              } else {
                ma = sqrtf(a0 * a0 + zn.re * zn.re);
                mb = sqrtf(b0 * b0 + zn.im * zn.im);
              }
            }
            mag0[i] = ma;
            mag1[i] = mb;
          } else {
            cf zl = a[0], zh = b[0];
#pragma unroll
            for (int t = 1; t < 8; t++) {
              zl = cfma(a[t], w[t], zl);
              zh = cfma(b[t], cconj(w[t]), zh);
            }
            mag0[i] = sqrtf(zl.re * zl.re + zl.im * zl.im);
            mag1[i] = sqrtf(zh.re * zh.re + zh.im * zh.im);
          }
        }
      }
      __syncthreads();  // B4: magnitudes visible; data tile free for the next pass 1

      // ---- windows: arm_max_f32 x 2 per history ------------------------------
      if (MODE == kModeRxReal) {
        const float* arr = wave == 0 ? mag0 : mag1;
        float mr, ml;
        int kr, kl;
        wave_window_max(arr, 0, bw2 - 1, true, lane, mr, kr);
        wave_window_max(arr, 1, bw2, false, lane, ml, kl);
        if (lane == 0) {
          res[4 * wave + 0] = mr;
          res[4 * wave + 1] = __int_as_float(kr);
          res[4 * wave + 2] = ml;
          res[4 * wave + 3] = __int_as_float(kl);
        }
      } else {
        float m;
        int k;
        if (wave == 0) wave_window_max(mag0, 0, bw2 - 1, true, lane, m, k);
        else wave_window_max(mag1, 1, bw2, false, lane, m, k);
        if (lane == 0) {
          res[4 * run + 2 * wave + 0] = m;
          res[4 * run + 2 * wave + 1] = __int_as_float(k);
        }
      }
    }
    fprev = f;
    have_prev = true;
  }
  __syncthreads();
  if (j == 0 && have_prev) finalise(fprev);
}

}  // namespace

template <int MODE, int DTYPE, int WAVES>
static int launch_one(const BandParams& p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL((band_kernel<MODE, DTYPE, WAVES>), dim3((unsigned)grid), dim3((unsigned)T), 0, stream, p);
  return (int)hipGetLastError();
}

template <int MODE, int DTYPE, int WAVES>
static int occupancy_one() {
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, band_kernel<MODE, DTYPE, WAVES>, T, 0);
  if (e != hipSuccess || nb <= 0) nb = 2 * WAVES;
  return nb;
}

#define UC_DISPATCH(FN, ...)                                                              \
  do {                                                                                    \
    if (mode == kModeRxReal) {                                                            \
      if (dtype == UC_DTYPE_I32) {                                                        \
        if (waves == 3) return FN<kModeRxReal, UC_DTYPE_I32, 3>(__VA_ARGS__);             \
        if (waves == 4) return FN<kModeRxReal, UC_DTYPE_I32, 4>(__VA_ARGS__);             \
        return FN<kModeRxReal, UC_DTYPE_I32, 2>(__VA_ARGS__);                             \
      }                                                                                   \
      if (waves == 3) return FN<kModeRxReal, UC_DTYPE_F32, 3>(__VA_ARGS__);               \
      if (waves == 4) return FN<kModeRxReal, UC_DTYPE_F32, 4>(__VA_ARGS__);               \
      return FN<kModeRxReal, UC_DTYPE_F32, 2>(__VA_ARGS__);                               \
    }                                                                                     \
    if (dtype == UC_DTYPE_I32) {                                                          \
      if (waves == 3) return FN<kModeCplx, UC_DTYPE_I32, 3>(__VA_ARGS__);                 \
      if (waves == 4) return FN<kModeCplx, UC_DTYPE_I32, 4>(__VA_ARGS__);                 \
      return FN<kModeCplx, UC_DTYPE_I32, 2>(__VA_ARGS__);                                 \
    }                                                                                     \
    if (waves == 3) return FN<kModeCplx, UC_DTYPE_F32, 3>(__VA_ARGS__);                   \
    if (waves == 4) return FN<kModeCplx, UC_DTYPE_F32, 4>(__VA_ARGS__);                   \
    return FN<kModeCplx, UC_DTYPE_F32, 2>(__VA_ARGS__);                                   \
  } while (0)

int launch_band(int mode, int dtype, int waves, const BandParams& p, int grid, hipStream_t stream) {
  if (grid <= 0) return (int)hipSuccess;
  UC_DISPATCH(launch_one, p, grid, stream);
}

int band_max_blocks_per_cu(int mode, int dtype, int waves) {
  UC_DISPATCH(occupancy_one);
}

}  // namespace uc
