// uc_iq_kernel.hip -- UC_IQ: I/Q down-conversion + chirp multiply + CFFT + band maxima.
//
// Replaces, per frame (reference lines, experiments/iq_modulation):
//   iq_demodulation(): x*carrier_sin / x*carrier_cos              Src/iq_modem.c:60-61
//                      2 x arm_fir_f32, 27 taps, state carried     Src/iq_modem.c:64-65, taps :18
//   mult_ref_chirp():  (I + jQ) * down_chirp                       Src/chirp.c:46-48   (Q7 fixed: interleaved)
//   Hann, arm_cfft_f32(len 2048), arm_cmplx_mag_f32(n/2)           Src/main.c:126-132
//   arm_max_f32 x 3 over [idx_left_zero, +bw4), [.., +bw2), [center, +bw2)   Src/main.c:283-285
//
// Design (MI355X): one 2-wave workgroup per frame (n = 2048; ONE wave per frame at n = 1024, below),
// persistent over round-robin groups of up to 64 frames.  The frame plus its 26-sample FIR
// history is mixed with the carrier while it is written into a padded LDS image (stride 17
// per 16 samples: conflict-free for the per-thread sliding window); every thread then filters
// 16 CONSECUTIVE outputs from a 42-sample register window (27 packed FMAs per output, I and Q
// together, eight accumulators in flight, two taps per SGPR pair), writes them in the
// XOR-swizzled layout of the band kernel's first exchange, and the 16 x 16 x 8 FFT of
// uc_band_kernel.hip follows with the chirp*Hann table multiplied in pass 1.  Only the
// 4*bandwidth bins the three windows look at are evaluated in the last pass.
// The loop loads nothing but frames: the next frame sits in 17 registers a frame time ahead,
// the carrier, chirp*Hann and twiddle entries of a thread are resident (vector loads return
// in order: a table load behind the prefetch would stall the arithmetic on HBM latency).
// Image and tile ping-pong: one barrier per exchange.  Window partials go to a ring in LDS
// that is finalised for 64 frames at once (one lane per frame, coalesced 32-byte records).
// Compute-bound by definition (the FIR alone is as many flops as the FFT); HBM traffic is
// 4 n + 104 B in, 32 B out per frame.
#include "uc_dev.hpp"
#include "uc_kernels.hpp"

#ifndef UC_IQ_KNOCK
#define UC_IQ_KNOCK 0  // diagnostic builds only (tools/ab_build.sh; results WRONG by construction, timing only):
                       // 1 = no carrier mix (the image holds (x, x)): the VALU a carrier folded into the taps would save
#endif

namespace uc {

namespace {

// "no group asked for": stays beyond every group count when gridDim.x is added (batches hold fewer than 2^31 frames)
constexpr unsigned kNoGroup = 0x7fffffffu;

constexpr int T = kBandThreads;          // 128
constexpr int kHalo = 26;                // FIR taps - 1
constexpr int kMixLen = kN + kHalo;      // 2074 mixed samples

template <int DTYPE>
__device__ __forceinline__ float cvt1(float raw) {
  if (DTYPE == UC_DTYPE_I32) return (float)__float_as_int(raw);
  return raw;
}

// padded index of mixed sample m (m = frame index + 26)
__device__ __forceinline__ int mix_idx(int m) { return m + (m >> 4); }

// this wave's first-maximum (smallest bin on ties) over its candidates inside [lo, hi)
__device__ __forceinline__ void window_first_max(float q0, int k0, float q1, int k1, int lo, int hi, float& v, int& k) {
  const float c0 = (k0 >= lo && k0 < hi) ? q0 : -INFINITY;
  const float c1 = (k1 >= lo && k1 < hi) ? q1 : -INFINITY;
  v = wave_max_f32(max_f32(c0, c1));
  int cand = (c0 == v) ? k0 : 0x7fffffff;
  const int cand1 = (c1 == v) ? k1 : 0x7fffffff;
  cand = cand1 < cand ? cand1 : cand;
  k = wave_min_u32(cand);
}

// UC_FLAG_IQ_BASEBAND (simulation/IQ_modulation.ipynb cells 28-31): one history of dsp() from the two window
// partials of one dechirp run; the windows straddle DC, searched and merged as receiver/Src/main.c:205-229 does.
// il is a bin of [n - bandwidth, n), ir an UNWRAPPED bin of [n, n + bandwidth).
struct BbHist {
  float mag_max, mag_left, mag_right, snr;
  int32_t f, fl, fr;
};
__device__ __forceinline__ int32_t idx2freq_n(uint32_t ifs, uint32_t idx, uint32_t n) {  // receiver/Src/main.c:154-160
  if (idx < n / 2) return (int32_t)(ifs * idx / n);
  return (int32_t)((ifs * (n - idx) / n) * 0xFFFFFFFFu);
}
__device__ __forceinline__ BbHist bb_hist(float ql, int il, float qr, int ir, float mm, uint32_t ifs, uint32_t n) {
  BbHist h;
  h.mag_left = sqrtf(ql);
  h.mag_right = sqrtf(qr);
  const uint32_t idx_l = (uint32_t)il, idx_r = (uint32_t)ir - n;
  uint32_t idx;
  if (h.mag_left > h.mag_right) { h.mag_max = h.mag_left; idx = idx_l; } else { h.mag_max = h.mag_right; idx = idx_r; }
  h.f = idx2freq_n(ifs, idx, n);
  h.fl = idx2freq_n(ifs, idx_l, n);
  h.fr = idx2freq_n(ifs, idx_r, n);
  h.snr = (h.mag_max - mm) / mm;
  return h;
}
__device__ __forceinline__ void bb_store(uc_stats* dst, const BbHist& h, float mm) {
  float4 a, b;
  a.x = h.mag_max; a.y = h.mag_left; a.z = h.mag_right; a.w = __int_as_float(h.f);
  b.x = __int_as_float(h.fl); b.y = __int_as_float(h.fr); b.z = mm; b.w = h.snr;
  float4* d = reinterpret_cast<float4*>(dst);
  d[0] = a;
  d[1] = b;
}
// both histories of one frame + the symbol decision (receiver/Src/main.c:518-531); e0 / e1 = (ql, il, qr, ir) of the runs
__device__ __forceinline__ void bb_finish(const IqParams& p, size_t ff, float ql0, int il0, float qr0, int ir0, float ql1,
                                          int il1, float qr1, int ir1, uint32_t n) {
  const float mm_up = p.mag_mean ? p.mag_mean[2 * ff] : p.mag_mean_scalar;
  const float mm_dn = p.mag_mean ? p.mag_mean[2 * ff + 1] : p.mag_mean_scalar;
  const BbHist h0 = bb_hist(ql0, il0, qr0, ir0, mm_up, p.ifs, n);
  const BbHist h1 = bb_hist(ql1, il1, qr1, ir1, mm_dn, p.ifs, n);
  if (p.stats) {
    bb_store(p.stats + 2 * ff, h0, mm_up);
    bb_store(p.stats + 2 * ff + 1, h1, mm_dn);
  }
  if (p.symbols) {
    uint8_t sym = (uint8_t)UC_SYM_NONE;
    if ((h0.snr >= p.snr_threshold) || (h1.snr >= p.snr_threshold))
      sym = (h1.snr > h0.snr) ? (uint8_t)UC_SYM_DOWN : (uint8_t)UC_SYM_UP;
    p.symbols[ff] = sym;
  }
}

// LDS: the padded mixed image (also exchange 1), a second tile (FIR outputs, exchange 2) and the
// ring of per-frame window partials.  Ping-ponging between image and tile leaves ONE barrier per
// exchange: a buffer is rewritten only after a barrier that follows its last read.
constexpr int kImg = T * 17 + ((T * 17) >> 4) + 4;  // 2316: every m = j + 128 u, u < 17, has a slot
constexpr int kRingFr = 64;                         // (base band: 32 frames of 2 runs each, same footprint)
constexpr int kRingSt = 11;                         // 2 waves x (vl, kl, vr, kr, flags) + 1 pad
constexpr int kRingStBb = 21;                       // 2 runs x 2 waves x 5 + 1 pad
constexpr int kRingFrBb = 32;
constexpr int kTileOff = 2 * kImg;
constexpr int kRingO = kTileOff + 2 * kN;
constexpr int kTw2O = kRingO + kRingFr * kRingSt;  // even: 8-byte aligned
constexpr int kNextO = kTw2O + 2 * 256;  // one word: the group a dynamic hand-out gave this workgroup next
constexpr int kLdsAll = kNextO + 2;
static_assert((kTw2O & 1) == 0, "complex alignment");
static_assert(kImg >= kN, "exchange 1 lives in the image area");
static_assert(kRingFrBb * kRingStBb <= kRingFr * kRingSt, "the base-band ring fits the same LDS");

// BB: 0 = the firmware's windows, 1 = UC_FLAG_IQ_BASEBAND, 2 = base band with windows of at most 64 bins each (61 at
// BASELINE configs[2]'s constants): wave 0 owns the left window's bins, wave 1 the right window's -- ONE pruned round,
// one wave reduction per dechirp run and wave.
// wave priority, as in the band kernel: low while a wave issues a burst of LDS stores, raised otherwise, highest for the
// pruned pass that ends a run (UC_IQ_PRIO_OFF: the A/B switch of round 6)
#ifdef UC_IQ_PRIO_OFF
#define UC_IQ_PRIO(n) do { } while (0)
#else
#define UC_IQ_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif

template <int DTYPE, int BB>
__global__ __launch_bounds__(T, 2) void iq_kernel(const IqParams p) {
  UC_CLOCK_BEGIN();  // diagnostic build only (uc_dev.hpp)
  __shared__ __attribute__((aligned(16))) float lds[kLdsAll];
  float* img = lds;
  float* tile = lds + kTileOff;
  float* ring = lds + kRingO;
  float* tw2l = lds + kTw2O;

  const int j = threadIdx.x;
  const int lane = j & 63;
  const int wave = j >> 6;

  // frames are dealt in groups of G = 2^gsh consecutive frames (32-bit bookkeeping: the host rejects batches of 2^31
  // frames or more; the frame ADDRESS is 64-bit).  Static deal: round robin over the workgroups.  Dynamic hand-out
  // (p.work_ctr, G >= 2) as in iq1024_kernel below: thread 0 asks for the next group behind the FIR of a group's
  // second-to-last frame; at the top of the last frame it passes the id to both waves through one LDS word.
  const unsigned nfr = (unsigned)p.n_frames;
  const unsigned gsh = (unsigned)__builtin_ctz(p.group), gmask = (1u << gsh) - 1u;
  const unsigned ngroups = (nfr + gmask) >> gsh;
  unsigned grp = blockIdx.x;
  const bool dyn = p.work_ctr != nullptr;
  if (grp >= ngroups) {  // (the host never launches more workgroups than groups)
    if (dyn && j == 0) handout_leave(p.work_ctr);
    return;
  }
  unsigned f = grp << gsh;
  unsigned fetched = kNoGroup;  // thread 0: what the atomic in flight returns; kNoGroup = none asked for (the ragged last group)
  unsigned* next_slot = reinterpret_cast<unsigned*>(lds + kNextO);

  const __amdgpu_buffer_rsrc_t rs_car = make_rsrc(p.carrier, kN * 8);
  const __amdgpu_buffer_rsrc_t rs_ch = make_rsrc(p.chirp_hann, kN * 8);
  const __amdgpu_buffer_rsrc_t rs_ch2 = make_rsrc(BB ? p.chirp_hann2 : p.chirp_hann, kN * 8);
  const __amdgpu_buffer_rsrc_t rs_tw = make_rsrc(p.tw, kN * 8);
  const int voff8 = j * 8;
  const v2f K = mkv(kCos8, kSin8), H = mkv(kSqrtHalfF, kSqrtHalfF);

  const int lo = (int)p.idx_left_zero, center = (int)p.center;
  const int bw2 = (int)p.bw2, bw4 = (int)p.bw4;

  // ---- resident tables (the loop loads nothing but the frames: vector loads return in order) ----
  // carrier of mixed sample m = j + 128 u (frame index i = m - 26).  History was mixed with the
  // TAIL of the table, as the previous back-to-back block's samples were (iq_modem.c:60-61 with
  // the state carried in i_state / q_state).  m >= 2074 reads past the table: (0, 0).
  v2f cs[17];
#pragma unroll
  for (int u = 0; u < 17; u++) {
    const int i = j + T * u - kHalo;
    const int ci = i < 0 ? kN + i : i;
    cs[u] = buf_ld64(rs_car, ci * 8, 0);
  }
  v2f ch[16];  // chirp*hann of sample n = j + 128 t
#pragma unroll
  for (int t = 0; t < 16; t++) ch[t] = buf_ld64(rs_ch, voff8, T * 8 * t);
  // pass 2: W_256^(t k), k = j & 15 -- 16 x 16 entries shared through LDS (read with the pass-2 tile reads)
  for (int e = j; e < 256; e += T) lds_st(tw2l, e, buf_ld64(rs_tw, ((8 * (e >> 4) * (e & 15)) & (kN - 1)) * 8, 0));
  const int tw2o = j & 15;  // + 16 t
  // pass-3 twiddles of bins k = lo + j and lo + 128 + j: W^k and W^2k (Horner form)
  // (BB == 2: one bin per thread -- lo + lane on wave 0, center + lane on wave 1)
  const int kk2 = (wave == 0) ? lo + lane : center + lane;
  v2f t3w1[2], t3w2[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int k = (BB == 2) ? kk2 : lo + T * r + j;
    t3w1[r] = buf_ld64(rs_tw, (k & (kN - 1)) * 8, 0);
    t3w2[r] = buf_ld64(rs_tw, ((2 * k) & (kN - 1)) * 8, 0);
  }
  // the 27 real taps, two per SGPR pair
  v2f taps[14];
#pragma unroll
  for (int i = 0; i < 14; i++) taps[i] = mkv(p.fir[2 * i], 2 * i + 1 < kFirTapsDev ? p.fir[2 * i + 1] : 0.f);

  const int s1 = j & 15;
  const int wr1 = 16 * j;                                   // exchange-1 layout: + (t ^ s1)
  const int rd1e = (j & ~15) + ((j & 15) ^ (j >> 4));       // + 128 t, t even
  const int rd1o = (j & ~15) + ((j & 15) ^ (j >> 4) ^ 8);   // + 128 t, t odd
  const int wr2 = (j >> 4) * 256 + (j & 15);                // exchange-2 layout: + 16 t

  // the frame's 2074 samples (26 of history first) as raw words, one frame ahead
  float xn[17];
  auto load_frame = [&](size_t fr) {
    const __amdgpu_buffer_rsrc_t rx =
        make_rsrc(reinterpret_cast<const char*>(p.frames) + (fr * p.stride) * 4 - kHalo * 4, kMixLen * 4);
#pragma unroll
    for (int u = 0; u < 17; u++) xn[u] = buf_ld32_stream(rx, j * 4, T * 4 * u);  // past the end: 0
  };
  load_frame(f);

  // finaliser: lane L of wave 0 merges the two waves' partials of ring slot L into the history
  // record of frame f0 + L (experiments/iq_modulation/Src/main.c:283-303)
  auto finalise = [&](size_t f0, int count) {
    if (BB) {
      if (lane < count) {
        const float* e = ring + lane * kRingStBb;
        float ql[2], qr[2];
        int il[2], ir[2];
#pragma unroll
        for (int run = 0; run < 2; run++) {
          const float* er = e + 10 * run;
          const float a0 = er[0], a1 = er[5], b0 = er[2], b1 = er[7];
          const int ka0 = __float_as_int(er[1]), ka1 = __float_as_int(er[6]);
          const int kb0 = __float_as_int(er[3]), kb1 = __float_as_int(er[8]);
          const int flags = __float_as_int(er[4]) | __float_as_int(er[9]);
          ql[run] = a0; qr[run] = b0; il[run] = ka0; ir[run] = kb0;
          if (a1 > a0 || (a1 == a0 && ka1 < ka0)) { ql[run] = a1; il[run] = ka1; }
          if (b1 > b0 || (b1 == b0 && kb1 < kb0)) { qr[run] = b1; ir[run] = kb1; }
          if (flags & 1) { ql[run] = __int_as_float(0x7fc00000); il[run] = lo; }
          if (flags & 2) { qr[run] = __int_as_float(0x7fc00000); ir[run] = center; }
        }
        bb_finish(p, f0 + (size_t)lane, ql[0], il[0], qr[0], ir[0], ql[1], il[1], qr[1], ir[1], (uint32_t)kN);
      }
      return;
    }
    if (lane < count) {
      const float* e = ring + lane * kRingSt;
      const size_t ff = f0 + (size_t)lane;
      const float a0 = e[0], a1 = e[5], b0 = e[2], b1 = e[7];
      const int ka0 = __float_as_int(e[1]), ka1 = __float_as_int(e[6]);
      const int kb0 = __float_as_int(e[3]), kb1 = __float_as_int(e[8]);
      const int flags = __float_as_int(e[4]) | __float_as_int(e[9]);
      // merge waves: larger value, ties -> smaller bin
      float ql = a0, qr = b0;
      int il = ka0, ir = kb0;
      if (a1 > a0 || (a1 == a0 && ka1 < ka0)) { ql = a1; il = ka1; }
      if (b1 > b0 || (b1 == b0 && kb1 < kb0)) { qr = b1; ir = kb1; }
      if (flags & 1) { ql = __int_as_float(0x7fc00000); il = lo; }
      if (flags & 2) { qr = __int_as_float(0x7fc00000); ir = center; }
      const float ml = sqrtf(ql), mr = sqrtf(qr);
      // full window [lo, lo + bw4) = left then right: the right part wins only if strictly greater
      float mx = ml;
      int ix = il;
      if (!(ml != ml) && mr > ml) { mx = mr; ix = ir; }
      if (p.stats) {
        const float mm = p.mag_mean ? p.mag_mean[2 * ff] : p.mag_mean_scalar;
        // idx2freq of this experiment: (uint32)(sampling_rate * idx / n), Src/main.c:112-114
        const float fsn = p.fs;
        float4 sa, sb;
        sa.x = mx; sa.y = ml; sa.z = mr;
        sa.w = __int_as_float((int)(unsigned)(fsn * (float)ix / (float)kN));
        sb.x = __int_as_float((int)(unsigned)(fsn * (float)il / (float)kN));
        sb.y = __int_as_float((int)(unsigned)(fsn * (float)ir / (float)kN));
        sb.z = mm;
        sb.w = (mx - mm) / mm;
        float4* d = reinterpret_cast<float4*>(p.stats + ff);
        d[0] = sa;
        d[1] = sb;
      }
      if (p.symbols) p.symbols[ff] = (uint8_t)UC_SYM_NONE;
    }
  };

  unsigned ring_f0 = f;
  int ring_n = 0;

  for (;;) {
    unsigned fnext = f + 1;
    if ((fnext & gmask) == 0 || fnext >= nfr) {
      if (dyn) {
        if (j == 0) *next_slot = fetched;
        fetched = kNoGroup;
        __syncthreads();  // (the slot is next written a whole group later)
        grp = (unsigned)__builtin_amdgcn_readfirstlane((int)*next_slot) + gridDim.x;
      } else {
        grp += gridDim.x;
      }
      fnext = grp << gsh;
    }
    const bool has_next = grp < ngroups;
    int s1v = s1;
    asm volatile("" : "+v"(s1v));

    // ---- stage 0: carrier mix into the padded image (iq_modem.c:60-61) ------------------------
    UC_IQ_PRIO(0);
#pragma unroll
    for (int u = 0; u < 17; u++) {
      const float x = cvt1<DTYPE>(xn[u]);
      lds_st(img, mix_idx(j + T * u), (UC_IQ_KNOCK & 1) ? mkv(x, x) : mkv(x * cs[u].x, x * cs[u].y));
    }
    UC_IQ_PRIO(2);
    if (has_next) load_frame(fnext);
    __syncthreads();  // B1: image complete; the previous frame's pruned-pass reads of the tile are done
    if (ring_n > 0 && (f & gmask) == 0) {  // a new group starts: drain the last one
      if (wave == 0) finalise(ring_f0, ring_n);
      ring_f0 = f;
      ring_n = 0;
    }

    // ---- stage 1: FIR (iq_modem.c:64-65), 16 consecutive outputs per thread --------------------
    // output o = 16 j + u needs mixed[o + 26 - k], k = 0..26: window w[d] = mixed[16 j + d], d < 42.
    // Two halves of eight outputs; within a half the eight accumulators advance tap by tap.
    {
      v2f accA[8], accB[8];
      v2f w[42];
#pragma unroll
      for (int d = 0; d < 34; d++) w[d] = lds_ld(img, 17 * j + d + (d >> 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; u++) accA[u] = pk_mul_slo(w[u + kHalo], taps[0]);
#pragma unroll
      for (int k = 1; k < kFirTapsDev; k += 2)  // taps k (high half of pair k/2) and k + 1 (low half of the next)
        pk_tap8x2(accA, &w[kHalo - k - 1], taps[k >> 1], taps[(k >> 1) + 1]);
#pragma unroll
      for (int d = 34; d < 42; d++) w[d] = lds_ld(img, 17 * j + d + (d >> 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; u++) accB[u] = pk_mul_slo(w[8 + u + kHalo], taps[0]);
#pragma unroll
      for (int k = 1; k < kFirTapsDev; k += 2)
        pk_tap8x2(accB, &w[8 + kHalo - k - 1], taps[k >> 1], taps[(k >> 1) + 1]);
      UC_IQ_PRIO(0);
#pragma unroll
      for (int u = 0; u < 16; u++) lds_st(tile, wr1 + (u ^ s1v), u < 8 ? accA[u & 7] : accB[u & 7]);
      UC_IQ_PRIO(2);
    }
    __syncthreads();  // B2: filtered frame in the tile; every window read of the image is done
    if (dyn && ((f + 2) & gmask) == 0 && j == 0) fetched = atomicAdd(p.work_ctr, 1u);  // (+ gridDim.x where it is read)

    // ---- stage 2: (I + jQ) * down_chirp * hann, FFT pass 1 -------------------------------
    // Base band: two dechirp runs (conj(up), conj(down)) over the SAME filtered frame, which is read from the tile
    // ONCE and stays in vr[]: the tile is free again behind B3.
    constexpr int kRuns = BB ? 2 : 1;
    v2f vr[16];
#pragma unroll
    for (int t = 0; t < 16; t++) vr[t] = lds_ld(tile, ((t & 1) ? rd1o : rd1e) + 128 * t);
    // Base band: the second run's chirp*hann entries (16 KiB per workgroup, out of L2) are requested HERE, a whole run
    // ahead of their use -- issued where run 1 needs them they cost their full cache latency on both waves of the
    // workgroup at once (measured: the second run took 3.6 x its own issue time).  They queue behind the frame
    // prefetch of stage 0, which is half a frame old by now.
    v2f c2[BB ? 16 : 1];
    if (BB) {
#pragma unroll
      for (int t = 0; t < 16; t++) c2[t] = buf_ld64(rs_ch2, voff8, T * 8 * t);
    }
#pragma unroll
    for (int run = 0; run < kRuns; run++) {
    v2f v[16];
    __builtin_amdgcn_sched_barrier(0);
    if (run == 0) {
#pragma unroll
      for (int t = 0; t < 16; t++) v[t] = pk_cmul(vr[t], ch[t]);
    } else {
#pragma unroll
      for (int t = 0; t < 16; t++) v[t] = pk_cmul(vr[t], c2[BB ? t : 0]);
    }
    pk_dft16(v, K, H);
    // (run 1: every pass-2 read of the image area by run 0 sits in front of run 0's B4)
    UC_IQ_PRIO(0);
#pragma unroll
    for (int t = 0; t < 16; t++) lds_st(img, wr1 + (t ^ s1v), v[pk_slot16(t)]);
    UC_IQ_PRIO(2);
    __syncthreads();  // B3: every read of the tile (vr; run 0's pruned pass) is done

    // ---- FFT pass 2 -----------------------------------------------------------------------
    float* dst2 = tile;
#pragma unroll
    for (int t = 0; t < 16; t++) v[t] = lds_ld(img, ((t & 1) ? rd1o : rd1e) + 128 * t);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 1; t < 16; t++) v[t] = pk_cmul(v[t], lds_ld(tw2l, tw2o + 16 * t));
    pk_dft16(v, K, H);
    UC_IQ_PRIO(0);
#pragma unroll
    for (int t = 0; t < 16; t++) lds_st(dst2, wr2 + 16 * t, v[pk_slot16(t)]);
    UC_IQ_PRIO(2);
    __syncthreads();  // B4
    UC_IQ_PRIO(3);  // the pruned pass and the window search end the run: first in line

    if (BB == 2) {
      // ---- FFT pass 3, pruned to ONE bin per thread; this wave's window maximum and its first attaining bin --------
      v2f a[8];
      const int b = kk2 & 255;
#pragma unroll
      for (int t = 0; t < 8; t++) a[t] = lds_ld(dst2, b + 256 * t);
      __builtin_amdgcn_sched_barrier(0);
      const v2f w1 = t3w1[0], w2 = t3w2[0];
      v2f e_ = pk_cfma(a[6], w2, a[4]), o_ = pk_cfma(a[7], w2, a[5]);
      e_ = pk_cfma(e_, w2, a[2]); o_ = pk_cfma(o_, w2, a[3]);
      e_ = pk_cfma(e_, w2, a[0]); o_ = pk_cfma(o_, w2, a[1]);
      const v2f z = pk_cfma(o_, w1, e_);
      const float q = z.x * z.x + z.y * z.y;
      const bool valid = lane < bw2;
      const float c = valid ? q : -INFINITY;
      const float m = wave_max_f32(c);
      const unsigned long long hit = __ballot(valid && c == m);
      const unsigned long long nan = __ballot(q != q);
      if (lane == 0) {
        const int k = kk2 + (hit ? __ffsll((long long)hit) - 1 : 0);
        float* e = ring + ring_n * kRingStBb + 5 * wave + 10 * run;
        // wave 0 reports the left window, wave 1 the right one; the other half of the entry loses every merge
        e[0] = wave == 0 ? m : -INFINITY;
        e[1] = __int_as_float(wave == 0 ? k : 0x7fffffff);
        e[2] = wave == 0 ? -INFINITY : m;
        e[3] = __int_as_float(wave == 0 ? 0x7fffffff : k);
        e[4] = __int_as_float((int)(nan & 1ull) << wave);  // first element of this wave's window is NaN
      }
    } else {
    // ---- FFT pass 3, pruned: bins k = lo + j and k = lo + 128 + j (< lo + bw4) ------------
    float q[2] = {0.f, 0.f};
    {
      v2f a[2][8];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int b = (lo + T * r + j) & 255;
#pragma unroll
        for (int t = 0; t < 8; t++) a[r][t] = lds_ld(dst2, b + 256 * t);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const v2f w1 = t3w1[r], w2 = t3w2[r];
        v2f e = pk_cfma(a[r][6], w2, a[r][4]), o = pk_cfma(a[r][7], w2, a[r][5]);
        e = pk_cfma(e, w2, a[r][2]); o = pk_cfma(o, w2, a[r][3]);
        e = pk_cfma(e, w2, a[r][0]); o = pk_cfma(o, w2, a[r][1]);
        const v2f z = pk_cfma(o, w1, e);
        q[r] = z.x * z.x + z.y * z.y;  // |Z[k]|^2, the square root is taken for the winners only
      }
    }

    // ---- the three arm_max_f32 (first maximum wins): this wave's partials ----------------------
    {
      const int k0 = lo + j, k1 = lo + T + j;
      const float q1v = (k1 < lo + bw4) ? q[1] : -INFINITY;
      float vl, vr;
      int kl, kr;
      window_first_max(q[0], k0, q1v, k1, lo, lo + bw2, vl, kl);           // left  [lo, lo + bw2)
      window_first_max(q[0], k0, q1v, k1, center, center + bw2, vr, kr);   // right [center, center + bw2)
      // first elements (a NaN there sticks): bin lo is thread 0 / round 0; bin `center` wherever it sits
      const unsigned long long nl = __ballot(k0 == lo && q[0] != q[0]);
      const unsigned long long nr = __ballot((k0 == center && q[0] != q[0]) || (k1 == center && q1v != q1v));
      if (lane == 0) {
        float* e = ring + ring_n * (BB ? kRingStBb : kRingSt) + 5 * wave + 10 * run;
        e[0] = vl;
        e[1] = __int_as_float(kl);
        e[2] = vr;
        e[3] = __int_as_float(kr);
        e[4] = __int_as_float((nl ? 1 : 0) | (nr ? 2 : 0));
      }
    }
    }
    UC_IQ_PRIO(2);
    }  // run
    ring_n++;
    if (!has_next) break;
    f = fnext;
  }
  __syncthreads();
  if (ring_n > 0 && wave == 0) finalise(ring_f0, ring_n);
  if (dyn && j == 0) handout_leave(p.work_ctr);  // the last workgroup out leaves the counter at zero for the next launch
  UC_CLOCK_END(p.debug, 2);
}


// ---------------------------------------------------------------------------------------------
// n = 1024 (the README's intention for this experiment, README.md:64-68; BASELINE config 3):
// ONE WAVE per frame -- 64 threads x 16 samples, 1024 = 16 x 8 x 8.  A single-wave workgroup
// needs no s_barrier (LDS operations of one wave complete in order), so the whole pipeline
// (mix, FIR, three FFT passes, windows) runs without a single cross-wave wait.
//
// The loop touches HBM only for the frames themselves: the next frame's 1050 samples are
// prefetched into 17 registers a whole frame time ahead, and every table is resident (carrier
// 17 entries, chirp*hann 16, pass-2 and pass-3 twiddles) -- vector loads return in order, so a
// table load behind the prefetch would put the HBM latency into the arithmetic.  The FIR keeps
// eight independent accumulators in flight (a dependent v_pk_fma_f32 chain costs a wait state per
// tap), and the history record is finalised for 64 frames at once (ring of window partials in
// LDS, one lane per frame: square roots, idx2freq, snr, coalesced 32-byte stores).
// ---------------------------------------------------------------------------------------------
constexpr int kN1 = 1024;
constexpr int T1 = 64;
constexpr int kMixLen1 = kN1 + kHalo;                     // 1050
constexpr int kImg1 = T1 * 17 + ((T1 * 17) >> 4) + 4;     // 1160: every m = j + 64 u, u < 17, has a slot
// MFMA variant of the FIR: the image is padded by TWO slots per 16 samples, so that the B-operand reads of a
// 16 x 16 x 4 tile (lane (k, column): sample 16 block + 4 s + k of 16 blocks) are conflict-free ds_read_b64
constexpr int kImg1M = T1 * 17 + 2 * ((T1 * 17) >> 4) + 4;  // 1228
constexpr int kFirKSteps = 11;                            // 16 outputs need 42 inputs: 11 k-steps of 4
constexpr int kRing1 = 64;                                // frames per finaliser drain
constexpr int kRingStride1 = 5;                           // ql, il, qr, ir, flags (odd: conflict-free)
constexpr int kRingStride1Bb = 11;                        // base band: 2 runs x 5 + 1 pad
constexpr int kRingOff1 = 2 * kImg1;
constexpr int kLdsFloats1 = kRingOff1 + kRing1 * kRingStride1;
constexpr int kTab1Off = kRingOff1 + kRing1 * kRingStride1Bb;   // base band: the second chirp*hann table (8 KiB)
constexpr int kLdsFloats1Bb = kTab1Off + 2 * kN1;
static_assert(kImg1 >= kN1, "the FFT tile aliases the mixed image");
static_assert((kTab1Off & 1) == 0, "complex alignment");
constexpr int kImgGrow1 = 2 * (kImg1M - kImg1);           // floats the MFMA variant's image is longer by

typedef float v4acc __attribute__((ext_vector_type(4)));

// FIRM = 1: the 27-tap FIR on the MATRIX pipe (v_mfma_f32_16x16x4_f32, exact f32: a k-ordered fmaf chain), beside
// the transform's packed-VALU work of the SIMD's other waves: outputs 16 b + i of 16 blocks b = T^ x S, T the
// 16 x 44 Toeplitz matrix of the taps (rows i, columns = the 42 samples the 16 outputs see, zero padded to 44),
// S the mixed samples.  11 k-steps x (I, Q) x 4 tile pairs = 88 MFMAs per frame instead of 432 packed FMAs per lane.
// BB: 0 = the firmware's windows, 1 = UC_FLAG_IQ_BASEBAND, 2 = base band with windows of at most 32 bins each (BASELINE
// configs[2]: 30): lanes 0-31 own the left window's bins, lanes 32-63 the right window's, so ONE pruned round serves
// both windows and one half-wave DPP reduction per dechirp run finds both maxima.
template <int DTYPE, int BB, int FIRM>
__global__ __launch_bounds__(T1, 2) void iq1024_kernel(const IqParams p) {
  UC_CLOCK_BEGIN();  // diagnostic build only (uc_dev.hpp)
  constexpr int kGrow = FIRM ? kImgGrow1 : 0;
  __shared__ __attribute__((aligned(16))) float lds[(BB ? kLdsFloats1Bb : kLdsFloats1) + kGrow];
  float* ring = lds + kRingOff1 + kGrow;
  float* tab1l = lds + (BB ? kTab1Off + kGrow : 0);
  const int j = threadIdx.x;  // = lane

  // frames are dealt in groups of G = 2^gsh consecutive frames (32-bit bookkeeping: the host rejects batches of 2^31
  // frames or more; the frame ADDRESS is 64-bit).  Static deal: round robin over the workgroups.
  // Dynamic hand-out (p.work_ctr, G >= 2): the workgroups do not run at the same speed, so after its first group a
  // workgroup takes the next free one from an atomic counter.  The request goes out behind the FIR of a group's
  // second-to-last frame -- where the register file has room for the returning id -- and is read at the top of the
  // last frame, behind loads that are waited for there anyway.
  const unsigned nfr = (unsigned)p.n_frames;
  const unsigned gsh = (unsigned)__builtin_ctz(p.group), gmask = (1u << gsh) - 1u;
  const unsigned ngroups = (nfr + gmask) >> gsh;
  unsigned grp = blockIdx.x;
  const bool dyn = p.work_ctr != nullptr;
  if (grp >= ngroups) {  // (the host never launches more workgroups than groups)
    if (dyn && j == 0) handout_leave(p.work_ctr);
    return;
  }
  unsigned f = grp << gsh;
  unsigned fetched = kNoGroup;  // lane 0: what the atomic in flight returns; kNoGroup = none asked for (the ragged last group)

  const __amdgpu_buffer_rsrc_t rs_car = make_rsrc(p.carrier, kN1 * 8);
  const __amdgpu_buffer_rsrc_t rs_ch = make_rsrc(p.chirp_hann, kN1 * 8);
  const __amdgpu_buffer_rsrc_t rs_tw = make_rsrc(p.tw, kN1 * 8);  // exp(-2 pi i k / 1024)
  const v2f K = mkv(kCos8, kSin8), H = mkv(kSqrtHalfF, kSqrtHalfF);

  const int lo = (int)p.idx_left_zero, center = (int)p.center;
  const int bw2 = (int)p.bw2, bw4 = (int)p.bw4;

  // ---- resident tables ---------------------------------------------------------------------
  // carrier of mixed sample m = j + 64 u (frame index i = m - 26).  History was mixed with the
  // TAIL of the table, as the previous back-to-back block's samples were (iq_modem.c:60-61 with
  // the state carried in i_state / q_state).  m >= 1050 reads past the table: (0, 0).
  v2f cs[17];
#pragma unroll
  for (int u = 0; u < 17; u++) {
    const int i = j + T1 * u - kHalo;
    const int ci = i < 0 ? kN1 + i : i;
    cs[u] = buf_ld64(rs_car, ci * 8, 0);
  }
  v2f ch[16];  // chirp*hann of sample n = j + 64 t
#pragma unroll
  for (int t = 0; t < 16; t++) ch[t] = buf_ld64(rs_ch, j * 8, T1 * 8 * t);
  if (BB) {  // the second run's table lives in LDS (no registers left for it, and a load in the loop would queue
             // behind the frame prefetch): entry n at complex index n, read as n = j + 64 t
    const __amdgpu_buffer_rsrc_t rs_ch2 = make_rsrc(p.chirp_hann2, kN1 * 8);
#pragma unroll
    for (int t = 0; t < 16; t++) lds_st(tab1l, j + T1 * t, buf_ld64(rs_ch2, j * 8, T1 * 8 * t));
  }
  // pass-2 twiddles W_128^(t k) = W_1024^(8 t k), k = b & 15 = j & 15 for both butterflies b = j, j + 64
  v2f tw2[8];
#pragma unroll
  for (int t = 1; t < 8; t++) tw2[t] = buf_ld64(rs_tw, ((8 * t * (j & 15)) & (kN1 - 1)) * 8, 0);
  // pass-3 twiddles of bins k = lo + j and lo + 64 + j: W^k and W^2k (Horner form)
  // (BB == 2: one bin per lane -- lo + j on lanes 0-31, center + j - 32 on lanes 32-63)
  const int kk2 = (j < 32) ? lo + j : center + (j - 32);
  v2f t3w1[2], t3w2[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int k = (BB == 2) ? kk2 : lo + T1 * r + j;
    t3w1[r] = buf_ld64(rs_tw, (k & (kN1 - 1)) * 8, 0);
    t3w2[r] = buf_ld64(rs_tw, ((2 * k) & (kN1 - 1)) * 8, 0);
  }

  // the 27 real taps, two per SGPR pair
  v2f taps[14];
#pragma unroll
  for (int i = 0; i < 14; i++) taps[i] = mkv(p.fir[2 * i], 2 * i + 1 < kFirTapsDev ? p.fir[2 * i + 1] : 0.f);
  // MFMA variant: this lane's element of the Toeplitz A operand of every k-step, A[i = j & 15][k = j >> 4] =
  // fir[i + 26 - (4 s + k)] or 0 (host-built table, 11 x 64 floats); B-operand base (lane = (k, column)):
  // sample 16 b + 4 s + k of block b = 16 T + column sits at image index 18 b + 4 s + k + 2 (s >> 2)
  float fa[kFirKSteps];
  if (FIRM) {
    const __amdgpu_buffer_rsrc_t rs_fa = make_rsrc(p.fir_mfma, kFirKSteps * 64 * 4);
#pragma unroll
    for (int s = 0; s < kFirKSteps; s++) fa[s] = buf_ld32(rs_fa, j * 4, 64 * 4 * s);
  }
  const int fb_base = 18 * (j & 15) + (j >> 4);

  const int s1 = j & 15;
  const int wr1 = 16 * j;  // natural / exchange-1 layout: + (t ^ s1)
  int rdn[4];              // natural-order read n = j + 64 t: base for t & 3
#pragma unroll
  for (int m = 0; m < 4; m++) rdn[m] = (j & ~15) + ((j & 15) ^ ((j >> 4) + 4 * m));

  // the frame's 1050 samples (26 of history first) as raw words, one frame ahead
  float xn[17];
  auto load_frame = [&](size_t fr) {
    const __amdgpu_buffer_rsrc_t rx =
        make_rsrc(reinterpret_cast<const char*>(p.frames) + (fr * p.stride) * 4 - kHalo * 4, kMixLen1 * 4);
#pragma unroll
    for (int u = 0; u < 17; u++) xn[u] = buf_ld32_stream(rx, j * 4, T1 * 4 * u);  // past the end: 0
  };
  load_frame(f);

  // finaliser: lane L turns ring slot L into the history record of frame f0 + L
  // (experiments/iq_modulation/Src/main.c:283-303)
  auto finalise = [&](size_t f0, int count) {
    if (BB) {
      if (j < count) {
        const float* e = ring + j * kRingStride1Bb;
        float ql[2], qr[2];
        int il[2], ir[2];
#pragma unroll
        for (int run = 0; run < 2; run++) {
          const float* er = e + 5 * run;
          ql[run] = er[0]; qr[run] = er[2];
          il[run] = __float_as_int(er[1]); ir[run] = __float_as_int(er[3]);
          const int flags = __float_as_int(er[4]);
          if (flags & 1) { ql[run] = __int_as_float(0x7fc00000); il[run] = lo; }
          if (flags & 2) { qr[run] = __int_as_float(0x7fc00000); ir[run] = center; }
        }
        bb_finish(p, f0 + (size_t)j, ql[0], il[0], qr[0], ir[0], ql[1], il[1], qr[1], ir[1], (uint32_t)kN1);
      }
      return;
    }
    if (j < count) {
      const float* e = ring + j * kRingStride1;
      const size_t ff = f0 + (size_t)j;
      float ql = e[0], qr = e[2];
      int il = __float_as_int(e[1]), ir = __float_as_int(e[3]);
      const int flags = __float_as_int(e[4]);
      if (flags & 1) { ql = __int_as_float(0x7fc00000); il = lo; }
      if (flags & 2) { qr = __int_as_float(0x7fc00000); ir = center; }
      const float ml = sqrtf(ql), mr = sqrtf(qr);
      // full window [lo, lo + bw4) = left then right: the right part wins only if strictly greater
      float mx = ml;
      int ix = il;
      if (!(ml != ml) && mr > ml) { mx = mr; ix = ir; }
      if (p.stats) {
        const float mm = p.mag_mean ? p.mag_mean[2 * ff] : p.mag_mean_scalar;
        // idx2freq of this experiment: (uint32)(sampling_rate * idx / n), Src/main.c:112-114
        const float fsn = p.fs;
        float4 sa, sb;
        sa.x = mx; sa.y = ml; sa.z = mr;
        sa.w = __int_as_float((int)(unsigned)(fsn * (float)ix / (float)kN1));
        sb.x = __int_as_float((int)(unsigned)(fsn * (float)il / (float)kN1));
        sb.y = __int_as_float((int)(unsigned)(fsn * (float)ir / (float)kN1));
        sb.z = mm;
        sb.w = (mx - mm) / mm;
        float4* d = reinterpret_cast<float4*>(p.stats + ff);
        d[0] = sa;
        d[1] = sb;
      }
      if (p.symbols) p.symbols[ff] = (uint8_t)UC_SYM_NONE;
    }
  };

  unsigned ring_f0 = f;
  int ring_n = 0;

  if (FIRM) {
    // Experiment knob (p.stagger, default 0): delay the odd wave slots of a SIMD (HW_ID.WAVE_ID) at the start, so that
    // one wave's FIR (matrix pipe) would run beside its partner's transform (VALU).  Measured: no effect at any
    // delay -- an f32 MFMA and the partner's packed-f32 VALU do not execute concurrently on a SIMD at all
    // (profiles/r02_mfma_valu_probe.txt: MFMA wave 4.32 ms + VALU wave 0.88 ms alone, 5.20 ms together).
    const unsigned slot = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((4 - 1) << 11)) & 15u;  // HW_REG_HW_ID[3:0]
    if (slot & 1u) {
      for (unsigned w = 0; w < p.stagger; w++) __builtin_amdgcn_s_sleep(64);
    }
  }

  for (;;) {
    unsigned fnext = f + 1;
    if ((fnext & gmask) == 0 || fnext >= nfr) {
      if (dyn) {
        grp = (unsigned)__builtin_amdgcn_readfirstlane((int)fetched) + gridDim.x;
        fetched = kNoGroup;
      } else {
        grp += gridDim.x;
      }
      fnext = grp << gsh;
    }
    const bool has_next = grp < ngroups;
    int s1v = s1;
    asm volatile("" : "+v"(s1v));

    // ---- stage 0: carrier mix into the padded image (iq_modem.c:60-61) ------------------------
#pragma unroll
    for (int u = 0; u < 17; u++) {
      const float x = cvt1<DTYPE>(xn[u]);
      const int m = j + T1 * u;
      lds_st(lds, FIRM ? m + ((m >> 4) << 1) : mix_idx(m), (UC_IQ_KNOCK & 1) ? mkv(x, x) : mkv(x * cs[u].x, x * cs[u].y));
    }
    if (has_next) load_frame(fnext);
    __syncthreads();  // single wave: no s_barrier is emitted, only the LDS wait

    // ---- stage 1: FIR (iq_modem.c:64-65) ---------------------------------------------------------
    v2f accA[8], accB[8];   // the lane's 16 filtered (I, Q) points, in the order the exchange-1 stores below want them
    if (FIRM) {
      // matrix pipe: tile pair T = blocks 16 T .. 16 T + 15; lane (k = j >> 4, column = j & 15) feeds sample
      // 16 b + 4 s + k of its column's block as the B operand and receives outputs 16 b + 4 (j >> 4) + r, r < 4
      v4acc dI[4], dQ[4];
#pragma unroll
      for (int T = 0; T < 4; T++) {
        v2f smp[kFirKSteps];
#pragma unroll
        for (int sk = 0; sk < kFirKSteps; sk++) smp[sk] = lds_ld(lds, fb_base + 288 * T + 4 * sk + 2 * (sk >> 2));
        v4acc aI = {0.f, 0.f, 0.f, 0.f}, aQ = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sk = 0; sk < kFirKSteps; sk++) {
          aI = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sk], smp[sk].x, aI, 0, 0, 0);
          aQ = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sk], smp[sk].y, aQ, 0, 0, 0);
        }
        dI[T] = aI;
        dQ[T] = aQ;
      }
#pragma unroll
      for (int T = 0; T < 4; T++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const v2f pt = mkv(dI[T][r], dQ[T][r]);
          if (T < 2) accA[4 * T + r] = pt; else accB[4 * (T - 2) + r] = pt;
        }
      }
    } else {
      // output o = 16 j + u needs mixed[o + 26 - k], k = 0..26: window w[d] = mixed[16 j + d], d < 42.
      // Two halves of eight outputs; within a half the eight accumulators advance tap by tap.
      v2f w[42];
#pragma unroll
      for (int d = 0; d < 34; d++) w[d] = lds_ld(lds, 17 * j + d + (d >> 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; u++) accA[u] = pk_mul_slo(w[0 + u + kHalo], taps[0]);
#pragma unroll
      for (int k = 1; k < kFirTapsDev; k += 2)  // taps k (high half of pair k/2) and k + 1 (low half of the next)
        pk_tap8x2(accA, &w[0 + kHalo - k - 1], taps[k >> 1], taps[(k >> 1) + 1]);
#pragma unroll
      for (int d = 34; d < 42; d++) w[d] = lds_ld(lds, 17 * j + d + (d >> 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; u++) accB[u] = pk_mul_slo(w[8 + u + kHalo], taps[0]);
#pragma unroll
      for (int k = 1; k < kFirTapsDev; k += 2)  // taps k (high half of pair k/2) and k + 1 (low half of the next)
        pk_tap8x2(accB, &w[8 + kHalo - k - 1], taps[k >> 1], taps[(k >> 1) + 1]);
    }
    __syncthreads();
    if (dyn && ((f + 2) & gmask) == 0 && j == 0) fetched = atomicAdd(p.work_ctr, 1u);  // (+ gridDim.x where it is read)
    if (ring_n > 0 && (f & gmask) == 0) {  // a new group starts: drain the last one
      finalise(ring_f0, ring_n);
      ring_f0 = f;
      ring_n = 0;
    }
    // Exchange 1 (FIR order -> stride-64 order), ONCE per frame: both dechirp runs of the base-band mode start from the
    // same filtered samples, which stay in vr[] (the FIR's accumulators are dead from here on).
    if (FIRM) {
      // point u = 4 T + r of this lane is output o = 16 b + 4 (j >> 4) + r of block b = 16 T + (j & 15); the
      // exchange-1 layout puts o at (o & ~15) + ((o & 15) ^ (b & 15))
      const int ob = 16 * (j & 15), oq = 4 * (j >> 4), oc = s1v;
#pragma unroll
      for (int u = 0; u < 16; u++)
        lds_st(lds, 256 * (u >> 2) + ob + ((oq + (u & 3)) ^ oc), u < 8 ? accA[u & 7] : accB[u & 7]);
    } else {
#pragma unroll
      for (int u = 0; u < 16; u++) lds_st(lds, wr1 + (u ^ s1v), u < 8 ? accA[u & 7] : accB[u & 7]);
    }
    __syncthreads();
    v2f vr[16];
#pragma unroll
    for (int t = 0; t < 16; t++) vr[t] = lds_ld(lds, rdn[t & 3] + 64 * t);
    // Base band: two dechirp runs (conj(up), conj(down)) over the same filtered frame.
    constexpr int kRuns = BB ? 2 : 1;
    float qn[2] = {0.f, 0.f};  // BB == 2: |Z|^2 of this lane's bin in run 0 / run 1
#pragma unroll
    for (int run = 0; run < kRuns; run++) {
    // ---- pass 1: x chirp*hann, radix-16 (Ns = 1) -------------------------------------------
    v2f v[16];
    if (run == 0) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 16; t++) v[t] = pk_cmul(vr[t], ch[t]);
    } else {
      v2f c2[16];
#pragma unroll
      for (int t = 0; t < 16; t++) c2[t] = lds_ld(tab1l, j + T1 * t);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 16; t++) v[t] = pk_cmul(vr[t], c2[t]);
    }
    pk_dft16(v, K, H);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; t++) lds_st(lds, wr1 + (t ^ s1v), v[pk_slot16(t)]);
    __syncthreads();

    // ---- pass 2: radix-8 (Ns = 16), butterflies b = j and j + 64 ----------------------------
    v2f g[2][8];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int b = j + T1 * h;
      const int be = (b & ~15) + ((b & 15) ^ (b >> 4)), bo = (b & ~15) + ((b & 15) ^ (b >> 4) ^ 8);
#pragma unroll
      for (int t = 0; t < 8; t++) g[h][t] = lds_ld(lds, ((t & 1) ? bo : be) + 128 * t);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int t = 1; t < 8; t++) g[h][t] = pk_cmul(g[h][t], tw2[t]);
      // BB == 2: the pruned pass only reads columns 0 .. 31 and 96 .. 127 (bins within 32 of DC): outputs 0, 1, 6, 7
      if (BB == 2) pk_dft8_0167(g[h], H);
      else pk_dft8(g[h], H);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int b = j + T1 * h;
#pragma unroll
      for (int t = 0; t < 8; t++)
        if (BB != 2 || t < 2 || t >= 6) lds_st(lds, (b >> 4) * 128 + (b & 15) + 16 * t, g[h][pk_slot8(t)]);
    }
    __syncthreads();

    if (BB == 2) {
      // ---- pass 3 (radix-8, Ns = 128), pruned to ONE bin per lane: lanes 0-31 the left window, 32-63 the right ------
      v2f a[8];
      const int b = kk2 & 127;
#pragma unroll
      for (int t = 0; t < 8; t++) a[t] = lds_ld(lds, b + 128 * t);
      __builtin_amdgcn_sched_barrier(0);
      const v2f w1 = t3w1[0], w2 = t3w2[0];
      v2f e = pk_cfma(a[6], w2, a[4]), o = pk_cfma(a[7], w2, a[5]);
      e = pk_cfma(e, w2, a[2]); o = pk_cfma(o, w2, a[3]);
      e = pk_cfma(e, w2, a[0]); o = pk_cfma(o, w2, a[1]);
      const v2f z = pk_cfma(o, w1, e);
      qn[run] = z.x * z.x + z.y * z.y;
      __syncthreads();  // tile free for the next run / frame
    } else {
    // ---- pass 3 (radix-8, Ns = 128), pruned: bins k = lo + j, lo + 64 + j ---------------------
    float q[2] = {0.f, 0.f};
    {
      v2f a[2][8];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int b = (lo + T1 * r + j) & 127;
#pragma unroll
        for (int t = 0; t < 8; t++) a[r][t] = lds_ld(lds, b + 128 * t);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const v2f w1 = t3w1[r], w2 = t3w2[r];
        v2f e = pk_cfma(a[r][6], w2, a[r][4]), o = pk_cfma(a[r][7], w2, a[r][5]);
        e = pk_cfma(e, w2, a[r][2]); o = pk_cfma(o, w2, a[r][3]);
        e = pk_cfma(e, w2, a[r][0]); o = pk_cfma(o, w2, a[r][1]);
        const v2f z = pk_cfma(o, w1, e);
        q[r] = z.x * z.x + z.y * z.y;
      }
    }
    __syncthreads();  // tile free for the next run / frame

    // ---- windows (one wave: no merge step) -----------------------------------------------------
    {
      const int k0 = lo + j, k1 = lo + T1 + j;
      const float q1v = (k1 < lo + bw4) ? q[1] : -INFINITY;
      float ql, qr;
      int il, ir;
      window_first_max(q[0], k0, q1v, k1, lo, lo + bw2, ql, il);
      window_first_max(q[0], k0, q1v, k1, center, center + bw2, qr, ir);
      // first elements (a NaN there sticks): bin lo is lane 0 / round 0; bin `center` wherever it sits
      const unsigned long long nl = __ballot(k0 == lo && q[0] != q[0]);
      const unsigned long long nr = __ballot((k0 == center && q[0] != q[0]) || (k1 == center && q1v != q1v));
      if (j == 0) {
        float* e = ring + ring_n * (BB ? kRingStride1Bb : kRingStride1) + 5 * run;
        e[0] = ql;
        e[1] = __int_as_float(il);
        e[2] = qr;
        e[3] = __int_as_float(ir);
        e[4] = __int_as_float((nl ? 1 : 0) | (nr ? 2 : 0));
      }
    }
    }
    }  // run
    if (BB == 2) {
      // ---- windows of both runs: each half-wave holds one window's bins in ascending order.  Four row rotations and
      // one row broadcast leave the left maximum in lane 31 and the right one in lane 63 (as common_partial2 of
      // uc_band_kernel.hip); the first attaining bin of each comes from one ballot and a scalar bit scan.
      const bool valid = (j & 31) < bw2;
      float* e = ring + ring_n * kRingStride1Bb;
#pragma unroll
      for (int run = 0; run < 2; run++) {
        const float q = qn[run];
        const float c = valid ? q : -INFINITY;
        float m = c;
        asm("s_nop 1\n\t"
            "v_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"
            : "+v"(m));
        const float ml = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 31));
        const float mr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
        const unsigned long long hit = __ballot(valid && c == ((j < 32) ? ml : mr));
        const unsigned long long nan = __ballot(q != q);
        const unsigned hl = (unsigned)hit, hr = (unsigned)(hit >> 32);
        if (j == 0) {
          e[5 * run + 0] = ml;
          e[5 * run + 1] = __int_as_float(lo + (hl ? __ffs((int)hl) - 1 : 0));
          e[5 * run + 2] = mr;
          e[5 * run + 3] = __int_as_float(center + (hr ? __ffs((int)hr) - 1 : 0));
          e[5 * run + 4] = __int_as_float((int)((nan & 1ull) | ((nan >> 31) & 2ull)));  // first elements: lanes 0 and 32
        }
      }
    }
    ring_n++;
    if (!has_next) break;
    f = fnext;
  }
  __syncthreads();
  if (ring_n > 0) finalise(ring_f0, ring_n);
  if (dyn && j == 0) handout_leave(p.work_ctr);  // the last workgroup out leaves the counter at zero for the next launch
  UC_CLOCK_END(p.debug, 1);
}

}  // namespace

namespace {

template <typename F>
int occ(F kernel, int threads, int fallback) {
  int nb = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, 0);
  return (e != hipSuccess || nb <= 0) ? fallback : nb;
}

}  // namespace

UC_LAUNCH_BEGIN
#define UC_IQ_DISPATCH(CALL1024, CALL2048)                                  \
  do {                                                                      \
    const bool i32 = dtype == UC_DTYPE_I32;                                 \
    if (n == kN1 && mfma) {                                                 \
      if (bb) { if (i32) { CALL1024(UC_DTYPE_I32, 1, 1); } else { CALL1024(UC_DTYPE_F32, 1, 1); } }     \
      else    { if (i32) { CALL1024(UC_DTYPE_I32, 0, 1); } else { CALL1024(UC_DTYPE_F32, 0, 1); } }     \
    } else if (n == kN1 && bb && narrow) {                                  \
      if (i32) { CALL1024(UC_DTYPE_I32, 2, 0); } else { CALL1024(UC_DTYPE_F32, 2, 0); }                 \
    } else if (n == kN1) {                                                  \
      if (bb) { if (i32) { CALL1024(UC_DTYPE_I32, 1, 0); } else { CALL1024(UC_DTYPE_F32, 1, 0); } }     \
      else    { if (i32) { CALL1024(UC_DTYPE_I32, 0, 0); } else { CALL1024(UC_DTYPE_F32, 0, 0); } }     \
    } else if (bb && narrow2) {                                             \
      if (i32) { CALL2048(UC_DTYPE_I32, 2); } else { CALL2048(UC_DTYPE_F32, 2); }                       \
    } else {                                                                \
      if (bb) { if (i32) { CALL2048(UC_DTYPE_I32, 1); } else { CALL2048(UC_DTYPE_F32, 1); } }           \
      else    { if (i32) { CALL2048(UC_DTYPE_I32, 0); } else { CALL2048(UC_DTYPE_F32, 0); } }           \
    }                                                                       \
  } while (0)

int launch_iq(int dtype, const IqParams& p, int grid, hipStream_t stream, int n) {
  if (grid <= 0) return (int)hipSuccess;
  const bool bb = p.baseband != 0, mfma = p.fir_mfma != nullptr, narrow = p.bw2 <= 32, narrow2 = p.bw2 <= 64;
#define UC_L1024(D, B, M) hipLaunchKernelGGL((iq1024_kernel<D, B, M>), dim3((unsigned)grid), dim3((unsigned)T1), 0, stream, p)
#define UC_L2048(D, B) hipLaunchKernelGGL((iq_kernel<D, B>), dim3((unsigned)grid), dim3((unsigned)T), 0, stream, p)
  UC_IQ_DISPATCH(UC_L1024, UC_L2048);
#undef UC_L1024
#undef UC_L2048
  return (int)hipGetLastError();
}

int iq_max_blocks_per_cu(int dtype, int n, int baseband, int fir_mfma, int narrow_) {
  const bool bb = baseband != 0, mfma = fir_mfma != 0, narrow = narrow_ != 0, narrow2 = narrow_ != 0;
  int nb = 0;
#define UC_O1024(D, B, M) nb = occ(iq1024_kernel<D, B, M>, T1, 8)
#define UC_O2048(D, B) nb = occ(iq_kernel<D, B>, T, 4)
  UC_IQ_DISPATCH(UC_O1024, UC_O2048);
#undef UC_O1024
#undef UC_O2048
  return nb;
}

UC_LAUNCH_END

}  // namespace uc
