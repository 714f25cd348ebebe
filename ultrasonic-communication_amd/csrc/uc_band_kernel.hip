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
// Stockham: both radix-16 passes are register-resident and written in packed
// fp32 (uc_pk.hpp: one complex number = one VGPR pair, every butterfly add,
// rotation and twiddle product is 1-2 v_pk_* instructions), the two exchanges
// go through a 16 KiB LDS tile (XOR-swizzled so that every ds_write_b64 /
// ds_read_b64 lane group is conflict-free), and the last radix-8 pass is
// evaluated only for bins 0..bandwidth2 and their mirror images -- the only
// bins dsp() looks at.  RX_REAL: both real references ride in one complex FFT
// (re = x*up*hann, im = x*down*hann) and are separated by Hermitian symmetry
// inside the pruned last pass.  HBM traffic is the frame itself (8 KiB) plus
// one symbol byte; the window*chirp table (16 KiB) lives in VGPRs, twiddles too.
#include "uc_dev.hpp"
#include "uc_kernels.hpp"

#ifndef UC_BAND_KNOCK
// diagnostic builds only (tools/band_knock.sh; results WRONG by construction, timing only), default build of RX_REAL:
// 2 no pass-1 arithmetic, 4 no pass-2 arithmetic, 8 no pruned-pass arithmetic, 16 no exchange 1 (stores + reads),
// 32 no exchange-2 stores, 64 no pruned-pass reads, 128 no window search, 256 exchange 1 as 30 DPP row rotations (no B1, B2),
// 512 the lane -> sample map those rotations would need, 1024 the raw samples through LDS in front of pass 1, 2048 (with 1024) B4
// dropped (profiles/r04_dpp_exchange.txt)
#define UC_BAND_KNOCK 0
#endif
#ifndef UC_CPLX_RES3
#define UC_CPLX_RES3 1  // SYNC_CPLX at 3 waves/SIMD: tables kept resident (0, 1)
#endif
#ifndef UC_CPLX_NRES
#define UC_CPLX_NRES 14  // ... and how many of the 16 entries of that table
#endif

#ifndef UC_ROWS_CPOL
#define UC_ROWS_CPOL 0  // cache policy of the ROWS build's frame loads: every 256-sample segment of a block is read by up to 8
#endif                  // frames (the batch builds read every byte once and say so: UC_STREAM_CPOL = nt)

namespace uc {

namespace {

__device__ __forceinline__ float buf_ld32_rows(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, UC_ROWS_CPOL));
}

constexpr int T = kBandThreads;  // 128
// LDS: the 2048-point complex tile, then a ring of per-frame window partials that a
// 64-lane finaliser drains once per 64 frames.
constexpr int kRingFrames = 64;  // (32 in the 4-waves-per-SIMD build: groups of at most 32 frames there)
constexpr int kRingStride = 13;  // 2 waves x 6 words + 1 pad word (conflict-free lane-strided reads)
constexpr int kRingOff = 2 * kN;
// WAVES = 4 (8 workgroups per CU, 128 VGPRs): the 15 pass-2 twiddles of a thread do not fit the registers any more; the
// 16 x 16 table W_256^(t k) sits in LDS instead (2 KiB, read with broadcast inside the 16-lane rows), and the ring
// shrinks to 32 frames so that eight workgroups still fit the CU's 160 KiB
constexpr int ring_frames(int waves) { return waves >= 4 ? 32 : kRingFrames; }
constexpr int next_off(int waves) { return kRingOff + ring_frames(waves) * kRingStride; }  // one word: the next group
constexpr int tw2_off(int waves) { return (next_off(waves) + 2) & ~1; }                     // 256 complex, 8-byte aligned
constexpr int lds_floats(int waves) { return waves >= 4 ? tw2_off(waves) + 2 * 256 : next_off(waves) + 1; }
static_assert(lds_floats(4) * 4 * 8 <= 160 * 1024, "eight workgroups per CU");

constexpr float kCos16 = 0.98078528040323044913f;  // cos(pi/16)
constexpr float kSin16 = 0.19509032201612826785f;  // sin(pi/16)

// two adjacent complex values with one ds_write_b128 (cidx even)
__device__ __forceinline__ void lds_st2(float* lds, int cidx, v2f a, v2f b) {
  v4f q;
  q.x = a.x; q.y = a.y; q.z = b.x; q.w = b.y;
  *reinterpret_cast<v4f*>(lds + 2 * cidx) = q;
}

// idx2freq(): receiver/Src/main.c:154-160, 32-bit unsigned arithmetic as on the MCU
__device__ __forceinline__ int32_t idx2freq(uint32_t ifs, uint32_t idx) {
  if (idx < (uint32_t)kN / 2) return (int32_t)(ifs * idx / (uint32_t)kN);
  return (int32_t)((ifs * ((uint32_t)kN - idx) / (uint32_t)kN) * 0xFFFFFFFFu);
}

// This wave's share of the two arm_max_f32 of one history (receiver/Src/main.c:205-208).
// Candidates: slot 0 = bin k0 of every lane, slot 1 (HAS1R / HAS1L) = bin k1 = 128 + lane.
//   right window: bins [0, bw2)  -- ties resolve to the SMALLEST bin (ascending index)
//   left  window: bins [1, bw2]  -- index n-k ascending = bin descending: LARGEST bin wins
// vr*/vl* are the right-/left-window magnitudes of the bin (equal for RX_REAL).
struct Partial {
  float vr, vl;
  int kr, kl;
  unsigned nan_first;  // bit0: right window's first element (bin 0) is NaN, bit1: left's (bin bw2)
};

template <bool HAS1R, bool HAS1L>
__device__ __forceinline__ Partial window_partial(float vr0, float vl0, int k0, float vr1, float vl1, int k1,
                                                  int bw2) {
  const float ninf = -INFINITY;
  const float r0 = (k0 < bw2) ? vr0 : ninf;
  const float l0 = (k0 >= 1 && k0 <= bw2) ? vl0 : ninf;
  float r1 = ninf, l1 = ninf;
  if (HAS1R) r1 = (k1 < bw2) ? vr1 : ninf;
  if (HAS1L) l1 = (k1 <= bw2) ? vl1 : ninf;
  Partial o;
  // value first (v_max_f32 skips NaNs exactly as arm_max_f32's '<' update does) ...
  o.vr = wave_max_f32(HAS1R ? max_f32(r0, r1) : r0);
  o.vl = wave_max_f32(HAS1L ? max_f32(l0, l1) : l0);
  // ... then the winning bin: smallest for the right window, largest for the left
  int cr = (r0 == o.vr) ? k0 : 0x7fffffff;
  int cl = (l0 == o.vl) ? k0 : -1;
  if (HAS1R) {
    const int cr1 = (r1 == o.vr) ? k1 : 0x7fffffff;
    cr = cr1 < cr ? cr1 : cr;
  }
  if (HAS1L) {
    const int cl1 = (l1 == o.vl) ? k1 : -1;
    cl = cl1 > cl ? cl1 : cl;
  }
  o.kr = wave_min_u32(cr);
  o.kl = wave_max_i32(cl);
  if (o.kr == 0x7fffffff) o.kr = 0;
  if (o.kl < 0) o.kl = 0;
  // arm_max_f32 starts from src[0]: a NaN there never loses a '<' compare
  unsigned long long nr = __ballot(k0 == 0 && vr0 != vr0);
  unsigned long long nl = __ballot(k0 == bw2 && vl0 != vl0);
  if (HAS1L) nl |= __ballot(k1 == bw2 && vl1 != vl1);
  o.nan_first = (nr ? 1u : 0u) | (nl ? 2u : 0u);
  return o;
}

// WIDE build (bandwidth2 of 192 .. 319 bins: receivers below ~64 kHz for the 16-19 kHz band, receiver/Src/main.c:372-374
// derives `bandwidth` for any sampling rate): up to NS candidate slots per lane, k[s] = 0x3fffffff where a lane has none.
template <int NS>
__device__ __forceinline__ Partial window_partial_n(const float (&vr)[NS], const float (&vl)[NS], const int (&k)[NS],
                                                    int bw2) {
  const float ninf = -INFINITY;
  float r[NS], l[NS];
  float mr = ninf, ml = ninf;
#pragma unroll
  for (int s = 0; s < NS; s++) {
    r[s] = (k[s] < bw2) ? vr[s] : ninf;
    l[s] = (k[s] >= 1 && k[s] <= bw2) ? vl[s] : ninf;
    mr = max_f32(mr, r[s]);
    ml = max_f32(ml, l[s]);
  }
  Partial o;
  o.vr = wave_max_f32(mr);
  o.vl = wave_max_f32(ml);
  int cr = 0x7fffffff, cl = -1;
#pragma unroll
  for (int s = 0; s < NS; s++) {
    if (r[s] == o.vr && k[s] < cr) cr = k[s];
    if (l[s] == o.vl && k[s] > cl) cl = k[s];
  }
  o.kr = wave_min_u32(cr);
  o.kl = wave_max_i32(cl);
  if (o.kr == 0x7fffffff) o.kr = 0;
  if (o.kl < 0) o.kl = 0;
  bool nfr = false, nfl = false;
#pragma unroll
  for (int s = 0; s < NS; s++) {
    nfr = nfr || (k[s] == 0 && vr[s] != vr[s]);
    nfl = nfl || (k[s] == bw2 && vl[s] != vl[s]);
  }
  o.nan_first = (__ballot(nfr) ? 1u : 0u) | (__ballot(nfl) ? 2u : 0u);
  return o;
}

// RX_REAL: both windows of a history look at the SAME magnitudes; they differ only in
// bin 0 (right window only) and bin bw2 (left window only).  So one maximum over the
// common bins [1, bw2) with its smallest and its largest attaining bin serves both, and
// the two edge bins are folded in by the finaliser (3 wave reductions instead of 4).
struct Common {
  float m;
  int ks, kl;
};

// Both histories' common maxima with ONE reduction: v_permlane32_swap folds the upper half of the
// up candidates and the lower half of the down candidates across, so that after one v_max lanes 0-31
// carry the up search and lanes 32-63 the down search; four row rotations and one row broadcast then
// leave the up maximum in lane 31 and the down maximum in lane 63 (7 cross-lane steps instead of 12).
template <bool HAS1>
__device__ __forceinline__ void common_partial2(float a0, float b0, int k0, int base0, float a1, float b1, int k1, int bw2,
                                                Common& up, Common& dn) {
  const float ninf = -INFINITY;
  const bool in0 = (k0 >= 1 && k0 < bw2);
  const float ca0 = in0 ? a0 : ninf, cb0 = in0 ? b0 : ninf;
  float ca1 = ninf, cb1 = ninf;
  float ma = ca0, mb = cb0;
  if (HAS1) {
    const bool in1 = k1 < bw2;
    ca1 = in1 ? a1 : ninf;
    cb1 = in1 ? b1 : ninf;
    ma = max_f32(ca0, ca1);
    mb = max_f32(cb0, cb1);
  }
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ma), __float_as_uint(mb), false, false);
  float m = max_f32(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
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
  up.m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 31));
  dn.m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
  // the attaining bins from ballots and scalar bit scans (slot 0 holds the smaller bins)
  const int base = base0;
  const unsigned long long ua0 = __ballot(ca0 == up.m), ub0 = __ballot(cb0 == dn.m);
  unsigned long long ua1 = 0, ub1 = 0;
  if (HAS1) {
    ua1 = __ballot(ca1 == up.m);
    ub1 = __ballot(cb1 == dn.m);
  }
  up.ks = up.kl = dn.ks = dn.kl = 0;
  if (ua0) {
    up.ks = base + (__ffsll((long long)ua0) - 1);
    up.kl = base + (63 - __clzll((long long)ua0));
  }
  if (ub0) {
    dn.ks = base + (__ffsll((long long)ub0) - 1);
    dn.kl = base + (63 - __clzll((long long)ub0));
  }
  if (HAS1 && ua1) {
    if (!ua0) up.ks = 128 + (__ffsll((long long)ua1) - 1);
    up.kl = 128 + (63 - __clzll((long long)ua1));
  }
  if (HAS1 && ub1) {
    if (!ub0) dn.ks = 128 + (__ffsll((long long)ub1) - 1);
    dn.kl = 128 + (63 - __clzll((long long)ub1));
  }
}

// finaliser side of common_partial: merge the two waves' common maxima, then fold in the
// edge bins exactly as arm_max_f32 would meet them (first element wins ties, NaN sticks)
__device__ __forceinline__ void resolve_windows(float m0, int ks0, int kl0, float m1, int ks1, int kl1, float q_first,
                                                float q_last, int bw2, float& vr, int& kr, float& vl, int& kl) {
  const float a = (m0 != m0) ? -INFINITY : m0;
  const float b = (m1 != m1) ? -INFINITY : m1;
  float m;
  int ks, kg;
  if (b > a) { m = b; ks = ks1; kg = kl1; }
  else if (a > b) { m = a; ks = ks0; kg = kl0; }
  else { m = a; ks = ks0 < ks1 ? ks0 : ks1; kg = kl0 > kl1 ? kl0 : kl1; }
  // right window: bin 0 first, then ascending bins
  if (q_first != q_first || !(m > q_first)) { vr = q_first; kr = 0; } else { vr = m; kr = ks; }
  // left window: bin bw2 first, then descending bins
  if (q_last != q_last || !(m > q_last)) { vl = q_last; kl = bw2; } else { vl = m; kl = kg; }
}

struct Hist {
  float mag_max, mag_left, mag_right, snr;
  int32_t f, fl, fr;
};

// the tail of dsp(): receiver/Src/main.c:209-229
__device__ __forceinline__ Hist make_hist(float mr, int kr, float ml, int kl, float mm, uint32_t ifs,
                                          bool raw_idx) {
  Hist h;
  h.mag_left = ml;
  h.mag_right = mr;
  uint32_t idx_r = (uint32_t)kr, idx_l = (uint32_t)kN - (uint32_t)kl, idx;
  if (ml > mr) { h.mag_max = ml; idx = idx_l; } else { h.mag_max = mr; idx = idx_r; }
  if (raw_idx) {
    // chirp_compression_freq_domain/Src/main.c:152-156: raw indices, the left one as
    // bandwidth*8 - local index = the mirrored bin kl
    h.fr = kr;
    h.fl = kl;
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

// merge the two waves' partials of one window (ties: smallest / largest bin)
__device__ __forceinline__ void merge_window(float v0, int k0, float v1, int k1, bool prefer_small, bool nan_first,
                                             int first, float& v, int& k) {
  const float a = (v0 != v0) ? -INFINITY : v0;
  const float b = (v1 != v1) ? -INFINITY : v1;
  const bool pick1 = (b > a) || (b == a && (prefer_small ? (k1 < k0) : (k1 > k0)));
  v = pick1 ? b : a;
  k = pick1 ? k1 : k0;
  if (nan_first) {
    v = __int_as_float(0x7fc00000);
    k = first;
  }
}

// rotate a complex value S lanes to the right inside its row of 16 lanes (lane i of a row receives lane (i - S) & 15's)
template <int S>
__device__ __forceinline__ v2f row_ror(v2f a) {
  v2f r;
  // (old = the value itself: every lane of a rotation has a source, and the move then happens in place)
  r.x = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a.x), __float_as_int(a.x), 0x120 + S, 0xf, 0xf, false));
  r.y = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a.y), __float_as_int(a.y), 0x120 + S, 0xf, 0xf, false));
  return r;
}

__device__ __forceinline__ v2f ld_tw(const float2* tw, int idx) {
  const float2 w = tw[idx & (kN - 1)];
  return mkv(w.x, w.y);
}

// Diagnostic build only (-DUC_STAMPS): per-phase cycle sums of every wave go to
// p.debug (never read by the kernel, never part of an output).  See tools/phase_stamps.py.
#ifdef UC_STAMPS
#define UC_STAMP(k)                                               \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    acc_[k] += now_ - last_;                                      \
    last_ = now_;                                                 \
  } while (0)
#else
#define UC_STAMP(k) do { } while (0)
#endif

// WIDE: windows of up to 319 bins (three pruned-pass rounds, generic window search, 16-bit indices in the ring);
// built at 2 waves/SIMD only.  The default build (windows of up to 191 bins) is untouched by it.
// SPEC: the same build with the window bins' magnitudes stored as well (uc_window_spectrum; p.spectrum).  A separate
// instantiation, so that the throughput builds carry neither the branch nor the stores: band_kernel<.., false, false> is
// instruction for instruction what it was (tools/isa_hist.sh).  The WIDE build tests p.spectrum at run time.
// ROWS: the frames are the FIFO offsets a new block adds to a live receiver (uc_receive_streams[_next]; BandParams: `prev`,
// row_pitch ...): every frame is the tail of one block followed by the head of the next, two base addresses, the split a
// multiple of 256 samples -- which is a whole number of this kernel's 128-sample load instructions, so each load takes one
// of two buffer resources by a SCALAR select and nothing is ever copied together.  The m = 8 frame of a row's last block IS
// that block and stores it for the next call on its way through (p.save) -- or its m = 7 frame does, with the block's last 256
// samples in two registers more, where the switch cannot look at the m = 8 frame; one-block calls of live receivers WALK the
// offsets the switch can still look at (p.need: 3 or 5 of the 8 of an IDLE stream, main.c:447-453; round 6: what is passed over
// costs nothing -- the walk below).  A separate instantiation again: the batch builds carry none of it (their machine code is
// recorded: tests/golden/kernel_digests.json).
// FRAMES = 2 (OVERLAP): the batch addressing over frames that OVERLAP (stride < n: the reference's own FIFO reads, main.c:447-451,
// 256 or 512 samples apart): the same loads with the default cache policy instead of `nt` -- with `nt` a frame's bytes are
// fetched again by each of the up to 8 frames that share them (6384 instead of 1025 B of HBM traffic per frame at stride 256),
// and that traffic is power the clock does not get (profiles/r05_stride_power.txt: +18-23 % frames/s).  Nothing else differs.
enum { kFramesBatch = 0, kFramesRows = 1, kFramesOverlap = 2 };
// a frame's ring slot: the next free one -- ROWS: the frame's index in its group (the walk passes over units)
#define UC_RSLOT (ROWS ? (int)(f & gmask) : ring_n)
template <int MODE, int DTYPE, int WAVES, bool WIDE = false, bool SPEC = false, int FRAMES = kFramesBatch>
__global__ __launch_bounds__(T, WAVES) void band_kernel(const BandParams p) {
  constexpr bool ROWS = FRAMES == kFramesRows;
#ifdef UC_STAMPS
  unsigned long long acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = __builtin_readcyclecounter();
#endif
  UC_CLOCK_BEGIN();  // diagnostic build only (uc_dev.hpp)
  constexpr bool kTw2Lds = WAVES >= 4;
  constexpr int kNextOff = next_off(WAVES);
  __shared__ __attribute__((aligned(16))) float lds[lds_floats(WAVES)];
  float* tw2l = lds + tw2_off(WAVES);
  float* ring = lds + kRingOff;

  const int j = threadIdx.x;
  const int lane = j & 63;
  const int wave = __builtin_amdgcn_readfirstlane(j >> 6);  // scalar: the compiler cannot see that it is wave-uniform
  const int bw2 = (int)p.bw2;

  // Frames are dealt to workgroups in GROUPS of p.group consecutive frames: all resident workgroups
  // together sweep a window of a few hundred MiB through the batch instead of each walking its own
  // multi-MiB chunk of an 8+ GiB buffer (measured: the per-frame time grew with the batch size with
  // contiguous chunks), while the finaliser still owns consecutive frames (coalesced symbol stores).
  // Workgroup b starts with group b; every further group comes from an atomic counter (p.work_ctr):
  // the workgroups do NOT run at the same speed -- with one frame of prefetch per workgroup the ones whose
  // loads come back late are latency-bound (measured: the same share took 1.32 ms on the fastest and 2.20 ms
  // on the slowest workgroup, profiles/r02_v0_clock_skew.json) -- so a static deal ends with a long tail.
  // The id of the next group is fetched one group ahead and parked in LDS (no wait in the loop).
  // kModePair: the unit of work is a PAIR of frames (2u, 2u+1) riding in one complex transform
  constexpr bool kPair = MODE == kModePair;
  constexpr bool kReal = MODE != kModeCplx;  // real reference(s): Hermitian split in the pruned pass
  // (32-bit bookkeeping: the host rejects batches of 2^31 frames or more; the frame ADDRESS is 64-bit)
  // (UC_FLAG_NO_FRAME_PAIRS: p.unpaired -- every frame rides alone, its partner slot reads as zeros)
  const unsigned psh = (kPair && !p.unpaired) ? 1u : 0u;  // frames per unit = 1 << psh
  const unsigned nfr = (unsigned)((p.n_frames + psh) >> psh);
  const unsigned gsh = p.group_log2, gmask = (1u << gsh) - 1u;
  const unsigned ngroups = (nfr + gmask) >> gsh;
  unsigned grp = blockIdx.x;
  const bool dyn = p.work_ctr != nullptr;
  if (grp >= ngroups) {  // (the host never launches more workgroups than groups)
    if (dyn && j == 0) handout_leave(p.work_ctr);
    return;
  }
  unsigned f = grp << gsh;
  unsigned* next_slot = reinterpret_cast<unsigned*>(lds + kNextOff);
  // thread 0: the group id an in-flight atomic returns.  It is parked in LDS in the first frame of every group
  // (the frame after the one that issued it), once that frame's loads -- older than the atomic -- are consumed.
  unsigned fetched = 0;
  if (dyn && j == 0) fetched = atomicAdd(p.work_ctr, 1u);  // (+ gridDim.x where it is READ: nothing may wait for it here)

  // ---- per-thread constants, resident for the whole batch -----------------
  const __amdgpu_buffer_rsrc_t rs_tab0 = make_rsrc(p.tab0, kN * 8);
  const __amdgpu_buffer_rsrc_t rs_tab1 = make_rsrc(p.tab1, kN * 8);
#if UC_BAND_KNOCK & 512  // (pricing only: the lane -> sample map of the DPP exchange, lane = n1 + 16 n0 reads sample n0 + 8 n1)
  const int voff4 = 4 * ((j >> 4) + 8 * (j & 15)), voff8 = 2 * voff4;
#else
  const int voff8 = j * 8, voff4 = j * 4;
#endif
  v2f wt[16];  // RX_REAL only: window*chirp table entries of this thread's samples
  if (MODE == kModeRxReal) {
#pragma unroll
    for (int t = 0; t < 16; t++) wt[t] = buf_ld64(rs_tab0, voff8, T * 8 * t);
  }
  // CPLX: the two complex tables (up, down).  Table loads inside the loop are 32 KiB of cache traffic per
  // frame and sit behind the prefetch in the in-order vector-memory queue, so as many as the register budget
  // allows stay resident: both at 2 waves/SIMD, the first one at 3.
  constexpr int kCplxRes =
      (MODE == kModeCplx) ? (WAVES <= 2 ? 2 : (WAVES == 3 && DTYPE == UC_DTYPE_F32 ? UC_CPLX_RES3 : 0)) : 0;  // (int32: no room)
  v2f wc[2][16];
  if (kCplxRes >= 1) {
#pragma unroll
    for (int t = 0; t < 16; t++) wc[0][t] = buf_ld64(rs_tab0, voff8, T * 8 * t);
  }
  if (kCplxRes >= 2) {
#pragma unroll
    for (int t = 0; t < 16; t++) wc[1][t] = buf_ld64(rs_tab1, voff8, T * 8 * t);
  }
  v2f wr[8];   // PAIR only: the REAL window*chirp table, two samples per register pair
  if (kPair) {
#pragma unroll
    for (int m = 0; m < 8; m++)
      wr[m] = mkv(buf_ld32(rs_tab0, voff8, T * 8 * (2 * m)), buf_ld32(rs_tab0, voff8, T * 8 * (2 * m + 1)));
  }
  // pass-2 twiddles W_256^(t*k), k = j & 15: all 15 resident (at 3 waves/SIMD the
  // registers are there; the factored form wa[n2]*wb[n1] costs 9 more products per frame)
  v2f tw2[16];
  if (kTw2Lds) {
    for (int e = j; e < 256; e += T) lds_st(tw2l, e, ld_tw(p.tw, 8 * (e >> 4) * (e & 15)));  // entry 16 t + k
    __syncthreads();
  } else {
#pragma unroll
    for (int t = 1; t < 16; t++) tw2[t] = ld_tw(p.tw, 8 * t * (j & 15));
  }
  const int tw2o = j & 15;
  // pass-3 twiddles: W_2048^j and its square (the pruned pass is evaluated in Horner form)
  const v2f tw3_1 = ld_tw(p.tw, j), tw3_2 = ld_tw(p.tw, 2 * j);
  const v2f K = mkv(kCos8, kSin8), H = mkv(kSqrtHalfF, kSqrtHalfF);  // SGPR pairs
  // WIDE: W^i and W^2i of the bins i = 128 + j (both waves) and 256 + lane (wave 1) of rounds 1 and 2
  v2f tw3w[2][2];
  if (WIDE) {
    tw3w[0][0] = ld_tw(p.tw, 128 + j);
    tw3w[0][1] = ld_tw(p.tw, 2 * (128 + j));
    tw3w[1][0] = ld_tw(p.tw, 256 + lane);
    tw3w[1][1] = ld_tw(p.tw, 2 * (256 + lane));
  }

  // LDS addresses (complex units)
  // exchange 1: element o = 16 j + t lives at o ^ (((o >> 4) & 7) << 1): bit 0 untouched, so the
  // pairs (t, t+1) stay adjacent and go out as ds_write_b128 (8 instead of 16 stores per thread);
  // conflict-free for the 8-lane b128 write groups and for the 32-lane b64 read groups
  const int s1 = (j & 7) << 1;
  const int wr1 = 16 * j;                                        // + (t ^ s1), t even
  const int rd1 = j ^ (((j >> 4) & 7) << 1);                     // + 128 t
  const int wr2 = (j >> 4) * 256 + (j & 15);                     // + 16 t

  // the unit's 16 samples of this thread as raw words: two per register pair, or (PAIR) sample t of
  // frame 2u in the low and of frame 2u+1 in the high half of pair t
  constexpr int NX = kPair ? 16 : 8;
  v2f xp[NX];
  // SYNC_CPLX at 2 waves/SIMD: both runs of a frame read xp, so the next frame is prefetched into a second register set
  // at the START of the frame (run 0) instead of behind run 1's first pass: a whole frame time of latency hiding, as the
  // single-run modes have; the set is copied over at the end of the frame (8 packed moves).
  constexpr bool kDblX = (MODE == kModeCplx) && WAVES <= 2;
  v2f xq[kDblX ? NX : 1];
  // ROWS: which half of the state's newest-block store this launch reads (the other one it writes)
  const char* prev_rd = nullptr;
  char* prev_wr = nullptr;
  if (ROWS) {
    const unsigned par = p.parity ? (unsigned)__builtin_amdgcn_readfirstlane((int)*p.parity) & 1u : 0u;
    prev_rd = reinterpret_cast<const char*>(p.prev) + (size_t)par * p.prev_half * 4;
    prev_wr = reinterpret_cast<char*>(p.save_to) + (size_t)(par ^ 1u) * p.save_half * 4;
  }
  // ROWS: the walk visits only the units that are WANTED.  A group (<= 32 units here: the host caps it; whole rows when there
  // are need words) is described by two uniform words: rw_eval -- bit i: unit (group base + i) is transformed;
  // rw_dn -- bit r: row r of the group needs its DOWN statistics (SYNC_CPLX).  Without need words (recorded calls, chunks of
  // several blocks) every unit of the batch is transformed.  With them (one-block calls of a live state: row = stream) a
  // unit the switch cannot look at costs NOTHING: the successor of a unit is the next wanted bit of its group's word or
  // the first wanted bit of the next group's (the group's want bits in ONE scalar, fetched by scalar loads: no vector
  // memory wait), so every transformed frame asks for the samples of the next transformed frame -- of the same row or not --
  // a whole computed frame ahead; the ring slot of a frame is its index in the group and the finaliser writes zeros for
  // the slots nothing was stored to.  Masked launches are dealt statically (group g + gridDim.x follows g).
  unsigned rw_eval = 0xffffffffu, rw_dn = 0xfu;
  unsigned ring_eval = 0xffffffffu;  // rw_eval of the group whose entries the ring holds
  typedef const __attribute__((address_space(4))) unsigned int* need_ptr;  // (scalar loads: s_load_dword, lgkmcnt)
  const need_ptr need_c = (need_ptr)(unsigned long long)p.need;
  // the need words of the (up to four) rows of group g: requested here, combined by rows_masks() -- the caller puts work in between
  auto rows_fetch = [&](unsigned g, unsigned (&nw)[4]) {
    if constexpr (ROWS) {  // (the other builds capture nothing: their code is what it was)
    nw[0] = nw[1] = nw[2] = nw[3] = 0u;
    if (!p.need) return;
    const unsigned s0 = (g << gsh) >> 3, ns = nfr >> 3;
    nw[0] = need_c[s0];
    if (gsh > 3 && s0 + 1 < ns) nw[1] = need_c[s0 + 1];
    if (gsh > 4 && s0 + 2 < ns) nw[2] = need_c[s0 + 2];
    if (gsh > 4 && s0 + 3 < ns) nw[3] = need_c[s0 + 3];
    }
  };
  // the units of the group that starts at unit `base` (a multiple of the group size)
  auto rows_valid = [&](unsigned base) -> unsigned {
    const unsigned cnt = nfr - base;  // units from the group's first one to the end of the batch (>= 1)
    const unsigned gm = gsh >= 5u ? 0xffffffffu : (1u << (1u << gsh)) - 1u;
    return cnt >= 32u ? gm : (gm & ((1u << cnt) - 1u));
  };
  auto rows_masks = [&](unsigned g, const unsigned (&nw)[4], unsigned& ev, unsigned& dn) {
    if constexpr (ROWS) {
    const unsigned base = g << gsh;
    const unsigned valid = rows_valid(base);
    if (!p.need) {
      ev = valid; dn = 0xfu;
      return;
    }
    unsigned e = (nw[0] & 0xffu) | ((nw[1] & 0xffu) << 8) | ((nw[2] & 0xffu) << 16) | ((nw[3] & 0xffu) << 24);
    // The block a row hands to the state (p.save) rides on the row's m = 8 frame, or on its m = 7 frame plus two loads
    // (below): every need word the replay writes holds one of the two (uc_rx_kernel.hip: need_word -- the acquisition set of
    // either turn reaches m = 7 or 8); a word that held neither would get its m = 8 frame transformed for nothing.
    if (p.save) e |= (~(e | (e << 1)) & 0x80808080u);
    ev = e & valid;
    dn = ((nw[0] >> 8) & 1u) | (((nw[1] >> 8) & 1u) << 1) | (((nw[2] >> 8) & 1u) << 2) | (((nw[3] >> 8) & 1u) << 3);
    }
  };
  v2f xs = mkv(0.f, 0.f);  // ROWS: the last 256 samples of a block its m = 7 frame hands to the state (below)
  auto load_unit = [&](size_t u, v2f (&xp)[NX]) {
    if (kPair) {
      const size_t fa = u << psh;
      const bool has_b = psh && fa + 1 < p.n_frames;  // a ragged last pair (or no pairing): frame b reads as zeros
      const __amdgpu_buffer_rsrc_t ra = make_rsrc(reinterpret_cast<const char*>(p.frames) + fa * p.stride * 4, kN * 4);
      const __amdgpu_buffer_rsrc_t rb = make_rsrc(
          reinterpret_cast<const char*>(p.frames) + (fa + (has_b ? 1 : 0)) * p.stride * 4, has_b ? kN * 4 : 0);
#pragma unroll
      for (int t = 0; t < NX; t++) xp[t] = mkv(buf_ld32_stream(ra, voff4, T * 4 * t), buf_ld32_stream(rb, voff4, T * 4 * t));
    } else if (ROWS) {
      // unit u = (row s, block jb, m): samples [256 m, n) of the block in front of block jb, then [0, 256 m) of block jb
      const unsigned g8 = (unsigned)u >> 3;
      const unsigned m8 = ((unsigned)u & 7u) + 1u;
      const unsigned s = (__umulhi(g8, p.div_magic) + g8) >> p.div_shift;
      const unsigned jb = g8 - s * p.row_blocks;
      const char* blk = reinterpret_cast<const char*>(p.frames) + ((size_t)s * p.row_pitch + (size_t)jb * kN) * 4;
      const char* before = jb ? blk - kN * 4 : prev_rd + (size_t)s * p.prev_pitch * 4;
      // sample 0 of the frame as if the whole frame lay in the one block / in the other; load t covers samples
      // [128 t, 128 t + 128): the first 16 - 2 m loads belong to the block in front
      if (jb != 0) {
        // behind the row's first block the block in front is the row's previous one: the frame lies in memory as one piece
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(before + 1024 * m8, kN * 4);
#pragma unroll
        for (int m = 0; m < 8; m++)
          xp[m] = mkv(buf_ld32_rows(rx, voff4, T * 4 * (2 * m)), buf_ld32_rows(rx, voff4, T * 4 * (2 * m + 1)));
      } else {
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(before + 1024 * m8, kN * 4);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(blk - (kN * 4 - 1024 * (int)m8), kN * 4);
        const int tsplit = 16 - 2 * (int)m8;
        // (one-block calls) the m = 7 frame of a row whose m = 8 frame is passed over hands the row's block to the state: the
        // block's last 256 samples come along (rw_eval describes u's group by now)
        if (p.need && p.save && m8 == 7u && !((rw_eval >> (((unsigned)u & gmask) + 1u)) & 1u)) {
          const __amdgpu_buffer_rsrc_t rl = make_rsrc(blk, kN * 4);
          xs = mkv(buf_ld32_rows(rl, voff4, T * 4 * 14), buf_ld32_rows(rl, voff4, T * 4 * 15));
        }
#pragma unroll
        for (int m = 0; m < 8; m++)
          xp[m] = mkv(buf_ld32_rows(2 * m < tsplit ? ra : rb, voff4, T * 4 * (2 * m)),
                      buf_ld32_rows(2 * m + 1 < tsplit ? ra : rb, voff4, T * 4 * (2 * m + 1)));
      }
    } else {
      const __amdgpu_buffer_rsrc_t rx = make_rsrc(reinterpret_cast<const char*>(p.frames) + u * p.stride * 4, kN * 4);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        if (FRAMES == kFramesOverlap)
          xp[m] = mkv(buf_ld32_rows(rx, voff4, T * 4 * (2 * m)), buf_ld32_rows(rx, voff4, T * 4 * (2 * m + 1)));
        else
          xp[m] = mkv(buf_ld32_stream(rx, voff4, T * 4 * (2 * m)), buf_ld32_stream(rx, voff4, T * 4 * (2 * m + 1)));
      }
    }
  };
  if constexpr (ROWS) {
    // the first wanted unit of this workgroup's first group (a group nothing is wanted of still gets its zero records: its
    // first unit is then transformed for nothing -- no need word the replay writes is empty)
    unsigned nw[4];
    rows_fetch(grp, nw);
    rows_masks(grp, nw, rw_eval, rw_dn);
    if (!rw_eval) rw_eval = 1u;
    const unsigned want = rw_eval;
    f += (unsigned)__builtin_ctz(want);
  }
  load_unit(f, xp);

  // ROWS: the same finaliser for the launches of uc_receive_streams[_next], which take (up, down) mag_max and nothing else
  // (receiver/Src/main.c:209-215: the larger of the two window maxima, the right one on a tie).  It runs once per group: its
  // lane, addresses and constants are derived when it runs -- hoisted out of the frame loop they would sit in registers the
  // 168-register build does not have.
  auto rows_finalise = [&](size_t f0, int count, unsigned evalmask) {
    if constexpr (ROWS) {
      int ln = lane;
      unsigned poison = p.poison;
      asm volatile("" : "+v"(ln), "+s"(poison));
      if (ln >= count) return;
      float2 out;
      if (!((evalmask >> ln) & 1u)) {
        // a slot nothing was stored to (a unit the switch cannot look at): a zero record -- or, under the tests' poison
        // switch, statistics no signal reaches
        out.x = out.y = poison ? 1e15f : 0.f;
      } else {
        const float* e = ring + ln * kRingStride;
        const float mscale = kReal ? 0.5f : 1.0f;  // the ring holds squared magnitudes (x4 for RX_REAL)
        float mr[2], ml[2];
        int kr, kl;
        if (WIDE) {
          const unsigned a0 = __float_as_uint(e[4]), a1 = __float_as_uint(e[6 + 4]);
          const unsigned b0 = __float_as_uint(e[5]), b1 = __float_as_uint(e[6 + 5]);
          merge_window(e[0], a0 & 0x7fff, e[6 + 0], a1 & 0x7fff, true, ((a0 | a1) >> 15) & 1u, 0, mr[0], kr);
          merge_window(e[1], (a0 >> 16) & 0x7fff, e[6 + 1], (a1 >> 16) & 0x7fff, false, ((a0 | a1) >> 31) & 1u, bw2, ml[0], kl);
          merge_window(e[2], b0 & 0x7fff, e[6 + 2], b1 & 0x7fff, true, ((b0 | b1) >> 15) & 1u, 0, mr[1], kr);
          merge_window(e[3], (b0 >> 16) & 0x7fff, e[6 + 3], (b1 >> 16) & 0x7fff, false, ((b0 | b1) >> 31) & 1u, bw2, ml[1], kl);
        } else if (kReal) {
          const unsigned kp0 = __float_as_uint(e[2]), kp1 = __float_as_uint(e[6 + 2]);
          resolve_windows(e[0], kp0 & 255, (kp0 >> 8) & 255, e[6 + 0], kp1 & 255, (kp1 >> 8) & 255, e[3], e[6 + 3], bw2,
                          mr[0], kr, ml[0], kl);
          resolve_windows(e[1], (kp0 >> 16) & 255, (kp0 >> 24) & 255, e[6 + 1], (kp1 >> 16) & 255, (kp1 >> 24) & 255, e[4],
                          e[6 + 4], bw2, mr[1], kr, ml[1], kl);
        } else {
          const unsigned kp0 = __float_as_uint(e[4]), kp1 = __float_as_uint(e[6 + 4]);
          const unsigned fl = __float_as_uint(e[5]) | __float_as_uint(e[6 + 5]);
          merge_window(e[0], kp0 & 255, e[6 + 0], kp1 & 255, true, fl & 1u, 0, mr[0], kr);
          merge_window(e[1], (kp0 >> 8) & 255, e[6 + 1], (kp1 >> 8) & 255, false, fl & 2u, bw2, ml[0], kl);
          merge_window(e[2], (kp0 >> 16) & 255, e[6 + 2], (kp1 >> 16) & 255, true, fl & 4u, 0, mr[1], kr);
          merge_window(e[3], (kp0 >> 24) & 255, e[6 + 3], (kp1 >> 24) & 255, false, fl & 8u, bw2, ml[1], kl);
        }
        float mx[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const float r = mscale * sqrtf(mr[h]), l = mscale * sqrtf(ml[h]);
          mx[h] = (l > r) ? l : r;  // make_hist: mag_max
        }
        out.x = mx[0];
        out.y = mx[1];
      }
      p.magmax[f0 + (size_t)ln] = out;
    }
  };
  // Finaliser, vectorised over frames: lane L turns ring slot L into history[0],
  // history[1] and the symbol of frame f0 + L (receiver/Src/main.c:209-229, 518-531).
  auto finalise = [&](size_t f0, int count) {
    if constexpr (ROWS) {
      rows_finalise(f0, count, ring_eval);
      return;
    }
    if (lane < count) {
      const float* e = ring + lane * kRingStride;
      const size_t ff = f0 + (size_t)lane;
      float mm_up = p.mag_mean_scalar, mm_dn = p.mag_mean_scalar;
      if (p.mag_mean) {
        // two floats per frame; the single-history pipeline of PAIR uses the first of each frame
        const size_t fa = ff << psh;
        mm_up = p.mag_mean[kPair ? 2 * fa : 2 * ff];
        mm_dn = p.mag_mean[kPair ? ((psh && fa + 1 < p.n_frames) ? 2 * fa + 2 : 2 * fa) : 2 * ff + 1];
      }
      float mr, ml;
      int kr, kl;
      // the ring holds squared magnitudes (x4 for RX_REAL): |X| = mscale * sqrt(q)
      const float mscale = kReal ? 0.5f : 1.0f;
      unsigned kp0 = 0, kp1 = 0, fl = 0;
      if (WIDE) {
        // per wave: [up right, up left, down right, down left, (kr, kl) of up, (kr, kl) of down], 16 bits per index
        kp0 = __float_as_uint(e[4]);
        kp1 = __float_as_uint(e[6 + 4]);
        merge_window(e[0], kp0 & 0x7fff, e[6 + 0], kp1 & 0x7fff, true, ((kp0 | kp1) >> 15) & 1u, 0, mr, kr);
        merge_window(e[1], (kp0 >> 16) & 0x7fff, e[6 + 1], (kp1 >> 16) & 0x7fff, false, ((kp0 | kp1) >> 31) & 1u, bw2, ml, kl);
      } else if (kReal) {
        // per wave: [M_up, M_dn, kpack, edge_up, edge_dn]; wave 0's edge = bin 0, wave 1's = bin bw2
        kp0 = __float_as_uint(e[2]);
        kp1 = __float_as_uint(e[6 + 2]);
        resolve_windows(e[0], kp0 & 255, (kp0 >> 8) & 255, e[6 + 0], kp1 & 255, (kp1 >> 8) & 255, e[3], e[6 + 3],
                        bw2, mr, kr, ml, kl);
      } else {
        kp0 = __float_as_uint(e[4]);
        kp1 = __float_as_uint(e[6 + 4]);
        fl = __float_as_uint(e[5]) | __float_as_uint(e[6 + 5]);
        merge_window(e[0], kp0 & 255, e[6 + 0], kp1 & 255, true, fl & 1u, 0, mr, kr);
        merge_window(e[1], (kp0 >> 8) & 255, e[6 + 1], (kp1 >> 8) & 255, false, fl & 2u, bw2, ml, kl);
      }
      mr = mscale * sqrtf(mr);
      ml = mscale * sqrtf(ml);
      const Hist h0 = make_hist(mr, kr, ml, kl, mm_up, p.ifs, kPair);
      {
        if (WIDE) {
          kp0 = __float_as_uint(e[5]);
          kp1 = __float_as_uint(e[6 + 5]);
          merge_window(e[2], kp0 & 0x7fff, e[6 + 2], kp1 & 0x7fff, true, ((kp0 | kp1) >> 15) & 1u, 0, mr, kr);
          merge_window(e[3], (kp0 >> 16) & 0x7fff, e[6 + 3], (kp1 >> 16) & 0x7fff, false, ((kp0 | kp1) >> 31) & 1u, bw2, ml, kl);
        } else if (kReal) {
          resolve_windows(e[1], (kp0 >> 16) & 255, (kp0 >> 24) & 255, e[6 + 1], (kp1 >> 16) & 255, (kp1 >> 24) & 255,
                          e[4], e[6 + 4], bw2, mr, kr, ml, kl);
        } else {
          merge_window(e[2], (kp0 >> 16) & 255, e[6 + 2], (kp1 >> 16) & 255, true, fl & 4u, 0, mr, kr);
          merge_window(e[3], (kp0 >> 24) & 255, e[6 + 3], (kp1 >> 24) & 255, false, fl & 8u, bw2, ml, kl);
        }
        mr = mscale * sqrtf(mr);
        ml = mscale * sqrtf(ml);
        const Hist h1 = make_hist(mr, kr, ml, kl, mm_dn, p.ifs, kPair);
        if (kPair) {
          // h0 = the only history of frame 2 ff, h1 = of frame 2 ff + 1 (raw bin indices,
          // chirp_compression_freq_domain/Src/main.c:152-156); no symbol in this variant
          const size_t fa = ff << psh;
          const bool has_b = psh && fa + 1 < p.n_frames;
          if (p.stats) {
            store_hist(p.stats + fa, h0, mm_up);
            if (has_b) store_hist(p.stats + fa + 1, h1, mm_dn);
          }
          if (p.symbols) {
            p.symbols[fa] = (uint8_t)UC_SYM_NONE;
            if (has_b) p.symbols[fa + 1] = (uint8_t)UC_SYM_NONE;
          }
        } else {
        if (p.stats) {
          store_hist(p.stats + 2 * ff, h0, mm_up);
          store_hist(p.stats + 2 * ff + 1, h1, mm_dn);
        }
        if (p.magmax) p.magmax[ff] = make_float2(h0.mag_max, h1.mag_max);
        if (p.symbols) {
          // receiver/Src/main.c:521-531
          uint8_t sym = (uint8_t)UC_SYM_NONE;
          if ((h0.snr >= p.snr_threshold) || (h1.snr >= p.snr_threshold))
            sym = (h1.snr > h0.snr) ? (uint8_t)UC_SYM_DOWN : (uint8_t)UC_SYM_UP;
          p.symbols[ff] = sym;
        }
        }
      }
    }
  };

  // uc_window_spectrum: the magnitudes pipeline() leaves in signal[] (receiver/Src/main.c:176-179), for the bins dsp()
  // looks at: bin i of unit f, run `run`; qa / qb = the squared magnitudes the window search sees (x 4 for real references).
  // Real reference(s): both sides of DC carry the same value (Hermitian mirror, Q1).
  auto store_spectrum = [&](unsigned f, int run, int i, float qa, float qb) {
    const size_t wb = 2 * (size_t)bw2 + 1;
    const float va = (kReal ? 0.5f : 1.0f) * sqrtf(qa);  // (the finaliser's own expression: bit-equal to the stats)
    const float vb = (kReal ? 0.5f : 1.0f) * sqrtf(qb);
    if (MODE == kModeCplx) {
      float* o = p.spectrum + ((size_t)f * 2 + run) * wb + bw2;
      o[i] = va;
      if (i) o[-i] = vb;
    } else if (kPair) {
      const size_t fa = (size_t)f << psh;
      float* oa = p.spectrum + fa * wb + bw2;
      oa[i] = va;
      oa[-i] = va;
      if (psh && fa + 1 < p.n_frames) {
        float* ob = oa + wb;
        ob[i] = vb;
        ob[-i] = vb;
      }
    } else {
      float* o = p.spectrum + (size_t)f * 2 * wb + bw2;
      o[i] = va;
      o[-i] = va;
      o[wb + i] = vb;
      o[wb - i] = vb;
    }
  };

  unsigned ring_f0 = ROWS ? 0xffffffffu : f;  // frame held by ring slot 0 (ROWS: the first unit of the group the ring serves)
  int ring_n = 0;        // slots filled (ROWS: the slots of the group in the ring, set when the group is entered)

  for (;;) {
    // successor of frame f in this workgroup's visiting order
    unsigned fnext = f + 1;
    // ROWS: the group fnext opens -- its need words are requested here and combined (rows_masks) where the loads of fnext
    // go out, a first radix-16 half later: nothing waits for them
    bool next_first = false, nx_pending = false;
    unsigned nx_nw[4] = {0u, 0u, 0u, 0u};
    // ... and the group f opens (the first unit this workgroup visits of it): the ring changes hands here; what it held is
    // finalised behind the frame's first barrier
    bool grp_first = false, skip_down = false, save7 = false;
    unsigned old_f0 = 0u, old_eval = 0u;
    int old_n = 0;
    if constexpr (ROWS) {
      const unsigned fbase = f & ~gmask, bit = f & gmask;
      grp_first = ring_f0 != fbase;
      // the row's block goes to the state from its m = 7 frame when its m = 8 frame is not transformed (one-block calls)
      save7 = p.need && p.save && (bit & 7u) == 6u && !((rw_eval >> (bit + 1u)) & 1u);
      // SYNC_CPLX live receivers: acquisition looks at the UP reference only (receiver/Src/main.c:447-451); a stream that is
      // IDLE when its block arrives cannot read a DOWN statistic of that block before the block has left the FIFO (it takes
      // three evaluations, five blocks, to reach SYNCHRONIZED), so the second transform of its new offsets is skipped -- bit 8
      // of the row's need word, written by the replay kernel of the previous call (one-block calls only)
      skip_down = MODE == kModeCplx && !((rw_dn >> (bit >> 3)) & 1u);
      const unsigned rem = rw_eval & ~((2u << bit) - 1u);
      if (grp_first) {
        old_f0 = ring_f0;
        old_eval = ring_eval;
        old_n = ring_n;
        ring_f0 = fbase;
        ring_eval = rw_eval;
        ring_n = (int)(nfr - fbase < (1u << gsh) ? nfr - fbase : (1u << gsh));
      }
      if (rem) {
        fnext = fbase + (unsigned)__builtin_ctz(rem);
      } else {
        next_first = true;
        if (dyn) {
          if (grp_first) {  // (a group of one unit: the id parked in the very frame that reads it)
            if (j == 0) *next_slot = fetched;
            __syncthreads();
          }
          grp = (unsigned)__builtin_amdgcn_readfirstlane((int)*next_slot) + gridDim.x;
        } else {
          grp += gridDim.x;
        }
        fnext = grp << gsh;  // (the group's first WANTED unit once its masks are known: resolve_next)
        if (grp < ngroups) {
          rows_fetch(grp, nx_nw);
          nx_pending = true;
        }
      }
    } else {
    if ((fnext & gmask) == 0 || fnext >= nfr) {
      if (dyn) {
        // normally the slot was written several barriers ago; only a one-frame group (the ragged end of the
        // batch) gets here in the very frame that should park it
        if ((f & gmask) == 0) {
          if (j == 0) *next_slot = fetched;
          __syncthreads();
        }
        grp = (unsigned)__builtin_amdgcn_readfirstlane((int)*next_slot) + gridDim.x;
      } else {
        grp += gridDim.x;
      }
      fnext = grp << gsh;
    }
    }
    const bool has_next = grp < ngroups;
    // (from here on rw_eval / rw_dn describe the group of fnext: everything this frame needs of its own group is in the
    // flags above)
    auto resolve_next = [&]() {
      if constexpr (ROWS) {
        if (nx_pending) {
          rows_masks(grp, nx_nw, rw_eval, rw_dn);
          if (!rw_eval) rw_eval = 1u;  // (nothing wanted of a whole group: its first unit is transformed for nothing)
          fnext = (grp << gsh) + (unsigned)__builtin_ctz(rw_eval);
          nx_pending = false;
        }
      }
    };
    // Opaque re-definitions: stop LICM from hoisting the 16 swizzled store
    // addresses and the derived pass-3 twiddles out of the frame loop (they
    // would sit in registers for the whole batch).
    int s1v = s1;
    v2f t3a = tw3_1, t3b = tw3_2;
    if (kTw2Lds) {
      // (128-register build: W^2j is squared from W^j every frame instead of living in two more registers;
      // one rounding more than the table value, ~1e-7 relative on the twiddle)
      asm volatile("" : "+v"(s1v), "+v"(t3a));
      t3b = pk_cmul(t3a, t3a);
    } else {
      asm volatile("" : "+v"(s1v), "+v"(t3a), "+v"(t3b));
    }
    constexpr int kRuns = (MODE == kModeCplx) ? 2 : 1;
    float pv[4] = {0.f, 0.f, 0.f, 0.f};  // this wave's partials: up right/left, down right/left
    unsigned kpack = 0, flags = 0;
    unsigned kpw[2] = {0, 0};            // WIDE: (kr, kl) of a run as 2 x (15 bits + first-element-NaN flag)
    UC_STAMP(9);
    if (ROWS) {
      // the frame m = 8 of a row's last block IS that block: what the next call's first offsets still read of this call
      // goes to the state's other half straight from the registers (16 coalesced word stores on 1 frame in 8 row_blocks)
      if (p.save && (f & 7u) == 7u) {
        const unsigned g8 = f >> 3;
        const unsigned s = (__umulhi(g8, p.div_magic) + g8) >> p.div_shift;
        if (g8 - s * p.row_blocks == p.row_blocks - 1u) {
          const __amdgpu_buffer_rsrc_t rw = make_rsrc(prev_wr + (size_t)s * kN * 4, kN * 4);
#pragma unroll
          for (int m = 0; m < 8; m++) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xp[m].x), rw, voff4, T * 4 * (2 * m), 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xp[m].y), rw, voff4, T * 4 * (2 * m + 1), 0);
          }
        }
      }
      // ... and where the switch cannot look at that frame (one-block calls of a live state), the m = 7 frame hands the block
      // over: its last 14 loads ARE the block's first 1792 samples, the last 256 came with its loads in two registers more
      if (save7) {
        const unsigned s = f >> 3;  // (one-block calls: row = stream)
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(prev_wr + (size_t)s * kN * 4, kN * 4);
#pragma unroll
        for (int m = 1; m < 8; m++) {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xp[m].x), rw, voff4, T * 4 * (2 * m - 2), 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xp[m].y), rw, voff4, T * 4 * (2 * m - 1), 0);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xs.x), rw, voff4, T * 4 * 14, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xs.y), rw, voff4, T * 4 * 15, 0);
      }
    }
#pragma unroll
    for (int run = 0; run < kRuns; run++) {
      if (ROWS && MODE == kModeCplx && run == 1 && skip_down) {  // (uniform over the workgroup: no barrier is left behind)
        float* e = ring + UC_RSLOT * kRingStride + wave * 6;
        if (lane == 0) {
          e[0] = pv[0]; e[1] = pv[1]; e[2] = p.poison ? 1e30f : 0.f; e[3] = e[2];
          e[4] = __uint_as_float(WIDE ? kpw[0] : kpack);
          e[5] = __uint_as_float(WIDE ? 0u : flags);
        }
        continue;
      }
      v2f v[16];
      // ---- pass 1: window*chirp multiply, radix-16, Ns = 1 ------------------
#if UC_BAND_KNOCK & 1024  // (pricing only: the raw samples through LDS in front of pass 1 -- 4 b128 stores, a barrier, 4 b128 loads)
      if (MODE == kModeRxReal) {
        __syncthreads();
        v4f* st4 = reinterpret_cast<v4f*>(lds);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          v4f w4;
          w4.x = xp[2 * q].x; w4.y = xp[2 * q].y; w4.z = xp[2 * q + 1].x; w4.w = xp[2 * q + 1].y;
          st4[j + T * q] = w4;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const v4f w4 = st4[(j ^ 1) + T * q];
          xp[2 * q] = mkv(w4.x, w4.y);
          xp[2 * q + 1] = mkv(w4.z, w4.w);
        }
      }
#endif
      if (MODE == kModeRxReal && (UC_BAND_KNOCK & 2)) {
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = cvt_pair<DTYPE>(xp[t >> 1]);
      } else if (MODE == kModeRxReal) {
        // sample t of this thread = half (t & 1) of pair t >> 1; the table products ride in the first
        // butterfly additions (pk_dft4_scaled), so the pass starts at the second half of the DFT
        v2f xc[8];
#pragma unroll
        for (int m = 0; m < 8; m++) xc[m] = cvt_pair<DTYPE>(xp[m]);
        pk_dft4_scaled<0>(v[0], v[4], v[8], v[12], xc[0], xc[2], xc[4], xc[6], wt[0], wt[4], wt[8], wt[12]);
        pk_dft4_scaled<1>(v[1], v[5], v[9], v[13], xc[0], xc[2], xc[4], xc[6], wt[1], wt[5], wt[9], wt[13]);
        pk_dft4_scaled<0>(v[2], v[6], v[10], v[14], xc[1], xc[3], xc[5], xc[7], wt[2], wt[6], wt[10], wt[14]);
        pk_dft4_scaled<1>(v[3], v[7], v[11], v[15], xc[1], xc[3], xc[5], xc[7], wt[3], wt[7], wt[11], wt[15]);
      } else if (kPair) {
        // re = frame a * ref * hann, im = frame b * ref * hann: two real spectra in one transform
#pragma unroll
        for (int m = 0; m < 8; m++) {
          v[2 * m] = pk_scale_lo(cvt_pair<DTYPE>(xp[2 * m]), wr[m]);
          v[2 * m + 1] = pk_scale_hi(cvt_pair<DTYPE>(xp[2 * m + 1]), wr[m]);
        }
      } else {
        const __amdgpu_buffer_rsrc_t rt = run == 0 ? rs_tab0 : rs_tab1;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const v2f x2 = cvt_pair<DTYPE>(xp[m]);
          // (at 3 waves/SIMD the last entries of the resident table would spill: they are loaded instead)
          const bool res0 = run < kCplxRes && (WAVES <= 2 || 2 * m < UC_CPLX_NRES);
          const bool res1 = run < kCplxRes && (WAVES <= 2 || 2 * m + 1 < UC_CPLX_NRES);
          v[2 * m] = pk_scale_lo(res0 ? wc[run][2 * m] : buf_ld64(rt, voff8, T * 8 * (2 * m)), x2);
          v[2 * m + 1] = pk_scale_hi(res1 ? wc[run][2 * m + 1] : buf_ld64(rt, voff8, T * 8 * (2 * m + 1)), x2);
        }
      }
      // prefetch the next frame a whole frame time ahead (HBM latency under load is
      // microseconds; at 3 waves/SIMD the 16 registers are free)
      if (run == 0 && dyn && (ROWS ? grp_first : (f & gmask) == 0)) {
        // first frame of a group: every load of this frame has been consumed above, so the atomic issued behind
        // them has returned too (a one-frame group parked it at the loop top already: same value again)
        if (j == 0) *next_slot = fetched;
      }
#ifndef UC_KNOCK_NOLOAD  // (knock-out build: every frame re-uses the first frame's samples -- what the loop costs without HBM)
      if constexpr (kDblX) {
        if (run == 0 && has_next) {
          resolve_next();
          load_unit(fnext, xq);
        }
      } else {
        if (run == kRuns - 1 && has_next) {
          resolve_next();
          load_unit(fnext, xp);
        }
      }
#endif
      if (MODE == kModeRxReal && (UC_BAND_KNOCK & 2)) { }
      else if (MODE == kModeRxReal) pk_dft16_finish(v, K, H);
      else pk_dft16(v, K, H);
      UC_STAMP(0);
      // B4, placed AFTER the register-only part of pass 1: the wave that finished the previous
      // frame first computes ahead instead of idling (wave 1 carries the extra pruned round).
      // It frees the tile (all pruned-pass reads done) and publishes the ring entry.
#if !(UC_BAND_KNOCK & 2048)  // (pricing only, with 1024: the staging barrier at the loop top has taken its place)
      __syncthreads();
#endif
      UC_STAMP(7);
      if constexpr (ROWS) {
        // a new group starts: drain the last one
        if (run == 0 && grp_first && old_n > 0 && wave == 0) rows_finalise(old_f0, old_n, old_eval);
      } else {
      if (run == 0 && ring_n > 0 && (f & gmask) == 0) {  // a new group starts: drain the last one
        if (wave == 0) finalise(ring_f0, ring_n);
        ring_f0 = f;
        ring_n = 0;
      }
      }
      // (wave priority: low while it issues a burst of LDS stores, raised otherwise -- the SIMD's other waves get their
      // arithmetic issued ahead of the store burst; measured +0.5-0.7 %, the opposite assignment -1.5 %)
#if UC_BAND_KNOCK & 256  // (pricing only: exchange 1 as 30 row rotations instead of LDS stores + B1 + LDS reads + B2)
#define UC_ROR(S) v[pk_slot16(S)] = row_ror<S>(v[pk_slot16(S)]);
      UC_ROR(1) UC_ROR(2) UC_ROR(3) UC_ROR(4) UC_ROR(5) UC_ROR(6) UC_ROR(7) UC_ROR(8) UC_ROR(9) UC_ROR(10) UC_ROR(11)
      UC_ROR(12) UC_ROR(13) UC_ROR(14) UC_ROR(15)
#undef UC_ROR
#else
      __builtin_amdgcn_s_setprio(0);
#if !(UC_BAND_KNOCK & 16)
#pragma unroll
      for (int t = 0; t < 16; t += 2)
        lds_st2(lds, wr1 + (t ^ s1v), v[4 * (t & 3) + (t >> 2)], v[4 * ((t + 1) & 3) + ((t + 1) >> 2)]);
#endif
      __builtin_amdgcn_s_setprio(2);
      UC_STAMP(1);
      __syncthreads();  // B1
      UC_STAMP(2);
#endif

      // ---- pass 2: radix-16, Ns = 16 ----------------------------------------
      // all 16 reads are issued back to back (the fence keeps hipcc from sinking them
      // next to their uses, which serialises four load->wait round trips)
#if !(UC_BAND_KNOCK & (16 | 256))
#pragma unroll
      for (int t = 0; t < 16; t++) v[t] = lds_ld(lds, rd1 + 128 * t);
#endif
      __builtin_amdgcn_sched_barrier(0);
#if !(UC_BAND_KNOCK & 4)
#pragma unroll
      for (int t = 1; t < 16; t++) v[t] = pk_cmul(v[t], kTw2Lds ? lds_ld(tw2l, tw2o + 16 * t) : tw2[t]);
      pk_dft16(v, K, H);
#endif
      UC_STAMP(3);
#if !(UC_BAND_KNOCK & 256)
      __syncthreads();  // B2: every pass-2 read is done before the tile is overwritten
#endif
      UC_STAMP(4);
      __builtin_amdgcn_s_setprio(0);
#if !(UC_BAND_KNOCK & 32)
#pragma unroll
      for (int t = 0; t < 16; t++) lds_st(lds, wr2 + 16 * t, v[4 * (t & 3) + (t >> 2)]);
#endif
      __builtin_amdgcn_s_setprio(2);

      __syncthreads();  // B3
      UC_STAMP(5);
      __builtin_amdgcn_s_setprio(3);  // the pruned pass and the window search end the frame: first in line (+0.4 %)
      // ---- pass 3: radix-8, Ns = 256, only bins i in [0, bw2] and n - i -----
      // Round 0: bin j on every thread.  Round 1: bins 128 + lane on wave 1 only (wave 0
      // owns the finaliser).  A round is latency- not throughput-bound: splitting round 1
      // into one dot product per wave was measured SLOWER (both waves then pay a round).
      // RX_REAL: first = |A[i]| (up), second = |B[i]| (down); CPLX: |Z[i]|, |Z[n-i]|.
      if constexpr (WIDE) {
        // Windows of up to 319 bins: three rounds -- bins j and 128 + j on both waves, 256 + lane on wave 1 -- with the
        // twiddles of rounds 1 and 2 resident, then the generic window search over the lane's three candidates.
        float qa[3] = {0.f, 0.f, 0.f}, qb[3] = {0.f, 0.f, 0.f};
        int kb[3];
        kb[0] = j;
        kb[1] = 128 + j;
        kb[2] = (wave == 1) ? 256 + lane : 0x3fffffff;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const int i = kb[r];
          if (i <= bw2) {
            const int ia = i & 255, ib = (256 - i) & 255;
            v2f av[8], bv[8];
#pragma unroll
            for (int t = 0; t < 8; t++) {
              av[t] = lds_ld(lds, ia + 256 * t);
              bv[t] = lds_ld(lds, ib + 256 * t);
            }
            __builtin_amdgcn_sched_barrier(0);
            const v2f w1 = (r == 0) ? t3a : tw3w[r == 1 ? 0 : 1][0];
            const v2f w2 = (r == 0) ? t3b : tw3w[r == 1 ? 0 : 1][1];
            v2f el = pk_cfma(av[6], w2, av[4]), ol = pk_cfma(av[7], w2, av[5]);
            v2f eh = pk_cfmac(bv[6], w2, bv[4]), oh = pk_cfmac(bv[7], w2, bv[5]);
            el = pk_cfma(el, w2, av[2]); ol = pk_cfma(ol, w2, av[3]);
            eh = pk_cfmac(eh, w2, bv[2]); oh = pk_cfmac(oh, w2, bv[3]);
            el = pk_cfma(el, w2, av[0]); ol = pk_cfma(ol, w2, av[1]);
            eh = pk_cfmac(eh, w2, bv[0]); oh = pk_cfmac(oh, w2, bv[1]);
            const v2f zn = el - ol;               // bin 0 only: Z[n/2]
            const v2f zl = pk_cfma(ol, w1, el);   // Z[k]
            const v2f zh = pk_cfmac(oh, w1, eh);  // Z[n-k]
            if (kReal) {
              const v2f sa = pk_add_conj(zl, zh);  // 2 A[k]
              const v2f sb = pk_sub_conj(zl, zh);  // 2j B[k]
              // (|.|^2 as ONE product and ONE fused multiply-add, spelled out: left to the compiler's contraction the batch and the
              // ROWS instantiation of this branch rounded differently in the last bit -- round 6, tests/test_gpu_wide.py)
              float ma = __builtin_fmaf(sa.x, sa.x, sa.y * sa.y), mb = __builtin_fmaf(sb.x, sb.x, sb.y * sb.y);
              if (r == 0 && i == 0) {  // Q2, as in the default build below
                ma = sa.x * sa.x;
                mb = sb.y * sb.y;
                if (!p.true_dc) {
                  ma = __builtin_fmaf(4.0f * zn.x, zn.x, ma);
                  mb = __builtin_fmaf(4.0f * zn.y, zn.y, mb);
                }
              }
              qa[r] = ma;
              qb[r] = mb;
            } else {
              qa[r] = __builtin_fmaf(zl.x, zl.x, zl.y * zl.y);
              qb[r] = __builtin_fmaf(zh.x, zh.x, zh.y * zh.y);
            }
            if (p.spectrum) store_spectrum(f, run, i, qa[r], qb[r]);
          }
        }
        UC_STAMP(6);
        auto pack16 = [](const Partial& q) {
          return ((unsigned)q.kr | ((q.nan_first & 1u) << 15)) | (((unsigned)q.kl | ((q.nan_first & 2u) << 14)) << 16);
        };
        float* e = ring + UC_RSLOT * kRingStride + wave * 6;
        if (kReal) {
          // the up history looks at qa in both windows, the down history at qb
          const Partial qu = window_partial_n<3>(qa, qa, kb, bw2);
          const Partial qd = window_partial_n<3>(qb, qb, kb, bw2);
          if (lane == 0) {
            e[0] = qu.vr; e[1] = qu.vl; e[2] = qd.vr; e[3] = qd.vl;
            e[4] = __uint_as_float(pack16(qu));
            e[5] = __uint_as_float(pack16(qd));
          }
        } else {
          const Partial q = window_partial_n<3>(qa, qb, kb, bw2);
          pv[2 * run] = q.vr;
          pv[2 * run + 1] = q.vl;
          kpw[run] = pack16(q);
          if (run == kRuns - 1 && lane == 0) {
            e[0] = pv[0]; e[1] = pv[1]; e[2] = pv[2]; e[3] = pv[3];
            e[4] = __uint_as_float(kpw[0]);
            e[5] = __uint_as_float(kpw[1]);
          }
        }
      } else {
      float m_a[2] = {0.f, 0.f}, m_b[2] = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 2; r++) {
        int lane_r = lane;
        if (kTw2Lds) asm volatile("" : "+v"(lane_r));  // 128-register build: re-derive the round's addresses every frame
        const int i = (r == 0) ? j : 128 + lane_r;
        if (i <= bw2 && (r == 0 || wave == 1)) {
          // issue the 16 reads of the round first, derive the twiddles under their latency
          const int ib = (256 - i) & 255;
          v2f av[8], bv[8];
#pragma unroll
          for (int t = 0; t < 8; t++) {
#if UC_BAND_KNOCK & 64
            av[t] = v[t];
            bv[t] = v[8 + t];
            (void)ib;
#else
            av[t] = lds_ld(lds, i + 256 * t);
            bv[t] = lds_ld(lds, ib + 256 * t);
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
#if UC_BAND_KNOCK & 8
          if (kReal) {
            v2f sa = av[0], sb = bv[0];
#pragma unroll
            for (int t = 1; t < 8; t++) { sa = sa + av[t]; sb = sb + bv[t]; }   // (keeps the 16 values live: 14 packed adds)
            m_a[r] = sa.x + sa.y;
            m_b[r] = sb.x + sb.y;
            continue;
          }
#endif
          // Z = sum_t a_t w^t evaluated as E(w^2) + w O(w^2), E and O by Horner in w^2: seven fused
          // multiply-adds with only w and w^2 (no per-frame derivation of w^3..w^7)
          v2f w1, w2;
          if (r == 0) {
            w1 = t3a;
            w2 = t3b;
          } else {
            // wave 1: W_2048^(j+64) = W_2048^j * W_32, squared: * W_16
            w1 = pk_cmul_s(t3a, mkv(kCos16, -kSin16));
            w2 = pk_mul_w1(t3b, K);
          }
          constexpr bool do_a = true, do_b = true;
          if (kReal) {
            // A[k] = (Z[k] + conj Z[n-k]) / 2,  B[k] = (Z[k] - conj Z[n-k]) / 2j
            // Z[k] = sum a_t w_t and Z[n-k] = sum b_t conj(w_t) first (2 packed FMAs per term each),
            // then sa = Z[k] + conj Z[n-k] = 2 A[k], sb = Z[k] - conj Z[n-k] = 2j B[k]
            v2f el = pk_cfma(av[6], w2, av[4]), ol = pk_cfma(av[7], w2, av[5]);
            v2f eh = pk_cfmac(bv[6], w2, bv[4]), oh = pk_cfmac(bv[7], w2, bv[5]);
            el = pk_cfma(el, w2, av[2]); ol = pk_cfma(ol, w2, av[3]);
            eh = pk_cfmac(eh, w2, bv[2]); oh = pk_cfmac(oh, w2, bv[3]);
            el = pk_cfma(el, w2, av[0]); ol = pk_cfma(ol, w2, av[1]);
            eh = pk_cfmac(eh, w2, bv[0]); oh = pk_cfmac(oh, w2, bv[1]);
            // bin 0 (w = 1): el / ol are the sums of the even / odd terms, so Z[n/2] = sum (-1)^t a_t = el - ol
            const v2f zn = el - ol;
            const v2f zl = pk_cfma(ol, w1, el);   // Z[k]
            const v2f zh = pk_cfmac(oh, w1, eh);  // Z[n-k] (conjugated twiddles)
            const v2f sa = pk_add_conj(zl, zh);
            const v2f sb = pk_sub_conj(zl, zh);
            // Window search on q = 4 |X|^2 (monotonic in |X|); the finaliser takes the
            // square root of the four winners only: |X| = 0.5 sqrt(q).
            float ma = 0.f, mb = 0.f;
            // (plain expressions: which product the compiler fuses is its choice -- today the SAME one in the batch, overlap, SPEC
            // and ROWS instantiations of this branch, which tests/test_gpu_k6.py, test_gpu_receive_many.py and test_gpu_decision.py
            // hold bit for bit against each other; the WIDE branch above had to spell it out)
            if (do_a) ma = sa.x * sa.x + sa.y * sa.y;
            if (do_b) mb = sb.x * sb.x + sb.y * sb.y;
            if (r == 0 && i == 0) {
              // Q2: the packed RFFT stores Re X[n/2] in the imaginary slot of bin 0, so the
              // reference's mag[0] is hypot(X0, X[n/2]) (receiver/Src/main.c:178).
              // sa.x = 2 Re Z[0] = 2 X_up[0], sb.y = 2 Im Z[0] = 2 X_down[0], zn = Z[n/2].
              ma = sa.x * sa.x;
              mb = sb.y * sb.y;
              if (!p.true_dc) {
                ma += 4.0f * (zn.x * zn.x);
                mb += 4.0f * (zn.y * zn.y);
              }
            }
            m_a[r] = ma;
            m_b[r] = mb;
          } else {
            v2f el = pk_cfma(av[6], w2, av[4]), ol = pk_cfma(av[7], w2, av[5]);
            v2f eh = pk_cfmac(bv[6], w2, bv[4]), oh = pk_cfmac(bv[7], w2, bv[5]);
            el = pk_cfma(el, w2, av[2]); ol = pk_cfma(ol, w2, av[3]);
            eh = pk_cfmac(eh, w2, bv[2]); oh = pk_cfmac(oh, w2, bv[3]);
            el = pk_cfma(el, w2, av[0]); ol = pk_cfma(ol, w2, av[1]);
            eh = pk_cfmac(eh, w2, bv[0]); oh = pk_cfmac(oh, w2, bv[1]);
            const v2f zl = pk_cfma(ol, w1, el);
            const v2f zh = pk_cfmac(oh, w1, eh);
            if (do_a) m_a[r] = zl.x * zl.x + zl.y * zl.y;  // |Z|^2: the finaliser takes sqrt
            if (do_b) m_b[r] = zh.x * zh.x + zh.y * zh.y;
          }
          if constexpr (SPEC) {
            if (p.spectrum) store_spectrum(f, run, i, m_a[r], m_b[r]);
          }
        }
      }
      UC_STAMP(6);
      // ---- windows: this wave's partial arm_max_f32 results, in registers -----
      // (bins beyond bw2 fail the window predicates inside window_partial)
      const int k1 = 128 + lane;
      float* e = ring + UC_RSLOT * kRingStride + wave * 6;
      if (kReal) {
        // up history looks at m_a, down history at m_b; wave 1 also holds slot 1
        Common up, dn;
        // (base0 = the bin of the wave's lane 0 in slot 0: 64 * wave, a scalar)
#if UC_BAND_KNOCK & 128
        up.m = m_a[0] + m_a[1]; dn.m = m_b[0] + m_b[1]; up.ks = up.kl = dn.ks = dn.kl = j;
#else
        if (wave == 0) common_partial2<false>(m_a[0], m_b[0], j, 0, 0.f, 0.f, k1, bw2, up, dn);
        else common_partial2<true>(m_a[0], m_b[0], j, 64, m_a[1], m_b[1], k1, bw2, up, dn);
#endif
        if (lane == 0) {
          e[0] = up.m;
          e[1] = dn.m;
          e[2] = __uint_as_float((unsigned)up.ks | ((unsigned)up.kl << 8) | ((unsigned)dn.ks << 16) |
                                 ((unsigned)dn.kl << 24));
        }
        // edge bins go straight to the ring from the lanes that own them
        float* e0 = ring + UC_RSLOT * kRingStride;
        if (j == 0) { e0[3] = m_a[0]; e0[4] = m_b[0]; }
        if (j == bw2) { e0[6 + 3] = m_a[0]; e0[6 + 4] = m_b[0]; }
        if (wave == 1 && k1 == bw2) { e0[6 + 3] = m_a[1]; e0[6 + 4] = m_b[1]; }
      } else {
        // right window looks at |Z[k]|^2 (m_a), left window at |Z[n-k]|^2 (m_b)
        Partial q;
        if (wave == 0) q = window_partial<false, false>(m_a[0], m_b[0], j, 0.f, 0.f, k1, bw2);
        else q = window_partial<true, true>(m_a[0], m_b[0], j, m_a[1], m_b[1], k1, bw2);
        pv[2 * run] = q.vr;
        pv[2 * run + 1] = q.vl;
        kpack |= ((unsigned)q.kr | ((unsigned)q.kl << 8)) << (16 * run);
        flags |= q.nan_first << (2 * run);
        if (run == kRuns - 1 && lane == 0) {
          e[0] = pv[0]; e[1] = pv[1]; e[2] = pv[2]; e[3] = pv[3];
          e[4] = __uint_as_float(kpack);
          e[5] = __uint_as_float(flags);
        }
      }
      }
      UC_STAMP(8);
      __builtin_amdgcn_s_setprio(2);
      if (!ROWS && run == kRuns - 1) ring_n++;
    }
    if (!has_next) break;
    if (dyn && (ROWS ? next_first : (fnext & gmask) == 0) && j == 0) {
      // a new group was taken: ask for the one after it.  Issued here, where few registers are live.
      fetched = atomicAdd(p.work_ctr, 1u);
    }
    if constexpr (kDblX) {
#pragma unroll
      for (int m = 0; m < NX; m++) xp[m] = xq[m];
    }
    f = fnext;
  }
  __syncthreads();
  if (ring_n > 0 && wave == 0) finalise(ring_f0, ring_n);
  if (dyn && j == 0) handout_leave(p.work_ctr);  // the last workgroup out leaves the counter at zero for the next launch
#ifdef UC_STAMPS
  if (lane == 0 && p.debug) {
    for (int k = 0; k < 10; k++) p.debug[((size_t)blockIdx.x * 2 + wave) * 10 + k] = acc_[k];
  }
#endif
  UC_CLOCK_END(p.debug, 2);
}

template <int MODE, int DTYPE, int WAVES, bool WIDE = false, bool SPEC = false, int FRAMES = kFramesBatch>
static int launch_one(const BandParams& p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL((band_kernel<MODE, DTYPE, WAVES, WIDE, SPEC, FRAMES>), dim3((unsigned)grid), dim3((unsigned)T), 0, stream, p);
  return (int)hipGetLastError();
}

template <int MODE, int DTYPE, int WAVES, bool WIDE = false, bool SPEC = false, int FRAMES = kFramesBatch>
static int occupancy_one() {
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, band_kernel<MODE, DTYPE, WAVES, WIDE, SPEC, FRAMES>, T, 0);
  if (e != hipSuccess || nb <= 0) nb = 2 * WAVES;
  return nb;
}

}  // namespace

UC_LAUNCH_BEGIN
#define UC_DISPATCH(FN, ...)                                                              \
  do {                                                                                    \
    if (rows) { /* the FIFO offsets of live receivers (RX_REAL, SYNC_CPLX): each mode's default occupancy, or WIDE */ \
      if (mode == kModePair || spec) return (int)hipErrorInvalidValue;                    \
      if (mode == kModeRxReal) {                                                          \
        if (wide) {                                                                       \
          if (dtype == UC_DTYPE_I32) return FN<kModeRxReal, UC_DTYPE_I32, 2, true, false, kFramesRows>(__VA_ARGS__);   \
          return FN<kModeRxReal, UC_DTYPE_F32, 2, true, false, kFramesRows>(__VA_ARGS__);        \
        }                                                                                 \
        if (dtype == UC_DTYPE_I32) return FN<kModeRxReal, UC_DTYPE_I32, 3, false, false, kFramesRows>(__VA_ARGS__);    \
        return FN<kModeRxReal, UC_DTYPE_F32, 3, false, false, kFramesRows>(__VA_ARGS__);         \
      }                                                                                   \
      if (wide) {                                                                         \
        if (dtype == UC_DTYPE_I32) return FN<kModeCplx, UC_DTYPE_I32, 2, true, false, kFramesRows>(__VA_ARGS__);       \
        return FN<kModeCplx, UC_DTYPE_F32, 2, true, false, kFramesRows>(__VA_ARGS__);            \
      }                                                                                   \
      if (dtype == UC_DTYPE_I32) return FN<kModeCplx, UC_DTYPE_I32, 2, false, false, kFramesRows>(__VA_ARGS__);        \
      return FN<kModeCplx, UC_DTYPE_F32, 2, false, false, kFramesRows>(__VA_ARGS__);             \
    }                                                                                     \
    if (overlap && !spec && !wide && mode != kModePair) { /* overlapping frames (stride < n): default occupancy of each mode */ \
      if (mode == kModeRxReal && waves == 3) {                                            \
        if (dtype == UC_DTYPE_I32) return FN<kModeRxReal, UC_DTYPE_I32, 3, false, false, kFramesOverlap>(__VA_ARGS__);  \
        return FN<kModeRxReal, UC_DTYPE_F32, 3, false, false, kFramesOverlap>(__VA_ARGS__);                            \
      }                                                                                   \
      if (mode == kModeCplx && waves == 2) {                                              \
        if (dtype == UC_DTYPE_I32) return FN<kModeCplx, UC_DTYPE_I32, 2, false, false, kFramesOverlap>(__VA_ARGS__);    \
        return FN<kModeCplx, UC_DTYPE_F32, 2, false, false, kFramesOverlap>(__VA_ARGS__);                              \
      }                                                                                   \
    }                                                                                     \
    if (spec && !wide) { /* uc_window_spectrum on the default two-round build, at each mode's default occupancy */ \
      if (mode == kModePair) {                                                            \
        if (dtype == UC_DTYPE_I32) return FN<kModePair, UC_DTYPE_I32, 3, false, true>(__VA_ARGS__);   \
        return FN<kModePair, UC_DTYPE_F32, 3, false, true>(__VA_ARGS__);                  \
      }                                                                                   \
      if (mode == kModeRxReal) {                                                          \
        if (dtype == UC_DTYPE_I32) return FN<kModeRxReal, UC_DTYPE_I32, 3, false, true>(__VA_ARGS__); \
        return FN<kModeRxReal, UC_DTYPE_F32, 3, false, true>(__VA_ARGS__);                \
      }                                                                                   \
      if (dtype == UC_DTYPE_I32) return FN<kModeCplx, UC_DTYPE_I32, 2, false, true>(__VA_ARGS__);     \
      return FN<kModeCplx, UC_DTYPE_F32, 2, false, true>(__VA_ARGS__);                    \
    }                                                                                     \
    if (wide) { /* windows of 192 .. 319 bins: one build per mode and dtype */             \
      if (mode == kModePair) {                                                            \
        if (dtype == UC_DTYPE_I32) return FN<kModePair, UC_DTYPE_I32, 2, true>(__VA_ARGS__);   \
        return FN<kModePair, UC_DTYPE_F32, 2, true>(__VA_ARGS__);                         \
      }                                                                                   \
      if (mode == kModeRxReal) {                                                          \
        if (dtype == UC_DTYPE_I32) return FN<kModeRxReal, UC_DTYPE_I32, 2, true>(__VA_ARGS__); \
        return FN<kModeRxReal, UC_DTYPE_F32, 2, true>(__VA_ARGS__);                       \
      }                                                                                   \
      if (dtype == UC_DTYPE_I32) return FN<kModeCplx, UC_DTYPE_I32, 2, true>(__VA_ARGS__);     \
      return FN<kModeCplx, UC_DTYPE_F32, 2, true>(__VA_ARGS__);                           \
    }                                                                                     \
    if (mode == kModePair) { /* one build: 3 waves/SIMD */                                \
      if (dtype == UC_DTYPE_I32) return FN<kModePair, UC_DTYPE_I32, 3>(__VA_ARGS__);      \
      return FN<kModePair, UC_DTYPE_F32, 3>(__VA_ARGS__);                                 \
    }                                                                                     \
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
  const bool wide = p.wide != 0, spec = p.spectrum != nullptr, rows = p.row_blocks != 0;
  const bool overlap = !rows && p.stride < (size_t)kN;
  UC_DISPATCH(launch_one, p, grid, stream);
}

int band_max_blocks_per_cu(int mode, int dtype, int waves, bool wide, bool spec, bool rows, bool overlap) {
  UC_DISPATCH(occupancy_one);
}

UC_LAUNCH_END

}  // namespace uc
