// uc_full_kernel.hip -- UC_COMPRESS: chirp compression by FFT x H x IFFT.
//
// Replaces, per frame (reference lines, experiments/chirp_compression_time_domain):
//   windowing(pInOut)                          Src/chirp.c:47-50, 79   (symmetric Hann)
//   arm_rfft_fast_f32(&S, pInOut, pInOut, 0)   Src/chirp.c:80
//   arm_cmplx_mult_cmplx_f32(pInOut, H_down)   Src/chirp.c:81          (Q5 fixed: DC / Nyquist separately)
//   arm_rfft_fast_f32(&S, pInOut, pInOut, 1)   Src/chirp.c:82          (inverse, 1/N inside)
//   arm_max_f32(fft_inout, PCM_SAMPLES, ...)   Src/main.c:186-189      (signed maximum + index)
//
// Design (MI355X): the filter h is real, so two frames ride in ONE complex
// transform (re = frame 2q, im = frame 2q+1) and stay separate through
// FFT -> xH -> IFFT.  One 2-wave workgroup per frame pair, persistent.  Forward
// 2048 = 16 x 16 x 8 Stockham (as uc_band_kernel.hip, full last pass); the
// spectrum is multiplied by H/N in registers and -- because a Stockham first
// pass consumes exactly the stride-256 octets the forward last pass produced --
// the inverse transform (8 x 16 x 16) starts in the same registers: 4 LDS
// exchanges per pair, ping-ponging between two tiles (one barrier each).
// Inverse butterflies reuse the forward ones through IDFT_R[k] = DFT_R[-k mod R]
// with conjugated twiddles.  The next pair is prefetched a whole pair time ahead
// and nothing else is loaded from memory inside the loop (Hann, H/N and the
// twiddle seeds are register-resident, the small twiddle tables sit in LDS).
// HBM traffic: 2 x 8 KiB in, 2 x 32 B stats out per pair.
#include "uc_dev.hpp"
#include "uc_kernels.hpp"
#include "uc_xform.hpp"

namespace uc {

namespace {

constexpr int T = kBandThreads;  // 128
#ifndef UC_COMPRESS_WAVES
#define UC_COMPRESS_WAVES 2
#endif
#ifndef UC_COMPRESS_RESIDENT_TW
#define UC_COMPRESS_RESIDENT_TW 1
#endif
constexpr bool kResTw = UC_COMPRESS_RESIDENT_TW != 0;
constexpr int kRedOff = 4 * kN;  // floats: per-wave reduction results after the two tiles
constexpr int kTw2Off = kRedOff + 16;
constexpr int kTwBOff = kTw2Off + 2 * 256;
constexpr int kLdsFloats = kTwBOff + 2 * 128;

// five cross-lane steps that finish a reduction whose two 32-lane halves hold two independent searches:
// afterwards lane 31 has the result of lanes 0-31 and lane 63 that of lanes 32-63
#define UC_DPP_HALVES(OP, v)                                                  \
  asm("s_nop 1\n\t" OP " %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"  \
      "s_nop 1\n\t" OP " %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"  \
      "s_nop 1\n\t" OP " %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"  \
      "s_nop 1\n\t" OP " %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"  \
      "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"     \
      : "+v"(v))

__device__ __forceinline__ float max3_f32(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// arm_max_f32 over the 2048 signed values of BOTH frames of a pair (y[t].x = frame a, y[t].y = frame b,
// index j + 128 t): this wave's maxima and first indices.  NaNs are skipped as arm_max_f32's '<' update
// skips them (v_max_f32 returns the number), except a NaN in element 0, which sticks.
__device__ __forceinline__ void pair_max(const v2f (&y)[16], int j, float& va, int& ia, float& vb, int& ib) {
  // per thread: the maximum of its 16 values, then the FIRST t that attains it
  float ma = max3_f32(y[0].x, y[1].x, y[2].x), mb = max3_f32(y[0].y, y[1].y, y[2].y);
#pragma unroll
  for (int t = 3; t < 15; t += 2) {
    ma = max3_f32(ma, y[t].x, y[t + 1].x);
    mb = max3_f32(mb, y[t].y, y[t + 1].y);
  }
  ma = max_f32(ma, y[15].x);
  mb = max_f32(mb, y[15].y);
  int ta = 0, tb = 0;
#pragma unroll
  for (int t = 15; t >= 1; t--) {
    ta = (y[t].x == ma) ? t : ta;
    tb = (y[t].y == mb) ? t : tb;
  }
  ta = (y[0].x == ma) ? 0 : ta;
  tb = (y[0].y == mb) ? 0 : tb;
  const unsigned long long fna = __ballot(j == 0 && y[0].x != y[0].x), fnb = __ballot(j == 0 && y[0].y != y[0].y);
  // across the wave: frame a's search in lanes 0-31, frame b's in lanes 32-63 after one swap
  const float sa = (ma != ma) ? -INFINITY : ma, sb = (mb != mb) ? -INFINITY : mb;
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sa), __float_as_uint(sb), false, false);
  float m = max_f32(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  UC_DPP_HALVES("v_max_f32_dpp", m);
  va = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 31));
  vb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
  const int ca = (sa == va) ? j + T * ta : 0x7fffffff, cb = (sb == vb) ? j + T * tb : 0x7fffffff;
  const auto si = __builtin_amdgcn_permlane32_swap((unsigned)ca, (unsigned)cb, false, false);
  int c = (int)(si[0] < si[1] ? si[0] : si[1]);
  UC_DPP_HALVES("v_min_u32_dpp", c);
  ia = __builtin_amdgcn_readlane(c, 31);
  ib = __builtin_amdgcn_readlane(c, 63);
  if (fna) { va = __int_as_float(0x7fc00000); ia = 0; }
  if (fnb) { vb = __int_as_float(0x7fc00000); ib = 0; }
}

template <int DTYPE>
__global__ __launch_bounds__(T, UC_COMPRESS_WAVES) void compress_kernel(const FullParams p) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  float* ta = lds;             // the transforms ping-pong between two tiles: one barrier per exchange
  float* tb = lds + 2 * kN;
  float* red = lds + kRedOff;
  float* tw2t = lds + kTw2Off;  // W_256^(t k), t < 16, k < 16: forward pass 2
  float* twBt = lds + kTwBOff;  // W_128^(t k), t < 16, k < 8: inverse pass B

  const int j = threadIdx.x;
  const int lane = j & 63;
  const int wave = j >> 6;
  UC_CLOCK_BEGIN();  // diagnostic build only (uc_dev.hpp)

  // Static: a balanced contiguous partition of the frame pairs.  Dynamic (p.work_ctr): chunks of G = 2^chunk_log2
  // consecutive pairs; a workgroup starts with chunk blockIdx.x and takes every further one from an atomic counter
  // (the workgroups do not run at the same speed).  Thread 0 asks one pair before a chunk's last pair, right before
  // that pair's prefetch, and hands the id to the workgroup through one LDS word at the top of the chunk's last pair.
  // (32-bit bookkeeping: the host rejects batches of 2^31 frames or more; the frame ADDRESS is 64-bit)
  // (UC_FLAG_NO_FRAME_PAIRS: p.unpaired -- every frame rides alone, its partner slot reads as zeros)
  const unsigned psh = p.unpaired ? 0u : 1u;  // frames per unit = 1 << psh
  const unsigned npairs = (unsigned)((p.n_frames + psh) >> psh);
  const bool dyn = p.work_ctr != nullptr;
  const unsigned gsh = p.chunk_log2, gmask = (1u << gsh) - 1u;
  const unsigned nchunks = (npairs + gmask) >> gsh;
  unsigned q, qend;
  if (dyn) {
    if (blockIdx.x >= nchunks) {  // (the host never launches more workgroups than chunks)
      if (j == 0) handout_leave(p.work_ctr);
      return;
    }
    q = blockIdx.x << gsh;
    qend = q + gmask + 1u < npairs ? q + gmask + 1u : npairs;
  } else {
    const unsigned base = npairs / gridDim.x, rem = npairs % gridDim.x;
    const unsigned w_ = blockIdx.x;
    q = w_ * base + (w_ < rem ? w_ : rem);
    qend = q + base + (w_ < rem ? 1u : 0u);
    if (q >= qend) return;
  }
  constexpr unsigned kNoChunk = 0x7fffffffu;  // stays beyond every chunk count when gridDim.x is added
  unsigned fetched = kNoChunk;  // thread 0: what the atomic in flight returns; kNoChunk = none asked for (ragged last chunk)

  const __amdgpu_buffer_rsrc_t rs_hn = make_rsrc(p.hn, kN * 8);
  const __amdgpu_buffer_rsrc_t rs_tw = make_rsrc(p.tw, kN * 8);
  const int voff8 = j * 8, voff4 = j * 4;
  const v2f K = mkv(kCos8, kSin8), H = mkv(kSqrtHalfF, kSqrtHalfF);

  // Everything the loop needs besides the frames is resident (vector loads return in order: a table
  // load issued behind the next pair's prefetch would wait for it): Hann and H/N at this thread's
  // positions and three pass-3/C twiddle seeds in registers, the pass-2 / pass-B twiddles in LDS.
  v2f hw[8];  // symmetric Hann at this thread's samples j + 128 t, two per register pair
  {
    const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(p.hann, kN * 4);
#pragma unroll
    for (int m = 0; m < 8; m++) hw[m] = mkv(buf_ld32(rs_h, voff4, T * 4 * (2 * m)), buf_ld32(rs_h, voff4, T * 4 * (2 * m + 1)));
  }
  v2f hres[2][8];  // H[k]/N at this thread's bins k = j + 128 h + 256 t
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int t = 0; t < 8; t++) hres[h][t] = buf_ld64(rs_hn, voff8 + T * 8 * h, 256 * 8 * t);
  const v2f tw3_1 = buf_ld64(rs_tw, (j & (kN - 1)) * 8, 0);        // W_2048^j
  const v2f tw3_2 = buf_ld64(rs_tw, ((2 * j) & (kN - 1)) * 8, 0);  // W_2048^2j
  const v2f tw3_4 = buf_ld64(rs_tw, ((4 * j) & (kN - 1)) * 8, 0);  // W_2048^4j
  xf_fill_twiddle_tables(tw2t, twBt, rs_tw, j);
  const XfAddr xa = xf_addresses(j);
  // pass-3 and pass-C twiddles resident (the registers are there at 2 waves/SIMD): 44 fewer products per pair
  v2f w3r[2][8], wCr[16];
  if (kResTw) {
    xf_twiddles3(w3r[0], 0, tw3_1, tw3_2, tw3_4, K, H);
    xf_twiddles3(w3r[1], 1, tw3_1, tw3_2, tw3_4, K, H);
    xf_twiddlesC(wCr, tw3_1, tw3_2, tw3_4);
  }

  const bool has_mm = p.mag_mean != nullptr;

  // raw words of the pair: sample j + 128 t of frame 2q in the low, of frame 2q+1 in the high half
  v2f xp[16];
  auto load_pair = [&](unsigned uq) {
    const size_t fa = (size_t)uq << psh;
    const bool hb = psh && fa + 1 < p.n_frames;  // a ragged last pair (or no pairing): frame b reads as zeros
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(reinterpret_cast<const char*>(p.frames) + fa * p.stride * 4, kN * 4);
    const __amdgpu_buffer_rsrc_t rb =
        make_rsrc(reinterpret_cast<const char*>(p.frames) + (fa + (hb ? 1 : 0)) * p.stride * 4, hb ? kN * 4 : 0);
#pragma unroll
    for (int t = 0; t < 16; t++) xp[t] = mkv(buf_ld32_stream(ra, voff4, T * 4 * t), buf_ld32_stream(rb, voff4, T * 4 * t));
  };
  load_pair(q);

  // result of one pair: merged by thread 0/1 after the NEXT barrier (red is rewritten three barriers later)
  auto publish = [&](unsigned qu) {
    const size_t fa = (size_t)qu << psh;
    const bool hb = psh && fa + 1 < p.n_frames;
    if (j < 2 && (j == 0 || hb) && p.stats) {
      // merge the two waves: value, then smallest index; a NaN partial only survives
      // if element 0 was NaN (wave 0 reports it with index 0)
      const float v0 = red[2 * j], v1 = red[4 + 2 * j];
      const int i0 = __float_as_int(red[2 * j + 1]), i1 = __float_as_int(red[4 + 2 * j + 1]);
      float mx;
      int mi;
      if (v0 != v0) { mx = v0; mi = i0; }
      else if (v1 > v0 || (v1 == v0 && i1 < i0)) { mx = v1; mi = i1; }
      else { mx = v0; mi = i0; }
      const size_t ff = fa + j;
      const float mm = has_mm ? p.mag_mean[2 * ff] : p.mag_mean_scalar;
      float4 a, bq;
      a.x = mx; a.y = 0.0f; a.z = mx; a.w = __int_as_float(mi);
      bq.x = __int_as_float(0); bq.y = __int_as_float(mi); bq.z = mm; bq.w = (mx - mm) / mm;
      float4* d = reinterpret_cast<float4*>(p.stats + ff);
      d[0] = a;
      d[1] = bq;
    }
    if (j < 2 && (j == 0 || hb) && p.symbols) p.symbols[fa + j] = (uint8_t)UC_SYM_NONE;
  };

  bool pending = false;
  unsigned qprev = q;
  for (;;) {
    unsigned qn = q + 1;
    bool more = qn < qend;
    // last pair of a chunk: the next pair is the first of the chunk the hand-out gave (known behind the barrier below)
    const bool hop = dyn && !more;
    if (hop && j == 0) red[8] = __uint_as_float(fetched);
    if (hop) fetched = kNoChunk;
    int s1v = xa.s1;
    v2f t3a = tw3_1, t3b = tw3_2, t3c = tw3_4;
    asm volatile("" : "+v"(s1v), "+v"(t3a), "+v"(t3b), "+v"(t3c));

    // ---- window, forward pass 1 (registers -> tile A) -------------------------------
    {
      v2f v[16];
#pragma unroll
      for (int m = 0; m < 8; m++) {
        // hann * x, one float32 rounding, as windowing() does (chirp.c:47-50)
        v[2 * m] = pk_scale_lo(cvt_pair<DTYPE>(xp[2 * m]), hw[m]);
        v[2 * m + 1] = pk_scale_hi(cvt_pair<DTYPE>(xp[2 * m + 1]), hw[m]);
      }
      if ((!dyn || !hop) && more) load_pair(qn);  // a whole pair time ahead (inside a chunk the next pair is known here)
      pk_dft16(v, K, H);
      xf_store1(ta, xa, s1v, v);
    }
    __syncthreads();
    if (dyn) {
      if (hop) {
        const unsigned c = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(red[8])) + gridDim.x;
        more = c < nchunks;
        qn = c << gsh;
      }
      // one pair before a chunk's last pair: ask for the next chunk, ahead of the prefetch issued below
      if (((q + 2) & gmask) == 0 && j == 0) fetched = atomicAdd(p.work_ctr, 1u);  // (+ gridDim.x where it is read)
      if (hop && more) load_pair(qn);  // chunk boundary: the next pair was only known behind the barrier
    }
    if (pending) publish(qprev);

    xf_fwd2(ta, tb, tw2t, xa, j, K, H);                        // forward pass 2 (A -> B)
    __syncthreads();
    xf_fwd3_h_invA<kResTw>(tb, ta, hres, w3r, t3a, t3b, t3c, j, K, H);  // forward pass 3, x H/N, inverse pass A (B -> A)
    __syncthreads();
    xf_invB(ta, tb, twBt, xa, j, K, H);                        // inverse pass B (A -> B)
    __syncthreads();
    // inverse pass C (B -> registers): y[t] = sample j + 128 t; re = frame a, im = frame b
    v2f y[16];
    xf_invC<kResTw>(tb, y, xa, wCr, t3a, t3b, t3c, K, H);

    // ---- arm_max_f32 over the 2048 signed values of each frame -------------------
    float va, vb;
    int ia, ib;
    pair_max(y, j, va, ia, vb, ib);
    if (lane == 0) {
      red[4 * wave + 0] = va;
      red[4 * wave + 1] = __int_as_float(ia);
      red[4 * wave + 2] = vb;
      red[4 * wave + 3] = __int_as_float(ib);
    }
    pending = true;
    qprev = q;
    if (!more) break;
    q = qn;
    if (hop) qend = q + gmask + 1u < npairs ? q + gmask + 1u : npairs;
  }
  __syncthreads();
  if (pending) publish(qprev);
  if (dyn && j == 0) handout_leave(p.work_ctr);  // the last workgroup out leaves the counter at zero for the next launch
  UC_CLOCK_END(p.debug, 2);
}

}  // namespace

UC_LAUNCH_BEGIN
int launch_compress(int dtype, const FullParams& p, int grid, hipStream_t stream) {
  if (grid <= 0) return (int)hipSuccess;
  if (dtype == UC_DTYPE_I32)
    hipLaunchKernelGGL((compress_kernel<UC_DTYPE_I32>), dim3((unsigned)grid), dim3((unsigned)T), 0, stream, p);
  else
    hipLaunchKernelGGL((compress_kernel<UC_DTYPE_F32>), dim3((unsigned)grid), dim3((unsigned)T), 0, stream, p);
  return (int)hipGetLastError();
}

int compress_max_blocks_per_cu(int dtype) {
  int nb = 0;
  hipError_t e = dtype == UC_DTYPE_I32
                     ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, compress_kernel<UC_DTYPE_I32>, T, 0)
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, compress_kernel<UC_DTYPE_F32>, T, 0);
  if (e != hipSuccess || nb <= 0) nb = 4;
  return nb;
}

UC_LAUNCH_END

}  // namespace uc
