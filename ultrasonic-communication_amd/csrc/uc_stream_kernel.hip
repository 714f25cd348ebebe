// uc_stream_kernel.hip -- UC_STREAM: FIR-LPF decimating front-end + overlap-save chirp compression
// over one continuous sample stream (BASELINE config 4; definition in include/uchirp.h).
//
// Reference lines the stages restate:
//   carrier mix + 27-tap low-pass      experiments/iq_modulation/Src/iq_modem.c:55-75, taps :18
//   FFT, x H, inverse FFT (1/N)        experiments/chirp_compression_time_domain/Src/chirp.c:78-83
//   maximum + index of the result      experiments/chirp_compression_time_domain/Src/main.c:186-189
// The decimation and the overlap-save blocking are this build's (the firmware works frame by frame).
//
// Design (MI355X): one 2-wave workgroup per overlap-save block of 2048 decimated samples,
// persistent over a contiguous run of blocks (a block re-reads the (L-1) D + 26 input samples it
// shares with its predecessor: consecutive blocks on one CU find them in L2).
//   * The carrier is folded into the taps: sum_k fir[k] x[r-k] e^{-jw(r-k)} = e^{-jwr} sum_k c[k] x[r-k]
//     with 27 complex constants c[k] = fir[k] e^{jwk} held in SGPR pairs -- the FIR runs on the REAL
//     samples (one packed FMA per tap for I and Q together) and only the D-th outputs are computed;
//     e^{-jwr} is one table multiply per decimated sample (the block-constant part of that phase
//     drops out of |y|).
//   * Input is streamed in sub-tiles of 4096 samples: coalesced 16-byte loads into registers TWO
//     sub-tiles ahead (the next block's first two are in flight during the FFTs), written to a
//     padded LDS image (36-float rows per 32 samples: the per-thread sliding windows, 128 B apart,
//     are read with conflict-free ds_read_b128), 32/D consecutive outputs per thread.
//   * The decimated samples land in the FFT tile in natural order (linear writes, linear reads);
//     forward 16 x 16 x 8, x H/N in registers, inverse 8 x 16 x 16 as uc_full_kernel.hip, but
//     ping-ponging between the tile and the (then idle) image area: one barrier per exchange.
//   * |y| for the hop = 2049 - L valid outputs leaves as coalesced dword stores; the block maximum
//     is a DPP wave reduction on squared magnitudes.
// HBM traffic per input sample: 4 B in + 4/D B out (+ 8 B of peak record per block).
#include "uc_dev.hpp"
#include "uc_kernels.hpp"
#include "uc_xform.hpp"

#ifndef UC_STREAM_STORE_CPOL
#define UC_STREAM_STORE_CPOL UC_STREAM_CPOL  // cache policy of the |y| stores (2 = nt; 0 = default, for A/B)
#endif
#ifndef UC_STREAM_KNOCK
#define UC_STREAM_KNOCK 0  // diagnostic builds only: 1 no loads in the loop, 2 no FIR arithmetic, 4 no transforms, 8 no stores,
                           // 16 / 32 the round-2 nt mix / every load nt, 64 a whole block of loads in flight,
                           // 128 loads on the cache-line grid
#endif

namespace uc {

namespace {

constexpr int T = kBandThreads;  // 128
constexpr int kSubIn = 4096;     // input samples per sub-tile (32 per thread)
constexpr int kImgF4 = (kSubIn + 28) / 4 + ((kSubIn + 28) / 4 >> 3) + 1;  // padded image, float4 units
constexpr int kTileOff = kImgF4 * 4;                                        // floats
constexpr int kRedOff = kTileOff + 2 * kN;
constexpr int kTw2Off = kRedOff + 16;         // W_256^(t k), t < 16, k < 16: forward pass 2
constexpr int kTwBOff = kTw2Off + 2 * 256;    // W_128^(t k), t < 16, k < 8: inverse pass B
constexpr int kLdsFloats = kTwBOff + 2 * 128;

// acc += c * x.lo / c * x.hi: complex constant (SGPR pair) times a real sample broadcast from one
// half of a register pair -- one packed FMA for the I and the Q branch of the FIR
__device__ __forceinline__ void pk_fma_c_xlo(v2f& acc, v2f c, v2f x) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(c), "v"(x));
}
__device__ __forceinline__ void pk_fma_c_xhi(v2f& acc, v2f c, v2f x) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(c), "v"(x));
}

template <int DTYPE>
__device__ __forceinline__ v4f cvt4(v4u raw) {
  v4f r;
  if (DTYPE == UC_DTYPE_I32) {
    r.x = (float)(int)raw.x; r.y = (float)(int)raw.y; r.z = (float)(int)raw.z; r.w = (float)(int)raw.w;
  } else {
    r.x = __uint_as_float(raw.x); r.y = __uint_as_float(raw.y);
    r.z = __uint_as_float(raw.z); r.w = __uint_as_float(raw.w);
  }
  return r;
}

template <int DTYPE, int D>
__global__ __launch_bounds__(T, 2) void stream_kernel(const StreamParams p) {
  constexpr int L = kN / D;            // template length (decimated samples)
  constexpr int HOP = kN - (L - 1);    // valid outputs per block
  constexpr int OPT = 32 / D;          // FIR outputs per thread per sub-tile
  constexpr int NSUB = D / 2;          // sub-tiles per block: 2048 D / 4096
  constexpr int SUBOUT = kSubIn / D;   // decimated samples per sub-tile
  constexpr int WIN = 32 - D + kFirTapsDev;  // samples one thread's OPT outputs look at
  constexpr int WIN4 = (WIN + 3) / 4;
  static_assert(D == 4 || D == 8 || D == 16, "decimation");

  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  v4f* img = reinterpret_cast<v4f*>(lds);
  float* tile = lds + kTileOff;
  float* red = lds + kRedOff;

  const int j = threadIdx.x;
  const int wave = j >> 6;
  UC_CLOCK_BEGIN();  // diagnostic build only (uc_dev.hpp)

  // Static: a balanced contiguous partition of the blocks, workgroup w takes base (+1 for the first `rem`) blocks.
  // Dynamic (p.work_ctr): chunks of G = 2^chunk_log2 consecutive blocks; a workgroup starts with chunk blockIdx.x and
  // takes every further one from an atomic counter -- the workgroups do not run at the same speed.  Thread 0 asks one
  // block before a chunk's last block (right before a batch of input loads, which are waited for two sub-tiles
  // later) and hands the id to the workgroup through one LDS word at the top of the chunk's last block.
  // (32-bit block bookkeeping: the host rejects streams of 2^32 blocks; sample offsets are 64-bit)
  const unsigned nblk = (unsigned)p.n_blocks;
#ifdef UC_STREAM_NO_DYN  // (A/B build: the hand-out code compiled out)
  constexpr bool dyn = false;
#else
  const bool dyn = p.work_ctr != nullptr;
#endif
  const unsigned gsh = p.chunk_log2, gmask = (1u << gsh) - 1u;
  const unsigned nchunks = (nblk + gmask) >> gsh;
  unsigned b, bend;
  if (dyn) {
    if (blockIdx.x >= nchunks) {  // (the host never launches more workgroups than chunks)
      if (j == 0) handout_leave(p.work_ctr);
      return;
    }
    b = blockIdx.x << gsh;
    bend = b + gmask + 1u < nblk ? b + gmask + 1u : nblk;
  } else {
    const unsigned base = nblk / gridDim.x, rem = nblk % gridDim.x;
    const unsigned w_ = blockIdx.x;
    b = w_ * base + (w_ < rem ? w_ : rem);
    bend = b + base + (w_ < rem ? 1u : 0u);
    if (b >= bend) return;
  }
  constexpr unsigned kNoChunk = 0x7fffffffu;  // stays beyond every chunk count when gridDim.x is added
  unsigned fetched = kNoChunk;  // thread 0: what the atomic in flight returns; kNoChunk = none asked for (ragged last chunk)
  // (chunks of ONE block: every block asks for the one after its successor, so the first request goes out here)
  if (dyn && gsh == 0 && j == 0) fetched = atomicAdd(p.work_ctr, 1u);

  const __amdgpu_buffer_rsrc_t rs_hn = make_rsrc(p.hn, kN * 8);
  const __amdgpu_buffer_rsrc_t rs_tw = make_rsrc(p.tw, kN * 8);
  const __amdgpu_buffer_rsrc_t rs_rot = make_rsrc(p.rot, kN * 8);
  const int voff8 = j * 8, voff16 = j * 16;
  const v2f K = mkv(kCos8, kSin8), H = mkv(kSqrtHalfF, kSqrtHalfF);

  // Vector-memory loads return in order: a table load issued inside the block loop could not
  // complete before the input prefetch issued ahead of it, and would serialise the memory latency
  // into the transforms.  So everything the loop needs is made resident here -- registers for the
  // per-thread values, two small LDS tables for the pass-2 / pass-B twiddles -- and the loop's only
  // vector loads are the input stream itself.
  const v2f tw3_1 = buf_ld64(rs_tw, (j & (kN - 1)) * 8, 0);        // W_2048^j
  const v2f tw3_2 = buf_ld64(rs_tw, ((2 * j) & (kN - 1)) * 8, 0);  // W_2048^2j
  const v2f tw3_4 = buf_ld64(rs_tw, ((4 * j) & (kN - 1)) * 8, 0);  // W_2048^4j
  v2f hres[2][8];  // H[k]/N at this thread's bins k = j + 128 h + 256 t
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int t = 0; t < 8; t++) hres[h][t] = buf_ld64(rs_hn, voff8 + T * 8 * h, 256 * 8 * t);
  v2f rotu[OPT];   // e^{-jw D (OPT j + u)}: rotation of this thread's outputs inside a sub-tile
#pragma unroll
  for (int u = 0; u < OPT; u++) rotu[u] = buf_ld64(rs_rot, (OPT * j + u) * 8, 0);
  xf_fill_twiddle_tables(lds + kTw2Off, lds + kTwBOff, rs_tw, j);
  const float* tw2t = lds + kTw2Off;
  const float* twBt = lds + kTwBOff;
  const XfAddr xa = xf_addresses(j);  // LDS addresses of the transform, as uc_full_kernel.hip

  // one sub-tile of input: 4096 + 28 samples starting at sample (blk HOP D + sub 4096) of the buffer
  // (that sample is 26 taps behind the first output of the sub-tile); loads past the end of the
  // buffer, and the tail load of threads >= 7, fall outside the resource and return 0.
  // Two register sets: the loads run TWO sub-tiles ahead of the FIR (and through the transforms).
  constexpr int kDepth = (UC_STREAM_KNOCK & 64) ? NSUB : 2;  // (64: a whole block of input in flight -- diagnostic, with 2 + 4)
  v4u stg[kDepth][9];
  auto issue_loads = [&](unsigned blk, int sub, v4u (&dst)[9]) {
    size_t first = (size_t)blk * (size_t)(HOP * D) + (size_t)sub * kSubIn;
    if (UC_STREAM_KNOCK & 128) first &= ~(size_t)31;  // (128: loads on the 128-byte grid -- WRONG samples, timing only)
    const size_t left = p.n_samples > first ? p.n_samples - first : 0;
    const int recs = left < (size_t)(kSubIn + 28) ? (int)left : kSubIn + 28;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(reinterpret_cast<const char*>(p.samples) + first * 4, recs * 4);
    // Every load keeps the DEFAULT cache policy (round 3).  Rounds 1-2 streamed all but a block's last sub-tile past
    // the caches (nt): the last sub-tile holds the (L - 1) D samples the next block reads again.  Measured on two boxes
    // with the dynamic hand-out: default everywhere 1.833 / 1.836-1.858 ms per 2^31 samples, the nt mix 1.904 /
    // 1.847-1.917, nt everywhere 1.944 (profiles/r03_stream_knock.txt).
    if (!(UC_STREAM_KNOCK & 32) && (!(UC_STREAM_KNOCK & 16) || sub == NSUB - 1)) {  // (16: the nt mix of round 2, 32: every load nt)
#pragma unroll
      for (int r = 0; r < 8; r++) dst[r] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff16, T * 16 * r, 0);
      dst[8] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff16, kSubIn * 4, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 8; r++) dst[r] = buf_ld128_stream(rx, voff16, T * 16 * r);
      dst[8] = buf_ld128_stream(rx, voff16, kSubIn * 4);
    }
  };

#pragma unroll
  for (int s = 0; s < kDepth; s++) issue_loads(b, s, stg[s]);

  float* tb = lds;  // second transform tile: the image area is free once the last window is read

  while (true) {
    int s1v = xa.s1;
    v2f t3a = tw3_1, t3b = tw3_2, t3c = tw3_4;
    asm volatile("" : "+v"(s1v), "+v"(t3a), "+v"(t3b), "+v"(t3c));
    unsigned bn = b + 1;
    bool more = bn < bend;
    // last block of a chunk: the next block is the first of the chunk the hand-out gave (known behind the barrier below)
    const bool hop = dyn && !more;
    if (hop && j == 0) red[8] = __uint_as_float(fetched);
    if (hop) fetched = kNoChunk;

    // ---- front end: FIR + decimation, sub-tile by sub-tile, into the FFT tile ----------------
#pragma unroll
    for (int s = 0; s < NSUB; s++) {
      v4u (&cur)[9] = stg[s % kDepth];
      __syncthreads();  // the windows of the previous sub-tile (and the previous block's transforms) are read
      if (s == 0) {
        if (hop) {
          const unsigned c = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(red[8])) + gridDim.x;
          more = c < nchunks;
          bn = c << gsh;
        }
        // one block before a chunk's last block: ask for the next chunk, ahead of the loads issued below
        if (dyn && ((b + 2) & gmask) == 0 && j == 0) fetched = atomicAdd(p.work_ctr, 1u);  // (+ gridDim.x where it is read)
      }
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int q = j + T * r;  // float4 index inside the sub-tile
        img[q + (q >> 3)] = cvt4<DTYPE>(cur[r]);
      }
      if (j < 7) {
        const int q = kSubIn / 4 + j;
        img[q + (q >> 3)] = cvt4<DTYPE>(cur[8]);
      }
#if !(UC_STREAM_KNOCK & 1)  // (knock-out builds, tools/stream_knock.sh: what each stage costs; never shipped)
      if (s + kDepth < NSUB) issue_loads(b, s + kDepth, cur);
      else if (more) issue_loads(bn, s + kDepth - NSUB, cur);
#endif
      __syncthreads();

      // window: samples 32 j .. 32 j + WIN of the sub-tile image; output u sits on sample 26 + D u
      v2f xs[2 * WIN4];
#pragma unroll
      for (int d = 0; d < WIN4; d++) {
        const v4f w4 = img[9 * j + d + (d >> 3)];
        xs[2 * d] = mkv(w4.x, w4.y);
        xs[2 * d + 1] = mkv(w4.z, w4.w);
      }
      __builtin_amdgcn_sched_barrier(0);
      v2f acc[OPT];
#pragma unroll
      for (int u = 0; u < OPT; u++) acc[u] = mkv(0.f, 0.f);
#if UC_STREAM_KNOCK & 2
#pragma unroll
      for (int u = 0; u < OPT; u++) acc[u] = xs[u] + xs[WIN4 + u];
#endif
#pragma unroll
      for (int w = 0; w < ((UC_STREAM_KNOCK & 2) ? 0 : WIN); w++) {
#pragma unroll
        for (int u = 0; u < OPT; u++) {
          const int k = (kFirTapsDev - 1) + D * u - w;  // tap that multiplies sample w for output u
          if (k >= 0 && k < kFirTapsDev) {
            const v2f c = mkv(p.ctap[2 * k], p.ctap[2 * k + 1]);
            if (w & 1) pk_fma_c_xhi(acc[u], c, xs[w >> 1]);
            else pk_fma_c_xlo(acc[u], c, xs[w >> 1]);
          }
        }
      }
      // rotate by e^{-jw D m}, m = s SUBOUT + OPT j + u (the sub-tile's part of it is wave-uniform), and
      // store 2 complex samples per ds_write_b128, natural order
      const v2f rs = mkv(p.rots[2 * s], p.rots[2 * s + 1]);
#pragma unroll
      for (int h = 0; h < OPT / 2; h++) {
        const v2f z0 = pk_cmul(pk_cmul(acc[2 * h], rotu[2 * h]), rs);
        const v2f z1 = pk_cmul(pk_cmul(acc[2 * h + 1], rotu[2 * h + 1]), rs);
        v4f o;
        o.x = z0.x; o.y = z0.y; o.z = z1.x; o.w = z1.y;
        *reinterpret_cast<v4f*>(tile + 2 * (s * SUBOUT + OPT * j + 2 * h)) = o;
      }
    }
    __syncthreads();

    // The transforms (uc_xform.hpp) ping-pong between the two tiles (tile -> tb -> tile -> tb -> tile):
    // every exchange is write, ONE barrier, read.
#if UC_STREAM_KNOCK & 4
    v2f y16[16];
#pragma unroll
    for (int t = 0; t < 16; t++) y16[t] = lds_ld(tile, j + T * t);
#else
    {
      v2f v[16];  // forward pass 1
#pragma unroll
      for (int t = 0; t < 16; t++) v[t] = lds_ld(tile, j + T * t);
      __builtin_amdgcn_sched_barrier(0);
      pk_dft16(v, K, H);
      xf_store1(tb, xa, s1v, v);
    }
    __syncthreads();
    xf_fwd2(tb, tile, tw2t, xa, j, K, H);                       // forward pass 2
    __syncthreads();
    xf_fwd3_h_invA<false>(tile, tb, hres, hres, t3a, t3b, t3c, j, K, H);  // forward pass 3, x H/N, inverse pass A
    __syncthreads();
    xf_invB(tb, tile, twBt, xa, j, K, H);                       // inverse pass B
    __syncthreads();
    v2f y16[16];                                                // inverse pass C: y16[t] = output j + 128 t
    xf_invC<false>(tile, y16, xa, y16, t3a, t3b, t3c, K, H);
#endif

    // ---- |y[i]|, i = j + 128 t; outputs i >= L-1 are free of circular wrap-around ---------------------
    const size_t q0 = (size_t)b * (size_t)HOP;               // first output of this block
    const size_t room = p.n_out - q0;                        // > 0
    const int valid = room < (size_t)HOP ? (int)room : HOP;  // outputs of this block that exist
    // Branch-free: the stores go through a buffer resource that covers exactly this block's `valid`
    // outputs, so the wrapped-around head (o < 0: a huge unsigned offset) and the tail beyond the
    // stream are dropped by the range check; v_sqrt_f32 (1 ulp) is far inside the stated tolerance.
    float best = -1.0f;
    int best_i = 0x7fffffff;
    // no output buffer: a resource of zero records drops every store
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.compressed ? p.compressed + q0 : nullptr, p.compressed ? valid * 4 : 0);
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const v2f y = y16[t];
      const float m2 = y.x * y.x + y.y * y.y;
      const int o = j + T * t - (L - 1);  // offset inside the block's hop
#if !(UC_STREAM_KNOCK & 8)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_amdgcn_sqrtf(m2)), ro, o * 4, 0, UC_STREAM_STORE_CPOL);
#endif
      const bool take = (unsigned)o < (unsigned)valid && m2 > best;  // ascending offset: first maximum
      best = take ? m2 : best;
      best_i = take ? o : best_i;
    }
    if (p.peaks) {
      const float wm = wave_max_f32(best);
      const int cand = (best == wm) ? best_i : 0x7fffffff;
      const int wi = wave_min_u32(cand);
      if ((j & 63) == 0) {
        red[2 * wave] = wm;
        red[2 * wave + 1] = __int_as_float(wi);
      }
      __syncthreads();
      if (j == 0) {
        const float v0 = red[0], v1 = red[2];
        const int i0 = __float_as_int(red[1]), i1 = __float_as_int(red[3]);
        const bool second = v1 > v0 || (v1 == v0 && i1 < i0);
        uc_peak pk;
        pk.value = __builtin_sqrtf(second ? v1 : v0);
        pk.offset = (uint32_t)(second ? i1 : i0);
        p.peaks[b] = pk;
      }
    }
    if (!more) break;
    b = bn;
    if (hop) bend = b + gmask + 1u < nblk ? b + gmask + 1u : nblk;
  }
  if (dyn && j == 0) handout_leave(p.work_ctr);  // the last workgroup out leaves the counter at zero for the next launch
  UC_CLOCK_END(p.debug, 2);
}

template <int DTYPE, int D>
int launch_one(const StreamParams& p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL((stream_kernel<DTYPE, D>), dim3((unsigned)grid), dim3((unsigned)T), 0, stream, p);
  return (int)hipGetLastError();
}

template <int DTYPE, int D>
int occupancy_one() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stream_kernel<DTYPE, D>, T, 0) != hipSuccess || nb <= 0) nb = 4;
  return nb;
}

}  // namespace

UC_LAUNCH_BEGIN
int launch_stream(int dtype, int decim, const StreamParams& p, int grid, hipStream_t stream) {
  if (grid <= 0) return (int)hipSuccess;
  const bool i32 = dtype == UC_DTYPE_I32;
  switch (decim) {
    case 4: return i32 ? launch_one<UC_DTYPE_I32, 4>(p, grid, stream) : launch_one<UC_DTYPE_F32, 4>(p, grid, stream);
    case 8: return i32 ? launch_one<UC_DTYPE_I32, 8>(p, grid, stream) : launch_one<UC_DTYPE_F32, 8>(p, grid, stream);
    case 16: return i32 ? launch_one<UC_DTYPE_I32, 16>(p, grid, stream) : launch_one<UC_DTYPE_F32, 16>(p, grid, stream);
    default: return (int)hipErrorInvalidValue;
  }
}

int stream_max_blocks_per_cu(int dtype, int decim) {
  const bool i32 = dtype == UC_DTYPE_I32;
  switch (decim) {
    case 4: return i32 ? occupancy_one<UC_DTYPE_I32, 4>() : occupancy_one<UC_DTYPE_F32, 4>();
    case 8: return i32 ? occupancy_one<UC_DTYPE_I32, 8>() : occupancy_one<UC_DTYPE_F32, 8>();
    case 16: return i32 ? occupancy_one<UC_DTYPE_I32, 16>() : occupancy_one<UC_DTYPE_F32, 16>();
    default: return 4;
  }
}

UC_LAUNCH_END

}  // namespace uc
