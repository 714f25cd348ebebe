// uc_cic_kernel.hip -- the DFSDM front end as a kernel: sinc^5, decimate by 32, of a 1-bit PDM stream.
//
// Models the peripheral configured in receiver/Src/dfsdm.c:59-61 (SINC5, Oversampling 32,
// IntOversampling 1), :69 (bit clock 80 MHz / 32), :78 (RightBitShift 2) whose output words
// (24-bit result in bits 31:8) the ISR hands to the DSP (receiver/Src/main.c:659-668).
// Integer arithmetic: the result is exact (the parity tests demand bit equality with the CPU restatement).
//
// Design (MI355X): one output word per input word (32 PDM bits), 4 B in + 4 B out: HBM-bound
// integer work.  y[m] = sum_j h[j] s[32 m + 31 - j] (156 taps, s = +-1) touches words m-4 .. m.
// A lane owns FOUR consecutive words (one 16-byte load).  For each of its words it looks up, per byte,
// the contribution of that byte to each of the five outputs the word takes part in (a 20 KiB
// table in LDS: [byte position][byte value] -> 5 partial sums, read as one b128 + one b32), adds
// them up, and passes the four partial sums that belong to the NEXT lane's outputs along with one
// DPP wave shift each.  A wave covers 256 words and stores 252 outputs (lane 0 only feeds lane 1:
// its own outputs belong to the previous tile), coalesced 16-byte stores.
#include "uc_dev.hpp"
#include "uc_kernels.hpp"

namespace uc {

namespace {

#ifndef UC_CIC_KNOCK
#define UC_CIC_KNOCK 0  // diagnostic builds: 1 = no stores, 2 = no HBM reads (profiles/r02_sinc5_notes.txt)
#endif
#ifndef UC_CIC_THREADS
#define UC_CIC_THREADS 1024
#endif
constexpr int TC = UC_CIC_THREADS;  // one workgroup per CU shares one set of tables
constexpr int kTileWords = 256;     // words one wave loads
constexpr int kTileOut = 252;       // outputs one wave stores
// The lookups are indexed by DATA bytes, so with a plain table the bank a lane hits is random (58 % of the LDS
// cycles were bank conflicts, r01).  Here the bank is a function of the LANE alone:
//   * every entry is replicated (4 times in the 16-byte table, 8 times in the 4-byte table), a lane reads replica
//     lane & 3 resp. lane & 7;
//   * the lanes that share a replica inside one LDS access group never look at the same BYTE POSITION in the same
//     instruction: in step i a lane handles byte b = (i + (lane >> 3)) & 3 of its words.  A ds_read_b128 is served
//     in groups of 16 lanes -- {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS): four runs of four lanes with four
//     different lane >> 3, so (b, lane & 3) takes all 16 values; a ds_read_b32 in groups of 32 lanes: (b, lane & 7)
//     takes all 32 values;
//   * the tables are laid out so that (b, replica) picks the bank and the byte value only the ROW, and a row is 256
//     bytes in BOTH tables, so that each address is ONE v_perm_b32 of the data word (byte 1 = the data byte):
//       16-byte table: entry (b, v), replica r at byte            v * 256 + (4 b + r) * 16
//        4-byte table: entry (b, v), replica r at byte  0x10000 + v * 256 + (8 b + r) * 4      (half of each row unused)
// 64 KiB + 64 KiB of the CU's 160 KiB, one 1024-thread workgroup per CU.
constexpr int R4 = 4;
constexpr int R1 = 8;
constexpr size_t kT1Base = 0x10000;
constexpr size_t kCicLdsBytes = 2 * 0x10000;

typedef int v4i __attribute__((ext_vector_type(4)));

// LDS access by ABSOLUTE byte address: the v_perm_b32 result is the address itself (the kernel's dynamic LDS starts at
// byte 0 -- it declares no static LDS; checked once at kernel start), no base to add per lookup
template <typename V>
__device__ __forceinline__ const __attribute__((address_space(3))) V* lds_at(unsigned byte_address) {
  return reinterpret_cast<const __attribute__((address_space(3))) V*>(static_cast<uintptr_t>(byte_address));
}

// value of lane - 1 (lane 0 receives 0)
__device__ __forceinline__ int from_prev_lane(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// MULTI (uc_dfsdm_sinc5_streams): many microphones at once, the filter history of every one CARRIED between calls instead
// of lying in front of the samples.  Stream s = p.n_words new words at p.pdm + s * p.stride, its four history words at
// p.hist + 4 s, its p.n_words outputs at p.out + s * p.out_stride.  A stream is cut into tiles of 252 outputs as the single
// stream is; tile T = (stream T / tps, tile T % tps); lane 0 of a stream's FIRST tile takes its four words from the
// history array (a second, 16-byte resource: every other lane is out of its range and reads zeros, lane 0 is out of range
// of the sample resource -- the two loads are OR-ed, no branch), everything behind the load is the single-stream code.
// (The history array is brought up to date by hist_kernel behind this launch: no tile reads what another one writes.)
template <bool MULTI>
__global__ __launch_bounds__(TC) void sinc5_kernel(const CicParams p) {
#ifdef UC_CLOCKSTAMP
  const unsigned long long clk0_ = __builtin_readcyclecounter();
  const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char cic_lds[];
  if ((unsigned)reinterpret_cast<uintptr_t>(cic_lds) != 0u) __builtin_trap();  // (low half of a flat LDS address = the LDS offset)
  // entry e = 256 b + v of the host tables -> the replicated, bank-steered LDS layout.  The -2^24 of
  // y = 2 B - 2^25 rides in the table: every output sums exactly one "word m, byte 0, output m" entry.
  for (int i = threadIdx.x; i < 1024 * R1; i += TC) {
    const int e = i >> 3, r = i & 7, b = e >> 8, v = e & 255;
    if (r < R4) {
      v4i q = reinterpret_cast<const v4i*>(p.t4)[e];
      if (b == 0) q.x -= 1 << 24;
      *reinterpret_cast<v4i*>(cic_lds + v * 256 + (4 * b + r) * 16) = q;
    }
    *reinterpret_cast<int*>(cic_lds + kT1Base + v * 256 + (8 * b + r) * 4) = p.t1[e];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  // step i of a word: byte b = (i + (lane >> 3)) & 3.  Both addresses are v_perm_b32(word, constant, selector):
  // byte 0 and byte 2 from the lane's constant (slot inside the row, table base), byte 1 = data byte b, byte 3 = 0.
  unsigned sel[4], c4[4], c1[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const unsigned b = (unsigned)(i + (lane >> 3)) & 3u;
    sel[i] = 0x0c020000u | ((4u + b) << 8);              // {0, const.byte2, word.byte b, const.byte0}
    c4[i] = (4u * b + ((unsigned)lane & 3u)) << 4;
    c1[i] = (unsigned)kT1Base | ((8u * b + ((unsigned)lane & 7u)) << 2);
  }
  const size_t n_out = MULTI ? p.n_words : p.n_words - 4;  // (MULTI: per stream)
  const size_t tiles = MULTI ? (size_t)p.tps * p.n_streams : (n_out + kTileOut - 1) / kTileOut;
  // MULTI: tile -> (stream, tile of the stream), tile < 2^31
  auto split = [&](size_t tile, size_t& s, size_t& t) {
    const unsigned u = (unsigned)tile;
    const unsigned q = (__umulhi(u, p.div_magic) + u) >> p.div_shift;
    s = q;
    t = u - q * p.tps;
  };
  // (readfirstlane: the compiler cannot see that threadIdx.x >> 6 is the same in every lane; without it the tile
  // index, the buffer resources and all the 64-bit address arithmetic live in VGPRs and every buffer access is
  // wrapped in a waterfall loop)
  const size_t wave0 = (size_t)blockIdx.x * (TC / 64) + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t nwaves = (size_t)gridDim.x * (TC / 64);

  // words base + 4 lane .. + 3 of a tile; past the end of the buffer the resource returns 0 (those outputs
  // are not stored), and so does a tile beyond the last one
  auto load_tile = [&](size_t tile) -> v4u {
    if constexpr (MULTI) {
      if (tile >= tiles) return v4u{0u, 0u, 0u, 0u};
      size_t s, t;
      split(tile, s, t);
      const uint32_t* stream = p.pdm + s * p.stride;
      if (t == 0) {
        // words -4 .. -1 of the stream are its carried history: lane 0 reads them, lanes 1 .. 63 read words 4 (lane - 1) ..
        const int recs = p.n_words < (size_t)kTileOut ? (int)p.n_words : kTileOut;
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(stream, recs * 4);
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(p.hist + 4 * s, 16);
        const v4u a = __builtin_amdgcn_raw_buffer_load_b128(rin, (lane - 1) * 16, 0, UC_STREAM_CPOL);
        const v4u h = __builtin_amdgcn_raw_buffer_load_b128(rh, lane * 16, 0, 0);
        return a | h;
      }
      const size_t first = t * kTileOut - 4;
      const size_t left = p.n_words - first;
      const int recs = left < (size_t)kTileWords ? (int)left : kTileWords;
      const __amdgpu_buffer_rsrc_t rin = make_rsrc(stream + first, recs * 4);
      return __builtin_amdgcn_raw_buffer_load_b128(rin, lane * 16, 0, UC_STREAM_CPOL);
    }
    const size_t base = tile * kTileOut;  // first word of the tile = first output of the tile + 4 - 4
    const size_t left = tile < tiles ? p.n_words - base : 0;
    const int recs = left < (size_t)kTileWords ? (int)left : kTileWords;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(p.pdm + (tile < tiles ? base : 0), recs * 4);
#if UC_CIC_KNOCK == 2  // (knock-out build: every tile re-reads the stream's first KiB -- no HBM reads)
    const __amdgpu_buffer_rsrc_t rin0 = make_rsrc(p.pdm, recs * 4);
    return __builtin_amdgcn_raw_buffer_load_b128(rin0, lane * 16, 0, 0);
#else
    return __builtin_amdgcn_raw_buffer_load_b128(rin, lane * 16, 0, UC_STREAM_CPOL);
#endif
  };
  // One tile of a wave: 256 words in, 252 outputs out.
  auto process = [&](const v4u w, size_t tile) {
    size_t base = tile * kTileOut;
    const int32_t* out_base = p.out;
    if constexpr (MULTI) {
      size_t s = 0, t = 0;
      if (tile < tiles) split(tile, s, t);
      base = t * kTileOut;
      out_base = p.out + s * p.out_stride;
    }
    const unsigned wd[4] = {w.x, w.y, w.z, w.w};
    int g[4][5];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      int a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const v4i q = *lds_at<v4i>(__builtin_amdgcn_perm(wd[c], c4[i], sel[i]));
        a0 += q.x; a1 += q.y; a2 += q.z; a3 += q.w;
        a4 += *lds_at<int>(__builtin_amdgcn_perm(wd[c], c1[i], sel[i]));
      }
      g[c][0] = a0; g[c][1] = a1; g[c][2] = a2; g[c][3] = a3; g[c][4] = a4;
      if (c & 1) __builtin_amdgcn_sched_barrier(0);  // two words' lookups (16 reads, 40 result registers) at a time
    }
    // sums over this lane's own words, and what its words add to the next lane's four outputs
    int y0 = g[0][0];
    int y1 = g[1][0] + g[0][1];
    int y2 = g[2][0] + g[1][1] + g[0][2];
    int y3 = g[3][0] + g[2][1] + g[1][2] + g[0][3];
    const int c0 = g[3][1] + g[2][2] + g[1][3] + g[0][4];
    const int c1 = g[3][2] + g[2][3] + g[1][4];
    const int c2 = g[3][3] + g[2][4];
    const int c3 = g[3][4];
    y0 += from_prev_lane(c0);
    y1 += from_prev_lane(c1);
    y2 += from_prev_lane(c2);
    y3 += from_prev_lane(c3);
    // B = sum of the taps that met a 1 bit: y = 2 B - 2^25; result = clip(y >> 2) << 8  (bsum = B - 2^24: see the fill)
    auto word = [](int bsum) {
      int v = bsum >> 1;
      v = v > 8388607 ? 8388607 : v;
      v = v < -8388608 ? -8388608 : v;
      return v * 256;
    };
    // outputs base + 4 (lane - 1) .. + 3 through a buffer resource over this tile's outputs: lane 0 (its outputs
    // belong to the previous tile), lanes past the end of the stream and tiles past the last one fall outside the
    // resource and are dropped by the range check -- no branch, no 64-bit address registers.
    // The hardware contract this rests on (raw buffer, num_records in bytes): the range check of a b128 store is made
    // PER DWORD, so a lane whose four words straddle the end of a ragged stream (n_out % 4 != 0) stores the words
    // inside and drops the rest; lane 0's offset (lane - 1) * 16 = 0xFFFFFFF0 is out of range on purpose.
    // Pinned by tests/test_dfsdm.py::test_sinc5_ragged_tails_on_the_device_never_write_past_the_end (guard words).
    const size_t left = tile < tiles ? n_out - base : 0;
    const int recs = left < (size_t)kTileOut ? (int)left : kTileOut;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out_base + (tile < tiles ? base : 0), recs * 4);
    v4u r;
    r.x = (unsigned)word(y0); r.y = (unsigned)word(y1); r.z = (unsigned)word(y2); r.w = (unsigned)word(y3);
#if UC_CIC_KNOCK == 1  // (knock-out build: one lane of 63 stores)
    if (lane == 1) __builtin_amdgcn_raw_buffer_store_b128(r, rout, (lane - 1) * 16, 0, UC_STREAM_CPOL);  // 1/63 of the bytes
#else
    __builtin_amdgcn_raw_buffer_store_b128(r, rout, (lane - 1) * 16, 0, UC_STREAM_CPOL);  // written once, never read here
#endif
  };
  // Workgroup b owns the tiles b, b + B, b + 2 B, ... (B workgroups); its 16 waves do NOT run at the same speed (the
  // SIMD arbitration favours the older wave: with equal static shares the slowest wave of a workgroup ran 1.5 x as
  // long as the fastest, profiles/r02_sinc5_skew_static.json), so they draw tickets k = 0, 1, 2, ... from the
  // workgroup's counter in global memory (p.ctr[b], zero at launch; LDS is full): ticket k = tile b + k B.
  // The loads run two tiles ahead, the ticket for the next load one iteration ahead of its use; the returning
  // atomic is always OLDER than the loads still in flight, so reading it never drains the load pipeline.
  const size_t nblocks = gridDim.x;
  const unsigned my_tiles = (unsigned)((tiles - blockIdx.x + nblocks - 1) / nblocks);  // tickets of this workgroup
  auto tile_of = [&](unsigned k) { return k < my_tiles ? (size_t)blockIdx.x + (size_t)k * nblocks : tiles; };
  unsigned* ctr = p.ctr ? p.ctr + blockIdx.x : nullptr;
  const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // one ticket = kTicket consecutive tiles of the workgroup (an atomic takes several microseconds to return while
  // the chip streams: with a ticket per tile every iteration waited for its atomic, 1.3 ms instead of 0.49 ms)
  constexpr unsigned kTicket = 4;
  auto ticket = [&]() -> unsigned {  // returns in lane 0 of a VGPR; made scalar when it is used
    return (lane == 0) ? atomicAdd(ctr, 1u) + (unsigned)(TC / 64) : 0u;
  };
  if (ctr) {
    unsigned kl = kTicket * wv;     // tile (ticket-local numbering) of the next LOAD; every wave's first ticket is fixed
    unsigned kn_v = ticket();       // the ticket after it, in flight (issued before the loads: see below)
    auto next_k = [&]() {           // advance kl; at a ticket boundary take the ticket in flight and ask for another
      if (((kl + 1) & (kTicket - 1)) != 0) {
        kl++;
      } else {
        kl = kTicket * (unsigned)__builtin_amdgcn_readfirstlane((int)kn_v);
        kn_v = ticket();            // BEFORE the next load: vector-memory operations return in order, so reading
      }                             // this ticket later must not have to wait for loads issued after it
    };
    unsigned k0 = kl;
    v4u q0 = load_tile(tile_of(kl));
    next_k();
    unsigned k1 = kl;
    v4u q1 = load_tile(tile_of(kl));
    for (;;) {
      if (k0 >= my_tiles) break;
      const v4u w = q0;
      q0 = q1;
      next_k();
      const unsigned kn = kl;
      q1 = load_tile(tile_of(kn));
      process(w, tile_of(k0));
      __builtin_amdgcn_sched_barrier(0);
      k0 = k1;
      k1 = kn;
    }
  } else {
    // static deal (no counter): tiles wave, wave + W, ... of all W waves of the grid, loads two tiles ahead
    v4u q0 = load_tile(wave0), q1 = load_tile(wave0 + nwaves);
    for (size_t tile = wave0; tile < tiles; tile += nwaves) {
      const v4u w = q0;
      q0 = q1;
      q1 = load_tile(tile + 2 * nwaves);
      process(w, tile);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#ifdef UC_CLOCKSTAMP
  if (lane == 0 && p.debug) {
    const unsigned long long rt1_ = __builtin_amdgcn_s_memrealtime();
    unsigned long long* d_ = p.debug + ((size_t)blockIdx.x * (TC / 64) + (threadIdx.x >> 6)) * 4;
    d_[0] = __builtin_readcyclecounter() - clk0_;
    d_[1] = rt1_ - rt0_;
    d_[2] = rt0_;
    d_[3] = rt1_;
  }
#endif
}

// the filter history a call of uc_dfsdm_sinc5_streams leaves behind: the last four words of [history | the call's words]
__global__ __launch_bounds__(256) void hist_kernel(const CicParams p) {
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= p.n_streams) return;
  uint32_t* h = const_cast<uint32_t*>(p.hist) + 4 * s;
  const uint32_t* w = p.pdm + s * p.stride;
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {  // word n_words - 4 + k of the stream, counted from its first NEW word
    const long long i = (long long)p.n_words - 4 + k;
    v[k] = i >= 0 ? w[i] : h[4 + i];
  }
#pragma unroll
  for (int k = 0; k < 4; k++) h[k] = v[k];
}

}  // namespace

UC_LAUNCH_BEGIN
int launch_sinc5(const CicParams& p, int grid, hipStream_t stream) {
  if (p.n_streams) {  // uc_dfsdm_sinc5_streams
    if (grid <= 0 || p.n_words == 0) return (int)hipSuccess;
    hipLaunchKernelGGL(sinc5_kernel<true>, dim3((unsigned)grid), dim3((unsigned)TC), kCicLdsBytes, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)((p.n_streams + 255) / 256)), dim3(256), 0, stream, p);
    return (int)hipGetLastError();
  }
  if (grid <= 0 || p.n_words <= 4) return (int)hipSuccess;
  hipLaunchKernelGGL(sinc5_kernel<false>, dim3((unsigned)grid), dim3((unsigned)TC), kCicLdsBytes, stream, p);
  return (int)hipGetLastError();
}

// Called once per context on its device, before the first launch: more than the default 64 KiB of
// dynamic LDS needs the opt-in; returns the resident workgroups per CU (0 if the opt-in fails).
int sinc5_max_blocks_per_cu() {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(sinc5_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kCicLdsBytes) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(sinc5_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kCicLdsBytes) != hipSuccess)
    return 0;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sinc5_kernel<false>, TC, kCicLdsBytes) != hipSuccess || nb <= 0) nb = 1;
  return nb;
}

int sinc5_tile_outputs() { return kTileOut * (TC / 64); }
int sinc5_waves_per_block() { return TC / 64; }

UC_LAUNCH_END

}  // namespace uc
