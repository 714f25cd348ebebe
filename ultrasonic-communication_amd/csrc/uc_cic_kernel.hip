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
// DPP wave shift each.  A wave covers a tile of 256 words and stores 256 outputs, coalesced 16-byte
// stores; what the tile's last four words add to the next tile travels in four SGPRs (the same wave
// works on that tile next), what lies in front of a wave's first tile is looked up for 64 such
// places at once.
#include "uc_dev.hpp"
#include "uc_kernels.hpp"

namespace uc {

namespace {

#ifndef UC_CIC_LOAD_CPOL
#define UC_CIC_LOAD_CPOL UC_STREAM_CPOL
#endif
#ifndef UC_CIC_THREADS
#define UC_CIC_THREADS 1024
#endif
constexpr int TC = UC_CIC_THREADS;  // one workgroup per CU shares one set of tables
constexpr int kTileWords = 256;     // words one wave loads = outputs it stores
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

// what every lane carries: step i of a word handles byte b = (i + (lane >> 3)) & 3.  Both addresses of a
// step are v_perm_b32(word, constant, selector): byte 0 and byte 2 from the lane's constant (slot inside the row, table
// base), byte 1 = data byte b, byte 3 = 0.
struct Steer {
  unsigned sel[4], c4[4], c1[4];
};
__device__ __forceinline__ Steer make_steer(int lane) {
  Steer st;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const unsigned b = (unsigned)(i + (lane >> 3)) & 3u;
    st.sel[i] = 0x0c020000u | ((4u + b) << 8);              // {0, const.byte2, word.byte b, const.byte0}
    st.c4[i] = (4u * b + ((unsigned)lane & 3u)) << 4;
    st.c1[i] = (unsigned)kT1Base | ((8u * b + ((unsigned)lane & 7u)) << 2);
  }
  return st;
}

// entry e = 256 b + v of the host tables -> the replicated, bank-steered LDS layout.  The -2^24 of y = 2 B - 2^25 rides in
// the table: every output sums exactly one "word m, byte 0, output m" entry.
__device__ __forceinline__ void fill_tables(const CicParams& p, unsigned char* cic_lds) {
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
}

// The lookups of a lane's four words w: y[k] = what they add to the lane's own output k, c[k] = what they add to output k
// of the NEXT four words (the next lane's, or the next tile's first lane's).
__device__ __forceinline__ void lookups(const v4u w, const Steer& st, int y[4], int c[4]) {
  const unsigned wd[4] = {w.x, w.y, w.z, w.w};
  int g[4][5];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const v4i q = *lds_at<v4i>(__builtin_amdgcn_perm(wd[k], st.c4[i], st.sel[i]));
      a0 += q.x; a1 += q.y; a2 += q.z; a3 += q.w;
      a4 += *lds_at<int>(__builtin_amdgcn_perm(wd[k], st.c1[i], st.sel[i]));
    }
    g[k][0] = a0; g[k][1] = a1; g[k][2] = a2; g[k][3] = a3; g[k][4] = a4;
    if (k & 1) __builtin_amdgcn_sched_barrier(0);  // two words' lookups (16 reads, 40 result registers) at a time
  }
  y[0] = g[0][0];
  y[1] = g[1][0] + g[0][1];
  y[2] = g[2][0] + g[1][1] + g[0][2];
  y[3] = g[3][0] + g[2][1] + g[1][2] + g[0][3];
  c[0] = g[3][1] + g[2][2] + g[1][3] + g[0][4];
  c[1] = g[3][2] + g[2][3] + g[1][4];
  c[2] = g[3][3] + g[2][4];
  c[3] = g[3][4];
}

// B = sum of the taps that met a 1 bit: y = 2 B - 2^25; result = clip(y >> 2) << 8  (bsum = B - 2^24: see fill_tables)
__device__ __forceinline__ unsigned word24(int bsum) {
  int v = bsum >> 1;
  v = v > 8388607 ? 8388607 : v;
  v = v < -8388608 ? -8388608 : v;
  return (unsigned)(v * 256);
}

// Streams s = 0 .. p.n_streams - 1: p.n_words new words at p.pdm + s * p.stride, the four words in front of them -- the
// filter history, CARRIED between calls instead of lying in front of the samples -- at p.hist + 4 s, p.n_words outputs at
// p.out + s * p.out_stride.  (uc_dfsdm_sinc5, one recorded stream whose first four words are its history, is the case
// n_streams = 1, hist = the buffer, pdm = the buffer + 4.)
//
// No overlap between tiles: a live block is 2048 words = EIGHT tiles of 256 (tiles that overlap by the four words of
// history, rounds 1-4, made it eight tiles of 252 and a ninth that carried 32: 11 % of the lookups on lanes without
// outputs).  A stream is cut into SEGMENTS of p.tps tiles (<= 8) and ONE wave walks a segment front to back: what the last
// four words of a tile add to the next four outputs (lane 63's c[]) goes to lane 0 of the next tile through four SGPRs.
// What the four words IN FRONT of a segment add to its first four outputs -- the carried history (segment 0) or the tail
// of the segment before -- a wave works out for 64 of its segments at a time, one per lane (the boundary pass: one extra
// tile's worth of lookups per 64 segments), and hands to lane 0 of each segment's first tile with a v_readlane.
// Segments are dealt statically: wave w of W takes segments w, w + W, ...  (The history array is brought up to date by
// hist_kernel behind this launch: no wave reads what another one writes.)
__global__ __launch_bounds__(TC) void sinc5_kernel(const CicParams p) {
#ifdef UC_CLOCKSTAMP
  const unsigned long long clk0_ = __builtin_readcyclecounter();
  const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char cic_lds[];
  if ((unsigned)reinterpret_cast<uintptr_t>(cic_lds) != 0u) __builtin_trap();
  fill_tables(p, cic_lds);

  const int lane = threadIdx.x & 63;
  const Steer st = make_steer(lane);
  const unsigned W = gridDim.x * (unsigned)(TC / 64);
  const unsigned wave = blockIdx.x * (unsigned)(TC / 64) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned mine = wave < p.units ? (p.units - wave + W - 1) / W : 0;  // segments of this wave: wave + i W, i < mine
  const unsigned tp = p.tps;
  // segment u -> (stream, segment of the stream); u < 2^31
  auto split = [&](unsigned u, unsigned& s, unsigned& g) {
    s = (__umulhi(u, p.div_magic) + u) >> p.div_shift;
    g = u - s * p.nseg;
  };
  // tile slot j of this wave's i-th segment: where its words and outputs lie and how many there are (0: none -- the
  // slot lies behind the end of the stream, or the wave has run out of segments)
  struct Slot {
    const uint32_t* in;
    int32_t* out;
    int recs;
  };
  auto slot = [&](unsigned i, unsigned j) -> Slot {
    Slot r{p.pdm, p.out, 0};
    if (i < mine) {
      unsigned s, g;
      split(wave + i * W, s, g);
      const size_t first = ((size_t)g * tp + j) * kTileWords;
      if (first < p.n_words) {
        const size_t left = p.n_words - first;
        r.recs = left < (size_t)kTileWords ? (int)left : kTileWords;
        r.in = p.pdm + (size_t)s * p.stride + first;
        r.out = p.out + (size_t)s * p.out_stride + first;
      }
    }
    return r;
  };
  auto advance = [&](unsigned& i, unsigned& j) {
    if (++j == tp) {
      j = 0;
      i++;
    }
  };
  // words 4 lane .. 4 lane + 3 of a slot; behind the end of the stream (and in an empty slot) the resource returns 0
  auto load = [&](unsigned i, unsigned j) -> v4u {
    const Slot t = slot(i, j);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(t.in, t.recs * 4);
    return __builtin_amdgcn_raw_buffer_load_b128(rin, lane * 16, 0, UC_CIC_LOAD_CPOL);
  };

  int cin[4] = {0, 0, 0, 0};    // lane l: what lies in front of segment (i & ~63) + l adds to its first four outputs
  int carry[4] = {0, 0, 0, 0};  // scalars: lane 63's c[] of the tile before
  unsigned il = 0, jl = 0;      // the slot of the next load: two ahead of the one being worked on
  unsigned i = 0, j = 0;        // the slot being worked on
  // one slot: start the load two slots ahead into `fill` (the register set whose tile was consumed last), work on `w`
  auto step = [&](const v4u& w, v4u& fill) -> bool {
    advance(il, jl);
    fill = load(il, jl);
    if (j == 0 && (i & 63u) == 0) {
      // boundary pass: lane l looks at the four words in front of segment i + l
      const unsigned k = i + (unsigned)lane;
      v4u b = {0u, 0u, 0u, 0u};
      if (k < mine) {
        unsigned s, g;
        split(wave + k * W, s, g);
        const uint32_t* src = g ? p.pdm + (size_t)s * p.stride + (size_t)g * tp * kTileWords - 4 : p.hist + 4 * (size_t)s;
        b = *reinterpret_cast<const v4u*>(src);
      }
      int y[4];
      lookups(b, st, y, cin);
    }
    const Slot t = slot(i, j);
    if (t.recs) {
      int y[4], c[4];
      lookups(w, st, y, c);
      v4u r;
      unsigned out[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int in = j == 0 ? __builtin_amdgcn_readlane(cin[k], (int)(i & 63u)) : carry[k];
        // lane l > 0 takes c[k] of lane l - 1; lane 0 has no source lane and keeps the DPP's `old` operand: `in`
        y[k] += __builtin_amdgcn_update_dpp(in, c[k], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        carry[k] = __builtin_amdgcn_readlane(c[k], 63);
        out[k] = word24(y[k]);
      }
      r.x = out[0]; r.y = out[1]; r.z = out[2]; r.w = out[3];
      // The hardware contract this rests on (raw buffer, num_records in bytes): the range check of a b128 store is made PER
      // DWORD, so a lane whose four words straddle the end of a ragged stream (n_words % 4 != 0) stores the words inside and
      // drops the rest; lanes behind the end store nothing -- no branch, no 64-bit address registers.  Pinned by
      // tests/test_dfsdm.py::test_sinc5_ragged_tails_on_the_device_never_write_past_the_end (guard words).
      const __amdgpu_buffer_rsrc_t rout = make_rsrc(t.out, t.recs * 4);
      __builtin_amdgcn_raw_buffer_store_b128(r, rout, lane * 16, 0, UC_STREAM_CPOL);
    }
    __builtin_amdgcn_sched_barrier(0);
    advance(i, j);
    return i < mine;
  };
  // three register sets in rotation (a loop that hands q1 to q0 to w by copies has to wait for ALL loads in flight
  // before it may copy: the load just started would be waited for at once)
  v4u qa = load(il, jl), qb, qc;
  advance(il, jl);
  qb = load(il, jl);
  if (mine) {
    for (;;) {
      if (!step(qa, qc)) break;
      if (!step(qb, qa)) break;
      if (!step(qc, qb)) break;
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
  if (grid <= 0 || p.n_words == 0 || p.n_streams == 0) return (int)hipSuccess;
  hipLaunchKernelGGL(sinc5_kernel, dim3((unsigned)grid), dim3((unsigned)TC), kCicLdsBytes, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !p.update_hist) return (int)e;
  hipLaunchKernelGGL(hist_kernel, dim3((unsigned)((p.n_streams + 255) / 256)), dim3(256), 0, stream, p);
  return (int)hipGetLastError();
}

// Called once per context on its device, before the first launch: more than the default 64 KiB of
// dynamic LDS needs the opt-in; returns the resident workgroups per CU (0 if the opt-in fails).
int sinc5_max_blocks_per_cu() {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(sinc5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kCicLdsBytes) != hipSuccess)
    return 0;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sinc5_kernel, TC, kCicLdsBytes) != hipSuccess || nb <= 0) nb = 1;
  return nb;
}

int sinc5_waves_per_block() { return TC / 64; }

UC_LAUNCH_END

}  // namespace uc
