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

#ifndef UC_CIC_THREADS
#define UC_CIC_THREADS 1024
#endif
#ifndef UC_CIC_R4
#define UC_CIC_R4 8
#endif
#ifndef UC_CIC_R1
#define UC_CIC_R1 4
#endif
constexpr int TC = UC_CIC_THREADS;  // one workgroup per CU shares one set of tables
constexpr int kTileWords = 256;     // words one wave loads
constexpr int kTileOut = 252;       // outputs one wave stores
// The lookups are indexed by DATA bytes: lanes of one LDS access group that need different entries on
// the same banks are serialised (16 random bytes on 16 bank quads: ~3 per quad).  Replicating the
// tables -- entry e of replica r at e * R + r, a lane reads replica lane % R -- spreads a group over
// R times as many bank positions per entry: 2 lanes of a 16-lane group share a replica of the 16-byte
// table (R4 = 8, 128 KiB), 8 lanes of a 32-lane group a replica of the 4-byte table (R1 = 4, 16 KiB):
// measured best split of the 160 KiB (4/8: 43.7 %, 8/8: 45.0 %, 8/4: 46.9 % of 8 TB/s).
constexpr int R4 = UC_CIC_R4;
constexpr int R1 = UC_CIC_R1;
constexpr size_t kCicLdsBytes = 1024 * (size_t)R4 * 16 + 1024 * (size_t)R1 * 4;

typedef int v4i __attribute__((ext_vector_type(4)));

// value of lane - 1 (lane 0 receives 0)
__device__ __forceinline__ int from_prev_lane(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__global__ __launch_bounds__(TC) void sinc5_kernel(const CicParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cic_lds[];
  v4i* t4 = reinterpret_cast<v4i*>(cic_lds);              // [byte position b][value v][replica] -> outputs m .. m+3 of word m
  int* t1 = reinterpret_cast<int*>(cic_lds + 1024 * R4 * 16);  //                                 -> output m+4
  for (int i = threadIdx.x; i < 1024 * R4; i += TC) t4[i] = reinterpret_cast<const v4i*>(p.t4)[i / R4];
  for (int i = threadIdx.x; i < 1024 * R1; i += TC) t1[i] = p.t1[i / R1];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const size_t n_out = p.n_words - 4;
  const size_t tiles = (n_out + kTileOut - 1) / kTileOut;
  const size_t wave0 = (size_t)blockIdx.x * (TC / 64) + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (TC / 64);

  // words base + 4 lane .. + 3 of a tile; past the end of the buffer the resource returns 0 (those outputs
  // are not stored), and so does a tile beyond the last one
  auto load_tile = [&](size_t tile) {
    const size_t base = tile * kTileOut;  // first word of the tile = first output of the tile + 4 - 4
    const size_t left = tile < tiles ? p.n_words - base : 0;
    const int recs = left < (size_t)kTileWords ? (int)left : kTileWords;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(p.pdm + (tile < tiles ? base : 0), recs * 4);
    return __builtin_amdgcn_raw_buffer_load_b128(rin, lane * 16, 0, UC_STREAM_CPOL);
  };
  // A wave has 1 KiB of input in flight per outstanding load; 16 waves per CU with one load each cannot
  // cover the HBM latency at this rate, so the loads run TWO tiles ahead of the arithmetic.
  v4u w1 = load_tile(wave0), w2 = load_tile(wave0 + nwaves);
  for (size_t tile = wave0; tile < tiles; tile += nwaves) {
    const size_t base = tile * kTileOut;
    const v4u w = w1;
    w1 = w2;
    w2 = load_tile(tile + 2 * nwaves);
    const unsigned wd[4] = {w.x, w.y, w.z, w.w};
    int g[4][5];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      int a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int idx = b * 256 + (int)((wd[c] >> (8 * b)) & 255u);
        const v4i q = t4[idx * R4 + (lane & (R4 - 1))];
        a0 += q.x; a1 += q.y; a2 += q.z; a3 += q.w;
        a4 += t1[idx * R1 + (lane & (R1 - 1))];
      }
      g[c][0] = a0; g[c][1] = a1; g[c][2] = a2; g[c][3] = a3; g[c][4] = a4;
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
    // B = sum of the taps that met a 1 bit: y = 2 B - 2^25; result = clip(y >> 2) << 8
    auto word = [](int bsum) {
      int v = (bsum - (1 << 24)) >> 1;
      v = v > 8388607 ? 8388607 : v;
      v = v < -8388608 ? -8388608 : v;
      return v * 256;
    };
    if (lane > 0) {
      // outputs base + 4 (lane - 1) .. + 3
      const size_t o = base + 4 * (size_t)(lane - 1);
      if (o + 3 < n_out) {
        v4i r;
        r.x = word(y0); r.y = word(y1); r.z = word(y2); r.w = word(y3);
        __builtin_nontemporal_store(r, reinterpret_cast<v4i*>(p.out + o));  // written once, never read here
      } else {
        if (o < n_out) p.out[o] = word(y0);
        if (o + 1 < n_out) p.out[o + 1] = word(y1);
        if (o + 2 < n_out) p.out[o + 2] = word(y2);
      }
    }
  }
}

}  // namespace

int launch_sinc5(const CicParams& p, int grid, hipStream_t stream) {
  if (grid <= 0 || p.n_words <= 4) return (int)hipSuccess;
  hipLaunchKernelGGL(sinc5_kernel, dim3((unsigned)grid), dim3((unsigned)TC), kCicLdsBytes, stream, p);
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

int sinc5_tile_outputs() { return kTileOut * (TC / 64); }

}  // namespace uc
