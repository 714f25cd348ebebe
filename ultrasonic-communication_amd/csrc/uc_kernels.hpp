// uc_kernels.hpp -- launch interface between the C-ABI (uc_api_*.cpp) and the
// gfx950 kernels (uc_band_kernel.hip, uc_full_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/uchirp.h"

namespace uc {

constexpr int kN = 2048;          // frame length all kernels are specialised for
constexpr int kBandThreads = 128; // one 16x16x8 frame per 2-wave workgroup
constexpr int kBandNarrowMax = 191; // largest bandwidth2 of the default band-kernel build (bins 0 .. 191 in two rounds)
constexpr int kBandWideMax = 319;   // ... of the WIDE build (three rounds), the largest window the library evaluates

// Band pipeline: RX_REAL, SYNC_CPLX, DECHIRP_DOWN (windows around DC only).
struct BandParams {
  const void* frames;     // device, int32 or float
  size_t n_frames;
  union {
    size_t stride;        // elements between frame starts
    size_t save_half;     // the ROWS build (frames are not a strided batch there): see save_to
  };
  const float2* tab0;     // RX_REAL: (up*hann, down*hann)[n]   CPLX: (cos,sin)*hann of UP   PAIR: (down*hann, 0)
  const float2* tab1;     // CPLX: (cos,sin)*hann of DOWN
  const float2* tw;       // exp(-2 pi i k / 2048), k < 2048
  const float* mag_mean;  // device, 2 per frame {up,down}, or nullptr
  uint8_t* symbols;       // device or nullptr
  union {
    uc_stats* stats;      // device or nullptr
    void* save_to;        // the ROWS build (writes magmax only): (save != 0) the state's two halves, save_half elements apart,
  };                      // rows kN words apart -- a row's newest block goes to half (1 - *parity)
  float2* magmax;         // device or nullptr: (up, down) mag_max of every frame only (uc_receive_stream's replay needs no
                          // more: 8 instead of 64 bytes per frame to bring back); RX_REAL / SYNC_CPLX
  float* spectrum;        // device or nullptr: |X| of the window bins -bw2 .. +bw2 of every history,
                          // [frame][history][2 bw2 + 1], entry bw2 + k = bin k (k < 0: bin n + k) -- uc_window_spectrum
  uint32_t wide;          // 1 = the WIDE build (bw2 > kBandNarrowMax)
  float mag_mean_scalar;
  float snr_threshold;
  uint32_t bw2;           // window length (<= kBandWideMax; > kBandNarrowMax selects the WIDE build)
  uint32_t ifs;           // (uint32_t)(int32_t)fs for idx2freq
  uint32_t true_dc;       // UC_FLAG_TRUE_DC
  uint32_t group_log2;    // log2 of the frames (PAIR: frame pairs) per group: 1..6 with work_ctr, 0..6 without
  uint32_t unpaired;      // kModePair only: 1 = UC_FLAG_NO_FRAME_PAIRS, one frame per transform
  unsigned int* work_ctr; // two device words, zero at launch and left at zero: groups beyond the first one of each workgroup are handed out
                          // by atomic increments (the workgroups run at different speeds); nullptr = static round robin
  unsigned long long* debug;  // diagnostic builds only (UC_STAMPS), else nullptr
  // ---- the ROWS build only (uc_receive_streams[_next]: the FIFO offsets a NEW block adds, receiver/Src/main.c:659-668) ----
  // `frames` holds rows of row_blocks consecutive blocks of kN samples, row_pitch elements apart: the accepted blocks of
  // one microphone stream each.  The ISR shifts the FIFO by one block per accepted block (main.c:662), so of the 17
  // 256-sample offsets dsp() can visit in the new FIFO only 8 are new: unit u = (row s, block jb, m = 1 .. 8) =
  // ((s * row_blocks + jb) * 8 + m - 1) is the frame that starts 256 m samples into the block IN FRONT of block jb and ends
  // 256 m samples into block jb.  In front of block 0 of row s lies prev + s * prev_pitch (the newest block the FIFO
  // held before this call; prev_pitch = 0: one block of zeros for every row, the FIFO at power-on, main.c:94).
  // n_frames = rows * row_blocks * 8; `stride` and `stats` are not used (their slots carry save_half / save_to).
  // A live state keeps that block in TWO halves, prev_half elements apart, and one device word *parity says which half is
  // current: rows read half *parity, and (save != 0) the frame m = 8 of a row's last block -- which is that block, whole --
  // stores its raw words into the OTHER half of its row on the way through (the replay kernel behind the launch flips
  // *parity).  Nothing is copied by a kernel of its own, no frame reads what another one writes.  parity == nullptr: half 0.
  // Where the block is READ and where it is SAVED are named apart: a caller that keeps its previous chunk alive
  // (uc_rx_state_keep_previous) has the block in front read from that chunk (prev = its last block, prev_half = 0) and
  // nothing saved; save_to / save_half name the state's two halves (rows kN words apart).
  const void* prev;
  size_t row_pitch, prev_pitch, prev_half;
  const unsigned int* parity;
  const uint32_t* need;       // one-block calls of a live state: [rows] what the switch can still look at of the row's NEW
                              // block -- bits 0 .. 7: FIFO offset m = 1 .. 8 is needed, bit 8: the DOWN statistics are needed
                              // (uc_rx.hpp: RxParams::need; main.c:447-451).  nullptr = everything for every row
  uint32_t save;
  uint32_t poison;            // tests only (UC_TUNING=1 UC_RX_POISON=1): what is passed over gets a HUGE statistic instead of zero, so that
                              // a switch that looked at it after all could not fail to show in the trace (tests/test_gpu_receive_many.py)
  uint32_t row_blocks;
  uint32_t div_magic, div_shift;  // u / row_blocks for u < 2^31: (mulhi(u, div_magic) + u) >> div_shift (uc::rows_divisor)
};

// the (magic, shift) pair of BandParams for a divisor d >= 1: floor(u / d) == (mulhi(u, magic) + u) >> shift for every
// u < 2^31 (round-up method: shift = ceil(log2 d), magic = ceil(2^(32 + shift) / d) - 2^32)
inline void rows_divisor(uint32_t d, uint32_t* magic, uint32_t* shift) {
  uint32_t l = 0;  // (d < 2^31, so 2^(32 + l) fits 64 bits)
  while (((uint64_t)1 << l) < d) l++;
  const uint64_t num = (uint64_t)1 << (32 + l);
  const uint64_t m = (num + d - 1) / d - ((uint64_t)1 << 32);
  *magic = (uint32_t)m;
  *shift = l;
}

// kModePair (DECHIRP_DOWN): ONE real reference, so two frames share a complex transform; one
// history per frame with raw bin indices, stats[frame], symbols = UC_SYM_NONE
enum BandMode { kModeRxReal = 0, kModeCplx = 1, kModePair = 2 };

// returns hipError_t as int
// `waves` = min waves per SIMD the kernel was compiled for (2, 3 or 4): a tuning knob
int launch_band(int mode, int dtype, int waves, const BandParams& p, int grid, hipStream_t stream);
// spec: the instantiation that also stores the window bins (p.spectrum != nullptr on the default two-round build)
// rows: the ROWS instantiation (p.row_blocks != 0; RX_REAL / SYNC_CPLX at their default occupancy -- `waves` is not looked at)
// overlap: frames that overlap (p.stride < kN) on the default builds of RX_REAL / SYNC_CPLX: the same kernel with default-policy loads
int band_max_blocks_per_cu(int mode, int dtype, int waves, bool wide, bool spec = false, bool rows = false, bool overlap = false);

// Full-spectrum pipeline: UC_COMPRESS (FFT x H x IFFT, two frames per complex transform).
struct FullParams {
  const void* frames;     // device, int32 or float
  size_t n_frames;
  size_t stride;          // elements between frame starts
  const float* hann;      // symmetric Hann, n floats
  const float2* hn;       // full Hermitian spectrum of the reference, divided by n
  const float2* tw;       // exp(-2 pi i k / 2048)
  const float* mag_mean;  // device, 2 per frame (first used), or nullptr
  uint8_t* symbols;       // device or nullptr (always UC_SYM_NONE)
  uc_stats* stats;        // device or nullptr, 1 per frame
  float mag_mean_scalar;
  uint32_t unpaired;       // 1 = UC_FLAG_NO_FRAME_PAIRS: one frame per transform (the partner slot reads as zeros)
  unsigned int* work_ctr;  // device word, zero at launch: chunks of 2^chunk_log2 consecutive frame pairs, the ones after a
  uint32_t chunk_log2;     // workgroup's first handed out by atomic increments (chunk_log2 >= 1); nullptr = a balanced
                           // contiguous partition of the pairs
  unsigned long long* debug;  // diagnostic builds only (UC_CLOCKSTAMP), else nullptr
};
int launch_compress(int dtype, const FullParams& p, int grid, hipStream_t stream);
int compress_max_blocks_per_cu(int dtype);

// UC_IQ: carrier mix + 27-tap FIR + chirp multiply + CFFT + three band maxima.
constexpr int kFirTapsDev = 27;
struct IqParams {
  const void* frames;         // device; sample 0 of frame 0, 26 history samples sit in front of it
  size_t n_frames;
  size_t stride;
  const float2* carrier;      // (cos, sin) of the carrier, n entries
  const float2* chirp_hann;   // down chirp (cos, sin) * hann, n entries; base band: conj(up chirp) * hann
  const float2* chirp_hann2;  // base band only: conj(down chirp) * hann
  const float2* tw;
  const float* mag_mean;      // device, 2 per frame (first used), or nullptr
  uint8_t* symbols;
  uc_stats* stats;            // 1 per frame
  float fir[kFirTapsDev];
  const float* fir_mfma;      // n = 1024: the taps as the Toeplitz A operand of v_mfma_f32_16x16x4_f32, [11 k-steps][64 lanes];
                              // nullptr = the packed-VALU FIR
  uint32_t stagger;           // MFMA FIR: start delay of the odd wave slots, in units of s_sleep 64 (4096 clocks)
  float mag_mean_scalar;
  float fs;
  uint32_t idx_left_zero, center, bw2, bw4;
  uint32_t group;             // frames per round-robin group (power of two, <= 64)
  // UC_FLAG_IQ_BASEBAND: two histories {up, down} per frame and a symbol; the windows straddle DC:
  // idx_left_zero = n - bandwidth, center = n (bins are taken mod n), bw2 = bandwidth, bw4 = 2 bandwidth
  uint32_t baseband;
  uint32_t ifs;               // (uint32_t)(int32_t)fs for the receiver's integer idx2freq
  float snr_threshold;
  unsigned int* work_ctr;     // device word, zero at launch: groups beyond a workgroup's first are handed out
                              // dynamically, group id = gridDim.x + the value an atomic increment returns; nullptr = static deal
  unsigned long long* debug;  // diagnostic builds only (UC_CLOCKSTAMP), else nullptr
};
// n = 2048 (the committed firmware) or 1024 (one wave per frame)
int launch_iq(int dtype, const IqParams& p, int grid, hipStream_t stream, int n);
// narrow: base band with windows of at most 32 bins (n = 1024: one pruned round, half-wave reductions)
int iq_max_blocks_per_cu(int dtype, int n, int baseband, int fir_mfma, int narrow);

// UC_STREAM: FIR-LPF decimating front-end + overlap-save chirp compression (include/uchirp.h).
struct StreamParams {
  const void* samples;   // device, 16-byte aligned; the first `halo` samples are history
  size_t n_samples;
  size_t n_out;          // compressed outputs = (n_samples - halo) / D
  size_t n_blocks;       // ceil(n_out / hop)
  const float2* hn;      // FFT_n(template zero-padded) / n
  const float2* rot;     // e^{-j 2 pi carrier D i / fs}, i < n
  const float2* tw;
  float* compressed;     // device or nullptr
  uc_peak* peaks;        // device or nullptr
  float ctap[2 * kFirTapsDev];  // fir[k] e^{+j 2 pi carrier k / fs}, (re, im)
  float rots[16];               // rot[s * 4096 / D], s < D/2: rotation of sub-tile s, (re, im)
  unsigned int* work_ctr;       // device word, zero at launch: chunks of 2^chunk_log2 consecutive blocks, the ones after a
  uint32_t chunk_log2;          // workgroup's first handed out by atomic increments; nullptr = a
                                // balanced contiguous partition of the blocks
  unsigned long long* debug;    // diagnostic builds only (UC_CLOCKSTAMP), else nullptr
};
int launch_stream(int dtype, int decim, const StreamParams& p, int grid, hipStream_t stream);
int stream_max_blocks_per_cu(int dtype, int decim);

// DFSDM model: sinc^5, decimate by 32, of a packed 1-bit PDM stream (receiver/Src/dfsdm.c:59-61,69,78).
struct CicParams {
  // stream s = n_words NEW words at pdm + s * stride (bit t of a stream = bit (t & 31) of word t >> 5), n_words outputs
  // (24-bit result in bits 31:8) at out + s * out_stride; everything device memory, 16-byte aligned
  const uint32_t* pdm;
  size_t n_words;
  int32_t* out;
  const int32_t* t4;     // [4][256][4] per-byte contributions to outputs m .. m+3 of word m
  const int32_t* t1;     // [4][256]    per-byte contribution to output m+4
  unsigned long long* debug;  // diagnostic builds only (UC_CLOCKSTAMP), else nullptr
  size_t n_streams, stride, out_stride;
  // [n_streams][4] words, the filter history of every stream (the four words in front of its new ones): read by the
  // streams' first segments, then -- update_hist -- brought up to date (the last four words of [history | new words]) by a
  // second small kernel
  const uint32_t* hist;
  uint32_t update_hist;
  // a stream is cut into nseg SEGMENTS of tps tiles of 256 words (the last one may be shorter); one wave walks a segment
  uint32_t tps, nseg, units;       // units = n_streams * nseg < 2^31
  uint32_t div_magic, div_shift;   // unit / nseg (uc::rows_divisor)
};
int launch_sinc5(const CicParams& p, int grid, hipStream_t stream);
int sinc5_max_blocks_per_cu();
int sinc5_waves_per_block();

// ---- the clock-stamped twin of every kernel (diagnostic, shipped, costs nothing when off) ----------------------------
// Every .hip file is compiled TWICE into libuchirp.so: once as it is, once with -DUC_CLOCKSTAMP (Makefile: csrc/*.clk.o),
// which adds ONE s_memtime / s_memrealtime stamp pair per wave around the kernel's persistent loop (uc_dev.hpp:
// UC_CLOCK_BEGIN / UC_CLOCK_END; four words per wave go to `debug`).  The second build's launchers live in uc::clk; the
// kernels themselves have internal linkage in both.  uc_clock_probe() (include/uchirp.h) makes a context launch the twin:
// the shader clock the chip holds UNDER a kernel, measured in the run that quotes it, not taken from another box.
#ifdef UC_CLOCKSTAMP
#define UC_LAUNCH_BEGIN namespace clk {
#define UC_LAUNCH_END }
#else
#define UC_LAUNCH_BEGIN
#define UC_LAUNCH_END
#endif
namespace clk {
int launch_band(int mode, int dtype, int waves, const BandParams& p, int grid, hipStream_t stream);
// rows: the ROWS instantiation (p.row_blocks != 0; RX_REAL / SYNC_CPLX at their default occupancy -- `waves` is not looked at)
// overlap: frames that overlap (p.stride < kN) on the default builds of RX_REAL / SYNC_CPLX: the same kernel with default-policy loads
int band_max_blocks_per_cu(int mode, int dtype, int waves, bool wide, bool spec = false, bool rows = false, bool overlap = false);
int launch_compress(int dtype, const FullParams& p, int grid, hipStream_t stream);
int compress_max_blocks_per_cu(int dtype);
int launch_iq(int dtype, const IqParams& p, int grid, hipStream_t stream, int n);
int iq_max_blocks_per_cu(int dtype, int n, int baseband, int fir_mfma, int narrow);
int launch_stream(int dtype, int decim, const StreamParams& p, int grid, hipStream_t stream);
int stream_max_blocks_per_cu(int dtype, int decim);
int launch_sinc5(const CicParams& p, int grid, hipStream_t stream);
int sinc5_max_blocks_per_cu();
int sinc5_waves_per_block();
}  // namespace clk

}  // namespace uc
