// uc_rx_kernel.hip -- the receiver's main loop for MANY streams at once, recorded or live (uc_receive_streams[_next],
// include/uchirp.h).
//
// Replaces, per stream (reference lines):
//   the ISR's FIFO incl. its drop-on-busy            receiver/Src/main.c:659-668   -> accept_kernel (+ pack_kernel), last_kernel
//   4 dsp() calls per block                          receiver/Src/main.c:447-451, 493-504, 243-250
//                                                    -> ONE launch of the band kernel's ROWS build (uc_band_kernel.hip) over
//                                                       the 8 FIFO offsets every accepted block ADDS, (up, down) mag_max each
//   while (1) { switch (state) ... }  + resync()     receiver/Src/main.c:417-554, 243-273   -> replay_kernel
//
// The ISR shifts the FIFO by one block per accepted block (main.c:662), so 9 of the 17 offsets the switch can visit were
// evaluated when the previous block arrived: their records are CARRIED (uc_rx.hpp: RxParams), a block costs 8 transforms per
// reference, and no frame that straddles two streams is ever looked at.
// The switch is sequential per stream and tiny (about thirty words of state, a handful of 8-byte reads per block).  Up to
// 16 Ki streams it runs ONE WAVE PER STREAM (the lanes stage the next blocks' FIFO statistics in LDS while lane 0 runs the
// switch out of LDS); beyond that ONE LANE PER STREAM, 64 streams per wave, lanes diverging over the four states.  It is the code of
// include/uchirp_mainloop.hpp compiled for the device -- the very functions the host replays for uc_receive_stream and
// that tests/cpp/rx_main.cpp drives one GPU call per frame -- so the three cannot drift apart.
// Without a busy mask nothing is copied: the band kernel reads the caller's buffer as it lies.  With one, the ACCEPTED
// blocks of every stream are first laid out one behind the other (pack_kernel) -- the FIFO only ever holds accepted blocks.
#include <hip/hip_runtime.h>

#include <new>

#include "../../include/uchirp_mainloop.hpp"
#include "uc_rx.hpp"

namespace uc {

namespace {

constexpr int kPackThreads = 256;

// acc[s][k] = index of the k-th accepted block of stream s, na[s] = how many (busy[s][b] != 0: the ISR drops block b)
__global__ __launch_bounds__(64) void accept_kernel(const uint8_t* busy, uint32_t nb, uint32_t* acc, uint32_t* na) {
  const size_t s = blockIdx.x;
  const int lane = threadIdx.x;
  const uint8_t* bz = busy + s * nb;
  uint32_t* out = acc + s * nb;
  uint32_t count = 0;
  for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
    const uint32_t b = b0 + (uint32_t)lane;
    const bool ok = b < nb && bz[b] == 0;
    const unsigned long long m = __ballot(ok);
    if (ok) out[count + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = b;
    count += (uint32_t)__popcll(m);
  }
  if (lane == 0) na[s] = count;
}

// one workgroup per (stream, k): the k-th accepted block of the stream, 2048 words; VEC = words per access (4 when
// everything is 16-byte aligned, else 1)
template <int VEC>
__global__ __launch_bounds__(kPackThreads) void pack_kernel(const uint32_t* src, size_t src_stride, uint32_t n, uint32_t nb,
                                                            const uint32_t* acc, const uint32_t* na, uint32_t* dst, size_t pitch) {
  // (a flat grid: neither the streams nor the blocks of one stream are bounded by the 65 535 of gridDim.y)
  const size_t s = blockIdx.x / nb;
  const uint32_t k = blockIdx.x % nb;
  if (k >= na[s]) return;
  const uint32_t* from = src + s * src_stride + (size_t)acc[s * nb + k] * n;
  uint32_t* d = dst + s * pitch + (size_t)k * n;
  if (VEC == 4) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    for (uint32_t i = threadIdx.x; i < n / 4; i += kPackThreads) reinterpret_cast<v4u*>(d)[i] = reinterpret_cast<const v4u*>(from)[i];
  } else {
    for (uint32_t i = threadIdx.x; i < n; i += kPackThreads) d[i] = from[i];
  }
}

// dsp() (receiver/Src/main.c:183-231) over the record sequence of one stream (uc_rx.hpp: RxParams): the frame at FIFO
// offset pos of the call's accepted block `block` is record 8 block + pos / 256 of [9 carried | the call's new records]
struct RxSeq {
  typedef HistLite history_t;
  const float2* carry;  // 9 records
  const float2* rec;    // biased by -ncarry: entry q (q >= ncarry) is the call's new record q - ncarry
  uint32_t per_block, ncarry;  // n / 256, n / 256 + 1
  uint32_t block;
  __device__ float2 at(uint32_t q) const { return q < ncarry ? carry[q] : rec[q]; }
  __device__ void dsp(uint32_t pos, HistLite* h, float mag_mean, int updown) const {
    const float2 mm = at(block * per_block + (pos >> 8));
    const float m = updown == UC_UP_CHIRP ? mm.x : mm.y;
    h->mag_max = m;
    h->mag_mean = mag_mean;
    h->snr = (m - mag_mean) / mag_mean;  // main.c:229
  }
};

// main()'s locals (main.c:314-339) as they travel between calls: the object, then the number of blocks the stream has been
// offered since power-on (trace records carry stream-global block indices)
typedef uchirp::MainLoop<RxSeq> SeqLoop;
constexpr int kStateWords = (int)((sizeof(SeqLoop) + 3) / 4);
constexpr int kImageWords = kStateWords + 1;
static_assert(sizeof(SeqLoop) % 4 == 0, "main()'s locals are whole words");

// main()'s loop, one lane per stream
__global__ __launch_bounds__(64) void replay_kernel(const RxParams p) {
  const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (s >= p.n_streams) return;
  const uint32_t count = p.na ? p.na[s] : p.nb;
  const uint32_t per_block = p.n / 256, ncarry = per_block + 1;
  RxSeq rx{p.carry + s * p.carry_pitch, p.rec + ((ptrdiff_t)(s * p.rec_pitch) - (ptrdiff_t)ncarry), per_block, ncarry, 0};
  // main()'s locals live in LDS: the switch indexes history[] and mag_stat[] dynamically, which as a plain local object
  // would put them in scratch -- a global-memory round trip per access, ~1 us per pass (measured); odd word stride per lane
  constexpr int kLoopWords = kStateWords | 1;
  __shared__ uint32_t loop_mem[64 * kLoopWords];
  uint32_t* mine = loop_mem + threadIdx.x * kLoopWords;
  // a live stream: main()'s locals as the previous call left them -- their bytes go into the storage BEFORE the object is
  // looked at (memcpy + launder: copying words into an object the compiler has just seen constructed lets it keep the
  // constructor's values of the float members in registers -- it did)
  uint32_t block_base = 0;
  if (p.loop_state) {
    __builtin_memcpy(mine, p.loop_state + s * kImageWords, sizeof(SeqLoop));
    block_base = p.loop_state[s * kImageWords + kStateWords];
  } else {
    new (mine) SeqLoop(p.n, p.snr_threshold);
  }
  asm volatile("" ::: "memory");
  SeqLoop& loop = *__builtin_launder(reinterpret_cast<SeqLoop*>(mine));
  char* text = p.text + s * p.text_cap;
  uc_rx_event* trace = p.trace ? p.trace + s * p.trace_cap : nullptr;
  uint32_t ntext = p.fill ? p.fill[s] : 0u, nt = p.trace_start;
  auto put = [&](char ch) {
    if (ntext + 1 < p.text_cap) text[ntext++] = ch;
  };
  for (uint32_t i = 0; i < count; i++) {
    rx.block = i;
    const uchirp::loop_event le = loop.step(rx, put);
    if (trace && nt < p.trace_cap) {
      uc_rx_event ev;
      ev.block = block_base + (p.acc ? p.acc[s * p.nb + i] : i);
      ev.sync_position = le.sync_position;
      ev.state_before = (uint8_t)le.state_before;
      ev.state_after = (uint8_t)le.state_after;
      ev.bit = (int8_t)le.bit;
      ev.reserved = 0;
      ev.snr_up = le.snr_up;
      ev.snr_down = le.snr_down;
      trace[nt] = ev;
    }
    nt++;
  }
  text[ntext] = '\0';
  if (p.n_text) p.n_text[s] = ntext;
  if (p.fill) p.fill[s] = ntext;
  if (p.n_trace) p.n_trace[s] = nt;
  asm volatile("" ::: "memory");
  if (p.loop_state) {
    __builtin_memcpy(p.loop_state + s * kImageWords, mine, sizeof(SeqLoop));
    p.loop_state[s * kImageWords + kStateWords] = block_base + p.nb;
  }
  if (p.parity && s == 0) *p.parity ^= 1u;  // (nothing of this launch reads it)
  if (p.need) p.need[s] = p.need_force ? (p.need_force & 0x1ffu) : need_word(loop.state(), loop.turn(), loop.sync_position());
  if (p.carry_out && count) {
    // the FIFO after the call's last accepted block: its last 9 records are what the next block's FIFO starts with
    float2 keep[9];
    for (uint32_t k = 0; k < ncarry && k < 9; k++) keep[k] = rx.at(count * per_block + k);
    for (uint32_t k = 0; k < ncarry && k < 9; k++) p.carry_out[s * ncarry + k] = keep[k];
  }
}

// The same loop, ONE WAVE PER STREAM (up to a few thousand streams: a lane per stream would leave the chip empty and pay a
// global-memory round trip for every dsp() of the switch, ~1 us per block).  The FIFO of a block spans offsets
// pos = 0 .. 2 n in steps of 256: seventeen (up, down) records.  The lanes fetch the NEXT block's seventeen into LDS while
// lane 0 runs the switch over the current block's -- the switch itself never touches global memory.
struct RxWindow {
  typedef HistLite history_t;
  const float2* win;  // LDS: record pos / 256 of the current block
  __device__ void dsp(uint32_t pos, HistLite* h, float mag_mean, int updown) const {
    const float2 mm = win[pos >> 8];
    const float m = updown == UC_UP_CHIRP ? mm.x : mm.y;
    h->mag_max = m;
    h->mag_mean = mag_mean;
    h->snr = (m - mag_mean) / mag_mean;  // main.c:229
  }
};
static_assert(sizeof(uchirp::MainLoop<RxWindow>) == sizeof(SeqLoop), "one image for both replay kernels");

__global__ __launch_bounds__(64) void replay_wave_kernel(const RxParams p) {
  constexpr int kAhead = 8;                      // blocks whose statistics are in flight: a global load takes ~1 us here,
  __shared__ float2 win[2 * kAhead][32];         // a pass of the switch ~0.15 us
  const size_t s = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t count = p.na ? p.na[s] : p.nb;
  const uint32_t per_block = p.n / 256, span = 2 * per_block + 1, ncarry = per_block + 1;  // 8 offsets per block, 17 per FIFO
  const RxSeq rx{p.carry + s * p.carry_pitch, p.rec + ((ptrdiff_t)(s * p.rec_pitch) - (ptrdiff_t)ncarry), per_block, ncarry, 0};
  auto fetch = [&](uint32_t block) {  // record `lane` of the FIFO at accepted block `block`
    float2 v = make_float2(0.f, 0.f);
    if ((uint32_t)lane < span && block < count) v = rx.at(block * per_block + (uint32_t)lane);
    return v;
  };
  float2 pend[kAhead];
#pragma unroll
  for (int k = 0; k < kAhead; k++) pend[k] = fetch((uint32_t)k);
  // what the FIFO carries into the next call, read before anything of this call is written (carry_out may alias carry)
  float2 keep = make_float2(0.f, 0.f);
  if (p.carry_out && count && (uint32_t)lane < ncarry) keep = rx.at(count * per_block + (uint32_t)lane);
  if (lane < 32) {
#pragma unroll
    for (int k = 0; k < kAhead; k++) win[k][lane] = pend[k];
  }
#pragma unroll
  for (int k = 0; k < kAhead; k++) pend[k] = fetch((uint32_t)(kAhead + k));
  // main()'s locals in LDS, as in replay_kernel (only lane 0 touches them)
  typedef uchirp::MainLoop<RxWindow> Loop;
  __shared__ uint32_t loop_mem[kImageWords];
  Loop* loopp = reinterpret_cast<Loop*>(loop_mem);
  if (p.loop_state) {  // a live stream: main()'s locals as the previous call left them
    if (lane < kImageWords) loop_mem[lane] = p.loop_state[s * kImageWords + lane];
  } else if (lane == 0) {
    new (loop_mem) Loop(p.n, p.snr_threshold);
    loop_mem[kStateWords] = 0u;
  }
  static_assert(kImageWords <= 64, "one lane per word of main()'s locals");
  __syncthreads();
  Loop& loop = *__builtin_launder(loopp);
  const uint32_t block_base = loop_mem[kStateWords];
  char* text = p.text + s * p.text_cap;
  uc_rx_event* trace = p.trace ? p.trace + s * p.trace_cap : nullptr;
  uint32_t ntext = (p.fill && lane == 0) ? p.fill[s] : 0u, nt = p.trace_start;
  auto put = [&](char ch) {
    if (ntext + 1 < p.text_cap) text[ntext++] = ch;
  };
  for (uint32_t base = 0; base < count; base += kAhead) {
    if (lane == 0) {
      const uint32_t end = base + kAhead < count ? base + kAhead : count;
      for (uint32_t i = base; i < end; i++) {
        RxWindow w{win[i % (2 * kAhead)]};
        const uchirp::loop_event le = loop.step(w, put);
        if (trace && nt < p.trace_cap) {
          uc_rx_event ev;
          ev.block = block_base + (p.acc ? p.acc[s * p.nb + i] : i);
          ev.sync_position = le.sync_position;
          ev.state_before = (uint8_t)le.state_before;
          ev.state_after = (uint8_t)le.state_after;
          ev.bit = (int8_t)le.bit;
          ev.reserved = 0;
          ev.snr_up = le.snr_up;
          ev.snr_down = le.snr_down;
          trace[nt] = ev;
        }
        nt++;
      }
    }
    // the statistics of blocks base + 8 .. base + 15, requested a whole group of passes ago, go to the half of the ring the
    // passes above did not read; then the requests for the group after that go out
    if (lane < 32) {
#pragma unroll
      for (int k = 0; k < kAhead; k++) win[(base + kAhead + k) % (2 * kAhead)][lane] = pend[k];
    }
#pragma unroll
    for (int k = 0; k < kAhead; k++) pend[k] = fetch(base + 2 * kAhead + (uint32_t)k);
    __syncthreads();  // (one wave: an LDS wait, no s_barrier)
  }
  if (lane == 0) {
    text[ntext] = '\0';
    if (p.n_text) p.n_text[s] = ntext;
    if (p.fill) p.fill[s] = ntext;
    if (p.n_trace) p.n_trace[s] = nt;
    loop_mem[kStateWords] = block_base + p.nb;
    if (p.parity && s == 0) *p.parity ^= 1u;  // (nothing of this launch reads it)
    if (p.need) p.need[s] = p.need_force ? (p.need_force & 0x1ffu) : need_word(loop.state(), loop.turn(), loop.sync_position());
  }
  __syncthreads();
  if (p.loop_state && lane < kImageWords) p.loop_state[s * kImageWords + lane] = loop_mem[lane];
  if (p.carry_out && count && (uint32_t)lane < ncarry) p.carry_out[s * ncarry + (uint32_t)lane] = keep;
}

// live streams: main()'s locals at power-on (receiver/Src/main.c:314-339) and a block count of zero, one image per stream
__global__ __launch_bounds__(64) void state_init_kernel(uint32_t* loop_state, size_t n_streams, uint32_t n, float thr) {
  __shared__ uint32_t img[kImageWords];
  if (threadIdx.x == 0) {
    for (int w = 0; w < kImageWords; w++) img[w] = 0u;  // (padding bytes too: the images compare equal word for word)
    new (img) SeqLoop(n, thr);
  }
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n_streams * kImageWords; i += (size_t)gridDim.x * 64)
    loop_state[i] = img[i % kImageWords];
}

// the FIFO's newest block a busy-masked call leaves behind (uc_rx.hpp: launch_rx_last)
template <int VEC>
__global__ __launch_bounds__(kPackThreads) void last_kernel(const uint32_t* base, size_t pitch, const uint32_t* na, uint32_t nb,
                                                            uint32_t n, uint32_t* last, const unsigned int* parity, size_t half) {
  const size_t s = blockIdx.x;
  const uint32_t count = na ? na[s] : nb;
  const unsigned par = *parity & 1u;
  // every block of the stream dropped: the FIFO is as it was -- the block moves to the other half with the rest
  const uint32_t* from = count ? base + s * pitch + (size_t)(count - 1) * n : last + par * half + s * (size_t)n;
  uint32_t* d = last + (par ^ 1u) * half + s * (size_t)n;
  if (VEC == 4) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    for (uint32_t i = threadIdx.x; i < n / 4; i += kPackThreads) reinterpret_cast<v4u*>(d)[i] = reinterpret_cast<const v4u*>(from)[i];
  } else {
    for (uint32_t i = threadIdx.x; i < n; i += kPackThreads) d[i] = from[i];
  }
}

// uc_rx_state_keep_previous ahead of a busy-masked call (uc_rx.hpp: launch_rx_keep)
template <int VEC>
__global__ __launch_bounds__(kPackThreads) void keep_kernel(const uint32_t* kept, size_t pitch, uint32_t n, uint32_t* last,
                                                            const unsigned int* parity, size_t half) {
  const size_t s = blockIdx.x;
  const uint32_t* from = kept + s * pitch;
  uint32_t* d = last + (*parity & 1u) * half + s * (size_t)n;
  if (VEC == 4) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    for (uint32_t i = threadIdx.x; i < n / 4; i += kPackThreads) reinterpret_cast<v4u*>(d)[i] = reinterpret_cast<const v4u*>(from)[i];
  } else {
    for (uint32_t i = threadIdx.x; i < n; i += kPackThreads) d[i] = from[i];
  }
}

}  // namespace

int launch_rx_keep(const void* kept, size_t pitch, uint32_t n, size_t n_streams, void* last, const unsigned int* parity,
                   bool aligned16, hipStream_t stream) {
  if (n_streams == 0) return (int)hipSuccess;
  const size_t half = n_streams * (size_t)n;
  if (aligned16)
    hipLaunchKernelGGL(keep_kernel<4>, dim3((unsigned)n_streams), dim3(kPackThreads), 0, stream, (const uint32_t*)kept, pitch, n,
                       (uint32_t*)last, parity, half);
  else
    hipLaunchKernelGGL(keep_kernel<1>, dim3((unsigned)n_streams), dim3(kPackThreads), 0, stream, (const uint32_t*)kept, pitch, n,
                       (uint32_t*)last, parity, half);
  return (int)hipGetLastError();
}

int launch_rx_accept(const uint8_t* busy, size_t n_streams, uint32_t nb, uint32_t* acc, uint32_t* na, hipStream_t stream) {
  if (n_streams == 0) return (int)hipSuccess;
  hipLaunchKernelGGL(accept_kernel, dim3((unsigned)n_streams), dim3(64), 0, stream, busy, nb, acc, na);
  return (int)hipGetLastError();
}

int rx_loop_words() { return kImageWords; }

int launch_rx_state_init(uint32_t* loop_state, size_t n_streams, uint32_t n, float snr_threshold, hipStream_t stream) {
  if (n_streams == 0) return (int)hipSuccess;
  const size_t words = n_streams * (size_t)kImageWords;
  const unsigned grid = (unsigned)((words + 63) / 64 < 4096 ? (words + 63) / 64 : 4096);
  hipLaunchKernelGGL(state_init_kernel, dim3(grid), dim3(64), 0, stream, loop_state, n_streams, n, snr_threshold);
  return (int)hipGetLastError();
}

int launch_rx_last(const void* base, size_t pitch, const uint32_t* na, uint32_t nb, uint32_t n, size_t n_streams, void* last,
                   const unsigned int* parity, bool aligned16, hipStream_t stream) {
  if (n_streams == 0 || nb == 0) return (int)hipSuccess;
  const size_t half = n_streams * (size_t)n;
  if (aligned16)
    hipLaunchKernelGGL(last_kernel<4>, dim3((unsigned)n_streams), dim3(kPackThreads), 0, stream, (const uint32_t*)base, pitch, na,
                       nb, n, (uint32_t*)last, parity, half);
  else
    hipLaunchKernelGGL(last_kernel<1>, dim3((unsigned)n_streams), dim3(kPackThreads), 0, stream, (const uint32_t*)base, pitch, na,
                       nb, n, (uint32_t*)last, parity, half);
  return (int)hipGetLastError();
}

int launch_rx_pack(const void* src, size_t src_stride, uint32_t n, uint32_t nb, size_t n_streams, const uint32_t* acc,
                   const uint32_t* na, void* dst, size_t pitch, bool aligned16, hipStream_t stream) {
  if (n_streams == 0 || nb == 0) return (int)hipSuccess;
  const dim3 grid((unsigned)((size_t)nb * n_streams));  // < 2^28: the frame count of the launch behind it is 8 x this
  if (aligned16)
    hipLaunchKernelGGL(pack_kernel<4>, grid, dim3(kPackThreads), 0, stream, (const uint32_t*)src, src_stride, n, nb, acc, na,
                       (uint32_t*)dst, pitch);
  else
    hipLaunchKernelGGL(pack_kernel<1>, grid, dim3(kPackThreads), 0, stream, (const uint32_t*)src, src_stride, n, nb, acc, na,
                       (uint32_t*)dst, pitch);
  return (int)hipGetLastError();
}

// Up to kWaveStreams streams: a wave per stream (latency: the switch runs out of LDS).  Beyond that a lane per stream (the
// chip is full either way, and 64 streams share a wave's instruction issue).
constexpr size_t kWaveStreams = 16384;

int launch_rx_replay(const RxParams& p, hipStream_t stream) {
  if (p.n_streams == 0) return (int)hipSuccess;
  if (p.n_streams <= kWaveStreams && p.n <= 2048)  // (17 FIFO records fit the 32-entry window)
    hipLaunchKernelGGL(replay_wave_kernel, dim3((unsigned)p.n_streams), dim3(64), 0, stream, p);
  else
    hipLaunchKernelGGL(replay_kernel, dim3((unsigned)((p.n_streams + 63) / 64)), dim3(64), 0, stream, p);
  return (int)hipGetLastError();
}

}  // namespace uc
