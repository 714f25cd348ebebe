// uc_rx.hpp -- what the host replay (uc_api.cpp: uc_receive_stream) and the device replay (uc_rx_kernel.hip:
// uc_receive_streams) of main()'s switch share: the dsp() stand-in that looks a frame up in the statistics of the batched
// launch, and the launch interface of the rx kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/uchirp.h"
#include "../../include/uchirp_mainloop.hpp"

namespace uc {

struct HistLite {  // the members of struct history (receiver/Src/main.c:124-136) the switch reads
  float mag_max = 0.0f, mag_mean = 0.0f, snr = 0.0f;
};

// dsp() (receiver/Src/main.c:183-231) over a packed stream: the FIFO at accepted block b is packed[b n, b n + 3 n), so the
// frame at FIFO offset pos is frame (b n + pos) / 256 of the stride-256 launch
struct RxReplay {
  typedef HistLite history_t;
  const float2* magmax;  // (up, down) mag_max of every 256-sample offset of this stream's PACKED form (2 n zeros, then the
                         // accepted blocks): entry g = packed offset 256 g
  size_t n_frames;       // (host replay only: bound of magmax)
  uint32_t n;
  size_t block;          // current accepted block index b
  // uc_receive_streams without a packed copy: the offsets that reach into the zero prefix (g < head_count = 2 n / 256) come
  // from a small second launch over [2 n zeros | first block] of every stream, all others straight from the caller's
  // buffer; `magmax` is then biased so that entry g (g >= head_count) is stream offset 256 g - 2 n.  nullptr: packed form.
  const float2* head = nullptr;
  uint32_t head_count = 0;
  UC_HD void dsp(uint32_t pos, HistLite* h, float mag_mean, int updown) const {
    const size_t g = (block * (size_t)n + pos) / 256;
    const float2 mm = (head && g < head_count) ? head[g] : magmax[g];
    const float m = updown == UC_UP_CHIRP ? mm.x : mm.y;
    h->mag_max = m;
    h->mag_mean = mag_mean;
    h->snr = (m - mag_mean) / mag_mean;  // main.c:229
  }
};

struct RxParams {
  const float2* magmax;  // device: (up, down) mag_max of every 256-sample offset of the packed buffer (or, with `head`, of
                         // the caller's own buffer)
  const float2* head;    // device or nullptr: the same for [2 n zeros | first block] of every stream, 3 n samples apart
  size_t n_streams;
  size_t pitch;          // samples between streams in the buffer `magmax` was computed over: (2 + nb) n packed, else the
                         // caller's stream stride (a multiple of 256)
  uint32_t n, nb;
  float snr_threshold;
  const uint32_t* acc;   // device or nullptr (no busy mask): [n_streams][nb] indices of the accepted blocks
  const uint32_t* na;    // device or nullptr: accepted blocks per stream
  char* text;            // device: [n_streams][text_cap]
  uint32_t text_cap;
  uint32_t* n_text;      // device or nullptr
  uc_rx_event* trace;    // device or nullptr: [n_streams][trace_cap]
  uint32_t trace_cap;
  uint32_t* n_trace;     // device or nullptr
  // live streams (uc_rx_state): main()'s locals of every stream between calls, rx_loop_words() words each -- loaded at the
  // start of the replay and stored back at its end; nullptr = a recorded stream (the locals start as at power-on)
  uint32_t* loop_state;
  uint32_t block_base;   // blocks of every stream the earlier calls have seen (trace records carry stream-global indices)
};

int launch_rx_accept(const uint8_t* busy, size_t n_streams, uint32_t nb, uint32_t* acc, uint32_t* na, hipStream_t stream);
// prefix: device or nullptr -- what the first 2 n words of every packed stream are (2 n words per stream: the FIFO's tail
// of the previous call); nullptr = zeros (a stream that starts here)
int launch_rx_pack(const void* src, size_t src_stride, uint32_t n, uint32_t nb, size_t n_streams, const uint32_t* acc,
                   const uint32_t* na, const void* prefix, void* dst, size_t pitch, bool aligned16, hipStream_t stream);
int launch_rx_replay(const RxParams& p, hipStream_t stream);
// live streams: words of main()'s locals per stream; their power-on image; the FIFO tail a call leaves behind --
// tail[s] = 2 n words of `base` at s * pitch + (na ? na[s] : off_blocks) * n
int rx_loop_words();
int launch_rx_state_init(uint32_t* loop_state, size_t n_streams, uint32_t n, float snr_threshold, hipStream_t stream);
int launch_rx_tail(const void* base, size_t pitch, const uint32_t* na, uint32_t off_blocks, uint32_t n, size_t n_streams,
                   void* tail, hipStream_t stream);

}  // namespace uc
