// uc_rx.hpp -- what the host replay (uc_api_rx.cpp: uc_receive_stream) and the device replay (uc_rx_kernel.hip:
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

// dsp() (receiver/Src/main.c:183-231) over ONE packed stream (uc_receive_stream, replayed on the host): the FIFO at
// accepted block b is packed[b n, b n + 3 n), so the frame at FIFO offset pos is frame (b n + pos) / 256 of the
// stride-256 launch
struct RxReplay {
  typedef HistLite history_t;
  const float2* magmax;  // (up, down) mag_max of every 256-sample offset of this stream's PACKED form (2 n zeros, then the
                         // accepted blocks): entry g = packed offset 256 g
  size_t n_frames;       // bound of magmax
  uint32_t n;
  size_t block;          // current accepted block index b
  UC_HD void dsp(uint32_t pos, HistLite* h, float mag_mean, int updown) const {
    const size_t g = (block * (size_t)n + pos) / 256;
    const float2 mm = magmax[g];
    const float m = updown == UC_UP_CHIRP ? mm.x : mm.y;
    h->mag_max = m;
    h->mag_mean = mag_mean;
    h->snr = (m - mag_mean) / mag_mean;  // main.c:229
  }
};

// Many streams, on the device.  The ISR shifts the FIFO by ONE block per accepted block (main.c:662): of the 2 n / 256 + 1
// offsets dsp() can visit in the FIFO (pos = 0 .. 2 n in steps of 256) the first n / 256 + 1 were the LAST n / 256 + 1 of
// the FIFO before the shift -- evaluated then, carried since -- and only n / 256 are new.  So the (up, down) mag_max records
// of one stream form ONE sequence, record q = the frame that starts 256 q samples into [2 n zeros | accepted blocks]:
// accepted block i of the stream (counted from power-on) looks at records 8 i .. 8 i + 16 and adds records 8 i + 9 ..
// 8 i + 16 (n = 2048).  A call keeps the last 9 records of the sequence for the next one (`carry`: zeros at power-on, the
// FIFO being zeros, main.c:94) and evaluates 8 per accepted block (`rec`, the ROWS build of the band kernel).
struct RxParams {
  const float2* rec;     // device: the new records of this call, [n_streams][rec_pitch]; entry 8 k + m - 1 = offset 256 m
                         // behind the stream's k-th accepted block of this call (m = 1 .. 8)
  size_t rec_pitch;      // 8 nb
  const float2* carry;   // device: the 9 records every stream carries INTO this call, [n_streams][carry_pitch]
  uint32_t carry_pitch;  // 9 (a live state) or 0 (one set of zero records for all: streams that start here)
  float2* carry_out;     // device or nullptr: [n_streams][9], the records carried into the NEXT call (may alias carry)
  size_t n_streams;
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
  // live streams (uc_rx_state): main()'s locals of every stream between calls and the number of blocks the stream has been
  // offered so far (trace records carry stream-global block indices), rx_loop_words() words each -- loaded at the start of
  // the replay and stored back at its end; nullptr = a recorded stream (the locals start as at power-on, block 0)
  uint32_t* loop_state;
  // live streams: the word that says which half of the state's newest-block store is current (uc_kernels.hpp: BandParams);
  // the replay flips it on its way out -- the band launch in front of it has filled the other half.  nullptr: none.
  unsigned int* parity;
  // live streams: [n_streams] words the replay leaves for the NEXT call's band launch (one-block calls use them): what main()
  // can still look at of the stream's next block.  While a stream is IDLE the acquisition pass evaluates 4 positions a block,
  // the UP reference only, alternating between two interleaved sets (`turn`, main.c:447-453): of the 8 offsets the next block
  // adds, the pass of THAT block reads those <= 11 of its set, the pass of the block after it those >= 12 of the other set, and
  // nothing later can read them (SYNCHRONIZED is three evaluations = five blocks away; by then the block has left the FIFO):
  //   turn 0: offsets m = 2, 5, 7 (0x052)    turn 1: m = 1, 3, 4, 6, 8 (0x0AD)    bit 8 (DOWN statistics): 0
  // SYNCHRONIZING: 0x1FF (the lock may fall on any position).  SYNCHRONIZED / DATA_RECEIVING at sync_position = 256 ks: the
  // offsets ks - 1 .. ks + 1 of the next block's pass, ks + 6 .. ks + 10 (= ks - 2 .. ks + 2 one block later, resync moves one
  // step a block), the acquisition set the stream would use if it fell back to IDLE, and k = 16 when ks <= 3; both references:
  // 5 or 6 of the 8.  (uc_rx_kernel.hip: need_word.)  nullptr: none.
  uint32_t* need;
  // a call served as a SEQUENCE of one-block steps (uc_api_rx.cpp: receive_steps): the characters a stream has been given so far
  // in this call ([n_streams], in and out: the step appends behind them; nullptr: none) and the trace records every stream has
  // been given so far (one per block)
  uint32_t* fill;
  uint32_t trace_start;
  uint32_t need_force;   // pricing runs only (UC_TUNING=1 UC_RX_NEED_FORCE): bit 31 set = every stream's word is the low 9 bits of this
};

// (shared by the replay kernels, uc_rx_state_reset and the CPU harness that ties it to main()'s switch: tests/cpp/san_host.cpp)
// what the switch can still look at of a stream's NEXT block (uc_rx.hpp: RxParams::need).  FIFO offsets are counted in steps
// of 256 samples: k = pos / 256 = 0 .. 16; the next block's NEW offsets are k = 9 .. 16 (bit k - 9); one block later they sit at
// k - 8 = 1 .. 8, two blocks later only k = 16 is left (at 0).
UC_HD inline uint32_t need_word(int state, uint32_t turn, uint32_t sync_position) {
  if (state == UC_STATE_IDLE) return turn ? 0x0ADu : 0x052u;  // acquisition: k = 4 + turn + 2 i now, the other set next block
  if (state == UC_STATE_SYNCHRONIZING) return 0x1FFu;         // may lock onto any of the eight positions: everything
  // SYNCHRONIZED / DATA_RECEIVING at ks = sync_position / 256 (main.c:491-550, resync 243-273): the pass of the next block reads
  // ks - 1 .. ks + 1 (both references) and moves by at most one step; the pass after it therefore reads ks - 2 .. ks + 2 of ITS
  // FIFO = this block's k = ks + 6 .. ks + 10 -- or, if the stream falls back to IDLE in between, the acquisition set of the
  // turn it kept (k = 12 + turn, 14 + turn, 16 + turn); two blocks on only k = 16 is left, at position 0: reachable from ks <= 3
  const int ks = (int)(sync_position >> 8), t = (int)turn;
  uint32_t m = 0x100u;
  for (int k = 9; k <= 16; k++) {
    const bool now = k >= ks - 1 && k <= ks + 1;
    const bool next = k >= ks + 6 && k <= ks + 10;
    const bool idle_next = k == 12 + t || k == 14 + t || k == 16 + t;
    const bool later = k == 16 && ks <= 3;
#ifdef UC_NEED_BREAK  // (a deliberately WRONG mask, to show that the poisoned runs notice: tools/soak_live.py must fail with it)
    if (now || idle_next || later) m |= 1u << (k - 9);
#else
    if (now || next || idle_next || later) m |= 1u << (k - 9);
#endif
  }
  return m;
}

int launch_rx_accept(const uint8_t* busy, size_t n_streams, uint32_t nb, uint32_t* acc, uint32_t* na, hipStream_t stream);
// the accepted blocks of every stream laid out one behind the other: dst[s * pitch + k * n ..) = k-th accepted block of
// stream s (k < na[s]; the rest of the row is left as it is)
int launch_rx_pack(const void* src, size_t src_stride, uint32_t n, uint32_t nb, size_t n_streams, const uint32_t* acc,
                   const uint32_t* na, void* dst, size_t pitch, bool aligned16, hipStream_t stream);
int launch_rx_replay(const RxParams& p, hipStream_t stream);
// live streams: words of main()'s locals (+ the block counter) per stream; their power-on image.
int rx_loop_words();
int launch_rx_state_init(uint32_t* loop_state, size_t n_streams, uint32_t n, float snr_threshold, hipStream_t stream);
// The FIFO's newest block a BUSY-MASKED call leaves behind (without a mask the band kernel stores it on its way through):
// half (1 - *parity) of `last` ([2][n_streams][n] words) receives, for every stream, the stream's last accepted block of the
// call -- n words of `base` at s * pitch + (na[s] - 1) * n -- or, when every block of the stream was dropped, the block the
// current half holds.  Must run BEFORE the replay kernel of the call (which flips *parity).
int launch_rx_last(const void* base, size_t pitch, const uint32_t* na, uint32_t nb, uint32_t n, size_t n_streams,
                   void* last, const unsigned int* parity, bool aligned16, hipStream_t stream);
// uc_rx_state_keep_previous, ahead of a busy-masked call: the block every stream's FIFO holds last lies in the caller's kept
// chunk (n words at kept + s * pitch); it goes into the CURRENT half of `last` (half *parity), where such a call looks for it.
int launch_rx_keep(const void* kept, size_t pitch, uint32_t n, size_t n_streams, void* last, const unsigned int* parity,
                   bool aligned16, hipStream_t stream);

// uc_api_rx.cpp: the argument checks of uc_receive_streams[_next] alone (live: the uc_receive_streams_next form, st required)
int receive_streams_check(uc_ctx* c, uc_rx_state* st, bool live, const void* samples, int dtype, size_t n_streams, size_t n_samples,
                          size_t stream_stride_elems, const char* text, size_t text_cap, size_t trace_cap);

}  // namespace uc
