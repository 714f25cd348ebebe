// uc_api_internal.hpp -- what the translation units of the C-ABI share (uc_api_core.cpp: errors, contexts, tables, the hand-out
// counters, the band launch and uc_process_batch; uc_api_rx.cpp: the receivers; uc_api_stream.cpp: UC_STREAM; uc_api_cic.cpp:
// the DFSDM front end; uc_api_clock.cpp: the clock probe).  Nothing here is part of the ABI (include/uchirp.h is).
#pragma once
#include <errno.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/uchirp.h"
#include "../../include/uchirp_mainloop.hpp"
#include "uc_kernels.hpp"
#include "uc_rx.hpp"
#include "uc_tables.hpp"

namespace uc_api {

// the thread's uc_last_error() text; both return `code` (hip_fail: -EIO)
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_fail(hipError_t e, const char* what);

// set while a call records into a stream capture: scratch must not be (re)allocated there (hipMalloc / hipFree are not
// capturable, and a freed buffer may be baked into the graph) -- a buffer that would have to grow fails the call instead
extern thread_local bool g_capturing;
struct CaptureNoAlloc {
  const bool prev;
  explicit CaptureNoAlloc(bool on) : prev(g_capturing) { g_capturing = on || prev; }
  ~CaptureNoAlloc() { g_capturing = prev; }
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (g_capturing)
      return fail(-ENOBUFS, "scratch of %zu bytes would have to be allocated during a stream capture: make one eager call of "
                            "the same shape first (it sizes the scratch)", bytes);
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(staging)");
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// (Not cached: device and host allocations share one virtual address space, so an address that was device memory can
// later be host memory; a stale "device" verdict would hand a host pointer to a kernel.  The query costs a few
// microseconds per pointer argument, which only small batches notice.)
bool is_device_ptr(const void* p);

// what one uc_receive_streams[_next] call parks on the device between its kernels: staged host inputs, the accepted-block
// lists and the packed copy of a busy-masked call, the new (up, down) records, staged host outputs.  A live state owns a set
// of its own (calls on different states never share scratch); calls without a state use the context's, one stream at a time
// (rx_guard below).
struct RxScratch {
  DevBuf in, busy, acc, na, pad, rec, text, ntext, trace, ntrace;
  DevBuf pcm, hist;  // UC_DTYPE_PDM: the DFSDM words of the call's blocks; the filter history of streams without a state
  // a call served block by block (uc_api_rx.cpp: receive_steps): the characters every stream has been given so far in the call;
  // for streams WITHOUT a state also what a state would carry from step to step -- the 9 surviving records, main()'s locals, the
  // need words
  DevBuf fill, carry, loop, need;
  void release() {
    for (DevBuf* b : {&in, &busy, &acc, &na, &pad, &rec, &text, &ntext, &trace, &ntrace, &pcm, &hist, &fill, &carry, &loop, &need})
      b->release();
  }
};

// Event queries are "potentially unsafe" calls: while ANY stream of the thread is being captured in the global capture mode
// (torch.cuda.graph's default) they are refused AND invalidate that capture.  The guard's events have nothing to do with
// a capture in progress, so its calls run with the thread's capture mode switched to relaxed for their duration (what
// allocators that must touch the runtime during someone else's capture do).
struct RelaxedCapture {
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  bool ok;
  RelaxedCapture() { ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; if (!ok) (void)hipGetLastError(); }
  ~RelaxedCapture() { if (ok) (void)hipThreadExchangeStreamCaptureMode(&mode); }
};

}  // namespace uc_api

// Hand-out counters of the dynamically dealt launches (two 32-bit words each -- next ticket, workgroups gone -- every
// pair in its own 128-byte line; zeroed once at uc_create, left at zero by every launch's last workgroup):
//   slots [0, kWorkSlots)                          a ring for EAGER launches; a slot is reused only after the launch that
//                                                  last used it has finished (one hipEvent per slot, queried on reuse)
//   slots [kWorkSlots, kWorkSlots + kGraphSlots)   handed out ONCE each to launches recorded while their stream is being
//                                                  captured into a hipGraph: the graph owns that slot for the life of the
//                                                  context
constexpr unsigned kWorkSlots = 64, kGraphSlots = 960, kWorkStride = 128;
using uc_api::DevBuf;
using uc_api::RxScratch;

struct uc_ctx {
  uc_config cfg;
  uc::Tables tab;
  uc::StreamTables stab;  // UC_STREAM only
  int device = 0;
  int num_cu = 256;
  // device-resident tables
  float2* d_tab0 = nullptr;
  float2* d_tab1 = nullptr;
  float2* d_tab2 = nullptr;  // IQ base band: conj(down chirp) * hann
  float2* d_tw = nullptr;
  float* d_aux = nullptr;  // variant-specific (COMPRESS: H_down packed; IQ: carrier/fir/...)
  int32_t* d_cic4 = nullptr;  // sinc^5 byte tables, built on first use of uc_dfsdm_sinc5
  int32_t* d_cic1 = nullptr;
  int cic_blocks_per_cu = 0;
  DevBuf s_cic_in, s_cic_out, s_cic_hist;
  // staging for host-pointer calls
  DevBuf s_frames, s_mm, s_sym, s_stats;
  int band_blocks_per_cu[6][3][2] = {};  // [default / wide / default + spectrum stores / rows / rows + wide / overlapping frames][mode][dtype]: the instantiations differ in registers
  int full_blocks_per_cu[2] = {0, 0};    // [dtype]: the int32 / f32 instantiations differ in registers
  int iq_blocks_per_cu[2] = {0, 0};
  int stream_blocks_per_cu[2] = {0, 0};
  DevBuf s_comp, s_peaks, s_spec;
  DevBuf s_rx_pad, s_rx_mag;        // uc_receive_stream: the zero-prefixed stream, (up, down) mag_max per frame
  RxScratch rx;                     // uc_receive_streams (no live state): scratch of the call in flight
  void* d_zero_block = nullptr;     // n zero words: the block "in front of" a stream that starts (fifo_queue at power-on,
                                    // main.c:94) and the zero records such a stream carries in
  hipEvent_t rx_ev = nullptr;       // recorded behind the last uc_receive_streams call that used `rx`: a call on ANOTHER
  hipStream_t rx_stream = nullptr;  // stream waits for it (on the device) before it overwrites the scratch
  bool rx_used = false;
  std::vector<float2> h_rx_mag;
  int band_waves = 3;     // tuning knobs (env UC_BAND_WAVES / UC_GRID / UC_BAND_GROUP / UC_STATIC_DEAL): not part of the ABI
  bool band_waves_set = false;  // UC_BAND_WAVES given: use it for every mode (default: 3, SYNC_CPLX 2 -- see process_batch_impl)
  int grid_override = 0;
  int band_group = 32;    // frames per group handed to a workgroup at a time
  bool static_deal = false;
  long rx_step_min = -1;       // (env UC_RX_STEP_MIN) streams from which a call of several blocks is served block by block: -1 = by variant
  uint32_t rx_need_force = 0;  // (env UC_RX_NEED_FORCE=0x1..: pricing runs only) every stream's need word is this one: WRONG results
  bool rx_poison = false;   // (env UC_RX_POISON=1, tests) the statistics the live receivers pass over are huge instead of zero
  int compress_chunk = 8;   // (env UC_COMPRESS_CHUNK) frame pairs per hand-out chunk of the compress kernel: a power of two >= 2
  int stream_chunk = 2;     // (env UC_STREAM_CHUNK) blocks per hand-out chunk of the stream kernel: a power of two
  int iq_group = 16;        // (env UC_IQ_GROUP) frames per hand-out group of the IQ kernels: a power of two <= 64
                            // (16: +0.6 ... 1.2 % over 32 on all three IQ kernels, profiles/r04_knob_sweep.txt)
  unsigned iq_stagger = 0;  // (env UC_IQ_STAGGER, MFMA FIR only) start delay of every second wave on a SIMD, x 4096 clocks
  // UC_IQ at n = 1024, env UC_IQ_FIR=mfma: the FIR as v_mfma_f32_16x16x4_f32 Toeplitz tiles instead of packed VALU.
  // Off by default: an f32 MFMA and the partner wave's packed-f32 VALU do not overlap on a SIMD (tools/mfma_valu_probe.hip:
  // together they take the SUM of their times), and the Toeplitz padding makes the matrix form 1.6x the FMA count.
  bool iq_fir_mfma = false;
  // work counters for the dynamic group hand-out: one word per launch, a ring so that launches of one context that
  // overlap on different streams never share one (each word sits in its own 128-byte line)
  unsigned int* d_work = nullptr;
  void* h_slot = nullptr;  // uc_process_frame: pinned, device-mapped host memory for one frame and its results
  unsigned work_next = 0;
  unsigned graph_next = 0;              // graph-owned slots handed out so far (never recycled)
  hipEvent_t work_ev[kWorkSlots] = {};  // recorded behind the launch that used ring slot i (several streams only)
  bool work_busy[kWorkSlots] = {};      // slot i has been used and its event not yet seen complete
  // As long as every eager launch of the context goes to ONE stream, stream order alone keeps a slot from being
  // shared (its previous user finished 64 launches earlier on the same stream) and no event is recorded at all.
  // The first launch on a second stream records `switch_ev` on the first one -- it covers every slot used so far --
  // and from then on every launch records its slot's event.
  hipStream_t ring_stream = nullptr;
  bool ring_stream_set = false, multi_stream = false;
  hipEvent_t switch_ev = nullptr;
  bool wait_switch[kWorkSlots] = {};    // slot i was last used before the switch: free once switch_ev has completed
  bool slot_used[kWorkSlots] = {};
  bool switch_lost = false;             // the switch event could not be recorded: slots used before it never come back
  bool graph_slots_warned = false;
  // uc_clock_probe(): launches run the clock-stamped twin of their kernel (uc_kernels.hpp: uc::clk) and leave four words
  // per wave here; clock_waves = the waves of the LAST launch
  bool clock_probe = false;
  DevBuf s_clock;
  size_t clock_waves = 0;
  int clk_cic_blocks = 0;
};

namespace uc_api {

// uc_api_core.cpp
int upload(void** dst, const void* src, size_t bytes);   // hipMalloc + copy of a host table
// the hand-out counter of a dynamically dealt launch and the event behind it (comments at the definitions)
int take_work_counter(uc_ctx* c, hipStream_t stream, unsigned int** out, int* slot);
int work_counter_launched(uc_ctx* c, hipStream_t stream, int slot);
int clock_buffer(uc_ctx* c, size_t grid, int waves_per_wg, hipStream_t stream, unsigned long long** out);
int band_launch(uc_ctx* c, uc::BandParams& p, int dtype, hipStream_t stream);
int process_batch_impl(uc_ctx* c, const void* frames, int dtype, size_t n_frames, size_t stride_elems, const float* mag_mean,
                       uint8_t* symbols, uc_stats* stats, float2* d_magmax, void* hip_stream, bool mapped = false,
                       float* d_spectrum = nullptr);
// uc_api_cic.cpp
int sinc5_streams_launch(uc_ctx* c, const uint32_t* d_pdm, size_t n_streams, size_t n_words, size_t stride,
                         const uint32_t* d_hist, bool update_hist, int32_t* d_out, size_t out_stride, hipStream_t stream);

}  // namespace uc_api
