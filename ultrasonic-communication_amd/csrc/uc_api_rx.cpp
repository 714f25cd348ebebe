// uc_api_rx.cpp -- the receivers: uc_receive_stream[_isr] (one recorded stream, main()'s switch replayed on the host),
// uc_receive_streams (many recorded streams) and uc_rx_state / uc_receive_streams_next (live streams, everything on the device).
#include "uc_api_internal.hpp"

using namespace uc_api;

// ---------------------------------------------------------------------------
// uc_receive_stream: the receiver's main loop over a recorded stream.
// DSP: ONE batched launch over every 256-sample offset of the zero-prefixed
// stream; control: main()'s switch (include/uchirp_mainloop.hpp, shared with the
// C++ host layer) replayed on the host over the (up, down) mag_max of every frame.
// ---------------------------------------------------------------------------
using uc::RxReplay;

extern "C" int uc_receive_stream_isr(uc_ctx* c, const void* samples, int dtype, size_t n_samples, const uint8_t* busy,
                                     char* text, size_t text_cap, uc_rx_event* trace, size_t trace_cap, size_t* n_trace) {
  if (!c || !text || text_cap == 0) return fail(-EINVAL, "uc_receive_stream: NULL argument");
  text[0] = '\0';
  if (n_trace) *n_trace = 0;
  if (c->cfg.variant != UC_RX_REAL && c->cfg.variant != UC_SYNC_CPLX)
    return fail(-ENOTSUP, "uc_receive_stream: variant %d has no up/down state machine", (int)c->cfg.variant);
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32) return fail(-EINVAL, "uc_receive_stream: bad dtype %d", dtype);
  const uint32_t n = c->cfg.n;
  const size_t n_blocks = n_samples / n;
  if (n_blocks == 0) return 0;
  if (!samples) return fail(-EINVAL, "uc_receive_stream: samples is NULL");

  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  // The ISR (main.c:659-668) appends a block only when the main loop has consumed the previous one: a block that
  // arrives while `new_pcm_data` is still set is DROPPED, the FIFO is not shifted.  `busy[b] != 0` says the consumer
  // was still busy when block b arrived (NULL: never -- the GPU evaluates all of a block's dsp() calls at once).
  // The FIFO therefore only ever holds ACCEPTED blocks, in order: drop the others before the batched launch.
  std::vector<uint32_t> accepted;
  accepted.reserve(n_blocks);
  for (size_t b = 0; b < n_blocks; b++)
    if (!busy || !busy[b]) accepted.push_back((uint32_t)b);
  const size_t na = accepted.size();
  if (na == 0) return 0;
  // zero-prefixed copy of the accepted stream on the device: fifo_queue starts as 3n zeros (main.c:94)
  const size_t padded = (2 + na) * (size_t)n;
  const size_t n_frames = (padded - n) / 256 + 1;
  int rc = c->s_rx_pad.ensure(padded * 4);
  if (!rc) rc = c->s_rx_mag.ensure(n_frames * sizeof(float2));
  if (rc) return rc;
  char* d_pad = (char*)c->s_rx_pad.p;
  const hipMemcpyKind kind = is_device_ptr(samples) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  e = hipMemsetAsync(d_pad, 0, 2 * (size_t)n * 4, nullptr);
  for (size_t i = 0; e == hipSuccess && i < na;) {  // runs of consecutive accepted blocks: one copy each
    size_t jn = i + 1;
    while (jn < na && accepted[jn] == accepted[jn - 1] + 1) jn++;
    e = hipMemcpyAsync(d_pad + (2 + i) * (size_t)n * 4, (const char*)samples + (size_t)accepted[i] * n * 4,
                       (jn - i) * (size_t)n * 4, kind, nullptr);
    i = jn;
  }
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy(stream)");
  rc = process_batch_impl(c, d_pad, dtype, n_frames, 256, nullptr, nullptr, nullptr, (float2*)c->s_rx_mag.p, nullptr);
  if (rc) return rc;
  c->h_rx_mag.resize(n_frames);
  e = hipMemcpy(c->h_rx_mag.data(), c->s_rx_mag.p, n_frames * sizeof(float2), hipMemcpyDeviceToHost);  // syncs stream 0
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy(mag_max)");

  RxReplay rx{c->h_rx_mag.data(), n_frames, n, 0};
  uchirp::MainLoop<RxReplay> loop(n, c->cfg.snr_threshold);
  size_t nt = 0, ntext = 0;
  auto put = [&](char ch) {
    if (ntext + 1 < text_cap) text[ntext++] = ch;
  };
  for (size_t i = 0; i < na; i++) {
    rx.block = i;
    const uchirp::loop_event le = loop.step(rx, put);
    if (trace && nt < trace_cap) {
      uc_rx_event& ev = trace[nt];
      ev.block = accepted[i];
      ev.sync_position = le.sync_position;
      ev.state_before = (uint8_t)le.state_before;
      ev.state_after = (uint8_t)le.state_after;
      ev.bit = (int8_t)le.bit;
      ev.reserved = 0;
      ev.snr_up = le.snr_up;
      ev.snr_down = le.snr_down;
    }
    nt++;
  }
  text[ntext] = '\0';
  if (n_trace) *n_trace = nt;
  return (int)ntext;
}

extern "C" int uc_receive_stream(uc_ctx* c, const void* samples, int dtype, size_t n_samples, char* text,
                                 size_t text_cap, uc_rx_event* trace, size_t trace_cap, size_t* n_trace) {
  return uc_receive_stream_isr(c, samples, dtype, n_samples, nullptr, text, text_cap, trace, trace_cap, n_trace);
}

// ---------------------------------------------------------------------------
// uc_receive_streams[_next]: the same receiver for MANY streams at once, recorded or live.  Per call: (busy mask only:
// accept + pack) -> ONE launch of the band kernel's ROWS build over the 8 FIFO offsets every accepted block adds (it reads
// the caller's buffer as it lies) -> main()'s switch replayed on the device, one wave or one lane per stream
// (csrc/uc_rx_kernel.hip: include/uchirp_mainloop.hpp compiled for the device) -> (live: the stream's newest block is kept).
// ---------------------------------------------------------------------------
// live streams: what n_streams receivers carry from one call to the next, all of it on the device
struct uc_rx_state {
  uc_ctx* c = nullptr;
  int device = 0;
  size_t n_streams = 0;
  uint32_t* d_last = nullptr;   // [2][n_streams][n] words: the newest ACCEPTED block of every stream -- the part of the FIFO the
                                // next block's new offsets still read; zeros at power-on (fifo_queue, main.c:94).  Two
                                // halves: a call reads half *d_parity and fills the other one (the band kernel stores the
                                // block on its way through; no copy kernel, no frame reads what another writes)
  unsigned int* d_parity = nullptr;  // which half is current; flipped by the replay kernel of every call
  float2* d_carry = nullptr;    // [n_streams][9]: (up, down) mag_max of the 9 FIFO offsets that survive the ISR's shift
                                // (main.c:662): offsets n .. 2 n of the FIFO become 0 .. n of the next one; zeros at power-on
  uint32_t* d_loop = nullptr;   // [n_streams][rx_loop_words()]: main()'s locals (main.c:314-339) + blocks offered so far
  uint32_t* d_need = nullptr;   // [n_streams]: what the switch can still look at of every stream's NEXT block (which of its 8 new
                                // FIFO offsets, and the DOWN statistics: main.c:447-453; uc_rx.hpp); written by every replay
  uint32_t* d_hist = nullptr;   // [n_streams][4] PDM words: the DFSDM's sinc^5 history of every microphone (UC_DTYPE_PDM chunks)
  RxScratch rx;                 // scratch of the call in flight
  uint64_t blocks_seen = 0;     // host mirror of the block count (the overflow check only; a replayed graph does not bump it)
  int dtype = -1;               // of the words in d_last (the first call decides)
  // uc_rx_state_keep_previous: the caller keeps the chunk of every call alive and unchanged until the NEXT call on the state has
  // completed, so "the block in front" of a call's first block is read where the previous call's samples lie -- nothing is
  // saved into d_last.  kept = the last block of stream 0 of the previous call's chunk (device memory of the caller),
  // kept_pitch elements from stream to stream; nullptr: the FIFO's newest block is in d_last (power-on, after a busy-masked
  // call, after a call on host memory).
  bool keep = false;
  const void* kept = nullptr;
  size_t kept_pitch = 0;
};

// Everything uc_receive_streams[_next] refuses for its ARGUMENTS, and nothing else: no HIP call that enqueues, no allocation.
// receive_streams_impl runs it first; uc_group_receive_streams[_next] runs it for EVERY local device before it touches a stream
// (uc_group.cpp: a refused group call has started nothing -- ADVICE r5).  0, or the negative code the call would return.
int uc::receive_streams_check(uc_ctx* c, uc_rx_state* st, bool live, const void* samples, int dtype, size_t n_streams, size_t n_samples,
                              size_t stream_stride_elems, const char* text, size_t text_cap, size_t trace_cap) {
  if (!c || !text || text_cap == 0) return fail(-EINVAL, "uc_receive_streams: NULL argument");
  if (live && (!st || st->c != c)) return fail(-EINVAL, "uc_receive_streams_next: the state belongs to another context");
  if (c->cfg.variant != UC_RX_REAL && c->cfg.variant != UC_SYNC_CPLX)
    return fail(-ENOTSUP, "uc_receive_streams: variant %d has no up/down state machine", (int)c->cfg.variant);
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32 && dtype != UC_DTYPE_PDM)
    return fail(-EINVAL, "uc_receive_streams: bad dtype %d", dtype);
  if (n_streams == 0) return 0;
  const uint32_t n = c->cfg.n;
  if (stream_stride_elems == 0) stream_stride_elems = n_samples;
  if (stream_stride_elems < n_samples) return fail(-EINVAL, "uc_receive_streams: streams overlap (stride %zu < %zu samples)",
                                                   stream_stride_elems, n_samples);
  const size_t nb = n_samples / n;
  if (nb >= ((size_t)1 << 31) / 8 || text_cap >= ((size_t)1 << 31) || trace_cap >= ((size_t)1 << 31))
    return fail(-EINVAL, "uc_receive_streams: stream too long");
  if (st) {
    if (n_samples % n != 0) return fail(-EINVAL, "uc_receive_streams_next: %zu samples are not whole blocks of %u", n_samples, n);
    if (st->dtype >= 0 && st->dtype != dtype && st->blocks_seen)
      return fail(-EINVAL, "uc_receive_streams_next: the streams began as dtype %d", st->dtype);
    if (st->blocks_seen + nb >= ((uint64_t)1 << 32)) return fail(-EOVERFLOW, "uc_receive_streams_next: 2^32 blocks per stream");
  }
  if (nb) {
    if (!samples) return fail(-EINVAL, "uc_receive_streams: samples is NULL");
    if (n_streams * nb * (size_t)(n / 256) >= ((size_t)1 << 31))
      return fail(-EINVAL, "uc_receive_streams: %zu new FIFO offsets in one call (at most 2^31 - 1)", n_streams * nb * (size_t)(n / 256));
    // UC_DTYPE_PDM on device memory: the DFSDM kernel's alignment rules (a host buffer is staged into aligned scratch)
    if (dtype == UC_DTYPE_PDM && is_device_ptr(samples) &&
        (((uintptr_t)samples & 15u) != 0 || (n_streams > 1 && (stream_stride_elems & 3u) != 0)))
      return fail(-EINVAL, "uc_receive_streams: UC_DTYPE_PDM device buffers must be 16-byte aligned, the stride a multiple of 4 words");
  }
  return 0;
}

// A call of SEVERAL blocks served block by block: nb one-block steps of the live form, back to back on the stream -- every step
// evaluates only the FIFO offsets main()'s switch can still look at (the need words its predecessor left: 3 or 5 of the 8 new
// offsets of an idle stream, the UP reference only; 5 or 6 around a locked sync_position: receiver/Src/main.c:447-453, 491-550)
// where ONE launch over all nb x 8 offsets must evaluate everything, because what the switch will look at in block k is only
// known once block k - 1 has gone through it.  "The block in front" of step b >= 1 is block b - 1 of the SAME buffer (nothing is
// copied); a live state's newest block is handed over by the last step alone.  Texts and traces are the one-launch call's, bit
// for bit (tests/test_gpu_receive_many.py).  Worth it when a step fills enough of the chip (profiles/r06_steps_ab.txt, 176-block
// streams with one transmission each): the complex receiver from 1024 streams on (5.8 -> 5.2 ms; 4096 streams 21.7 -> 14.6 ms,
// 16 384 streams 86 -> 52 ms), RX_REAL from 8192 (19.2 -> 16.8 ms; 32 768 streams 75 -> 58 ms); below that one launch wins.
static int receive_steps(uc_ctx* c, uc_rx_state* st, RxScratch& sc, const void* d_in, int dtype, size_t n_streams, size_t nb,
                         size_t stride, bool keep_next, char* d_text, size_t text_cap, uint32_t* d_ntext, uc_rx_event* d_trace,
                         size_t trace_cap, uint32_t* d_ntrace, hipStream_t stream) {
  const uint32_t n = c->cfg.n, per_block = n / 256;
  float2* d_carry = st ? st->d_carry : (float2*)sc.carry.p;
  uint32_t* d_loop = st ? st->d_loop : (uint32_t*)sc.loop.p;
  uint32_t* d_need = st ? st->d_need : (uint32_t*)sc.need.p;
  hipError_t e = hipMemsetAsync(sc.fill.p, 0, n_streams * sizeof(uint32_t), stream);
  if (!st) {  // streams that start with this call: every receiver at power-on (uc_rx_state_reset does the same for a state)
    if (e == hipSuccess) e = hipMemsetAsync(d_carry, 0, n_streams * (size_t)(per_block + 1) * sizeof(float2), stream);
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)d_need, (int)uc::need_word(UC_STATE_IDLE, 0, 0), n_streams, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(stepped receivers)");
    const int lrc = uc::launch_rx_state_init(d_loop, n_streams, n, c->cfg.snr_threshold, stream);
    if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx state init kernel launch");
  }
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(stepped receivers)");
  for (size_t b = 0; b < nb; b++) {
    const bool last = b + 1 == nb;
    uc::BandParams bp;
    memset(&bp, 0, sizeof(bp));
    bp.frames = (const char*)d_in + b * (size_t)n * 4;
    bp.n_frames = n_streams * per_block;
    bp.magmax = (float2*)sc.rec.p;
    bp.row_pitch = stride;
    bp.row_blocks = 1;
    uc::rows_divisor(1u, &bp.div_magic, &bp.div_shift);
    bp.need = d_need;
    bp.poison = c->rx_poison ? 1u : 0u;
    if (b == 0) {  // in front of the call's first block: what the state holds (or a kept chunk), zeros for streams that start
      bp.prev = st ? (const void*)st->d_last : (const void*)c->d_zero_block;
      bp.prev_pitch = st ? (size_t)n : 0;
      bp.prev_half = st ? n_streams * (size_t)n : 0;
      bp.parity = st ? st->d_parity : nullptr;
      if (st && st->kept) {
        bp.prev = st->kept;
        bp.prev_pitch = st->kept_pitch;
        bp.prev_half = 0;
      }
    } else {       // ... of every other block: the block before it, where it lies
      bp.prev = (const char*)bp.frames - (size_t)n * 4;
      bp.prev_pitch = stride;
      bp.prev_half = 0;
      bp.parity = st ? st->d_parity : nullptr;
    }
    if (st && last && !keep_next) {  // the state's newest block: handed over by the last step (its m = 7 or 8 frames)
      bp.save = 1u;
      bp.save_to = st->d_last;
      bp.save_half = n_streams * (size_t)n;
    }
    if (int rc = band_launch(c, bp, dtype, stream)) return rc;
    uc::RxParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.rec = (const float2*)sc.rec.p;
    rp.rec_pitch = per_block;
    rp.carry = d_carry;
    rp.carry_pitch = per_block + 1;
    rp.carry_out = d_carry;
    rp.n_streams = n_streams;
    rp.n = n;
    rp.nb = 1;
    rp.snr_threshold = c->cfg.snr_threshold;
    rp.text = d_text;
    rp.text_cap = (uint32_t)text_cap;
    rp.n_text = d_ntext;
    rp.trace = d_trace;
    rp.trace_cap = (uint32_t)trace_cap;
    rp.n_trace = d_ntrace;
    rp.loop_state = d_loop;
    rp.parity = (st && last) ? st->d_parity : nullptr;  // (one flip per call, behind the step that filled the other half)
    rp.need = d_need;
    rp.need_force = c->rx_need_force;
    rp.fill = (uint32_t*)sc.fill.p;
    rp.trace_start = (uint32_t)b;
    const int lrc = uc::launch_rx_replay(rp, stream);
    if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx replay kernel launch");
  }
  return 0;
}

static int receive_streams_impl(uc_ctx* c, uc_rx_state* st, const void* samples, int dtype, size_t n_streams, size_t n_samples,
                                size_t stream_stride_elems, const uint8_t* busy, char* text, size_t text_cap,
                                uint32_t* n_text, uc_rx_event* trace, size_t trace_cap, uint32_t* n_trace,
                                void* hip_stream) {
  if (const int rc = uc::receive_streams_check(c, st, st != nullptr, samples, dtype, n_streams, n_samples, stream_stride_elems, text,
                                               text_cap, trace_cap))
    return rc;
  if (n_streams == 0) return 0;
  const uint32_t n = c->cfg.n;
  const uint32_t per_block = n / 256;  // new FIFO offsets per accepted block
  const bool pdm = dtype == UC_DTYPE_PDM;
  const int dtype_in = dtype;
  const size_t stream_stride_in = stream_stride_elems ? stream_stride_elems : n_samples;
  if (stream_stride_elems == 0) stream_stride_elems = n_samples;
  const size_t nb = n_samples / n;
  if (trace && trace_cap == 0) trace = nullptr;
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;
  bool host_out = false;
  RxScratch& sc = st ? st->rx : c->rx;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  // Capture is the live form's: a uc_rx_state owns its scratch.  The context's scratch serves every state-less call, on any
  // stream, ordered by an event a capture cannot carry -- a captured state-less call would share it with eager calls and
  // replays without any ordering
  if (!st && capturing)
    return fail(-ENOTSUP, "uc_receive_streams: the stream is being captured -- only uc_receive_streams_next (a uc_rx_state owns "
                          "its scratch) can be captured into a hipGraph");
  const CaptureNoAlloc no_alloc(capturing);
  if (!st && !capturing) {
    // the context's scratch serves one call at a time: a call on another stream than the last one waits, ON THE DEVICE, for
    // that one's kernels (a live state has scratch of its own and needs none of this)
    if (c->rx_used && c->rx_stream != stream) {
      const RelaxedCapture relaxed;
      e = hipStreamWaitEvent(stream, c->rx_ev, 0);
      if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent(receiver scratch)");
    }
  }

  // outputs: straight into device buffers of the caller, else through staging
  char* d_text = text;
  if (!is_device_ptr(text)) {
    if (int rc = sc.text.ensure(n_streams * text_cap)) return rc;
    d_text = (char*)sc.text.p;
    host_out = true;
  }
  uint32_t* d_ntext = n_text;
  if (n_text && !is_device_ptr(n_text)) {
    if (int rc = sc.ntext.ensure(n_streams * sizeof(uint32_t))) return rc;
    d_ntext = (uint32_t*)sc.ntext.p;
    host_out = true;
  }
  uc_rx_event* d_trace = trace;
  if (trace && !is_device_ptr(trace)) {
    if (int rc = sc.trace.ensure(n_streams * trace_cap * sizeof(uc_rx_event))) return rc;
    d_trace = (uc_rx_event*)sc.trace.p;
    host_out = true;
  }
  uint32_t* d_ntrace = n_trace;
  if (n_trace && !is_device_ptr(n_trace)) {
    if (int rc = sc.ntrace.ensure(n_streams * sizeof(uint32_t))) return rc;
    d_ntrace = (uint32_t*)sc.ntrace.p;
    host_out = true;
  }
  if (nb == 0) {  // nothing to process: empty texts, zero counts
    e = hipMemsetAsync(d_text, 0, n_streams * text_cap, stream);
    if (e == hipSuccess && d_ntext) e = hipMemsetAsync(d_ntext, 0, n_streams * sizeof(uint32_t), stream);
    if (e == hipSuccess && d_ntrace) e = hipMemsetAsync(d_ntrace, 0, n_streams * sizeof(uint32_t), stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(empty streams)");
  }

  if (nb) {
    const size_t n_frames = n_streams * nb * per_block;
    // (every scratch buffer of the call is sized before its first launch: a call that cannot be served -- out of memory, or a
    // capture that would have to allocate -- has enqueued nothing)
    if (int rc = sc.rec.ensure(n_frames * sizeof(float2))) return rc;
    // a call of several blocks without a busy mask is served block by block when a step fills the chip (receive_steps)
    const size_t step_min = c->rx_step_min >= 0 ? (size_t)c->rx_step_min : (c->cfg.variant == UC_SYNC_CPLX ? 1024u : 8192u);
    const bool stepped = !busy && nb > 1 && n_streams >= step_min && !capturing;
    if (stepped) {
      int rc = sc.fill.ensure(n_streams * sizeof(uint32_t));
      if (!rc && !st) rc = sc.carry.ensure(n_streams * (size_t)(per_block + 1) * sizeof(float2));
      if (!rc && !st) rc = sc.loop.ensure(n_streams * (size_t)uc::rx_loop_words() * 4);
      if (!rc && !st) rc = sc.need.ensure(n_streams * sizeof(uint32_t));
      if (rc) return rc;
    }
    if (pdm)
      if (int rc = sc.pcm.ensure(n_streams * nb * (size_t)n * 4)) return rc;
    if (busy) {
      int rc = sc.acc.ensure(n_streams * nb * sizeof(uint32_t));
      if (!rc) rc = sc.na.ensure(n_streams * sizeof(uint32_t));
      if (!rc) rc = sc.pad.ensure(n_streams * nb * (size_t)n * 4);
      if (!rc && !is_device_ptr(busy)) rc = sc.busy.ensure(n_streams * nb);
      if (rc) return rc;
    }
    // inputs
    const void* d_in = samples;
    if (!is_device_ptr(samples)) {
      const size_t span = (n_streams - 1) * stream_stride_elems + nb * (size_t)n;
      if (int rc = sc.in.ensure(span * 4)) return rc;
      e = hipMemcpyAsync(sc.in.p, samples, span * 4, hipMemcpyHostToDevice, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(streams)");
      d_in = sc.in.p;
    }
    if (pdm) {
      // The chain starts at the microphones' bit streams: the DFSDM (receiver/Src/dfsdm.c:59-61,69,78) turns every 32 PDM
      // bits into one word of `buf[]` -- whole blocks here, the filter history of every stream carried in the state (or,
      // for streams that start with this call, the bit pattern of a silent microphone) -- and the ISR sees int32 words.
      // The peripheral filters whether or not the ISR later drops the block: the busy mask applies behind it.
      uint32_t* d_hist = st ? st->d_hist : nullptr;
      if (!st) {
        if (int rc = sc.hist.ensure(n_streams * 16)) return rc;
        d_hist = (uint32_t*)sc.hist.p;
        e = hipMemsetD32Async((hipDeviceptr_t)d_hist, (int)UC_PDM_SILENCE, n_streams * 4, stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetD32Async(pdm history)");
      }
      if (int rc = sc.pcm.ensure(n_streams * nb * (size_t)n * 4)) return rc;
      if (int rc = sinc5_streams_launch(c, (const uint32_t*)d_in, n_streams, nb * (size_t)n, stream_stride_elems, d_hist, true,
                                        (int32_t*)sc.pcm.p, nb * (size_t)n, stream))
        return rc;
      d_in = sc.pcm.p;
      stream_stride_elems = nb * (size_t)n;
      dtype = UC_DTYPE_I32;
    }
    // The ISR (main.c:659-668) appends a block only when the main loop has consumed the previous one; a block that arrives
    // while it is busy is DROPPED, the FIFO is not shifted.  The FIFO therefore only ever holds ACCEPTED blocks: with a busy
    // mask they are first laid out one behind the other; without one the caller's buffer is read as it lies.
    const void* rows = d_in;
    size_t row_pitch = stream_stride_elems;
    const uint32_t* d_acc = nullptr;
    const uint32_t* d_na = nullptr;
    // uc_rx_state_keep_previous holds for chunks the caller owns on the device and calls that accept every block; a call on
    // host memory (staged by the library), from PDM bits (the DFSDM words are the library's) or with a busy mask (the newest
    // ACCEPTED block differs by stream) hands the FIFO's newest block to the state as ever
    const bool keep_next = st && st->keep && !busy && !pdm && d_in == samples;
    if (st && st->kept && busy) {
      // a busy-masked call behind kept chunks: a stream whose blocks are all dropped keeps its FIFO, so the kept blocks go
      // into the state's current half first -- from here on this call is an ordinary one
      const bool al16 = (((uintptr_t)st->kept | (uintptr_t)st->d_last) & 15u) == 0 && (st->kept_pitch & 3u) == 0 && (n & 3u) == 0;
      const int lrc = uc::launch_rx_keep(st->kept, st->kept_pitch, n, n_streams, st->d_last, st->d_parity, al16, stream);
      if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx keep kernel launch");
      st->kept = nullptr;
      st->kept_pitch = 0;
    }
    if (busy) {
      const uint8_t* d_busy = busy;
      if (!is_device_ptr(busy)) {
        if (int rc = sc.busy.ensure(n_streams * nb)) return rc;
        e = hipMemcpyAsync(sc.busy.p, busy, n_streams * nb, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(busy)");
        d_busy = (const uint8_t*)sc.busy.p;
      }
      int rc = sc.acc.ensure(n_streams * nb * sizeof(uint32_t));
      if (!rc) rc = sc.na.ensure(n_streams * sizeof(uint32_t));
      if (!rc) rc = sc.pad.ensure(n_streams * nb * (size_t)n * 4);
      if (rc) return rc;
      int lrc = uc::launch_rx_accept(d_busy, n_streams, (uint32_t)nb, (uint32_t*)sc.acc.p, (uint32_t*)sc.na.p, stream);
      if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx accept kernel launch");
      d_acc = (const uint32_t*)sc.acc.p;
      d_na = (const uint32_t*)sc.na.p;
      const bool al16 = (((uintptr_t)d_in | (uintptr_t)sc.pad.p) & 15u) == 0 && (stream_stride_elems & 3u) == 0 && (n & 3u) == 0;
      lrc = uc::launch_rx_pack(d_in, stream_stride_elems, n, (uint32_t)nb, n_streams, d_acc, d_na, sc.pad.p, nb * (size_t)n, al16,
                               stream);
      if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx pack kernel launch");
      rows = sc.pad.p;
      row_pitch = nb * (size_t)n;
    }
    if (stepped) {
      if (int rc = receive_steps(c, st, sc, rows, dtype, n_streams, nb, row_pitch, keep_next, d_text, text_cap, d_ntext, d_trace,
                                 trace_cap, d_ntrace, stream))
        return rc;
    } else {
    if (int rc = sc.rec.ensure(n_frames * sizeof(float2))) return rc;
    // dsp() at the 8 FIFO offsets every accepted block ADDS (the other 9 of its FIFO were evaluated when the block before
    // it arrived, main.c:662): frame (s, k, m) = the last n - 256 m samples of the block in front of the stream's k-th
    // block followed by the first 256 m of that block -- in front of block 0: the newest block of the previous call (a
    // live state) or zeros (power-on).  (Rows of a busy-masked call beyond na[s] hold stale words: evaluated, never read.)
    {
      uc::BandParams bp;
      memset(&bp, 0, sizeof(bp));
      bp.frames = rows;
      bp.n_frames = n_frames;
      bp.magmax = (float2*)sc.rec.p;
      bp.prev = st ? (const void*)st->d_last : (const void*)c->d_zero_block;
      bp.prev_pitch = st ? (size_t)n : 0;
      bp.prev_half = st ? n_streams * (size_t)n : 0;
      bp.parity = st ? st->d_parity : nullptr;
      bp.save = (st && !busy) ? 1u : 0u;  // (with a busy mask the last ACCEPTED block differs by stream: launch_rx_last)
      bp.save_to = st ? (void*)st->d_last : nullptr;
      bp.save_half = bp.prev_half;
      if (st && st->kept) {
        // the caller has kept the previous chunk (uc_rx_state_keep_previous): the block in front is read where it lies
        bp.prev = st->kept;
        bp.prev_pitch = st->kept_pitch;
        bp.prev_half = 0;
      }
      if (keep_next) bp.save = 0u;  // ... and this call's chunk will be kept for the next one: nothing to hand over
      // acquisition evaluates 4 positions a block, the UP reference only: a stream that is IDLE when its ONE new block arrives
      // gets the 3 or 5 transforms the switch can still look at (SYNC_CPLX: of the UP reference only) instead of 8
      bp.need = (st && nb == 1) ? st->d_need : nullptr;
      bp.poison = c->rx_poison ? 1u : 0u;
      bp.row_pitch = row_pitch;
      bp.row_blocks = (uint32_t)nb;
      uc::rows_divisor((uint32_t)nb, &bp.div_magic, &bp.div_shift);
      if (int rc = band_launch(c, bp, dtype, stream)) return rc;
    }
    uc::RxParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.rec = (const float2*)sc.rec.p;
    rp.rec_pitch = nb * per_block;
    rp.carry = st ? st->d_carry : (const float2*)c->d_zero_block;
    rp.carry_pitch = st ? per_block + 1 : 0;
    rp.carry_out = st ? st->d_carry : nullptr;
    rp.n_streams = n_streams;
    rp.n = n;
    rp.nb = (uint32_t)nb;
    rp.snr_threshold = c->cfg.snr_threshold;
    rp.acc = d_acc;
    rp.na = d_na;
    rp.text = d_text;
    rp.text_cap = (uint32_t)text_cap;
    rp.n_text = d_ntext;
    rp.trace = d_trace;
    rp.trace_cap = (uint32_t)trace_cap;
    rp.n_trace = d_ntrace;
    rp.loop_state = st ? st->d_loop : nullptr;
    rp.parity = st ? st->d_parity : nullptr;
    rp.need = st ? st->d_need : nullptr;
    rp.need_force = c->rx_need_force;
    int lrc;
    if (st && busy) {
      // what the next call's new offsets still read of this one: every stream's newest ACCEPTED block
      const bool al16 = (((uintptr_t)rows | (uintptr_t)st->d_last) & 15u) == 0 && (row_pitch & 3u) == 0 && (n & 3u) == 0;
      lrc = uc::launch_rx_last(rows, row_pitch, d_na, (uint32_t)nb, n, n_streams, st->d_last, st->d_parity, al16, stream);
      if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx last-block kernel launch");
    }
    lrc = uc::launch_rx_replay(rp, stream);
    if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx replay kernel launch");
    }
    if (st) {
      st->blocks_seen += nb;
      st->dtype = dtype_in;
      st->kept = keep_next ? (const void*)((const char*)samples + (nb - 1) * (size_t)n * 4) : nullptr;
      st->kept_pitch = keep_next ? stream_stride_in : 0;
    }
  }
  if (host_out) {
    if (d_text != text) e = hipMemcpyAsync(text, d_text, n_streams * text_cap, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && n_text && d_ntext != n_text)
      e = hipMemcpyAsync(n_text, d_ntext, n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && trace && d_trace != trace)
      e = hipMemcpyAsync(trace, d_trace, n_streams * trace_cap * sizeof(uc_rx_event), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && n_trace && d_ntrace != n_trace)
      e = hipMemcpyAsync(n_trace, d_ntrace, n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "uc_receive_streams: copy back");
  }
  if (!st && !capturing) {
    const RelaxedCapture relaxed;
    if (hipEventRecord(c->rx_ev, stream) == hipSuccess) {
      c->rx_used = true;
      c->rx_stream = stream;
    } else {
      // (a capture elsewhere on this thread forbids the call): nothing can be waited for later -- drain now
      (void)hipGetLastError();
      (void)hipStreamSynchronize(stream);
      c->rx_used = false;
    }
  }
  return 0;
}

extern "C" int uc_receive_streams(uc_ctx* c, const void* samples, int dtype, size_t n_streams, size_t n_samples,
                                  size_t stream_stride_elems, const uint8_t* busy, char* text, size_t text_cap,
                                  uint32_t* n_text, uc_rx_event* trace, size_t trace_cap, uint32_t* n_trace,
                                  void* hip_stream) {
  return receive_streams_impl(c, nullptr, samples, dtype, n_streams, n_samples, stream_stride_elems, busy, text, text_cap, n_text,
                              trace, trace_cap, n_trace, hip_stream);
}

// ---- live streams: the same call, chunk after chunk ----------------------------------------------------------------
extern "C" void uc_rx_state_destroy(uc_rx_state* st) {
  if (!st) return;
  (void)hipSetDevice(st->device);  // (not through st->c: a state may outlive its context by mistake; its memory is its own)
  if (st->d_last) (void)hipFree(st->d_last);
  if (st->d_carry) (void)hipFree(st->d_carry);
  if (st->d_loop) (void)hipFree(st->d_loop);
  if (st->d_parity) (void)hipFree(st->d_parity);
  if (st->d_hist) (void)hipFree(st->d_hist);
  if (st->d_need) (void)hipFree(st->d_need);
  st->rx.release();
  delete st;
}

extern "C" size_t uc_rx_state_streams(const uc_rx_state* st) { return st ? st->n_streams : 0; }

extern "C" int uc_rx_state_keep_previous(uc_rx_state* st, int on) {
  if (!st) return fail(-EINVAL, "uc_rx_state_keep_previous: NULL state");
  st->keep = on != 0;  // (switched off: the NEXT call still reads the kept chunk in front of its own, and saves its own)
  return 0;
}

extern "C" int uc_rx_state_reset(uc_rx_state* st, void* hip_stream) {
  if (!st) return fail(-EINVAL, "uc_rx_state_reset: NULL state");
  uc_ctx* c = st->c;
  hipError_t e = hipSetDevice(st->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;
  const uint32_t n = c->cfg.n;
  e = hipMemsetAsync(st->d_last, 0, 2 * st->n_streams * (size_t)n * 4, stream);
  if (e == hipSuccess) e = hipMemsetAsync(st->d_parity, 0, sizeof(unsigned int), stream);
  if (e == hipSuccess)  // (power-on: IDLE, turn 0 -- the word the replay would have left)
    e = hipMemsetD32Async((hipDeviceptr_t)st->d_need, (int)uc::need_word(UC_STATE_IDLE, 0, 0), st->n_streams, stream);
  if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)st->d_hist, (int)UC_PDM_SILENCE, st->n_streams * 4, stream);
  if (e == hipSuccess) e = hipMemsetAsync(st->d_carry, 0, st->n_streams * (size_t)(n / 256 + 1) * sizeof(float2), stream);
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(FIFO state)");
  const int lrc = uc::launch_rx_state_init(st->d_loop, st->n_streams, n, c->cfg.snr_threshold, stream);
  if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx state init kernel launch");
  st->blocks_seen = 0;
  st->dtype = -1;
  st->kept = nullptr;
  st->kept_pitch = 0;
  return 0;
}

extern "C" int uc_rx_state_create(uc_ctx* c, size_t n_streams, uc_rx_state** out) {
  if (!c || !out) return fail(-EINVAL, "uc_rx_state_create: NULL argument");
  *out = nullptr;
  if (c->cfg.variant != UC_RX_REAL && c->cfg.variant != UC_SYNC_CPLX)
    return fail(-ENOTSUP, "uc_rx_state_create: variant %d has no up/down state machine", (int)c->cfg.variant);
  if (n_streams == 0) return fail(-EINVAL, "uc_rx_state_create: no streams");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  uc_rx_state* st = new (std::nothrow) uc_rx_state();
  if (!st) return fail(-ENOMEM, "uc_rx_state_create: out of memory");
  st->c = c;
  st->device = c->device;
  st->n_streams = n_streams;
  const uint32_t n = c->cfg.n;
  e = hipMalloc((void**)&st->d_last, 2 * n_streams * (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&st->d_parity, 256);
  if (e == hipSuccess) e = hipMalloc((void**)&st->d_need, n_streams * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&st->d_hist, n_streams * 16);
  if (e == hipSuccess) e = hipMalloc((void**)&st->d_carry, n_streams * (size_t)(n / 256 + 1) * sizeof(float2));
  if (e == hipSuccess) e = hipMalloc((void**)&st->d_loop, n_streams * (size_t)uc::rx_loop_words() * 4);
  if (e != hipSuccess) {
    uc_rx_state_destroy(st);
    return hip_fail(e, "hipMalloc(rx state)");
  }
  int rc = uc_rx_state_reset(st, nullptr);
  if (!rc) {
    e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
  }
  if (rc) {
    uc_rx_state_destroy(st);
    return rc;
  }
  *out = st;
  return 0;
}

extern "C" int uc_receive_streams_next(uc_ctx* c, uc_rx_state* st, const void* samples, int dtype, size_t n_samples,
                                       size_t stream_stride_elems, const uint8_t* busy, char* text, size_t text_cap,
                                       uint32_t* n_text, uc_rx_event* trace, size_t trace_cap, uint32_t* n_trace,
                                       void* hip_stream) {
  if (!st || !c || st->c != c) return fail(-EINVAL, "uc_receive_streams_next: the state belongs to another context");
  return receive_streams_impl(c, st, samples, dtype, st->n_streams, n_samples, stream_stride_elems, busy, text, text_cap, n_text,
                              trace, trace_cap, n_trace, hip_stream);
}
