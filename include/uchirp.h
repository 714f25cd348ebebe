/*
 * uchirp.h -- C-ABI of the MI355X chirp-demodulation engine (libuchirp.so).
 *
 * This is the drop-in boundary for the per-frame DSP path of
 * araobp/ultrasonic-communication.  The reference has no FFI layer: the path
 * is a set of file-scope C functions over globals.  Every entry point below
 * names the reference symbol(s) whose call contract it replaces
 * (paths relative to the reference checkout).
 *
 *   uc_create            <- inline init in main():  receiver/Src/main.c:367-393
 *                           (fs, bandwidth, idx_left_zero, arm_rfft_fast_init_f32,
 *                           Hann table) + init_ref_chirp(), receiver/Src/chirp.c:16-45
 *   uc_process_frame     <- symbol_snr(pos,&h[0],UP) + symbol_snr(pos,&h[1],DOWN)
 *                           + the bit decision, receiver/Src/main.c:233-236,518-531
 *   uc_process_batch     <- N x dsp(), receiver/Src/main.c:183-231, with the
 *                           int32->float ingest cast of the DFSDM ISR fused in,
 *                           receiver/Src/main.c:659-668
 *   uc_idx2freq          <- idx2freq(), receiver/Src/main.c:154-160
 *   uc_get_table         <- the globals up_chirp/down_chirp/hann_window,
 *   uc_set_table            receiver/Src/chirp.c:13-14, receiver/Src/main.c:99
 *   uc_window_spectrum   <- pipeline(), receiver/Src/main.c:163-180: the magnitudes it leaves in
 *                           signal[0..NN), for the bins dsp() then looks at (main.c:205-208)
 *   uc_destroy           <- (none; the firmware never frees)
 *
 * Variants select the sibling pipelines:
 *   UC_RX_REAL       receiver/Src/main.c:163-231            (shipping receiver)
 *   UC_SYNC_CPLX     experiments/synchronization/Src/main.c:144-213
 *   UC_COMPRESS      experiments/chirp_compression_time_domain/Src/chirp.c:78-83
 *   UC_DECHIRP_DOWN  experiments/chirp_compression_freq_domain/Src/main.c:113-160
 *   UC_IQ            experiments/iq_modulation/Src/main.c:117-134 + iq_modem.c:55-75
 *   UC_STREAM        streaming front-end + overlap-save chirp compression (BASELINE config 4):
 *                    carrier mix + the 27-tap FIR of iq_modem.c:18,55-75, decimation by
 *                    cfg.decim, then FFT x H x IFFT as compress_chirp()
 *                    (chirp_compression_time_domain/Src/chirp.c:78-83) run as overlap-save
 *                    over a continuous stream -- uc_process_stream()
 *
 * Conventions: 0 on success, negative errno-style code on failure, never
 * aborts.  The caller owns every buffer.  `frames`, `mag_mean`, `symbols`
 * and `stats` of uc_process_batch may each be host or device pointers
 * (detected with hipPointerGetAttributes); with device pointers the call is
 * asynchronous on `hip_stream`.  A uc_ctx is not thread-safe; distinct
 * contexts are independent.  There is NO CPU backend: uc_create fails with
 * -ENODEV when no HIP device is usable.
 */
#ifndef UCHIRP_H_
#define UCHIRP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UC_ABI_VERSION 7

/* pipeline variants */
enum {
  UC_RX_REAL      = 0,
  UC_SYNC_CPLX    = 1,
  UC_COMPRESS     = 2,
  UC_DECHIRP_DOWN = 3,
  UC_IQ           = 4,
  UC_STREAM       = 5,
  UC_NUM_VARIANTS = 6
};

/* `typedef enum {DOWN_CHIRP, UP_CHIRP} chirp;`  receiver/Inc/chirp.h:12-14 */
enum { UC_DOWN_CHIRP = 0, UC_UP_CHIRP = 1 };

/* element type of `frames` */
enum {
  UC_DTYPE_I32 = 0, /* raw DFSDM words, cast with (float) -- main.c:664 */
  UC_DTYPE_F32 = 1, /* already-cast fifo_queue contents -- main.c:94   */
  UC_DTYPE_PDM = 2  /* uc_receive_streams[_next] and their uc_group_ forms only: the microphones' 1-bit PDM streams, 32 bits
                       per word (uc_dfsdm_sinc5's packing), one word per sample -- the DFSDM in front of the ISR
                       (receiver/Src/dfsdm.c:59-61,69,78) runs on the device, its filter history travels in the uc_rx_state */
};

/* four PDM words of a silent microphone (alternating bits: the sinc^5 output is 0 +- 1 LSB): the filter history of a stream
 * that STARTS (uc_rx_state at power-on; every stream of uc_receive_streams) */
#define UC_PDM_SILENCE 0xAAAAAAAAu

/* symbol_out values */
#define UC_SYM_DOWN 0u    /* bit low : snr_down >  snr_up  (main.c:523-526) */
#define UC_SYM_UP   1u    /* bit high: otherwise           (main.c:527-531) */
#define UC_SYM_NONE 0xFFu /* neither snr >= SNR_THRESHOLD  (main.c:521,539) */

/* flags: each documents a quirk decision of SURVEY.md section 8a */
#define UC_FLAG_LIBM_TRIG   (1u << 0) /* build tables with exact sin/cos instead of
                                         the CMSIS-DSP LUT+interpolation restatement */
#define UC_FLAG_TRUE_DC     (1u << 1) /* Q2 fix: mag[0] = |X0| instead of the packed
                                         hypot(Re X0, Re X[N/2]) the reference computes */
#define UC_FLAG_FUSED_TABLE (1u << 2) /* informational: the HIP path always multiplies by
                                         ONE fp32 table ref*hann (one rounding) where the
                                         reference rounds (x*ref) then (*hann) */

#define UC_FLAG_STREAM_UP   (1u << 3) /* UC_STREAM: convolve with the UP template (compresses
                                         down chirps) instead of the default DOWN template */

#define UC_FLAG_IQ_BASEBAND (1u << 4) /* UC_IQ: the INTENDED maths of the experiment, simulation/IQ_modulation.ipynb
                                         cells 16-31, instead of the unfinished firmware windows:
                                         after mix + low-pass, R = I + jQ is multiplied by the CONJUGATE of the
                                         base-band up chirp (history 0) and of the base-band down chirp (history 1),
                                         band edges f0 - carrier .. f1 - carrier; Hann; complex FFT; the two windows
                                         are the `bandwidth` bins either side of DC, [0, bandwidth) and
                                         [n - bandwidth, n); signed idx2freq as the receiver's; two uc_stats per
                                         frame {up, down} and an up/down symbol, as RX_REAL / SYNC_CPLX.  The
                                         modulation is the notebook's (cell 4): x = A cos(2 pi (carrier - f_b(t)) t) */
#define UC_FLAG_NO_FRAME_PAIRS (1u << 5) /* UC_DECHIRP_DOWN / UC_COMPRESS: one frame per transform instead of two
                                         * (strict independence of the frames, at half the throughput: see
                                         * uc_process_batch) */

/* table ids for uc_get_table */
enum {
  UC_TABLE_UP        = 0, /* n floats (RX_REAL, DECHIRP_DOWN uses DOWN only) or 2n (re,im) */
  UC_TABLE_DOWN      = 1,
  UC_TABLE_HANN      = 2, /* n floats */
  UC_TABLE_H_UP      = 3, /* COMPRESS: packed RFFT of hann*up chirp, n floats   */
  UC_TABLE_H_DOWN    = 4, /* COMPRESS: packed RFFT of hann*down chirp, n floats */
  UC_TABLE_CARRIER_C = 5, /* IQ: n floats */
  UC_TABLE_CARRIER_S = 6, /* IQ: n floats */
  UC_TABLE_FIR       = 7, /* IQ, STREAM: 27 taps */
  UC_TABLE_TEMPLATE  = 8  /* STREAM: base-band template g, n/decim complex (re,im) pairs */
};

/* mirrors the compile-time #defines and the derived values of the firmware */
typedef struct uc_config {
  uint32_t n;             /* NN 2048            -- receiver/Inc/main.h:97         */
  float    fs;            /* 78125.0f           -- receiver/Src/main.c:367-369    */
  float    f0, f1;        /* 16000, 19000       -- receiver/Inc/chirp.h:18-19     */
  float    time_frame;    /* TIME_FRAME 0.0205f -- receiver/Inc/chirp.h:16 (Q4);
                             <= 0 selects n/fs, which COMPRESS / DECHIRP_DOWN use */
  float    phase_deg;     /* -90                -- receiver/Src/chirp.c:43-44     */
  float    snr_threshold; /* 2.0f               -- receiver/Inc/main.h:98         */
  float    mag_mean;      /* noise floor used when the per-frame pointer is NULL  */
  float    carrier;       /* CARRIER 18000 (IQ) -- iq_modulation/Inc/iq_modem.h:10 */
  int32_t  variant;       /* UC_RX_REAL ...                                       */
  int32_t  device;        /* HIP device ordinal, >= 0                             */
  uint32_t flags;         /* UC_FLAG_*                                            */
  uint32_t decim;         /* UC_STREAM: decimation factor D after the FIR (4, 8 or 16);
                             0 = default (8).  Ignored by the other variants.      */
} uc_config;

/* = struct history (receiver/Src/main.c:124-136) minus ticks and rank */
typedef struct uc_stats {
  float   mag_max;        /* max magnitude in the two windows          */
  float   mag_max_left;   /* window [n - bandwidth2, n)                */
  float   mag_max_right;  /* window [0, bandwidth2)                    */
  int32_t max_freq;       /* idx2freq of the winning peak              */
  int32_t max_freq_left;
  int32_t max_freq_right;
  float   mag_mean;       /* noise floor the snr was computed against  */
  float   snr;            /* (mag_max - mag_mean) / mag_mean           */
} uc_stats;

typedef struct uc_ctx uc_ctx;

/* Fill *cfg with the shipping receiver's constants for `variant`. */
int uc_default_config(int32_t variant, uc_config* cfg);

int uc_create(const uc_config* cfg, uc_ctx** out);
void uc_destroy(uc_ctx* ctx);

/*
 * One frame: n raw DFSDM words in, one symbol out.
 * st (nullable) receives st[0] = up-chirp history, st[1] = down-chirp history
 * (history[0] / history[1] of main(), receiver/Src/main.c:518-519).
 * For the single-reference variants (COMPRESS, DECHIRP_DOWN, IQ) only st[0]
 * is written and *symbol_out is UC_SYM_NONE.
 */
int uc_process_frame(uc_ctx* ctx, const int32_t* pcm_in, float mag_mean,
                     uint8_t* symbol_out, uc_stats st[2]);

/*
 * n_frames frames, frame i starting at element i*stride_elems of `frames`
 * (stride_elems < n expresses the overlapping FIFO reads of the firmware,
 * receiver/Src/main.c:447-451; stride_elems == 0 means n).
 * mag_mean : NULL (cfg.mag_mean for all) or 2 floats per frame {up, down}.
 * symbols  : n_frames bytes, nullable.
 * stats    : 2 uc_stats per frame {up, down} for RX_REAL / SYNC_CPLX / IQ with UC_FLAG_IQ_BASEBAND,
 *            1 per frame otherwise; nullable.
 * UC_IQ reads 26 samples of FIR history in front of every frame: frames must
 * point 26 elements into the buffer (see uc_iq_halo()).
 * UC_DECHIRP_DOWN and UC_COMPRESS (one real reference) transform frames 2u and 2u + 1 in ONE complex FFT: a
 * frame's float32 round-off then scales with the larger frame of its pair (about 1e-7 of it), and a NaN / Inf
 * sample makes the records of BOTH frames NaN (UC_FLAG_NO_FRAME_PAIRS gives every frame its own transform).
 * Every other variant treats frames independently.
 * With device pointers the call enqueues its work on hip_stream and returns; it may be captured into a
 * hipGraph (all buffers device-resident) and replayed over new contents of the same buffers.
 * Work distribution: a large launch hands its frame groups to the workgroups through one atomic counter.  Eager
 * launches take that counter from a ring of 64 per context; a slot is reused only once the launch that last used it
 * has finished (otherwise the launch falls back to the static round-robin deal: same results, a few percent slower),
 * so any number of launches of one context may be in flight on any number of streams.  (A context that launches on
 * one stream only pays nothing for this; from the first launch on a second stream on, every launch records an event.)
 * A launch recorded during stream capture gets a counter that its graph owns for the life of the context (960 per
 * context, then static deal).
 * Every launch leaves its counter at zero (its last workgroup resets it), so nothing but the kernel node is recorded.
 * A graph's counter belongs to the context: destroy (or stop replaying) the graphs captured from a context before
 * uc_destroy() frees it.  The counter guard never waits for the device and never fails a launch: if an event call is
 * refused (a stream of this thread is under a global-mode capture) the launch is dealt statically instead.
 */
int uc_process_batch(uc_ctx* ctx, const void* frames, int dtype,
                     size_t n_frames, size_t stride_elems,
                     const float* mag_mean, uint8_t* symbols, uc_stats* stats,
                     void* hip_stream);

/* Diagnostic: the number of hand-out counter words of this context that are not zero although nothing is in flight (waits
 * for the device first).  Always 0 -- a launch leaves its counter at zero; anything else means a kernel returned without
 * passing the hand-out's exit, and the next launch on that counter would silently skip frame groups. */
int uc_debug_busy_counters(uc_ctx* ctx);

/* number of uc_stats records uc_process_batch writes per frame (1 or 2) */
int uc_stats_per_frame(const uc_ctx* ctx);

/* samples of history UC_IQ needs in front of frame 0 (26), 0 for the others */
int uc_iq_halo(const uc_ctx* ctx);

/* copy a host copy of a reference table; returns the element count or <0 */
int uc_get_table(const uc_ctx* ctx, int table_id, float* out, size_t cap);

/*
 * Replace a reference table: the firmware's tables are plain globals (up_chirp / down_chirp, receiver/Src/chirp.c:13-14;
 * hann_window, receiver/Src/main.c:99) that a host program may fill with a reference of its own -- a measured chirp,
 * all ones (then RX_REAL is Hann -> RFFT -> magnitude, the `basic` experiment, experiments/basic/Src/main.c:107-131), a
 * complex exponential (SYNC_CPLX: the windows then look at any part of the spectrum).
 * table_id: UC_TABLE_UP, UC_TABLE_DOWN or UC_TABLE_HANN; count must equal what uc_get_table returns for it.
 * RX_REAL, SYNC_CPLX and DECHIRP_DOWN only (-ENOTSUP otherwise).  Synchronous: waits for the device, then rebuilds the
 * fused reference * Hann tables the kernels read.  Window geometry (bandwidth, idx_left_zero) is unchanged.
 * All or nothing: on failure both the host copy (uc_get_table) and the device tables keep the old reference.  The wait
 * covers launches already enqueued; a graph captured from this context must not be REPLAYED while the call runs.
 */
int uc_set_table(uc_ctx* ctx, int table_id, const float* data, size_t count);

/*
 * pipeline() for a batch: the magnitudes it leaves in signal[] (receiver/Src/main.c:176-179), restricted to the bins
 * dsp() looks at.  mags: n_frames x uc_stats_per_frame() x uc_window_bins() floats, host or device;
 * record [frame][history][bandwidth2 + k] = |X[k]| for k = -bandwidth2 .. +bandwidth2 (k < 0: bin n + k), history 0 = up
 * reference, 1 = down (one history for DECHIRP_DOWN).  Bin 0 follows UC_FLAG_TRUE_DC (Q2).  With real references both
 * sides of DC hold the same value (Q1: Hermitian mirror).  RX_REAL, SYNC_CPLX, DECHIRP_DOWN; same frame addressing,
 * dtype and stream semantics as uc_process_batch.  A diagnostic / capture-comparison path, not the throughput path: it runs
 * the build uc_process_batch runs for this geometry (two pruned rounds up to bandwidth2 = 191, three beyond) with the stores
 * of the window bins added, so its values are bit for bit the ones the statistics are the maxima of.
 */
int uc_window_bins(const uc_ctx* ctx);   /* 2 * bandwidth2 + 1, or <0 */
int uc_window_spectrum(uc_ctx* ctx, const void* frames, int dtype, size_t n_frames, size_t stride_elems,
                       float* mags, void* hip_stream);

/* derived integers of main(): bandwidth, bandwidth2, idx_left_zero */
int uc_get_windows(const uc_ctx* ctx, uint32_t* bandwidth, uint32_t* bandwidth2,
                   uint32_t* idx_left_zero);

/* idx2freq(), integer arithmetic -- receiver/Src/main.c:154-160 */
int32_t uc_idx2freq(const uc_ctx* ctx, uint32_t idx);

/*
 * The receiver's main loop over a recorded sample stream: ISR FIFO (receiver/Src/main.c:659-668),
 * IDLE -> SYNCHRONIZING -> SYNCHRONIZED -> DATA_RECEIVING (main.c:417-554), resync (main.c:243-273,
 * Q8 fixed: bounds are checked before evaluating), bit -> byte assembly MSB first (main.c:523-537).
 * All dsp() calls of all blocks are evaluated in ONE batched launch (every FIFO offset the state
 * machine can visit is a multiple of 256 samples: stride_elems = 256), then the switch() is
 * replayed on the host over the statistics.  RX_REAL and SYNC_CPLX only.
 */
enum { UC_STATE_IDLE = 0, UC_STATE_SYNCHRONIZING = 1, UC_STATE_SYNCHRONIZED = 2, UC_STATE_DATA_RECEIVING = 3 };

typedef struct uc_rx_event {      /* one per processed block */
  uint32_t block;                 /* index of the 2048-sample block just appended to the FIFO */
  uint32_t sync_position;         /* after the block was processed */
  uint8_t  state_before, state_after;
  int8_t   bit;                   /* 0 / 1 if a data bit was decoded in this block, else -1 */
  uint8_t  reserved;
  float    snr_up, snr_down;      /* of SYNCHRONIZED / DATA_RECEIVING blocks, else 0 */
} uc_rx_event;

/* samples: n_samples int32 / float words, host or device.  text receives the decoded characters
 * ('\n' ends a message) and is always NUL-terminated; returns the number of characters or <0. */
int uc_receive_stream(uc_ctx* ctx, const void* samples, int dtype, size_t n_samples,
                      char* text, size_t text_cap,
                      uc_rx_event* trace /*nullable*/, size_t trace_cap, size_t* n_trace /*nullable*/);

/* The same with the ISR's drop-on-busy (receiver/Src/main.c:661: `if (!new_pcm_data && ...)`): busy[b] != 0 says the
 * main loop had not yet consumed the previous block when block b arrived, so the ISR drops block b (the FIFO is not
 * shifted, the block is lost, no pass of the switch runs for it).  busy: n_samples / n bytes, host; NULL = never busy
 * (= uc_receive_stream).  Trace records carry the index of the ACCEPTED block they belong to. */
int uc_receive_stream_isr(uc_ctx* ctx, const void* samples, int dtype, size_t n_samples,
                          const uint8_t* busy /*nullable*/, char* text, size_t text_cap,
                          uc_rx_event* trace /*nullable*/, size_t trace_cap, size_t* n_trace /*nullable*/);

/*
 * The same receiver for MANY recorded streams at once ("one independent stream per GPU (replicas
 * across streams)" -- here thousands per GPU): stream s = n_samples words at samples + s * stream_stride_elems
 * (stream_stride_elems == 0: n_samples; n_samples / n blocks each).  busy: n_streams x (n_samples / n) bytes, nullable --
 * busy[s][b] != 0 drops block b of stream s as the ISR would (receiver/Src/main.c:661).
 * The ISR shifts the FIFO by ONE block per accepted block (main.c:662), so of the 17 offsets dsp() can visit in the new FIFO
 * (pos = 0 .. 2 n in steps of 256) 9 were evaluated when the block before it arrived: ONE launch of the band kernel evaluates
 * the 8 offsets every accepted block ADDS -- each frame is the tail of one block and the head of the next, read through two
 * base addresses from the caller's buffer as it lies (no packed copy, no frame that straddles two streams; with a busy mask
 * the accepted blocks are first laid out one behind the other) -- and main()'s switch (main.c:417-554) + resync()
 * (main.c:243-273) are replayed ON THE DEVICE, one wave or one lane per stream: include/uchirp_mainloop.hpp compiled for the
 * device, the code uc_receive_stream replays on the host, so a stream's text and trace are the ones uc_receive_stream[_isr]
 * gives for it alone, bit for bit.
 *   text    n_streams x text_cap bytes: the decoded characters of stream s at text + s * text_cap, NUL-terminated
 *   n_text  (nullable) characters per stream
 *   trace   (nullable) n_streams x trace_cap records, one per processed block of the stream; n_trace (nullable) how many
 * Every pointer may be host or device memory; with device pointers only, the call is asynchronous on hip_stream.
 * dtype: UC_DTYPE_I32 / UC_DTYPE_F32 samples, or UC_DTYPE_PDM (the microphones' bit streams, one 32-bit word per sample: the
 * DFSDM runs on the device first; device buffers 16-byte aligned, the stride a multiple of 4 words).
 * From 1024 streams (UC_SYNC_CPLX) / 8192 streams (UC_RX_REAL) on, a call of several blocks without a busy mask is served block
 * by block -- as many one-block steps of the live form, back to back on hip_stream: each evaluates only the offsets the switch
 * can still look at once the block before has gone through it (an idle stream 3 or 5 of 8, UP only), where one launch over all
 * blocks must evaluate everything; same texts and traces, bit for bit (4096 recorded streams of 176 blocks: 21.7 -> 14.6 ms).
 * 8 bytes per 256 samples of statistics are parked in the context between the kernels of a call.  That scratch serves one
 * call at a time: calls of one context on DIFFERENT streams are ordered by the library (the later one waits, on the device,
 * for the earlier one's kernels) -- use one context per stream, or live states, for calls that should overlap.
 * For the same reason the state-less call cannot be captured into a hipGraph (-ENOTSUP while hip_stream is capturing): the
 * live form below can.
 */
int uc_receive_streams(uc_ctx* ctx, const void* samples, int dtype, size_t n_streams, size_t n_samples,
                       size_t stream_stride_elems, const uint8_t* busy /*nullable*/, char* text, size_t text_cap,
                       uint32_t* n_text /*nullable*/, uc_rx_event* trace /*nullable*/, size_t trace_cap,
                       uint32_t* n_trace /*nullable*/, void* hip_stream);

/*
 * Live streams: the firmware does not process recordings -- its ISR appends a block every 26.2 ms and main() makes one pass
 * of its switch per block, for ever (receiver/Src/main.c:417-578, 659-668).  A uc_rx_state holds, on the device, what n_streams
 * such receivers carry from one block to the next: the FIFO's newest accepted block, the (up, down) statistics of the 9 FIFO
 * offsets that survive the ISR's shift, main()'s locals (mag_stat[], history[], state, sync_position, the byte being
 * assembled ...), the number of blocks seen, the DFSDM's filter history, and the scratch of the call in flight.
 * uc_receive_streams_next() is uc_receive_streams() for the NEXT n_samples (whole blocks) of every stream: the chunks of a
 * stream, of any sizes, give exactly the text and trace of the whole stream in one call (trace records carry stream-global
 * block indices; `text` receives the characters decoded during THIS call).  busy as in uc_receive_streams (flags of this
 * chunk's blocks).  One new block of every stream costs at most 8 transforms per stream and reference; one-block calls evaluate
 * only the offsets main()'s switch can still look at -- an IDLE stream the 3 - 5 of the 8 that acquisition reaches, the UP
 * reference only (receiver/Src/main.c:447-453), a SYNCHRONIZED / DATA_RECEIVING stream the 5 - 6 around its sync_position
 * (main.c:491-550, 243-273); what the switch cannot look at costs nothing (no loads, no transform, no loop iteration: the kernel
 * walks the wanted offsets of a group of streams by bit scans of ONE word) --, 2 kernel launches (5 with a busy mask, +2 for
 * UC_DTYPE_PDM), no copy kernel; see uc_rx_state_keep_previous for the one copy that is left.
 * Everything a step carries lives on the device, so with device pointers the call can be captured into a hipGraph and the
 * graph replayed for every block that arrives (make one eager call of the same shape first: it sizes the scratch; nothing is
 * allocated during a capture -- a call that would have to answers -ENOBUFS and records nothing; uc_rx_state_reset puts the
 * receivers back to power-on).
 * Errors: every argument check and every scratch allocation of a call comes before its first launch -- a call refused with
 * -EINVAL / -ENOTSUP / -EOVERFLOW / -ENOBUFS / -ENOMEM has enqueued nothing and the state is as it was (the next call
 * continues the streams).  Any OTHER negative return (-EIO: a HIP launch or copy failed in mid-call) leaves the state
 * UNDEFINED -- some of its kernels may have run, the half-flag and the block counts may not match -- until uc_rx_state_reset
 * puts every receiver back to power-on; do not continue the streams on it.
 * A state belongs to the context that made it and must be destroyed BEFORE that context (uc_destroy / uc_group_destroy);
 * calls on different states of one context may be in flight on different streams at once, calls on one state are the
 * caller's to order (one stream).
 */
typedef struct uc_rx_state uc_rx_state;
int uc_rx_state_create(uc_ctx* ctx, size_t n_streams, uc_rx_state** out);   /* every receiver at power-on */
int uc_rx_state_reset(uc_rx_state* st, void* hip_stream);                   /* back to power-on */
void uc_rx_state_destroy(uc_rx_state* st);
size_t uc_rx_state_streams(const uc_rx_state* st);                          /* how many receivers it holds (0 for NULL) */
/*
 * "The caller keeps the previous chunk" (on != 0; off by default).  What a call's first new FIFO offsets still read of the call
 * before it is that call's last block of every stream (the ISR keeps it in fifo_queue, receiver/Src/main.c:662).  By default
 * the library copies that block into the state on its way through the kernel (8 KiB read once more + 8 KiB written per stream
 * and call) because the caller may overwrite `samples` as soon as the call has been enqueued.  A caller that receives into a
 * ring of two or more chunk buffers -- what a DMA engine fills anyway -- can promise instead that the `samples` of every call
 * stay where they are, UNCHANGED, until the NEXT uc_receive_streams_next on the state that brings at least one block has
 * completed on the device (a call of zero blocks reads nothing and releases nothing): the next
 * call then reads "the block in front" from where the previous call's samples lie, and nothing is copied.  Results are the
 * same bit for bit.  The promise covers calls on device memory without a busy mask; a call on host memory (the library stages
 * it), from UC_DTYPE_PDM bits (the DFSDM words are the library's) or with a busy mask is served as ever -- such calls may
 * be mixed in freely (a busy-masked call first copies the kept blocks into the state: one more kernel).  Switching the
 * contract off takes effect with the chunk of the next call (which still reads the kept one).  Captured steps bake the
 * two addresses in: capture one step per buffer of the ring (A after B, B after A) and replay them in that order.
 */
int uc_rx_state_keep_previous(uc_rx_state* st, int on);
int uc_receive_streams_next(uc_ctx* ctx, uc_rx_state* st, const void* samples, int dtype, size_t n_samples,
                            size_t stream_stride_elems, const uint8_t* busy /*nullable*/, char* text, size_t text_cap,
                            uint32_t* n_text /*nullable*/, uc_rx_event* trace /*nullable*/, size_t trace_cap,
                            uint32_t* n_trace /*nullable*/, void* hip_stream);

/*
 * UC_STREAM -- BASELINE config 4: streaming FIR-LPF decimate front-end + overlap-save
 * frequency-domain chirp compression over ONE continuous real sample stream x[r].
 *
 *   z[p] = sum_k fir[k] * x[r-k] * exp(-j 2 pi carrier (r-k) / fs),  r = halo + p*D   (k < 27)
 *          -- iq_demodulation(): mix, then the 27-tap low-pass, iq_modem.c:55-75, taps :18; kept
 *             only at every D-th sample (the decimation is this build's: the reference has none)
 *   y[q] = sum_{i<L} g[i] * z[q-i],   L = n/D
 *          -- compress_chirp(): FFT, x H, IFFT (chirp_compression_time_domain/Src/chirp.c:78-83),
 *             evaluated as overlap-save: FFT size n = 2048, hop = n - (L-1) outputs per block
 *   g[i] = hann_sym_L[i] * exp(j (2 pi ((f1-carrier) t - k t^2/2) - pi/2)),  t = i*D/fs,
 *          k = (f1-f0)/(n/fs): the base-band DOWN chirp of one n-sample symbol, symmetric Hann and
 *          -pi/2 phase as init_ref_chirp() (same file :52-75), WITH the 1/2 in the sweep that
 *          file's `freq = f + k t` lacks (it sweeps twice the band); UC_FLAG_STREAM_UP mirrors it.
 *   compressed[q] = |y[q]|,  q = 0 .. n_out-1,   n_out = (n_samples - halo) / D
 *
 * The first `halo` = (L-1)*D + 26 samples of the buffer are history (zeros at the start of a
 * stream, the tail of the previous chunk otherwise), so consecutive calls continue one another
 * exactly: output q belongs to input sample halo + q*D.
 * peaks (nullable): one record per overlap-save block b = q / hop: the largest compressed value
 * of the block (first one on ties) and its offset q - b*hop.
 * `samples` must be 16-byte aligned when it is a device pointer.  Host or device pointers;
 * asynchronous on hip_stream with device pointers; fixed shapes, so the call can be captured
 * into a hipGraph and replayed chunk after chunk.
 */
typedef struct uc_peak {
  float    value;
  uint32_t offset;
} uc_peak;

/* sizes for a buffer of n_samples (all nullable): history length, number of compressed outputs,
 * number of overlap-save blocks (= peak records), outputs per block */
int uc_stream_geometry(const uc_ctx* ctx, size_t n_samples, size_t* halo, size_t* n_out,
                       size_t* n_blocks, size_t* hop);

int uc_process_stream(uc_ctx* ctx, const void* samples, int dtype, size_t n_samples,
                      float* compressed /*n_out, nullable*/, uc_peak* peaks /*n_blocks, nullable*/,
                      void* hip_stream);

/*
 * The DFSDM peripheral in front of the ISR, as configured in receiver/Src/dfsdm.c:59-61 (SINC5,
 * Oversampling 32, IntOversampling 1), :69 (bit clock = 80 MHz / 32 = 2.5 MHz -> 78125 words/s) and
 * :78 (RightBitShift 2): sinc^5 filter, decimation by 32, of the microphone's 1-bit PDM stream.
 * pdm_words: n_words x 32 bits, bit t of the stream = bit (t & 31) of word t >> 5, 1 -> +1, 0 -> -1.
 * words_out: n_words - 4 int32 words, 24-bit result in bits 31:8 (what `buf[]` of receiver/Src/main.c:91
 * holds: feed them to uc_process_batch / uc_receive_stream / uc_process_stream as UC_DTYPE_I32).
 * The first 4 words only fill the filter (history of the previous chunk, or anything at stream start):
 * words_out[q] is the conversion that ends with pdm_words[q + 4].  Integer arithmetic, exact.
 * Any context will do (the variant is irrelevant).  Host or device pointers (device: 16-byte aligned);
 * asynchronous on hip_stream with device pointers.
 */
int uc_dfsdm_sinc5(uc_ctx* ctx, const uint32_t* pdm_words, size_t n_words, int32_t* words_out,
                   void* hip_stream);

/*
 * The same peripheral for MANY microphones at once, block after block: stream s = n_words NEW words at
 * pdm_words + s * stride_words (no history in the buffer), its n_words DFSDM words go to words_out + s * out_stride_words
 * (strides 0: n_words).  history: n_streams x 4 words, in and out -- the last four PDM words every stream had before this
 * call (UC_PDM_SILENCE x 4 for a stream that starts); the call leaves the last four words of [history | new words] there, so
 * that the chunks of a stream, of any sizes (one word included), give exactly uc_dfsdm_sinc5 of the whole stream behind its
 * first history: words_out[s][q] is the conversion that ends with new word q.  Integer arithmetic, exact.
 * Host or device pointers, each on its own (device: 16-byte aligned, strides multiples of 4 words); asynchronous on hip_stream
 * when all three are device memory.  The live receivers take PDM words directly: UC_DTYPE_PDM.
 */
int uc_dfsdm_sinc5_streams(uc_ctx* ctx, const uint32_t* pdm_words, size_t n_streams, size_t n_words, size_t stride_words,
                           uint32_t* history, int32_t* words_out, size_t out_stride_words, void* hip_stream);

/*
 * ---- Frame sharding across the GPUs of a node (BASELINE.json configs[4]) -----------------------
 *
 * The reference's host is a C program (receiver/Src/main.c:311-587) that owns ONE sample stream; on a node of MI355X the
 * frames of a batch are independent, so the frame index space is block-partitioned over the GPUs (tables replicated, no
 * data-path collective) and the only exchange is the all-gather of the decoded symbol stream, 1 byte per frame, over
 * RCCL / xGMI.  A uc_group is that arrangement behind this C-ABI: one uc_ctx per device, one RCCL communicator, one
 * gather stream per device.  Two ways to build one:
 * (Frames: uc_group_process_batch.  Whole microphone streams: uc_group_receive_streams[_next].  The blocks of ONE long
 * UC_STREAM stream: uc_group_process_stream.)
 *   uc_group_create        ONE process drives n_devices GPUs (ncclCommInitAll); rank r = devices[r]
 *   uc_group_create_rank   one process per GPU (how bench.py is launched): rank `rank` of `world`, device cfg->device;
 *                          rank 0 calls uc_group_unique_id and hands the 128 bytes to the others by any means it has
 *                          (a file, MPI, torch.distributed's store ...)
 * librccl.so.1 is loaded when the first group is created (dlopen: a process that never builds a group never loads it,
 * and one that already holds an RCCL -- PyTorch's -- shares it); -ENOSYS if it cannot be found.
 */
typedef struct uc_group uc_group;
#define UC_GROUP_ID_BYTES 128

/* The contiguous block partition of n_units units of work (frames; overlap-save blocks) over `world` ranks: rank r owns
 * [*first, *first + *count), sizes differ by at most one, the first n_units % world ranks get the extra unit.
 * Pure arithmetic (no GPU needed). */
int uc_partition(size_t n_units, int world, int rank, size_t* first, size_t* count);

/* What a rank must HOLD of the sample buffer to process frames [first_frame, first_frame + count) of a batch whose frame i
 * starts at element halo + i * stride_elems (halo = uc_iq_halo(): the FIR history in front of frame 0; stride_elems < n =
 * the overlapping FIFO reads): elements [*first_elem, *first_elem + *n_elems) of the buffer.  Neighbouring shards overlap
 * by n - stride_elems + halo elements -- read-only duplication of the input, never an exchange.  The rank's `frames`
 * argument is then its copy's element `halo`.  stride_elems == 0 means n.  count == 0 gives an empty span. */
int uc_frame_span(uint32_t n, size_t stride_elems, size_t halo, size_t first_frame, size_t count,
                  size_t* first_elem, size_t* n_elems);

/* UC_STREAM: the overlap-save BLOCKS of a stream are independent, so a stream of n_samples (its first `halo` samples are
 * history, uc_stream_geometry) shards like frames do: rank processes samples [*first_sample, *first_sample + *n_shard) --
 * the first `halo` of them are history it shares, read-only, with the previous rank -- and produces outputs
 * [*first_out, *first_out + *n_out) of the whole stream.  Block boundaries coincide with the one-GPU run. */
int uc_stream_span(const uc_ctx* ctx, size_t n_samples, int world, int rank,
                   size_t* first_sample, size_t* n_shard, size_t* first_out, size_t* n_out);

int uc_group_unique_id(void* id, size_t cap);  /* writes UC_GROUP_ID_BYTES bytes (ncclGetUniqueId) */
/* cfg->device is ignored by uc_group_create (devices[] names them) and names THE device for uc_group_create_rank */
int uc_group_create(const uc_config* cfg, const int32_t* devices, int n_devices, uc_group** out);
int uc_group_create_rank(const uc_config* cfg, const void* id, int world, int rank, uc_group** out);
void uc_group_destroy(uc_group* g);
/* Everything uc_group_create_rank does on this rank EXCEPT the communicator's rendezvous (RCCL loads, the device answers, a
 * context and the group's streams can be made); 0 or < 0.  A launcher calls it on every rank and agrees on the results with
 * a collective of its own BEFORE any rank calls uc_group_create_rank -- a rank that fails before ncclCommInitRank would leave
 * the others waiting in it (bench.py does exactly this). */
int uc_group_preflight(const uc_config* cfg);
int uc_group_world(const uc_group* g);        /* ranks in the communicator */
int uc_group_local_count(const uc_group* g);  /* devices this process drives (n_devices, or 1) */
int uc_group_first_rank(const uc_group* g);   /* rank of local device 0 (local device l is rank first + l) */
uc_ctx* uc_group_ctx(uc_group* g, int local); /* the context of local device l (tables, windows, uc_process_batch ...) */

/*
 * One step of the frame-sharded pipeline, for a batch of n_frames_total frames in all:
 *   frames[l]    sample 0 of the FIRST frame rank (first + l) owns (uc_partition of n_frames_total; uc_frame_span says
 *                what the rank must hold around it); device memory of that device, or host memory
 *   gathered[l]  n_frames_total bytes, device memory of that device (or host): receives the WHOLE symbol stream.
 * Every local device decodes its shard straight into its slice of gathered[l] on hip_streams[l] (NULL array or NULL
 * entry: the group's own stream for that device), then the slices are all-gathered IN PLACE on the group's gather
 * stream of that device (ncclAllGather when world divides n_frames_total, one grouped ncclBroadcast per rank otherwise),
 * behind an event: the call returns at once and the gather of step k overlaps the kernels of the steps after it.
 * Buffer hazards are the library's: a later step that writes a `gathered` buffer waits (on the device) for the gather
 * that last used it, so rotating two or three buffers is all a caller does (three keep a persistent kernel from
 * waiting for the previous gather's copy kernel: bench.py NBUF).  Read a gathered buffer after uc_group_synchronize(),
 * or make a stream of yours wait for its gather with uc_group_wait_gather().
 * Host pointers: the shard is staged through the context (synchronous copy-in), the stream is gathered in a device
 * buffer of the group and copied out; the call then blocks until gathered[l] is complete.
 * All ranks of the communicator must make the same sequence of calls (it is a collective).  Arguments are checked for EVERY
 * local device before anything is enqueued: a call refused for its arguments (a NULL shard or buffer, a bad dtype, a state of
 * the wrong size or of another context, and everything uc_receive_streams[_next] itself refuses a share for: partial blocks,
 * overlapping streams, the state's dtype lock, a misaligned UC_DTYPE_PDM buffer ... -- the group runs that call's own argument
 * check for every local device first) has touched no stream and started no collective -- the group is as it was and stays usable.
 * Any other negative return (a HIP or RCCL error in mid-step) may leave this rank's part of the step half enqueued (the other
 * ranks then wait in the collective): treat it as fatal for the group -- uc_group_destroy (which never waits for a peer) and
 * rebuild -- not as something to retry.
 * The gather stream of every device is created with the highest stream priority the device offers: the decode kernels are
 * persistent and own every CU, so RCCL's kernel starts when workgroups retire -- ahead of the next decode launch's.
 */
int uc_group_process_batch(uc_group* g, const void* const* frames, int dtype, size_t n_frames_total,
                           size_t stride_elems, uint8_t* const* gathered, void* const* hip_streams);
/* make hip_stream (of local device l) wait for the most recent gather into `gathered` (device pointer); no host wait */
int uc_group_wait_gather(uc_group* g, int local, const uint8_t* gathered, void* hip_stream);
/* wait for everything the group has enqueued on every local device */
int uc_group_synchronize(uc_group* g);

/*
 * The multi-stream receiver over the GPUs of a node -- SURVEY.md section 8e: the sequential state machine of ONE stream does
 * not shard, so "run one independent stream per GPU (replicas across streams)"; here thousands per GPU.  The n_streams_total
 * streams are block-partitioned over the ranks (uc_partition), every local device runs uc_receive_streams() over its share
 * (the ISR FIFO, dsp() at every 256-sample offset, main()'s switch replayed on the device: receiver/Src/main.c:417-554,
 * 243-273, 659-668) and writes into its slice of the gathered arrays, which are then all-gathered in place on the gather
 * stream, behind an event, exactly as uc_group_process_batch gathers the symbol stream (same hazard guard, same rules):
 *   samples[l]  stream 0 of the share rank (first + l) owns: count x n_samples words, stream_stride_elems apart, on that
 *               device (or host); busy (nullable array, nullable entries): count x (n_samples / n) flags of that share
 *   text[l]     n_streams_total x text_cap bytes on that device (or host): receives the text of EVERY stream of the node
 *   n_text      (nullable; all ranks alike) n_text[l]: n_streams_total uint32 on that device (or host), characters per stream
 * A stream's text is the one uc_receive_stream[_isr] gives for it alone, bit for bit, wherever it ran.
 * uc_group_receive_streams_next is the LIVE form (uc_receive_streams_next): states[l] = uc_rx_state_create(uc_group_ctx(g, l),
 * count of rank first + l) holds that share's receivers between calls (NULL for a rank that owns no stream: fewer streams than
 * GPUs); text receives what was decoded during THIS call.
 * (A block completes at most ONE character per stream, plus the newline that ends a message: for one-block calls text_cap = 4
 * is plenty and keeps what is gathered per step at 8 bytes per stream of the node.)
 */
int uc_group_receive_streams(uc_group* g, const void* const* samples, int dtype, size_t n_streams_total, size_t n_samples,
                             size_t stream_stride_elems, const uint8_t* const* busy /*nullable*/, char* const* text,
                             size_t text_cap, uint32_t* const* n_text /*nullable*/, void* const* hip_streams /*nullable*/);
int uc_group_receive_streams_next(uc_group* g, uc_rx_state* const* states, const void* const* samples, int dtype,
                                  size_t n_streams_total, size_t n_samples, size_t stream_stride_elems,
                                  const uint8_t* const* busy /*nullable*/, char* const* text, size_t text_cap,
                                  uint32_t* const* n_text /*nullable*/, void* const* hip_streams /*nullable*/);

/*
 * UC_STREAM over the GPUs of a node (a group made from a UC_STREAM config).  The overlap-save blocks of the stream are
 * independent: rank r processes the samples uc_stream_span() names (its first `halo` samples are history it shares, read-only,
 * with the previous rank -- no exchange) and the block boundaries are those of the one-GPU run, so every value is the one
 * uc_process_stream gives for the whole stream.
 *   samples[l]     the shard of rank (first + l): sample *first_sample of the stream, *n_shard samples (uc_stream_span)
 *   compressed     (nullable array, nullable entries) compressed[l]: the rank's OWN *n_out outputs -- outputs *first_out ... of
 *                  the stream; the envelope stays where it was computed (4 / D bytes per input sample)
 *   peaks[l]       n_blocks records (uc_stream_geometry of n_samples_total): the peak record of EVERY block of the stream,
 *                  all-gathered in place (8 bytes per block) as uc_group_process_batch gathers symbols; same rules
 */
int uc_group_process_stream(uc_group* g, const void* const* samples, int dtype, size_t n_samples_total,
                            float* const* compressed /*nullable*/, uc_peak* const* peaks, void* const* hip_streams /*nullable*/);

/* Plain-C hosts without the HIP headers: device memory by ordinal.  uc_device_copy: either side may be host or device
 * memory (hipMemcpyDefault), synchronous. */
int uc_device_count(void);
int uc_device_malloc(int device, size_t bytes, void** out);
int uc_device_free(int device, void* ptr);
int uc_device_copy(void* dst, const void* src, size_t bytes);

/*
 * The shader clock the chip holds UNDER a kernel of this library, measured in the process that quotes it (diagnostic; costs
 * nothing when off).  libuchirp.so carries every kernel twice: as it is, and with ONE s_memtime / s_memrealtime stamp pair
 * per wave around the kernel's persistent loop.  After uc_clock_probe(ctx, 1) the launches of this context run the stamped
 * twin (same grid, same work distribution, same results; a memset of the stamp buffer rides in front of each launch, so do
 * not time with it); uc_clock_read() waits for the device and reduces the stamps of the LAST launch; uc_clock_probe(ctx, 0)
 * goes back to the throughput build.  MI355X clocks down to its power cap under these kernels (1.9 - 2.4 GHz by kernel and
 * data), so a VALU-issue roofline needs THIS number, not the 2.4 GHz of the data sheet.
 */
typedef struct uc_clock {
  double   shader_ghz;   /* median over waves of (shader cycles of the wave's loop) / (the same span in the 100 MHz clock) */
  double   wave_cycles;  /* median shader cycles a wave spent in its loop */
  double   span_us;      /* first wave's start to last wave's end */
  uint32_t waves;        /* waves that stamped */
} uc_clock;
int uc_clock_probe(uc_ctx* ctx, int on);
int uc_clock_read(uc_ctx* ctx, uc_clock* out);
/* the raw stamps of the last launch, four words per wave: shader cycles of the wave's loop (low 40 bits; bits 40..51 name
 * the CU), the same span in 100 MHz ticks, absolute start tick, absolute end tick (0 0 0 0: the wave had nothing to do).
 * Returns the number of words (also when words is NULL or cap_words too small: a size query), or < 0. */
int uc_clock_stamps(uc_ctx* ctx, uint64_t* words, size_t cap_words);

/* human-readable text of the last error on this thread ("" if none) */
const char* uc_last_error(void);

/* UC_ABI_VERSION of the loaded library */
int uc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* UCHIRP_H_ */
