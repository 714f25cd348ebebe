/*
 * uc_oracle.h -- CPU restatement of the reference's per-frame DSP path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it, and only as the checker / the timed CPU baseline.
 * libuchirp.so never links or calls it.
 *
 * Parity status: the arithmetic below follows the reference's own C
 * (file:line cited at each function).  The numeric primitives the reference
 * delegates to CMSIS-DSP V1.4.5 (ARM Ltd, $Date 20. October 2015; only
 * headers and a Cortex-M4 binary archive are vendored, the source is absent
 * from the reference checkout) are restated from their published algorithms:
 *   - Hann + RFFT + magnitude are PINNED against the on-device captures
 *     agent/ (K6) -- tests/test_oracle_golden.py;
 *   - arm_cfft_f32, the inverse RFFT, arm_fir_f32, arm_max_f32 tie-breaking
 *     and arm_sin_cos_f32 have no on-device vector in the reference:
 *     for those, parity is UNPINNED by the reference and is pinned against
 *     exact float64 mathematics within the stated tolerance instead.
 * The C reference itself is unbuildable here (needs CMSIS-DSP + STM32 HAL).
 */
#ifndef UC_ORACLE_H_
#define UC_ORACLE_H_

#include "../include/uchirp.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uco_ctx uco_ctx;

/* precision of the FFT arithmetic */
enum {
  UCO_F32 = 32, /* float32 butterflies, CMSIS-like structure (RFFT = N/2 CFFT + split) */
  UCO_F64 = 64  /* float64 butterflies on the float32 inputs: the tolerance anchor      */
};

int  uco_default_config(int32_t variant, uc_config* cfg);
int  uco_create(const uc_config* cfg, uco_ctx** out);
void uco_destroy(uco_ctx* ctx);

/* same argument meaning as uc_process_batch (include/uchirp.h); host pointers only */
int uco_process_batch(uco_ctx* ctx, const void* frames, int dtype,
                      size_t n_frames, size_t stride_elems,
                      const float* mag_mean, uint8_t* symbols, uc_stats* stats,
                      int precision, int threads);

/*
 * Full magnitude spectrum(s) of one frame as float64 (tolerance work):
 * RX_REAL/SYNC_CPLX: out[0..n) = up spectrum, out[n..2n) = down spectrum,
 * in the index space of the reference's `time_frame` after pipeline()
 * (Q1 fix: bins >= n/2 of RX_REAL are the Hermitian mirror).
 * DECHIRP_DOWN / IQ: out[0..n).  COMPRESS: out[0..n) = compressed signal.
 */
int uco_spectrum(uco_ctx* ctx, const void* frame, int dtype, int precision,
                 double* out);

int uco_stats_per_frame(const uco_ctx* ctx);
int uco_get_table(const uco_ctx* ctx, int table_id, float* out, size_t cap);
/* same meaning as uc_set_table (include/uchirp.h): UC_TABLE_UP / _DOWN / _HANN of RX_REAL, SYNC_CPLX, DECHIRP_DOWN */
int uco_set_table(uco_ctx* ctx, int table_id, const float* data, size_t count);
int uco_get_windows(const uco_ctx* ctx, uint32_t* bandwidth, uint32_t* bandwidth2,
                    uint32_t* idx_left_zero);
int32_t uco_idx2freq(const uco_ctx* ctx, uint32_t idx);

/* Literal, sequential restatement of the receiver's main loop over a recorded stream
 * (receiver/Src/main.c:417-554, 243-273, 659-668): one dsp() per call site, in program order.
 * Same outputs as uc_receive_stream (include/uchirp.h). */
int uco_receive_stream(uco_ctx* ctx, const void* samples, int dtype, size_t n_samples, int precision,
                       char* text, size_t text_cap, uc_rx_event* trace, size_t trace_cap, size_t* n_trace);
/* ... with the ISR's drop-on-busy (main.c:661): same outputs as uc_receive_stream_isr */
int uco_receive_stream_isr(uco_ctx* ctx, const void* samples, int dtype, size_t n_samples, const uint8_t* busy, int precision,
                           char* text, size_t text_cap, uc_rx_event* trace, size_t trace_cap, size_t* n_trace);
/* the same, plus per processed block the relative gap of the block's closest decision (uc_oracle.c); margin: trace_cap floats, nullable */
int uco_receive_stream_diag(uco_ctx* ctx, const void* samples, int dtype, size_t n_samples, const uint8_t* busy, int precision,
                            char* text, size_t text_cap, uc_rx_event* trace, size_t trace_cap, size_t* n_trace, float* margin);

/* UC_STREAM (BASELINE config 4): same outputs as uc_stream_geometry / uc_process_stream,
 * evaluated as the direct float64 time-domain sums of the definition in include/uchirp.h
 * (build-defined pipeline: UNPINNED by the reference, pinned against numpy in the CPU tests). */
int uco_stream_geometry(const uco_ctx* ctx, size_t n_samples, size_t* halo, size_t* n_out,
                        size_t* n_blocks, size_t* hop);
int uco_process_stream(uco_ctx* ctx, const void* samples, int dtype, size_t n_samples,
                       float* compressed, uc_peak* peaks, int threads);

/* DFSDM sinc^5 / 32 model (receiver/Src/dfsdm.c:59-61,69,78): same outputs as uc_dfsdm_sinc5 */
int uco_dfsdm_sinc5(const uint32_t* pdm, size_t n_words, int32_t* out);
/* test helper: 2nd-order delta-sigma modulator, n_bits (multiple of 32) samples in [-1,1] -> words */
int uco_pdm_modulate(const float* x, size_t n_bits, uint32_t* words);

/* the CMSIS-DSP primitives, restated (exposed for the golden-vector tests) */
float uco_arm_cos_f32(float x);
void  uco_arm_sin_cos_f32(float theta_deg, float* sin_val, float* cos_val);
void  uco_arm_max_f32(const float* src, uint32_t n, float* out, uint32_t* idx);
/* packed RFFT: [Re X0, Re X(n/2), Re X1, Im X1, ...]  (float32 butterflies) */
void  uco_rfft_fast_f32(const float* in, float* out_packed, uint32_t n);
void  uco_hann_periodic(float* w, uint32_t n, int libm);

#ifdef __cplusplus
}
#endif
#endif
