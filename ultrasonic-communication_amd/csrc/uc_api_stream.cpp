// uc_api_stream.cpp -- UC_STREAM (BASELINE configs[3]): uc_stream_geometry / uc_stream_span / uc_process_stream, the streaming
// FIR-decimate front end + overlap-save chirp compression over one continuous sample stream.
#include "uc_api_internal.hpp"

using namespace uc_api;

int uc_stream_geometry(const uc_ctx* c, size_t n_samples, size_t* halo, size_t* n_out, size_t* n_blocks,
                       size_t* hop) {
  if (!c) return fail(-EINVAL, "uc_stream_geometry: NULL ctx");
  if (c->cfg.variant != UC_STREAM) return fail(-EINVAL, "uc_stream_geometry: the context is not UC_STREAM");
  const size_t h = c->stab.halo, hp = c->stab.hop, D = c->stab.decim;
  const size_t no = n_samples > h ? (n_samples - h) / D : 0;
  if (halo) *halo = h;
  if (n_out) *n_out = no;
  if (n_blocks) *n_blocks = (no + hp - 1) / hp;
  if (hop) *hop = hp;
  return 0;
}

int uc_stream_span(const uc_ctx* c, size_t n_samples, int world, int rank, size_t* first_sample, size_t* n_shard,
                   size_t* first_out, size_t* n_out) {
  if (!c) return fail(-EINVAL, "uc_stream_span: NULL ctx");
  if (c->cfg.variant != UC_STREAM) return fail(-EINVAL, "uc_stream_span: the context is not UC_STREAM");
  const size_t h = c->stab.halo, hp = c->stab.hop, D = c->stab.decim;
  const size_t no = n_samples > h ? (n_samples - h) / D : 0;
  const size_t nb = (no + hp - 1) / hp;
  size_t b0 = 0, bc = 0;
  const int rc = uc_partition(nb, world, rank, &b0, &bc);  // whole overlap-save blocks: boundaries as in the one-GPU run
  if (rc) return rc;
  size_t q0 = b0 * hp, q1 = (b0 + bc) * hp;
  if (q0 > no) q0 = no;
  if (q1 > no) q1 = no;
  const bool empty = q1 <= q0;
  if (first_sample) *first_sample = empty ? 0 : q0 * D;
  if (n_shard) *n_shard = empty ? 0 : h + (q1 - q0) * D;
  if (first_out) *first_out = q0;
  if (n_out) *n_out = empty ? 0 : q1 - q0;
  return 0;
}

int uc_process_stream(uc_ctx* c, const void* samples, int dtype, size_t n_samples, float* compressed,
                      uc_peak* peaks, void* hip_stream) {
  if (!c) return fail(-EINVAL, "uc_process_stream: NULL ctx");
  if (c->cfg.variant != UC_STREAM) return fail(-EINVAL, "uc_process_stream: the context is not UC_STREAM");
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32)
    return fail(-EINVAL, "uc_process_stream: dtype %d is neither UC_DTYPE_I32 nor UC_DTYPE_F32", dtype);
  size_t n_out = 0, n_blocks = 0;
  uc_stream_geometry(c, n_samples, nullptr, &n_out, &n_blocks, nullptr);
  if (n_out == 0) return 0;
  if (!samples) return fail(-EINVAL, "uc_process_stream: samples is NULL");

  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;

  const void* d_samples = samples;
  if (!is_device_ptr(samples)) {
    int rc = c->s_frames.ensure(n_samples * 4);
    if (rc) return rc;
    e = hipMemcpyAsync(c->s_frames.p, samples, n_samples * 4, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(samples)");
    d_samples = c->s_frames.p;
  } else if (((uintptr_t)samples & 15u) != 0) {
    return fail(-EINVAL, "uc_process_stream: a device `samples` pointer must be 16-byte aligned");
  }
  bool any_host_out = false;
  float* d_comp = compressed;
  if (compressed && !is_device_ptr(compressed)) {
    int rc = c->s_comp.ensure(n_out * sizeof(float));
    if (rc) return rc;
    d_comp = (float*)c->s_comp.p;
    any_host_out = true;
  }
  uc_peak* d_peaks = peaks;
  if (peaks && !is_device_ptr(peaks)) {
    int rc = c->s_peaks.ensure(n_blocks * sizeof(uc_peak));
    if (rc) return rc;
    d_peaks = (uc_peak*)c->s_peaks.p;
    any_host_out = true;
  }

  uc::StreamParams sp;
  memset(&sp, 0, sizeof(sp));
  sp.samples = d_samples;
  sp.n_samples = n_samples;
  sp.n_out = n_out;
  sp.n_blocks = n_blocks;
  sp.hn = c->d_tab0;
  sp.rot = c->d_tab1;
  sp.tw = c->d_tw;
  sp.compressed = d_comp;
  sp.peaks = d_peaks;
  for (int k = 0; k < 2 * uc::kFirTapsDev; k++) sp.ctap[k] = c->stab.ctap[k];
  const int D = (int)c->stab.decim;
  for (int sub = 0; sub < D / 2; sub++) {
    sp.rots[2 * sub] = c->stab.rot[2 * (size_t)(sub * (4096 / D))];
    sp.rots[2 * sub + 1] = c->stab.rot[2 * (size_t)(sub * (4096 / D)) + 1];
  }
  int& st_bpc = c->stream_blocks_per_cu[dtype == UC_DTYPE_I32 ? 0 : 1];
  if (st_bpc == 0) st_bpc = uc::stream_max_blocks_per_cu(dtype, D);
  size_t grid = (size_t)c->num_cu * (size_t)st_bpc;
  if (c->grid_override > 0) grid = (size_t)c->grid_override;
  if (grid > n_blocks) grid = n_blocks;
  if (n_blocks >= ((size_t)1 << 32)) return fail(-EINVAL, "uc_process_stream: at most 2^32 - 1 blocks per call");
  sp.work_ctr = nullptr;
  sp.chunk_log2 = 0;
  int wslot = -1;
  // (tickets only from sixteen chunks per workgroup on: 2^26 samples +50 % dealt statically, 2^28 +9 %, 2^31 -11 %)
  if (!c->static_deal && n_blocks >= (size_t)16 * (size_t)c->stream_chunk * grid) {
    // dynamic hand-out of chunks of consecutive blocks
    const int wrc = take_work_counter(c, stream, &sp.work_ctr, &wslot);
    if (wrc) return wrc;
    if (sp.work_ctr) {
      while ((1u << sp.chunk_log2) < (unsigned)c->stream_chunk) sp.chunk_log2++;
      const size_t nchunks = (n_blocks + ((size_t)1 << sp.chunk_log2) - 1) >> sp.chunk_log2;
      if (grid > nchunks) grid = nchunks;
    }
  }
  if (int crc = clock_buffer(c, grid, 2, stream, &sp.debug)) return crc;
  int lrc = (c->clock_probe ? uc::clk::launch_stream : uc::launch_stream)(dtype, D, sp, (int)grid, stream);
  if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "stream kernel launch");
  if (int erc = work_counter_launched(c, stream, wslot)) return erc;

  if (any_host_out) {
    if (compressed && d_comp != compressed) {
      e = hipMemcpyAsync(compressed, d_comp, n_out * sizeof(float), hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(compressed)");
    }
    if (peaks && d_peaks != peaks) {
      e = hipMemcpyAsync(peaks, d_peaks, n_blocks * sizeof(uc_peak), hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(peaks)");
    }
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  }
  return 0;
}
