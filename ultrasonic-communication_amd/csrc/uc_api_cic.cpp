// uc_api_cic.cpp -- uc_dfsdm_sinc5 / uc_dfsdm_sinc5_streams: the DFSDM peripheral in front of the ISR (receiver/Src/dfsdm.c:59-61,
// 69, 78) on the device; the live receivers call sinc5_streams_launch for UC_DTYPE_PDM chunks (uc_api_rx.cpp).
#include "uc_api_internal.hpp"

using namespace uc_api;

// the sinc^5 byte tables on the device and the kernel's LDS opt-in, once per context
static int sinc5_prepare(uc_ctx* c) {
  if (!c->d_cic4) {
    std::vector<int32_t> t4, t1;
    uc::build_sinc5_tables(t4, t1);
    int rc = upload((void**)&c->d_cic4, t4.data(), t4.size() * sizeof(int32_t));
    if (!rc) rc = upload((void**)&c->d_cic1, t1.data(), t1.size() * sizeof(int32_t));
    if (rc) return rc;
  }
  if (c->cic_blocks_per_cu == 0) {
    c->cic_blocks_per_cu = uc::sinc5_max_blocks_per_cu();
    if (c->cic_blocks_per_cu <= 0) {
      c->cic_blocks_per_cu = 0;
      return fail(-ENOMEM, "uc_dfsdm_sinc5: the kernel's LDS tables do not fit this device");
    }
  }
  return 0;
}

// uc_dfsdm_sinc5_streams on device buffers: n_words NEW words of every stream, history carried in d_hist ([n_streams][4])
// (update_hist false: d_hist is only read -- uc_dfsdm_sinc5, where it is the head of the caller's input)
int uc_api::sinc5_streams_launch(uc_ctx* c, const uint32_t* d_pdm, size_t n_streams, size_t n_words, size_t stride,
                                const uint32_t* d_hist, bool update_hist, int32_t* d_out, size_t out_stride,
                                hipStream_t stream) {
  if (n_streams == 0 || n_words == 0) return 0;
  if ((((uintptr_t)d_pdm | (uintptr_t)d_out | (uintptr_t)d_hist) & 15u) != 0 ||
      (n_streams > 1 && ((stride & 3u) != 0 || (out_stride & 3u) != 0)))
    return fail(-EINVAL, "uc_dfsdm_sinc5_streams: device buffers must be 16-byte aligned and the strides multiples of 4 words");
  if (int rc = sinc5_prepare(c)) return rc;
  uc::CicParams cp;
  memset(&cp, 0, sizeof(cp));
  cp.pdm = d_pdm;
  cp.n_words = n_words;
  cp.out = d_out;
  cp.t4 = c->d_cic4;
  cp.t1 = c->d_cic1;
  cp.n_streams = n_streams;
  cp.stride = stride;
  cp.out_stride = out_stride;
  cp.hist = d_hist;
  cp.update_hist = update_hist ? 1u : 0u;
  size_t grid = (size_t)c->num_cu * (size_t)c->cic_blocks_per_cu;
  if (c->grid_override > 0) grid = (size_t)c->grid_override;
  const size_t per_block = (size_t)uc::sinc5_waves_per_block();
  // Tiles of 256 words; one wave walks a SEGMENT of up to 8 tiles front to back (uc_cic_kernel.hip).  A live block (2048
  // words) is one segment.  Few streams: shorter segments, so that every wave of the grid has one.
  const size_t tiles = (n_words + 255) / 256;
  size_t nseg = (tiles + 7) / 8;
  const size_t spread = (grid * per_block + n_streams - 1) / n_streams;  // segments per stream that fill the grid
  if (nseg < spread) nseg = spread < tiles ? spread : tiles;
  const size_t tps = (tiles + nseg - 1) / nseg;
  nseg = (tiles + tps - 1) / tps;
  if (nseg * n_streams >= ((size_t)1 << 31)) return fail(-EINVAL, "uc_dfsdm_sinc5_streams: too many segments in one call");
  cp.tps = (uint32_t)tps;
  cp.nseg = (uint32_t)nseg;
  cp.units = (uint32_t)(nseg * n_streams);
  uc::rows_divisor(cp.nseg, &cp.div_magic, &cp.div_shift);
  const size_t need = ((size_t)cp.units + per_block - 1) / per_block;
  if (grid > need) grid = need;
  if (c->clock_probe) {
    if (c->clk_cic_blocks == 0) c->clk_cic_blocks = uc::clk::sinc5_max_blocks_per_cu();  // (the twin needs the same LDS opt-in)
    if (c->clk_cic_blocks <= 0) return fail(-ENOMEM, "uc_dfsdm_sinc5: the clock-stamped kernel's LDS tables do not fit");
    if (int crc = clock_buffer(c, grid, uc::clk::sinc5_waves_per_block(), stream, &cp.debug)) return crc;
  }
  const int lrc = (c->clock_probe ? uc::clk::launch_sinc5 : uc::launch_sinc5)(cp, (int)grid, stream);
  if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "sinc5 kernel launch");
  return 0;
}

int uc_dfsdm_sinc5_streams(uc_ctx* c, const uint32_t* pdm_words, size_t n_streams, size_t n_words, size_t stride_words,
                           uint32_t* history, int32_t* words_out, size_t out_stride_words, void* hip_stream) {
  if (!c) return fail(-EINVAL, "uc_dfsdm_sinc5_streams: NULL ctx");
  if (n_streams == 0 || n_words == 0) return 0;
  if (!pdm_words || !words_out || !history) return fail(-EINVAL, "uc_dfsdm_sinc5_streams: NULL buffer");
  if (stride_words == 0) stride_words = n_words;
  if (out_stride_words == 0) out_stride_words = n_words;
  if (stride_words < n_words || out_stride_words < n_words)
    return fail(-EINVAL, "uc_dfsdm_sinc5_streams: streams overlap (stride < %zu words)", n_words);
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;
  // host buffers are staged through the context (padded to whole 16-byte rows), device buffers are used where they lie
  const bool in_host = !is_device_ptr(pdm_words), hist_host = !is_device_ptr(history), out_host = !is_device_ptr(words_out);
  const uint32_t* d_in = pdm_words;
  size_t in_stride = stride_words;
  if (in_host) {
    in_stride = (n_words + 3) & ~(size_t)3;
    if (int rc = c->s_cic_in.ensure(n_streams * in_stride * 4)) return rc;
    e = hipMemcpy2DAsync(c->s_cic_in.p, in_stride * 4, pdm_words, stride_words * 4, n_words * 4, n_streams, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy2DAsync(pdm)");
    d_in = (const uint32_t*)c->s_cic_in.p;
  }
  uint32_t* d_hist = history;
  if (hist_host) {
    if (int rc = c->s_cic_hist.ensure(n_streams * 16)) return rc;
    e = hipMemcpyAsync(c->s_cic_hist.p, history, n_streams * 16, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(history)");
    d_hist = (uint32_t*)c->s_cic_hist.p;
  }
  int32_t* d_out = words_out;
  size_t o_stride = out_stride_words;
  if (out_host) {
    o_stride = (n_words + 3) & ~(size_t)3;
    if (int rc = c->s_cic_out.ensure(n_streams * o_stride * 4)) return rc;
    d_out = (int32_t*)c->s_cic_out.p;
  }
  if (int rc = sinc5_streams_launch(c, d_in, n_streams, n_words, in_stride, d_hist, true, d_out, o_stride, stream)) return rc;
  if (out_host || hist_host) {
    if (out_host) {
      e = hipMemcpy2DAsync(words_out, out_stride_words * 4, d_out, o_stride * 4, n_words * 4, n_streams, hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpy2DAsync(words_out)");
    }
    if (hist_host) {
      e = hipMemcpyAsync(history, d_hist, n_streams * 16, hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(history)");
    }
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  }
  return 0;
}

int uc_dfsdm_sinc5(uc_ctx* c, const uint32_t* pdm_words, size_t n_words, int32_t* words_out, void* hip_stream) {
  if (!c) return fail(-EINVAL, "uc_dfsdm_sinc5: NULL ctx");
  if (n_words <= 4) return 0;
  if (!pdm_words || !words_out) return fail(-EINVAL, "uc_dfsdm_sinc5: NULL buffer");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;
  const size_t n_out = n_words - 4;
  const uint32_t* d_in = pdm_words;
  if (!is_device_ptr(pdm_words)) {
    int rc = c->s_cic_in.ensure(n_words * 4);
    if (rc) return rc;
    e = hipMemcpyAsync(c->s_cic_in.p, pdm_words, n_words * 4, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(pdm)");
    d_in = (const uint32_t*)c->s_cic_in.p;
  } else if (((uintptr_t)pdm_words & 15u) != 0) {
    return fail(-EINVAL, "uc_dfsdm_sinc5: a device `pdm_words` pointer must be 16-byte aligned");
  }
  int32_t* d_out = words_out;
  const bool host_out = !is_device_ptr(words_out);
  if (host_out) {
    int rc = c->s_cic_out.ensure(n_out * 4);
    if (rc) return rc;
    d_out = (int32_t*)c->s_cic_out.p;
  } else if (((uintptr_t)words_out & 15u) != 0) {
    return fail(-EINVAL, "uc_dfsdm_sinc5: a device `words_out` pointer must be 16-byte aligned");
  }
  // one stream whose history lies in front of it: words 0 .. 3 are the history, words 4 .. the stream
  if (int rc = sinc5_streams_launch(c, d_in + 4, 1, n_out, n_out, d_in, false, d_out, n_out, stream)) return rc;
  if (host_out) {
    e = hipMemcpyAsync(words_out, d_out, n_out * 4, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(words_out)");
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  }
  return 0;
}
