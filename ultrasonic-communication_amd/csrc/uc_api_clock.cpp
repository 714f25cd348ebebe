// uc_api_clock.cpp -- uc_clock_probe / uc_clock_read / uc_clock_stamps: the shader clock a kernel holds, read from the stamps its
// clock-stamped twin leaves (uc_kernels.hpp: uc::clk; the launch sites put the stamp buffer in place: uc_api::clock_buffer).
#include "uc_api_internal.hpp"

using namespace uc_api;

int uc_clock_probe(uc_ctx* c, int on) {
  if (!c) return fail(-EINVAL, "uc_clock_probe: NULL ctx");
  c->clock_probe = on != 0;
  c->clock_waves = 0;
  return 0;
}

int uc_clock_read(uc_ctx* c, uc_clock* out) {
  if (!c || !out) return fail(-EINVAL, "uc_clock_read: NULL argument");
  memset(out, 0, sizeof(*out));
  if (!c->clock_probe || c->clock_waves == 0) return fail(-ENODATA, "uc_clock_read: no launch since uc_clock_probe(ctx, 1)");
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) return hip_fail(e, "hipDeviceSynchronize");
  std::vector<unsigned long long> w(c->clock_waves * 4);
  e = hipMemcpy(w.data(), c->s_clock.p, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy(clock stamps)");
  // per wave: [0] shader cycles of its loop (low 40 bits; the bits above name the CU), [1] the same span in ticks of the
  // constant 100 MHz clock, [2] / [3] absolute start / end ticks
  std::vector<double> ghz, cyc;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t i = 0; i < c->clock_waves; i++) {
    const unsigned long long cycles = w[4 * i] & 0xffffffffffull, ticks = w[4 * i + 1];
    if (ticks == 0) continue;  // a wave that had nothing to do
    ghz.push_back((double)cycles / (double)ticks * 0.1);
    cyc.push_back((double)cycles);
    if (w[4 * i + 2] < t0) t0 = w[4 * i + 2];
    if (w[4 * i + 3] > t1) t1 = w[4 * i + 3];
  }
  if (ghz.empty()) return fail(-ENODATA, "uc_clock_read: the last launch stamped no wave");
  std::sort(ghz.begin(), ghz.end());
  std::sort(cyc.begin(), cyc.end());
  out->shader_ghz = ghz[ghz.size() / 2];
  out->wave_cycles = cyc[cyc.size() / 2];
  out->span_us = (double)(t1 - t0) * 0.01;
  out->waves = (uint32_t)ghz.size();
  return 0;
}

int uc_clock_stamps(uc_ctx* c, uint64_t* words, size_t cap_words) {
  if (!c) return fail(-EINVAL, "uc_clock_stamps: NULL ctx");
  if (!c->clock_probe || c->clock_waves == 0) return fail(-ENODATA, "uc_clock_stamps: no launch since uc_clock_probe(ctx, 1)");
  const size_t nw = c->clock_waves * 4;
  if (!words || cap_words < nw) return (int)nw;  // (size query)
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(words, c->s_clock.p, nw * sizeof(uint64_t), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "uc_clock_stamps");
  return (int)nw;
}
