// uchirp_receiver.hpp -- the reference's per-frame functions, same names and argument
// meaning, on top of the C-ABI (include/uchirp.h).  Header-only C++ host layer.
//
//   reference symbol (receiver/...)                          here
//   -------------------------------------------------------  ---------------------------------
//   float fifo_queue[NN*3]            Src/main.c:94          Receiver::fifo_queue
//   struct history                    Src/main.c:124-136     uchirp::history
//   init (fs, bandwidth, Hann, ...)   Src/main.c:367-393     Receiver::Receiver(fs)
//   init_ref_chirp(fs)                Src/chirp.c:42-45      (inside uc_create)
//   HAL_DFSDM_FilterRegConvCpltCallback  Src/main.c:659-668  Receiver::HAL_DFSDM_FilterRegConvCpltCallback(buf)
//   idx2freq(idx)                     Src/main.c:154-160     Receiver::idx2freq
//   dsp(pos, phist, mag_mean, updown) Src/main.c:183-231     Receiver::dsp
//   symbol_snr(pos, phist, updown)    Src/main.c:233-236     Receiver::symbol_snr
//   resync(snr, hist, offset, &pos, updown)  Src/main.c:243-273   Receiver::resync   (Q8: bounds first)
//
// Error behaviour: the firmware's functions return void and cannot fail; here a failing
// uc_* call throws std::runtime_error carrying uc_last_error() (there is no CPU fallback).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "uchirp.h"
#include "uchirp_mainloop.hpp"

namespace uchirp {

constexpr uint32_t NN = 2048;        // receiver/Inc/main.h:97
constexpr float SNR_THRESHOLD = 2.0f;  // receiver/Inc/main.h:98
enum chirp { DOWN_CHIRP = UC_DOWN_CHIRP, UP_CHIRP = UC_UP_CHIRP };  // receiver/Inc/chirp.h:12-14

struct history {  // receiver/Src/main.c:124-136
  float mag_max, mag_max_left, mag_max_right;
  int32_t max_freq, max_freq_left, max_freq_right;
  uint32_t start_time, finish_time;
  float mag_mean, snr;
  char rank;
};

class Receiver {
 public:
  typedef history history_t;          // (uchirp_mainloop.hpp: what MainLoop<Receiver> keeps per slot)
  float fifo_queue[NN * 3] = {0.0f};  // main.c:94
  bool new_pcm_data = false;          // main.c:84
  uint32_t bandwidth = 0, bandwidth2 = 0, idx_left_zero = 0;  // main.c:138-140

  explicit Receiver(float fs = 78125.0f, int variant = UC_RX_REAL, int device = 0) {
    uc_config cfg;
    check(uc_default_config(variant, &cfg), "uc_default_config");
    cfg.fs = fs;
    cfg.device = device;
    check(uc_create(&cfg, &ctx_), "uc_create");
    if (uc_stats_per_frame(ctx_) != 2) {  // dsp() reads the up AND the down history of a frame
      uc_destroy(ctx_);
      ctx_ = nullptr;
      throw std::invalid_argument("Receiver: the variant has no up/down history pair (UC_RX_REAL or UC_SYNC_CPLX)");
    }
    uc_get_windows(ctx_, &bandwidth, &bandwidth2, &idx_left_zero);
  }
  ~Receiver() { uc_destroy(ctx_); }
  Receiver(const Receiver&) = delete;
  Receiver& operator=(const Receiver&) = delete;

  // ISR: shift the FIFO by one block, append (float)buf[i]; drops the block if the consumer is busy
  void HAL_DFSDM_FilterRegConvCpltCallback(const int32_t* buf) {
    if (!new_pcm_data) {
      std::memmove(fifo_queue, fifo_queue + NN, sizeof(float) * 2 * NN);
      for (uint32_t i = 0; i < NN; i++) fifo_queue[2 * NN + i] = (float)buf[i];
      new_pcm_data = true;
    }
  }

  int32_t idx2freq(uint32_t idx) const { return uc_idx2freq(ctx_, idx); }

  void dsp(uint32_t sync_position, history* phist, float mag_mean, int updown) {
    if (sync_position + NN > 3 * NN) throw std::out_of_range("dsp: sync_position beyond the FIFO");
    uc_stats st[2];
    const float mm[2] = {mag_mean, mag_mean};
    check(uc_process_batch(ctx_, &fifo_queue[sync_position], UC_DTYPE_F32, 1, NN, mm, nullptr, st, nullptr),
          "uc_process_batch");
    const uc_stats& s = st[updown == UP_CHIRP ? 0 : 1];
    phist->mag_max = s.mag_max;
    phist->mag_max_left = s.mag_max_left;
    phist->mag_max_right = s.mag_max_right;
    phist->max_freq = s.max_freq;
    phist->max_freq_left = s.max_freq_left;
    phist->max_freq_right = s.max_freq_right;
    phist->start_time = phist->finish_time = 0;
    phist->mag_mean = mag_mean;
    phist->snr = s.snr;
  }

  float symbol_snr(uint32_t sync_position, history* phist, int updown) {
    return uchirp::symbol_snr(*this, sync_position, phist, updown);
  }

  void resync(float snr, history hist[], uint32_t offset, uint32_t* sync_position, int updown) {
    uchirp::resync(*this, NN, snr, hist, offset, sync_position, updown);
  }

  uc_ctx* ctx() { return ctx_; }

 private:
  static void check(int rc, const char* what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + uc_last_error());
  }
  uc_ctx* ctx_ = nullptr;
};

}  // namespace uchirp
