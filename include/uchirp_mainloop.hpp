// uchirp_mainloop.hpp -- the receiver's `while (1)` (receiver/Src/main.c:417-554) and resync() (main.c:243-273) as
// ONE header-only implementation over an abstract dsp().  Both users replay exactly this code:
//   * libuchirp.so's uc_receive_stream (csrc/uc_api_rx.cpp): dsp() looks the frame up in the statistics of one batched
//     launch over every FIFO offset;
//   * the C++ host layer (include/uchirp_receiver.hpp, tests/cpp/rx_main.cpp): dsp() is one GPU call per frame;
//   * uc_receive_streams (csrc/uc_rx_kernel.hip): the SAME code compiled for the device, one lane per recorded stream
//     (every function below is __host__ __device__ under hipcc).
// (The CPU oracle keeps its own, independent C restatement: oracle/uc_oracle.c, uco_receive_stream.)
//
// `Dsp` provides
//     typedef ... history_t;      with float members mag_max, mag_mean, snr   (struct history, main.c:124-136)
//     void dsp(uint32_t sync_position, history_t* phist, float mag_mean, int updown);      (main.c:183-231)
// Quirk decision Q8: resync() checks the FIFO bounds BEFORE evaluating a neighbour position (the
// firmware evaluates first, which reads outside the FIFO).
#pragma once
#include <cstdint>

#include "uchirp.h"

#if defined(__HIP__) || defined(__HIPCC__)
#define UC_HD __host__ __device__
#else
#define UC_HD
#endif

namespace uchirp {

// symbol_snr(): main.c:233-236
template <class Dsp>
UC_HD inline float symbol_snr(Dsp& d, uint32_t sync_position, typename Dsp::history_t* phist, int updown) {
  d.dsp(sync_position, phist, phist->mag_mean, updown);
  return phist->snr;
}

// resync(): main.c:243-273
template <class Dsp>
UC_HD inline void resync(Dsp& d, uint32_t n, float snr, typename Dsp::history_t hist[], uint32_t offset, uint32_t* sync_position,
                   int updown) {
  const int32_t sync_position_l = (int32_t)*sync_position - (int32_t)offset;
  const int32_t sync_position_r = (int32_t)*sync_position + (int32_t)offset;
  float snr_l = -1e38f, snr_r = -1e38f;
  if (sync_position_l >= 0) snr_l = symbol_snr(d, (uint32_t)sync_position_l, &hist[2], updown);
  if (sync_position_r <= (int32_t)(2 * n)) snr_r = symbol_snr(d, (uint32_t)sync_position_r, &hist[3], updown);
  if ((snr > snr_l) && (snr > snr_r)) return;
  if (snr_l >= snr_r) {
    if (sync_position_l >= 0) *sync_position = (uint32_t)sync_position_l;
  } else if (snr_l < snr_r) {
    if (sync_position_r <= (int32_t)(2 * n)) *sync_position = (uint32_t)sync_position_r;
  }
}

// what one pass through the switch did (uc_rx_event of include/uchirp.h without the block index)
struct loop_event {
  int state_before, state_after;
  int bit;                 // 0 / 1 if a data bit was decoded, else -1
  uint32_t sync_position;  // after the pass
  float snr_up, snr_down;  // of SYNCHRONIZED / DATA_RECEIVING passes, else 0
};

// main()'s locals (main.c:314-339) and one pass of its switch per new block (`new_pcm_data` set)
template <class Dsp>
class MainLoop {
 public:
  typedef typename Dsp::history_t history_t;

  UC_HD MainLoop(uint32_t n, float snr_threshold) : n_(n), thr_(snr_threshold), offset_(n / 8), shift_(n / 4), sync_position_(n / 2) {
    for (float& v : mag_stat_) v = 1E37f;  // main.c:321
    for (history_t& h : history_) h = history_t();
  }

  UC_HD int state() const { return state_; }
  UC_HD uint32_t turn() const { return turn_; }  // which of the two interleaved position sets the NEXT acquisition pass visits
  UC_HD uint32_t sync_position() const { return sync_position_; }

  // `put(char)` receives the decoded characters ('\n' ends a message), as the firmware's printf does
  template <class Put>
  UC_HD loop_event step(Dsp& d, Put&& put) {
    loop_event ev{state_, state_, -1, sync_position_, 0.0f, 0.0f};
    switch (state_) {
      case UC_STATE_IDLE: {
        sync_cnt_ = 0;
        float sum = 0.0f;  // arm_mean_f32(&mag_stat[4], 8, &mag_mean): main.c:431
        for (int i = 4; i < 12; i++) sum += mag_stat_[i];
        mag_mean_ = sum / 8.0f;
      }
        [[fallthrough]];  // "intentionally no break here": main.c:434
      case UC_STATE_SYNCHRONIZING: {
        for (uint32_t i = 0; i < 4; i++) {  // main.c:447-451
          sync_position_ = n_ / 2 + turn_ * offset_ + shift_ * i;
          d.dsp(sync_position_, &history_[i * 2 + turn_], mag_mean_, UC_UP_CHIRP);
        }
        turn_ = (turn_ == 0) ? 1 : 0;
        if (turn_ == 1) {  // main.c:455-487
          for (int i = 10; i >= 0; i--) mag_stat_[i + 1] = mag_stat_[i];
          float mag_max_max = 0.0f;
          for (int i = 0; i < 8; i++) {
            const float mag_max = history_[i].mag_max;
            if (mag_max > mag_max_max) { mag_max_max = mag_max; max_idx_ = (uint32_t)i; }
          }
          mag_stat_[0] = mag_max_max;
          const float snr = (mag_max_max - mag_mean_) / mag_mean_;
          if (snr >= thr_) {
            state_ = UC_STATE_SYNCHRONIZING;
            if (++sync_cnt_ >= 3) {
              state_ = UC_STATE_SYNCHRONIZED;
              sync_position_ = n_ / 2 + max_idx_ * offset_;
            }
          } else {
            state_ = UC_STATE_IDLE;
          }
        }
        break;
      }
      case UC_STATE_SYNCHRONIZED:  // main.c:491-510
        ev.snr_up = symbol_snr(d, sync_position_, &history_[0], UC_UP_CHIRP);
        ev.snr_down = symbol_snr(d, sync_position_, &history_[1], UC_DOWN_CHIRP);
        if ((ev.snr_up >= thr_) || (ev.snr_down >= thr_)) {
          if (ev.snr_down > ev.snr_up) {
            resync(d, n_, ev.snr_down, history_, offset_, &sync_position_, UC_DOWN_CHIRP);
            state_ = UC_STATE_DATA_RECEIVING;
          } else {
            resync(d, n_, ev.snr_up, history_, offset_, &sync_position_, UC_UP_CHIRP);
          }
        } else {
          state_ = UC_STATE_IDLE;
        }
        break;
      case UC_STATE_DATA_RECEIVING:  // main.c:512-550
        ev.snr_up = symbol_snr(d, sync_position_, &history_[0], UC_UP_CHIRP);
        ev.snr_down = symbol_snr(d, sync_position_, &history_[1], UC_DOWN_CHIRP);
        if ((ev.snr_up >= thr_) || (ev.snr_down >= thr_)) {
          if (ev.snr_down > ev.snr_up) {
            ev.bit = 0;
            msg_ = (unsigned char)((msg_ << 1) + 0);
            resync(d, n_, ev.snr_down, history_, offset_, &sync_position_, UC_DOWN_CHIRP);
          } else {
            ev.bit = 1;
            msg_ = (unsigned char)((msg_ << 1) + 1);
            resync(d, n_, ev.snr_up, history_, offset_, &sync_position_, UC_UP_CHIRP);
          }
          if (++msg_cnt_ >= 8) {
            put((char)msg_);
            msg_ = 0;
            msg_cnt_ = 0;
          }
        } else {
          put('\n');
          state_ = UC_STATE_IDLE;
          msg_ = 0;
          msg_cnt_ = 0;
        }
        break;
    }
    ev.state_after = state_;
    ev.sync_position = sync_position_;
    return ev;
  }

 private:
  uint32_t n_;
  float thr_;
  uint32_t offset_, shift_;  // main.c:406-407
  uint32_t max_idx_ = 0, turn_ = 0;
  history_t history_[8];
  float mag_stat_[12];
  float mag_mean_ = 0.0f;
  uint32_t sync_cnt_ = 0, sync_position_;
  int state_ = UC_STATE_IDLE;
  unsigned char msg_ = 0;
  int msg_cnt_ = 0;
};

}  // namespace uchirp
