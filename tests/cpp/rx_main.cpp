// rx_main.cpp -- the firmware's main loop written against include/uchirp_receiver.hpp:
// the same functions the reference calls (dsp / symbol_snr / resync / the ISR callback),
// one frame at a time, each running on the GPU.  Reads int32 DFSDM words from a file,
// prints the decoded characters like the firmware's printf (receiver/Src/main.c:417-554).
#include <cstdio>
#include <vector>

#include "uchirp_receiver.hpp"

using namespace uchirp;

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: rx_main words.i32 [variant]\n"); return 2; }
  std::FILE* fp = std::fopen(argv[1], "rb");
  if (!fp) { std::perror(argv[1]); return 2; }
  const int variant = argc > 2 ? std::atoi(argv[2]) : UC_RX_REAL;
  try {
    Receiver rx(78125.0f, variant, 0);
    enum state_t { IDLE, SYNCHRONIZING, SYNCHRONIZED, DATA_RECEIVING } state = IDLE;
    uint32_t max_idx = 0, turn = 0;
    const uint32_t offset = NN / 8, shift = NN / 4;
    history hist[8];
    std::memset(hist, 0, sizeof(hist));
    float mag_stat[12];
    for (float& v : mag_stat) v = 1E37f;
    float mag_mean = 0.0f;
    uint32_t sync_cnt = 0, sync_position = NN / 2;
    unsigned char msg = 0;
    int msg_cnt = 0;
    std::vector<int32_t> buf(NN);

    while (std::fread(buf.data(), sizeof(int32_t), NN, fp) == NN) {
      rx.HAL_DFSDM_FilterRegConvCpltCallback(buf.data());
      if (!rx.new_pcm_data) continue;
      switch (state) {
        case IDLE: {
          sync_cnt = 0;
          float sum = 0.0f;
          for (int i = 4; i < 12; i++) sum += mag_stat[i];
          mag_mean = sum / 8.0f;
        }
          [[fallthrough]];
        case SYNCHRONIZING:
          for (uint32_t i = 0; i < 4; i++) {
            sync_position = NN / 2 + turn * offset + shift * i;
            rx.dsp(sync_position, &hist[i * 2 + turn], mag_mean, UP_CHIRP);
          }
          turn = (turn == 0) ? 1 : 0;
          if (turn == 1) {
            for (int i = 10; i >= 0; i--) mag_stat[i + 1] = mag_stat[i];
            float mag_max_max = 0.0f;
            for (int i = 0; i < 8; i++)
              if (hist[i].mag_max > mag_max_max) { mag_max_max = hist[i].mag_max; max_idx = (uint32_t)i; }
            mag_stat[0] = mag_max_max;
            const float snr = (mag_max_max - mag_mean) / mag_mean;
            if (snr >= SNR_THRESHOLD) {
              state = SYNCHRONIZING;
              if (++sync_cnt >= 3) { state = SYNCHRONIZED; sync_position = NN / 2 + max_idx * offset; }
            } else {
              state = IDLE;
            }
          }
          break;
        case SYNCHRONIZED: {
          const float snr_up = rx.symbol_snr(sync_position, &hist[0], UP_CHIRP);
          const float snr_down = rx.symbol_snr(sync_position, &hist[1], DOWN_CHIRP);
          if ((snr_up >= SNR_THRESHOLD) || (snr_down >= SNR_THRESHOLD)) {
            if (snr_down > snr_up) { rx.resync(snr_down, hist, offset, &sync_position, DOWN_CHIRP); state = DATA_RECEIVING; }
            else rx.resync(snr_up, hist, offset, &sync_position, UP_CHIRP);
          } else {
            state = IDLE;
          }
          break;
        }
        case DATA_RECEIVING: {
          const float snr_up = rx.symbol_snr(sync_position, &hist[0], UP_CHIRP);
          const float snr_down = rx.symbol_snr(sync_position, &hist[1], DOWN_CHIRP);
          if ((snr_up >= SNR_THRESHOLD) || (snr_down >= SNR_THRESHOLD)) {
            if (snr_down > snr_up) { msg = (unsigned char)((msg << 1) + 0); rx.resync(snr_down, hist, offset, &sync_position, DOWN_CHIRP); }
            else { msg = (unsigned char)((msg << 1) + 1); rx.resync(snr_up, hist, offset, &sync_position, UP_CHIRP); }
            if (++msg_cnt >= 8) { std::printf("%c", msg); msg = 0; msg_cnt = 0; }
          } else {
            std::printf("\n");
            state = IDLE; msg = 0; msg_cnt = 0;
          }
          break;
        }
      }
      rx.new_pcm_data = false;
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rx_main: %s\n", e.what());
    return 1;
  }
  std::fclose(fp);
  return 0;
}
