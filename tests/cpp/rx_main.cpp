// rx_main.cpp -- the firmware's main loop run against include/uchirp_receiver.hpp: the same functions the
// reference calls (dsp / symbol_snr / resync / the ISR callback), one frame at a time, each running on the GPU,
// driven by the switch of include/uchirp_mainloop.hpp -- the very code uc_receive_stream replays.  Reads int32
// DFSDM words from a file, prints the decoded characters like the firmware's printf (receiver/Src/main.c:417-554).
// A third argument "busy" keeps the consumer busy on every 7th block: the ISR then drops that block (main.c:661).
#include <cstdio>
#include <cstring>
#include <vector>

#include "uchirp_receiver.hpp"

using namespace uchirp;

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: rx_main words.i32 [variant] [busy]\n"); return 2; }
  std::FILE* fp = std::fopen(argv[1], "rb");
  if (!fp) { std::perror(argv[1]); return 2; }
  const int variant = argc > 2 ? std::atoi(argv[2]) : UC_RX_REAL;
  const bool busy_mode = argc > 3 && std::strcmp(argv[3], "busy") == 0;
  try {
    Receiver rx(78125.0f, variant, 0);
    MainLoop<Receiver> loop(NN, SNR_THRESHOLD);
    std::vector<int32_t> buf(NN);
    size_t b = 0;
    while (std::fread(buf.data(), sizeof(int32_t), NN, fp) == NN) {
      const bool busy = busy_mode && (b % 7 == 3);
      b++;
      if (busy) rx.new_pcm_data = true;                   // the main loop is still working on the previous block
      rx.HAL_DFSDM_FilterRegConvCpltCallback(buf.data());  // ... so the ISR drops this one (main.c:661)
      if (busy) { rx.new_pcm_data = false; continue; }     // the late consumer finishes: nothing new to process
      if (!rx.new_pcm_data) continue;
      loop.step(rx, [](char c) { std::printf("%c", c); });
      rx.new_pcm_data = false;
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rx_main: %s\n", e.what());
    return 1;
  }
  std::fclose(fp);
  return 0;
}
