// need_check.cpp -- ties uc::need_word (csrc/uc_rx.hpp: what a live receiver's NEXT block may be evaluated for) to main()'s
// switch itself (include/uchirp_mainloop.hpp = receiver/Src/main.c:417-554, resync 243-273).  The masks encode the acquisition
// pattern (4 positions a block, two interleaved sets, UP only, main.c:447-453), the lock rule (three evaluations) and resync's
// one step per block; nothing else connects them to the loop -- an edit to the loop would silently drop statistics the switch
// reads.  Here the loop runs over random statistics (noise, preambles, data, drop-outs, every threshold) with a dsp() that
// records WHICH FIFO position and reference every call reads; every read must lie inside the need word that was emitted when
// the frame's block was still to come:
//   FIFO offset k = pos / 256 at block b:   k = 9 .. 16  is a new offset of block b      -> need(b)      bit k - 9
//                                           k = 1 .. 8   was offset k + 8 of block b - 1 -> need(b - 1)  bit k - 1
//                                           k = 0        was offset 16 of block b - 2    -> need(b - 2)  bit 7
//   a DOWN read additionally needs bit 8 of that word.  need(b) = need_word(state, turn, sync_position) after block b - 1,
//   need(0) = need_word(IDLE, 0, 0) = what uc_rx_state_reset writes; blocks before power-on are zeros (nothing to evaluate).
// Also: every word holds offset m = 7 or m = 8 (the frame that hands the block to the state rides on one of them:
// uc_band_kernel.hip, rows_masks) and is never empty.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/uchirp.h"
#include "../../include/uchirp_mainloop.hpp"
#include "../../ultrasonic-communication_amd/csrc/uc_rx.hpp"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(rng_state >> 33);
}

struct Hist {
  float mag_max = 0.f, mag_mean = 0.f, snr = 0.f;
};
struct Read {
  uint32_t k;
  int updown;
};
struct SpyDsp {
  typedef Hist history_t;
  float level_up = 1.f, level_dn = 1.f;
  std::vector<Read> reads;
  void dsp(uint32_t pos, Hist* h, float mag_mean, int updown) {
    reads.push_back({pos >> 8, updown});
    const float jitter = 0.75f + (float)(rnd() & 1023) / 2048.0f;
    const float m = (updown == UC_UP_CHIRP ? level_up : level_dn) * jitter;
    h->mag_max = m;
    h->mag_mean = mag_mean;
    h->snr = (m - mag_mean) / mag_mean;
  }
};

int main() {
  uint64_t reads = 0, down_reads = 0, tracking_blocks = 0;
  uint64_t by_state[4] = {0, 0, 0, 0};
  for (int run = 0; run < 60; run++) {
    SpyDsp d;
    const float thr = run % 3 == 0 ? 2.0f : (run % 3 == 1 ? 0.5f : 6.0f);
    uchirp::MainLoop<SpyDsp> loop(2048, thr);
    auto put = [](char) {};
    std::vector<uint32_t> need;  // need[b]: the word in force for block b's new offsets
    need.push_back(uc::need_word(UC_STATE_IDLE, 0, 0));
    if (need[0] != 0x052u) { fprintf(stderr, "need_check: the power-on word is 0x%x\n", need[0]); return 1; }
    int phase = 0, left = 0;
    for (int b = 0; b < 30000; b++) {
      if (left-- <= 0) {
        phase = (int)(rnd() % 4);
        left = 1 + (int)(rnd() % 60);
      }
      const float noise = 1.0f + (float)(rnd() % 100) / 100.0f;
      d.level_up = phase == 1 || (phase == 2 && (rnd() & 1)) ? noise * (3.0f + (float)(rnd() % 40)) : noise;
      d.level_dn = phase == 2 && d.level_up <= noise * 2.0f ? noise * (3.0f + (float)(rnd() % 40)) : noise;
      if (phase == 3) d.level_up = d.level_dn = noise * (rnd() % 7 == 0 ? 2.9f : 1.0f);
      d.reads.clear();
      const int state_before = loop.state();
      loop.step(d, put);
      by_state[state_before]++;
      if (state_before >= UC_STATE_SYNCHRONIZED) tracking_blocks++;
      for (const Read& r : d.reads) {
        reads++;
        if (r.k > 16) { fprintf(stderr, "need_check: a read outside the FIFO (k = %u)\n", r.k); return 1; }
        int src;
        uint32_t bit;
        if (r.k >= 9) { src = b; bit = r.k - 9; }
        else if (r.k >= 1) { src = b - 1; bit = r.k - 1; }
        else { src = b - 2; bit = 7; }
        if (src < 0) continue;  // a block from before power-on: zeros
        const uint32_t w = need[(size_t)src];
        if (!((w >> bit) & 1u)) {
          fprintf(stderr, "need_check: run %d block %d (state %d): FIFO offset k = %u is read, but the need word of block %d "
                          "(0x%03x) passes its offset m = %u over\n", run, b, state_before, r.k, src, w, bit + 1);
          return 1;
        }
        if (r.updown == UC_DOWN_CHIRP) {
          down_reads++;
          if (!((w >> 8) & 1u)) {
            fprintf(stderr, "need_check: run %d block %d (state %d): a DOWN statistic of k = %u is read, but the need word of "
                            "block %d (0x%03x) has no DOWN bit\n", run, b, state_before, r.k, src, w);
            return 1;
          }
        }
      }
      const uint32_t w = uc::need_word(loop.state(), loop.turn(), loop.sync_position());
      if ((w & 0xC0u) == 0 || (w & 0xFFu) == 0 || w > 0x1FFu) {
        fprintf(stderr, "need_check: need word 0x%x (state %d) holds neither m = 7 nor m = 8\n", w, loop.state());
        return 1;
      }
      need.push_back(w);
    }
  }
  if (reads < 1000000 || down_reads < 10000 || tracking_blocks < 10000 || !by_state[1] || !by_state[2] || !by_state[3]) {
    fprintf(stderr, "need_check: the scenario did not visit every state (reads %llu, down %llu, tracking %llu)\n",
            (unsigned long long)reads, (unsigned long long)down_reads, (unsigned long long)tracking_blocks);
    return 1;
  }
  printf("need_word ok: %llu reads (%llu of the DOWN reference) over %llu tracking blocks, all inside the need words in force; "
         "blocks by state %llu / %llu / %llu / %llu\n", (unsigned long long)reads, (unsigned long long)down_reads,
         (unsigned long long)tracking_blocks, (unsigned long long)by_state[0], (unsigned long long)by_state[1],
         (unsigned long long)by_state[2], (unsigned long long)by_state[3]);
  return 0;
}
