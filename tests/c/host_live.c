/* host_live.c -- a plain C99 host that runs libuchirp.so the way the firmware runs: a new 2048-sample block of every
 * microphone arrives, one call makes one pass of main()'s switch per stream (receiver/Src/main.c:417-578; the ISR's FIFO
 * :659-668), characters are printed as they complete (main.c:533).  The receivers' state -- FIFO tails, mag_stat[],
 * history[], sync_position, the byte being assembled -- stays on the device between the calls (uc_rx_state).
 * The microphones here are synthetic: noise, then the K7 transmission (generator/ChirpGenerator.ipynb: G, 7 x H, L, the
 * message MSB first, 12 x G), one symbol per block, a different message and noise per stream.
 *
 * usage: host_live [n_streams=3] [pdm]
 * With `pdm` the chain starts where the board's does: at the microphones' 1-bit PDM streams (2.5 Mbit/s each, a second-order
 * delta-sigma modulator here), 2048 words of 32 bits per block and microphone, handed over as UC_DTYPE_PDM -- the DFSDM
 * (receiver/Src/dfsdm.c:59-61,69,78) runs on the device and its filter history travels in the uc_rx_state with the rest.
 * Prints every stream's characters as they are decoded and, at the end, what each stream received.  Exit 0 when every
 * stream received its message.  Without a GPU: prints uc_create's error and exits 0 (there is no CPU path). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uchirp.h"

#define NN 2048
#define MAXS 16
#define LEAD 44 /* blocks of noise in front: mag_mean needs 24 of them (main.c:321,431) */

static uint64_t lcg = 0x9E3779B97F4A7C15ull;
static double uniform01(void) {
  lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
  return ((double)(lcg >> 11) + 0.5) / 9007199254740992.0;
}
static void block(int32_t* out, const uc_config* cfg, int kind /* 1 H, 0 L, 2 G */, double amp, double sigma) {
  const double fs = cfg->fs, T = cfg->n / fs, k = (cfg->f1 - cfg->f0) / T, pi = 3.14159265358979323846;
  uint32_t i;
  for (i = 0; i < cfg->n; i += 2) {
    const double r = sigma * sqrt(-2.0 * log(uniform01())), a = 2.0 * pi * uniform01();
    const double nz[2] = {r * cos(a), r * sin(a)};
    int h;
    for (h = 0; h < 2; h++) {
      const double t = (i + h) / fs;
      double s = 0.0;
      if (kind != 2) {
        const double f = kind ? cfg->f0 + k * t / 2.0 : cfg->f1 - k * t / 2.0;
        const double arg = 2.0 * pi * f * t - pi / 2.0;
        s = amp * (cos(arg) + sin(arg));
      }
      out[i + h] = (int32_t)lrint(s + nz[h]) * 256; /* a DFSDM word: 24-bit sample in bits 31:8 */
    }
  }
}

/* the same block as the microphone's bit stream: the waveform at the PDM bit rate (32 x fs) through a second-order
 * delta-sigma modulator whose integrators (dsm[0], dsm[1]) run on from block to block; full scale = 1 */
static void block_pdm(uint32_t* out, const uc_config* cfg, int kind, double amp, double sigma, double dsm[2]) {
  const double fs = cfg->fs, T = cfg->n / fs, k = (cfg->f1 - cfg->f0) / T, pi = 3.14159265358979323846;
  uint32_t i;
  int j;
  for (i = 0; i < cfg->n; i++) {
    const double nz = sigma * sqrt(-2.0 * log(uniform01())) * cos(2.0 * pi * uniform01());
    uint32_t w = 0;
    for (j = 0; j < 32; j++) {
      const double t = (i + j / 32.0) / fs;
      double x = nz, y;
      if (kind != 2) {
        const double f = kind ? cfg->f0 + k * t / 2.0 : cfg->f1 - k * t / 2.0;
        const double arg = 2.0 * pi * f * t - pi / 2.0;
        x += amp * (cos(arg) + sin(arg));
      }
      y = dsm[1] >= 0.0 ? 1.0 : -1.0;
      dsm[0] += x - y;
      dsm[1] += dsm[0] - y;
      if (y > 0.0) w |= (uint32_t)1 << j; /* bit t of the stream = bit (t & 31) of word t >> 5 */
    }
    out[i] = w;
  }
}

int main(int argc, char** argv) {
  const int ns = argc > 1 ? atoi(argv[1]) : 3;
  const int pdm = argc > 2 && strcmp(argv[2], "pdm") == 0;
  static double dsm[MAXS][2];
  static const char* const MSGS[4] = {"Hello World!", "uchirp", "MI355X", "0123456789"};
  uc_config cfg;
  uc_ctx* uc = NULL;
  uc_rx_state* rx = NULL;
  int32_t* words;
  char text[MAXS][32], got[MAXS][64];
  uint32_t n_text[MAXS];
  int s, b, total = 0, ok = 1;
  if (ns < 1 || ns > MAXS) return 2;
  if (uc_default_config(UC_SYNC_CPLX, &cfg) != 0) return 1; /* the complex reference: decodes whole texts */
  {
    const int rc = uc_create(&cfg, &uc);
    if (rc != 0) {
      printf("uc_create: %d (%s)\n", rc, uc_last_error());
      return uc_device_count() == 0 ? 0 : 1;
    }
  }
  if (uc_rx_state_create(uc, (size_t)ns, &rx) != 0) { printf("uc_rx_state_create: %s\n", uc_last_error()); return 1; }
  words = (int32_t*)malloc(sizeof(int32_t) * NN * (size_t)ns);
  if (!words) return 1;
  memset(got, 0, sizeof(got));
  for (s = 0; s < ns; s++) {
    const int len = (int)strlen(MSGS[s % 4]);
    const int blocks = LEAD + 1 + 7 + 1 + 8 * len + 12 + 6;
    if (blocks > total) total = blocks;
  }
  for (b = 0; b < total; b++) { /* "every 26.2 ms": one new block of every microphone */
    for (s = 0; s < ns; s++) {
      const char* m = MSGS[s % 4];
      const int len = (int)strlen(m), q = b - LEAD;
      int kind = 2;
      if (q >= 1 && q <= 7) kind = 1;
      else if (q == 8) kind = 0;
      else if (q >= 9 && q < 9 + 8 * len) kind = (m[(q - 9) / 8] >> (7 - (q - 9) % 8)) & 1;
      if (pdm) block_pdm((uint32_t*)words + (size_t)s * NN, &cfg, kind, 0.25, 0.00625, dsm[s]);
      else block(words + (size_t)s * NN, &cfg, kind, 2000.0, 50.0);
    }
    if (uc_receive_streams_next(uc, rx, words, pdm ? UC_DTYPE_PDM : UC_DTYPE_I32, NN, NN, NULL, &text[0][0], sizeof(text[0]), n_text, NULL, 0, NULL,
                                NULL) != 0) {
      printf("uc_receive_streams_next: %s\n", uc_last_error());
      return 1;
    }
    for (s = 0; s < ns; s++)
      if (n_text[s]) {
        uint32_t c;
        for (c = 0; c < n_text[s]; c++)
          if (text[s][c] != '\n') printf("block %3d  stream %d  '%c'\n", b, s, text[s][c]);
        {
          const size_t have = strlen(got[s]);
          if (have + n_text[s] < sizeof(got[s])) memcpy(got[s] + have, text[s], n_text[s]); /* got[] was zeroed: stays terminated */
        }
      }
  }
  for (s = 0; s < ns; s++) {
    char* nl = strchr(got[s], '\n');
    if (nl) *nl = 0;
    printf("stream %d received \"%s\" (sent \"%s\")\n", s, got[s], MSGS[s % 4]);
    if (strcmp(got[s], MSGS[s % 4]) != 0) ok = 0;
  }
  free(words);
  uc_rx_state_destroy(rx);
  uc_destroy(uc);
  return ok ? 0 : 1;
}
