/* host_node_live.c -- a plain C99 host that serves the microphones of a NODE: one process, n_devices MI355X, the live
 * streams block-partitioned over them (the state machine of one stream does not shard, so "run one
 * independent stream per GPU" -- here many per GPU), every device running the firmware's loop over its share
 * (receiver/Src/main.c:417-578, ISR FIFO :659-668; state on the device: uc_rx_state), the decoded characters of ALL streams
 * gathered to every device over RCCL after each block (uc_group_receive_streams_next).
 * The microphones are synthetic: noise, then the K7 transmission (generator/ChirpGenerator.ipynb: G, 7 x H, L, the message
 * MSB first, 12 x G), one symbol per block, a different message and noise per stream.
 *
 * usage: host_node_live [n_streams=5] [n_devices=1] [pdm]
 * With `pdm` the node's chain starts at the microphones' 1-bit PDM streams (UC_DTYPE_PDM: 2048 words of 32 bits per block and
 * microphone from a second-order delta-sigma modulator; the DFSDM of receiver/Src/dfsdm.c:59-61,69,78 runs on the devices,
 * every microphone's filter history travels in its device's uc_rx_state).
 * Prints what each stream received, as seen in the gathered arrays of EVERY local device.  Exit 0 when every device holds
 * every stream's message.  Without a GPU: prints uc_group_create's error and exits 0 (there is no CPU path). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uchirp.h"

#define NN 2048
#define MAXS 64
#define MAXDEV 16
#define CAP 8   /* characters per stream and call (one block can complete at most one) */
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

/* the same block as the microphone's bit stream (see tests/c/host_live.c): the waveform at 32 x fs through a second-order
 * delta-sigma modulator whose integrators run on from block to block; full scale = 1 */
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
      if (y > 0.0) w |= (uint32_t)1 << j;
    }
    out[i] = w;
  }
}

int main(int argc, char** argv) {
  const int ns = argc > 1 ? atoi(argv[1]) : 5;
  const int nd = argc > 2 ? atoi(argv[2]) : 1;
  const int pdm = argc > 3 && strcmp(argv[3], "pdm") == 0;
  static double dsm[MAXS][2];
  static const char* const MSGS[4] = {"Hello World!", "uchirp", "MI355X", "0123456789"};
  uc_config cfg;
  uc_group* g = NULL;
  uc_rx_state* rx[MAXDEV];
  int32_t devs[MAXDEV];
  int32_t* words;                     /* [ns][NN]: this block of every microphone, in stream order */
  const void* share[MAXDEV];          /* where each local device's share starts in words[] */
  static char text[MAXDEV][MAXS][CAP]; /* per local device: the gathered characters of this call */
  static uint32_t n_text[MAXDEV][MAXS];
  static char got[MAXDEV][MAXS][64];
  char* textp[MAXDEV];
  uint32_t* cntp[MAXDEV];
  int s, b, l, total = 0, ok = 1;
  if (ns < 1 || ns > MAXS || nd < 1 || nd > MAXDEV) return 2;
  if (uc_default_config(UC_SYNC_CPLX, &cfg) != 0) return 1; /* the complex reference: decodes whole texts */
  for (l = 0; l < nd; l++) devs[l] = l;
  {
    const int rc = uc_group_create(&cfg, devs, nd, &g);
    if (rc != 0) {
      printf("uc_group_create: %d (%s)\n", rc, uc_last_error());
      return uc_device_count() == 0 ? 0 : 1;
    }
  }
  words = (int32_t*)malloc(sizeof(int32_t) * NN * (size_t)ns);
  if (!words) return 1;
  for (l = 0; l < nd; l++) {
    size_t first = 0, count = 0;
    uc_partition((size_t)ns, uc_group_world(g), uc_group_first_rank(g) + l, &first, &count);
    rx[l] = NULL;
    if (count && uc_rx_state_create(uc_group_ctx(g, l), count, &rx[l]) != 0) {
      printf("uc_rx_state_create: %s\n", uc_last_error());
      return 1;
    }
    share[l] = words + first * NN;
    textp[l] = &text[l][0][0];
    cntp[l] = &n_text[l][0];
    printf("device %d serves streams %u .. %u\n", (int)devs[l], (unsigned)first, (unsigned)(first + count) - 1u);
  }
  for (l = 0; l < nd; l++)
    if (!rx[l]) { printf("more devices than streams\n"); return 2; }
  memset(got, 0, sizeof(got));
  for (s = 0; s < ns; s++) {
    const int len = (int)strlen(MSGS[s % 4]);
    const int blocks = LEAD + 1 + 7 + 1 + 8 * len + 12 + 6;
    if (blocks > total) total = blocks;
  }
  for (b = 0; b < total; b++) { /* "every 26.2 ms": one new block of every microphone of the node */
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
    /* host buffers: the call returns with the gathered characters of every stream in every device's arrays */
    if (uc_group_receive_streams_next(g, rx, share, pdm ? UC_DTYPE_PDM : UC_DTYPE_I32, (size_t)ns, NN, NN, NULL, textp, CAP, cntp,
                                      NULL) != 0) {
      printf("uc_group_receive_streams_next: %s\n", uc_last_error());
      return 1;
    }
    for (l = 0; l < nd; l++)
      for (s = 0; s < ns; s++)
        if (n_text[l][s]) {
          const size_t have = strlen(got[l][s]);
          if (have + n_text[l][s] < sizeof(got[l][s])) memcpy(got[l][s] + have, text[l][s], n_text[l][s]);
        }
  }
  for (s = 0; s < ns; s++) {
    for (l = 0; l < nd; l++) {
      char* nl = strchr(got[l][s], '\n');
      if (nl) *nl = 0;
      if (strcmp(got[l][s], MSGS[s % 4]) != 0) ok = 0;
      if (l && strcmp(got[l][s], got[0][s]) != 0) ok = 0;
    }
    printf("stream %d received \"%s\" (sent \"%s\"), the same on %d device(s)\n", s, got[0][s], MSGS[s % 4], nd);
  }
  free(words);
  for (l = 0; l < nd; l++) uc_rx_state_destroy(rx[l]);
  uc_group_destroy(g);
  return ok ? 0 : 1;
}
