/* host_main.c -- a plain C99 host of libuchirp.so: the "C host through a thin C-ABI" of BASELINE.json's north_star,
 * in the shape of the receiver's own call sequence (init once, then one dsp() per frame: receiver/Src/main.c:367-393,
 * 183-231).  Usage: host_main [n_frames]
 *   without a GPU: prints the ABI version and the error uc_create reports (exit 0: that IS the expected behaviour,
 *                  there is no CPU fallback);
 *   with a GPU   : one up chirp and one down chirp frame through uc_process_frame, then a batch of n_frames through
 *                  uc_process_batch; prints the symbols (1 = up, 0 = down). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "uchirp.h"

static void chirp_frame(int32_t* pcm, const uc_config* cfg, int up) {
  /* the transmitter's orthogonal chirp A (cos(theta) + sin(theta)), simulation/signal.py:45-53, as DFSDM words */
  const double fs = cfg->fs, T = cfg->n / fs, k = (cfg->f1 - cfg->f0) / T;
  for (uint32_t i = 0; i < cfg->n; i++) {
    const double t = i / fs;
    const double f = up ? cfg->f0 + k * t / 2.0 : cfg->f1 - k * t / 2.0;
    const double arg = 2.0 * 3.14159265358979323846 * f * t - 3.14159265358979323846 / 2.0;
    pcm[i] = (int32_t)lrint(1000.0 * (cos(arg) + sin(arg))) * 256;
  }
}

int main(int argc, char** argv) {
  const size_t n_frames = argc > 1 ? (size_t)strtoul(argv[1], NULL, 10) : 8;
  uc_config cfg;
  uc_ctx* ctx = NULL;
  printf("uc_abi_version %d (header %d)\n", uc_abi_version(), UC_ABI_VERSION);
  if (uc_default_config(UC_RX_REAL, &cfg) != 0) { printf("uc_default_config: %s\n", uc_last_error()); return 1; }
  cfg.time_frame = (float)cfg.n / cfg.fs; /* sweep matched to the frame */
  cfg.mag_mean = 1000.0f;
  const int rc = uc_create(&cfg, &ctx);
  if (rc != 0) {
    printf("uc_create: %d (%s)\n", rc, uc_last_error());
    return 0;
  }
  int32_t* pcm = (int32_t*)malloc(sizeof(int32_t) * cfg.n * n_frames);
  uint8_t* sym = (uint8_t*)malloc(n_frames);
  if (!pcm || !sym) return 1;
  for (int up = 1; up >= 0; up--) {
    uint8_t s = UC_SYM_NONE;
    uc_stats st[2];
    chirp_frame(pcm, &cfg, up);
    if (uc_process_frame(ctx, pcm, cfg.mag_mean, &s, st) != 0) { printf("uc_process_frame: %s\n", uc_last_error()); return 1; }
    printf("frame %s: symbol %d  snr_up %.1f  snr_down %.1f\n", up ? "up  " : "down", (int)s, st[0].snr, st[1].snr);
  }
  for (size_t f = 0; f < n_frames; f++) chirp_frame(pcm + f * cfg.n, &cfg, (int)(f & 1));
  if (uc_process_batch(ctx, pcm, UC_DTYPE_I32, n_frames, 0, NULL, sym, NULL, NULL) != 0) {
    printf("uc_process_batch: %s\n", uc_last_error());
    return 1;
  }
  printf("batch:");
  for (size_t f = 0; f < n_frames; f++) printf(" %d", (int)sym[f]);
  printf("\n");
  free(pcm);
  free(sym);
  uc_destroy(ctx);
  return 0;
}
