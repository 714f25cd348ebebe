/* host_multi.c -- a plain C99 host that drives N MI355X through libuchirp.so's group API (include/uchirp.h, uc_group_*):
 * BASELINE.json configs[4] -- the K7 "Hello World!" wire format (generator/ChirpGenerator.ipynb cells 1-3: G, 7 x H, L,
 * 96 data bits MSB first, 12 x G) repeated over the GLOBAL frame index, one pre-aligned 2048-sample frame per symbol,
 * AWGN at -10 dB, reference sweep matched to the frame -- block-partitioned over the devices, every device decoding its
 * shard, the symbol stream all-gathered over RCCL every step (three buffers in rotation: the gather of step k overlaps
 * the kernels behind it), then decoded to text the way main() assembles bytes (receiver/Src/main.c:523-537).
 * No HIP header, no C++: uc_device_malloc / uc_device_copy move the data.
 *
 * usage: host_multi [-d n_devices] [-f frames_total] [-k steps] [-i frames.f32]
 *   -i  read frames_total x 2048 float32 frames from a file instead of generating them (the pytest harness hands the
 *       frames it also gives the Python path, and compares digests)
 * Prints: world, frames/s over the timed steps, sha256 of the gathered symbol stream, transmissions decoded, first text.
 * Exit 0 when every device holds the same gathered stream and (generated frames) every transmission decodes.
 * Without a GPU: prints uc_group_create's error and exits 0 (there is no CPU path; that IS the expected behaviour). */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "uchirp.h"

#define NN 2048
#define NBUF 3
#define MAXDEV 16
static const char MSG[] = "Hello World!";
#define PER_TX (1 + 7 + 1 + 8 * 12 + 12) /* 117 frames per transmission */

/* ---- sha256 (FIPS 180-4), own code ---- */
typedef struct { uint32_t h[8]; unsigned char buf[64]; size_t n; uint64_t bits; } sha256_t;
static const uint32_t K256[64] = {
  0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,
  0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,
  0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,0x27b70a85,
  0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,
  0xd192e819,0xd6990624,0xf40e3585,0x106aa070,0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,
  0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
static uint32_t ror(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
static void sha_block(sha256_t* s, const unsigned char* p) {
  uint32_t w[64], a[8];
  int i;
  for (i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
  for (i = 16; i < 64; i++)
    w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
           (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
  for (i = 0; i < 8; i++) a[i] = s->h[i];
  for (i = 0; i < 64; i++) {
    const uint32_t t1 = a[7] + (ror(a[4], 6) ^ ror(a[4], 11) ^ ror(a[4], 25)) + ((a[4] & a[5]) ^ (~a[4] & a[6])) + K256[i] + w[i];
    const uint32_t t2 = (ror(a[0], 2) ^ ror(a[0], 13) ^ ror(a[0], 22)) + ((a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]));
    a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
  }
  for (i = 0; i < 8; i++) s->h[i] += a[i];
}
static void sha_init(sha256_t* s) {
  static const uint32_t h0[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
  memcpy(s->h, h0, sizeof(h0));
  s->n = 0;
  s->bits = 0;
}
static void sha_update(sha256_t* s, const unsigned char* p, size_t len) {
  s->bits += (uint64_t)len * 8;
  while (len) {
    size_t take = 64 - s->n;
    if (take > len) take = len;
    memcpy(s->buf + s->n, p, take);
    s->n += take; p += take; len -= take;
    if (s->n == 64) { sha_block(s, s->buf); s->n = 0; }
  }
}
static void sha_final(sha256_t* s, char hex[65]) {
  unsigned char pad[72] = {0x80};
  unsigned char lenb[8];
  const uint64_t bits = s->bits;
  int i;
  const size_t padn = (s->n < 56) ? 56 - s->n : 120 - s->n;
  for (i = 0; i < 8; i++) lenb[i] = (unsigned char)(bits >> (56 - 8 * i));
  sha_update(s, pad, padn);
  sha_update(s, lenb, 8);
  for (i = 0; i < 8; i++) sprintf(hex + 8 * i, "%08x", s->h[i]);
}

/* ---- the transmitter, one frame per symbol ---- */
static int frame_kind(size_t global_frame) { /* 1 = H (up chirp), 0 = L (down chirp), 2 = G (silence) */
  const size_t q = global_frame % PER_TX;
  if (q == 0 || q >= 9 + 96) return 2;
  if (q <= 7) return 1;
  if (q == 8) return 0;
  return (MSG[(q - 9) / 8] >> (7 - (q - 9) % 8)) & 1;
}
static uint64_t lcg_state = 0x243F6A8885A308D3ull;
static double uniform01(void) {
  lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
  return ((double)(lcg_state >> 11) + 0.5) / 9007199254740992.0;
}
static void make_frame(float* x, const uc_config* cfg, int kind, double sigma) {
  /* A (cos(theta) + sin(theta)), simulation/signal.py:45-53, sweep over the whole frame; Box-Muller noise */
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
        s = 1000.0 * (cos(arg) + sin(arg));
      }
      x[i + h] = (float)(s + nz[h]);
    }
  }
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define CHECK(call)                                                        \
  do {                                                                     \
    const int rc_ = (call);                                                \
    if (rc_ != 0) { printf("%s: %d (%s)\n", #call, rc_, uc_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  int n_dev = 1, steps = 6, a, l;
  size_t total = (size_t)PER_TX * 70; /* 8190 frames */
  const char* file = NULL;
  uc_config cfg;
  uc_group* g = NULL;
  int32_t devs[MAXDEV];
  void* d_frames[MAXDEV];
  const void* frames[MAXDEV];
  uint8_t* d_gath[NBUF][MAXDEV];
  uint8_t* gptr[MAXDEV];
  for (a = 1; a + 1 < argc; a += 2) {
    if (!strcmp(argv[a], "-d")) n_dev = atoi(argv[a + 1]);
    else if (!strcmp(argv[a], "-f")) total = (size_t)strtoull(argv[a + 1], NULL, 10);
    else if (!strcmp(argv[a], "-k")) steps = atoi(argv[a + 1]);
    else if (!strcmp(argv[a], "-i")) file = argv[a + 1];
  }
  if (n_dev < 1 || n_dev > MAXDEV || steps < 1 || total == 0) { printf("bad arguments\n"); return 2; }
  printf("uc_abi_version %d, devices visible %d\n", uc_abi_version(), uc_device_count());
  CHECK(uc_default_config(UC_RX_REAL, &cfg));
  cfg.time_frame = (float)cfg.n / cfg.fs; /* sweep matched to the frame */
  cfg.mag_mean = 1000.0f;
  for (l = 0; l < n_dev; l++) devs[l] = l;
  {
    const int rc = uc_group_create(&cfg, devs, n_dev, &g);
    if (rc != 0) {
      printf("uc_group_create: %d (%s)\n", rc, uc_last_error());
      return uc_device_count() == 0 ? 0 : 1; /* no GPU: the expected outcome, there is no CPU path */
    }
  }
  printf("group: world %d, %d local device(s), first rank %d\n", uc_group_world(g), uc_group_local_count(g), uc_group_first_rank(g));

  /* every device gets ITS shard of the frame index space (uc_partition), generated or read on the host, resident in HBM */
  {
    const double sigma = 1000.0 * pow(10.0, 10.0 / 20.0); /* -10 dB */
    FILE* fp = file ? fopen(file, "rb") : NULL;
    if (file && !fp) { printf("cannot open %s\n", file); return 2; }
    for (l = 0; l < n_dev; l++) {
      size_t first = 0, count = 0, e0 = 0, ne = 0, f;
      float* host;
      CHECK(uc_partition(total, n_dev, l, &first, &count));
      CHECK(uc_frame_span(cfg.n, 0, 0, first, count, &e0, &ne));
      host = (float*)malloc((ne ? ne : 1) * sizeof(float));
      if (!host) return 1;
      if (fp) {
        if (fseek(fp, (long)(e0 * sizeof(float)), SEEK_SET) != 0 || fread(host, sizeof(float), ne, fp) != ne) { printf("short read\n"); return 2; }
      } else {
        for (f = 0; f < count; f++) make_frame(host + f * cfg.n, &cfg, frame_kind(first + f), sigma);
      }
      CHECK(uc_device_malloc(devs[l], ne * sizeof(float), &d_frames[l]));
      CHECK(uc_device_copy(d_frames[l], host, ne * sizeof(float)));
      frames[l] = d_frames[l];
      free(host);
      for (a = 0; a < NBUF; a++) {
        void* p = NULL;
        CHECK(uc_device_malloc(devs[l], total, &p));
        d_gath[a][l] = (uint8_t*)p;
      }
      printf("rank %d: frames [%lu, %lu) on device %d\n", l, (unsigned long)first, (unsigned long)(first + count), (int)devs[l]);
    }
    if (fp) fclose(fp);
  }

  /* warm-up, then the timed steps: decode + gather, buffers in rotation, nothing waits on the host until the end */
  {
    int k;
    double t0, dt;
    for (k = 0; k < 2; k++) {
      for (l = 0; l < n_dev; l++) gptr[l] = d_gath[k % NBUF][l];
      CHECK(uc_group_process_batch(g, frames, UC_DTYPE_F32, total, 0, gptr, NULL));
    }
    CHECK(uc_group_synchronize(g));
    t0 = now_s();
    for (k = 0; k < steps; k++) {
      for (l = 0; l < n_dev; l++) gptr[l] = d_gath[k % NBUF][l];
      CHECK(uc_group_process_batch(g, frames, UC_DTYPE_F32, total, 0, gptr, NULL));
    }
    CHECK(uc_group_synchronize(g));
    dt = now_s() - t0;
    printf("%d steps of %lu frames over %d device(s): %.3f ms per step, %.4g frames/s (decode + RCCL gather every step)\n", steps,
           (unsigned long)total, n_dev, 1e3 * dt / steps, (double)total * steps / dt);
  }

  /* every device holds the whole stream: same digest everywhere; decode the text as main() assembles bytes */
  {
    uint8_t* sym = (uint8_t*)malloc(total);
    char hex0[65], hex[65];
    size_t ntx = total / PER_TX, q, good = 0;
    char first_text[13];
    int same = 1;
    if (!sym) return 1;
    for (l = n_dev - 1; l >= 0; l--) { /* (device 0 last: its copy stays in sym for the decode below) */
      sha256_t s;
      CHECK(uc_device_copy(sym, d_gath[(steps - 1) % NBUF][l], total));
      sha_init(&s);
      sha_update(&s, sym, total);
      sha_final(&s, hex);
      if (l == n_dev - 1) memcpy(hex0, hex, 65);
      else if (strcmp(hex0, hex) != 0) same = 0;
    }
    printf("sha256 of the gathered symbol stream: %s%s\n", hex0, same ? "" : "  (DEVICES DISAGREE)");
    memset(first_text, 0, sizeof(first_text));
    for (q = 0; q < ntx; q++) {
      char text[13];
      int c, b;
      for (c = 0; c < 12; c++) {
        unsigned char msg = 0;
        for (b = 0; b < 8; b++) msg = (unsigned char)((msg << 1) + (sym[q * PER_TX + 9 + 8 * c + b] != 0 ? 1 : 0));
        text[c] = (char)msg;
      }
      text[12] = 0;
      if (q == 0) memcpy(first_text, text, 13);
      if (!strcmp(text, MSG)) good++;
    }
    printf("transmissions decoded exactly: %lu of %lu, first: \"%s\"\n", (unsigned long)good, (unsigned long)ntx, first_text);
    free(sym);
    for (l = 0; l < n_dev; l++) {
      uc_device_free(devs[l], d_frames[l]);
      for (a = 0; a < NBUF; a++) uc_device_free(devs[l], d_gath[a][l]);
    }
    uc_group_destroy(g);
    if (!same) return 1;
    if (!file && good != ntx) return 1;
  }
  return 0;
}
