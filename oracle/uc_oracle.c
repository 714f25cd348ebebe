/*
 * uc_oracle.c -- CPU restatement of the reference's per-frame DSP path.
 * TEST INFRASTRUCTURE ONLY (see uc_oracle.h for the rules and the parity
 * status).  Own code; every function cites the reference lines it follows
 * (paths relative to the reference checkout).
 */
#include "uc_oracle.h"

#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define UCO_FIR_TAPS 27
#define UCO_TABLE_SIZE 512 /* FAST_MATH_TABLE_SIZE, CMSIS/Include/arm_math.h:341 */

struct uco_ctx {
  uc_config cfg;
  uint32_t n;
  uint32_t bandwidth, bandwidth2, idx_left_zero; /* receiver/Src/main.c:372-374 */
  uint32_t center, bandwidth4;                   /* iq_modulation/Src/main.c:215-219 */
  float *up, *down;     /* n floats (real refs) or 2n (interleaved complex refs) */
  float *hann;          /* n */
  float *h_up, *h_down; /* COMPRESS: packed RFFT of windowed reference chirps */
  float *carrier_c, *carrier_s;
  float fir[UCO_FIR_TAPS];
  /* UC_STREAM */
  uint32_t decim, tmpl_len; /* D, L = n/D */
  double *tmpl64;           /* L complex: base-band template g */
  float  *tmpl32;           /* the same rounded to float32 (uco_get_table) */
  /* twiddles */
  double *tw64;         /* n/2 complex: exp(-2 pi i k / n) */
  float  *tw32;         /* same, float32 */
  float  *tw32h;        /* n/4 complex for the n/2-point CFFT of the RFFT path */
  uint32_t *rev, *revh; /* bit reversal for n and n/2 */
};

/* ------------------------------------------------------------------------- */
/* CMSIS-DSP V1.4.5 primitives, restated from the published algorithms        */
/* ------------------------------------------------------------------------- */

static float sin_table[UCO_TABLE_SIZE + 1];
static int sin_table_ready = 0;

static void init_sin_table(void) {
  /* sinTable_f32[k] = sin(2*pi*k/512), k = 0..512 (arm_common_tables.h:132) */
  if (sin_table_ready) return;
  for (int k = 0; k <= UCO_TABLE_SIZE; k++)
    sin_table[k] = (float)sin(2.0 * M_PI * (double)k / (double)UCO_TABLE_SIZE);
  sin_table_ready = 1;
}

/* arm_cos_f32: 512-entry sine table, +0.25 turn, linear interpolation
 * (contract: CMSIS/Include/arm_math.h "Fast Math" group; pinned by K6). */
float uco_arm_cos_f32(float x) {
  init_sin_table();
  float in = x * 0.159154943092f + 0.25f;
  int32_t nn = (int32_t)in;
  if (in < 0.0f) nn--;
  in = in - (float)nn;
  float findex = (float)UCO_TABLE_SIZE * in;
  uint16_t index = ((uint16_t)findex) & 0x1ff;
  float fract = findex - (float)index;
  float a = sin_table[index];
  float b = sin_table[index + 1];
  return (1.0f - fract) * a + fract * b;
}

/* arm_sin_cos_f32: argument in DEGREES (arm_math.h:4627-4637); 512-entry table
 * with cubic (Hermite) interpolation using the table itself as derivative.
 * UNPINNED: no on-device vector of this function exists in the reference. */
void uco_arm_sin_cos_f32(float theta, float* sin_val, float* cos_val) {
  init_sin_table();
  float in = theta * 0.00277777777778f;
  int32_t nn = (int32_t)in;
  if (in < 0.0f) nn--;
  in = in - (float)nn;
  float findex = (float)UCO_TABLE_SIZE * in;
  uint16_t index_s = ((uint16_t)findex) & 0x1ff;
  uint16_t index_c = (index_s + (UCO_TABLE_SIZE / 4)) & 0x1ff;
  float fract = findex - (float)index_s;
  const float dn = 0.0122718463030f; /* 2*pi/512 */
  float f1, f2, d1, d2, df, temp;

  f1 = sin_table[index_c];
  f2 = sin_table[index_c + 1];
  d1 = -sin_table[index_s];
  d2 = -sin_table[index_s + 1];
  df = f2 - f1;
  temp = dn * (d1 + d2) - 2 * df;
  temp = fract * temp + (3 * df - (d2 + 2 * d1) * dn);
  temp = fract * temp + d1 * dn;
  *cos_val = fract * temp + f1;

  f1 = sin_table[index_s];
  f2 = sin_table[index_s + 1];
  d1 = sin_table[index_c];
  d2 = sin_table[index_c + 1];
  df = f2 - f1;
  temp = dn * (d1 + d2) - 2 * df;
  temp = fract * temp + (3 * df - (d2 + 2 * d1) * dn);
  temp = fract * temp + d1 * dn;
  *sin_val = fract * temp + f1;
}

/* arm_max_f32: first maximum wins (strict '<' update), arm_math.h:6530-6541 */
void uco_arm_max_f32(const float* src, uint32_t n, float* out, uint32_t* idx) {
  float m = src[0];
  uint32_t mi = 0;
  for (uint32_t i = 1; i < n; i++) {
    if (m < src[i]) { m = src[i]; mi = i; }
  }
  *out = m;
  *idx = mi;
}

static float trig_cos(float x, int libm) {
  return libm ? (float)cos((double)x) : uco_arm_cos_f32(x);
}

static void trig_sin_cos_deg(float theta, int libm, float* s, float* c) {
  if (libm) {
    double r = (double)theta * (M_PI / 180.0);
    *s = (float)sin(r);
    *c = (float)cos(r);
  } else {
    uco_arm_sin_cos_f32(theta, s, c);
  }
}

/* ------------------------------------------------------------------------- */
/* tables                                                                     */
/* ------------------------------------------------------------------------- */

/* Hann, periodic: receiver/Src/main.c:99,390-393 (same in synchronization,
 * chirp_compression_freq_domain, iq_modulation, basic). */
void uco_hann_periodic(float* w, uint32_t n, int libm) {
  const float window_scale = (float)(2.0f * M_PI / (float)n);
  for (uint32_t i = 0; i < n; i++)
    w[i] = 0.5f - 0.5f * trig_cos((float)i * window_scale, libm);
}

/* Hann, symmetric: chirp_compression_time_domain/Src/chirp.c:13,63-65
 * (WINDOW_SCALE uses the float macro PI, arm_math.h:334) */
static void hann_symmetric(float* w, uint32_t n, int libm) {
  const float window_scale = 2.0f * 3.14159265358979f / (float)(n - 1);
  for (uint32_t i = 0; i < n; i++)
    w[i] = 0.5f - 0.5f * trig_cos((float)i * window_scale, libm);
}

/* generate_ref_chirp of the receiver: receiver/Src/chirp.c:16-40.
 * complex_out = 0: table[n] = sin(theta)            (Q3: the cos store is overwritten)
 * complex_out = 1: table[2n] = cos(theta), table[2n+1] = sin(theta)
 *                  (experiments/synchronization/Src/chirp.c:16-45,
 *                   experiments/iq_modulation/Src/chirp.c:16-40) */
static void gen_ref_chirp_deg(float* ref, uint32_t n, int up, float f0, float f1,
                              float time_frame, float fs, float phase,
                              int complex_out, int libm) {
  float freq, theta, t = 0.0f;
  float sin_val, cos_val;
  float delta_f = (float)(f1 - f0) / time_frame;
  float delta_t = time_frame / (time_frame * fs);
  for (uint32_t i = 0; i < n; i++) {
    if (up) freq = (float)(f0 + delta_f * t / 2.0);
    else    freq = (float)(f1 - delta_f * t / 2.0);
    theta = (float)(360.0 * freq * t + phase);
    t = t + delta_t;
    trig_sin_cos_deg(theta, libm, &sin_val, &cos_val);
    if (complex_out) {
      ref[2 * i] = cos_val * 1.0f;
      ref[2 * i + 1] = sin_val * 1.0f;
    } else {
      ref[i] = sin_val * 1.0f;
    }
  }
}

/* generate_ref_chirp of the two chirp_compression experiments (radians, real
 * cosine, NO 1/2 in freq):
 *   time domain: chirp_compression_time_domain/Src/chirp.c:25-46 (phase used)
 *   freq domain: chirp_compression_freq_domain/Src/chirp.c:15-36 (phase ignored) */
static void gen_ref_chirp_rad(float* ref, uint32_t n, int up, float f1, float f2,
                              float fs, float phase, int use_phase, int libm) {
  float freq, arg, t = 0.0f;
  float time_frame = (float)n / (float)fs;
  float delta_f = (f2 - f1) / time_frame;
  float delta_t = time_frame / (time_frame * (float)fs);
  const float pi_f = 3.14159265358979f;
  for (uint32_t i = 0; i < n; i++) {
    if (up) freq = f1 + delta_f * t;
    else    freq = f2 - delta_f * t;
    if (use_phase) arg = (float)(2.0 * pi_f * freq * t + phase);
    else           arg = (float)(2.0 * pi_f * freq * t);
    t = t + delta_t;
    ref[i] = trig_cos(arg, libm) * 1.0f;
  }
}

/* init_iq_modem carrier tables: experiments/iq_modulation/Src/iq_modem.c:34-45 */
static void gen_carrier(float* c, float* s, uint32_t n, float carrier, float fs,
                        float time_frame, int libm) {
  float theta, t = 0.0f;
  float delta_t = time_frame / (time_frame * fs);
  for (uint32_t i = 0; i < n; i++) {
    theta = (float)(360.0 * carrier * t);
    trig_sin_cos_deg(theta, libm, &s[i], &c[i]);
    t = t + delta_t;
  }
}

/* 27-tap LPF: experiments/iq_modulation/Src/iq_modem.c:18 (= K4, FIR LPF design.ipynb) */
static const float fir_taps[UCO_FIR_TAPS] = {
    0.01560757f, 0.02043850f, 0.02535792f, 0.03027307f, 0.03508888f, 0.03971022f,
    0.04404423f, 0.04800257f, 0.05150362f, 0.05447453f, 0.05685299f, 0.05858884f,
    0.05964532f, 0.06000000f, 0.05964532f, 0.05858884f, 0.05685299f, 0.05447453f,
    0.05150362f, 0.04800257f, 0.04404423f, 0.03971022f, 0.03508888f, 0.03027307f,
    0.02535792f, 0.02043850f, 0.01560757f};

/* ------------------------------------------------------------------------- */
/* FFTs (own code: iterative radix-2 DIT, bit-reversed input)                 */
/* ------------------------------------------------------------------------- */

static uint32_t* make_rev(uint32_t n) {
  uint32_t bits = 0;
  while ((1u << bits) < n) bits++;
  uint32_t* rev = (uint32_t*)malloc(sizeof(uint32_t) * n);
  for (uint32_t i = 0; i < n; i++) {
    uint32_t r = 0;
    for (uint32_t b = 0; b < bits; b++)
      if (i & (1u << b)) r |= 1u << (bits - 1 - b);
    rev[i] = r;
  }
  return rev;
}

/* in-place complex FFT, float64; data interleaved; tw = exp(-2 pi i k/n), k < n/2;
 * inverse != 0 conjugates the twiddles (no scaling) */
static void cfft64(double* d, uint32_t n, const double* tw, const uint32_t* rev,
                   int inverse) {
  for (uint32_t i = 0; i < n; i++) {
    uint32_t j = rev[i];
    if (j > i) {
      double tr = d[2 * i], ti = d[2 * i + 1];
      d[2 * i] = d[2 * j]; d[2 * i + 1] = d[2 * j + 1];
      d[2 * j] = tr; d[2 * j + 1] = ti;
    }
  }
  for (uint32_t len = 2; len <= n; len <<= 1) {
    uint32_t half = len >> 1, step = n / len;
    for (uint32_t base = 0; base < n; base += len) {
      for (uint32_t k = 0; k < half; k++) {
        double wr = tw[2 * k * step], wi = tw[2 * k * step + 1];
        if (inverse) wi = -wi;
        uint32_t a = base + k, b = a + half;
        double xr = d[2 * b] * wr - d[2 * b + 1] * wi;
        double xi = d[2 * b] * wi + d[2 * b + 1] * wr;
        d[2 * b] = d[2 * a] - xr; d[2 * b + 1] = d[2 * a + 1] - xi;
        d[2 * a] += xr; d[2 * a + 1] += xi;
      }
    }
  }
}

static void cfft32(float* d, uint32_t n, const float* tw, const uint32_t* rev,
                   int inverse) {
  for (uint32_t i = 0; i < n; i++) {
    uint32_t j = rev[i];
    if (j > i) {
      float tr = d[2 * i], ti = d[2 * i + 1];
      d[2 * i] = d[2 * j]; d[2 * i + 1] = d[2 * j + 1];
      d[2 * j] = tr; d[2 * j + 1] = ti;
    }
  }
  for (uint32_t len = 2; len <= n; len <<= 1) {
    uint32_t half = len >> 1, step = n / len;
    for (uint32_t base = 0; base < n; base += len) {
      for (uint32_t k = 0; k < half; k++) {
        float wr = tw[2 * k * step], wi = tw[2 * k * step + 1];
        if (inverse) wi = -wi;
        uint32_t a = base + k, b = a + half;
        float xr = d[2 * b] * wr - d[2 * b + 1] * wi;
        float xi = d[2 * b] * wi + d[2 * b + 1] * wr;
        d[2 * b] = d[2 * a] - xr; d[2 * b + 1] = d[2 * a + 1] - xi;
        d[2 * a] += xr; d[2 * a + 1] += xi;
      }
    }
  }
}

/* arm_rfft_fast_f32 forward, restated: n/2-point CFFT of the even/odd packing
 * followed by the split stage; output packed [Re X0, Re X(n/2), Re X1, Im X1, ...]
 * (layout: experiments/chirp_compression_freq_domain/README.md:11-23).
 * tw_n = exp(-2 pi i k/n) for k < n/2, tw_h for the n/2-point transform. */
static void rfft32_core(const float* in, float* out, uint32_t n, const float* tw_n,
                        const float* tw_h, const uint32_t* rev_h) {
  uint32_t h = n / 2;
  memcpy(out, in, sizeof(float) * n); /* (x[2m], x[2m+1]) is already interleaved complex */
  cfft32(out, h, tw_h, rev_h, 0);
  /* split: X[k] = (Z[k] + conj(Z[h-k]))/2 - i/2 * W^k * (Z[k] - conj(Z[h-k])) */
  float z0r = out[0], z0i = out[1];
  out[0] = z0r + z0i; /* X[0]   */
  out[1] = z0r - z0i; /* X[n/2] */
  for (uint32_t k = 1; k <= h / 2; k++) {
    uint32_t m = h - k;
    float ar = out[2 * k], ai = out[2 * k + 1];
    float br = out[2 * m], bi = out[2 * m + 1];
    float er = 0.5f * (ar + br), ei = 0.5f * (ai - bi);  /* even part  */
    float orr = 0.5f * (ai + bi), oi = -0.5f * (ar - br); /* -i/2*(A - conj B) */
    float wr = tw_n[2 * k], wi = tw_n[2 * k + 1];
    float tr = orr * wr - oi * wi, ti = orr * wi + oi * wr;
    out[2 * k] = er + tr; out[2 * k + 1] = ei + ti;
    /* X[h-k] = conj(E[k]) - conj(W^k O[k]) ... computed from the mirrored pair */
    float er2 = er, ei2 = -ei;
    float wr2 = -wr, wi2 = wi; /* W^(h-k) = -conj(W^k) */
    float orr2 = 0.5f * (bi + ai), oi2 = -0.5f * (br - ar);
    float tr2 = orr2 * wr2 - oi2 * wi2, ti2 = orr2 * wi2 + oi2 * wr2;
    if (m != k) { out[2 * m] = er2 + tr2; out[2 * m + 1] = ei2 + ti2; }
  }
}

void uco_rfft_fast_f32(const float* in, float* out_packed, uint32_t n) {
  uint32_t h = n / 2;
  float* tw_n = (float*)malloc(sizeof(float) * n);
  float* tw_h = (float*)malloc(sizeof(float) * h);
  for (uint32_t k = 0; k < h; k++) {
    tw_n[2 * k] = (float)cos(-2.0 * M_PI * k / n);
    tw_n[2 * k + 1] = (float)sin(-2.0 * M_PI * k / n);
  }
  for (uint32_t k = 0; k < h / 2; k++) {
    tw_h[2 * k] = (float)cos(-2.0 * M_PI * k / h);
    tw_h[2 * k + 1] = (float)sin(-2.0 * M_PI * k / h);
  }
  uint32_t* rev_h = make_rev(h);
  rfft32_core(in, out_packed, n, tw_n, tw_h, rev_h);
  free(rev_h); free(tw_h); free(tw_n);
}

/* ------------------------------------------------------------------------- */
/* context                                                                    */
/* ------------------------------------------------------------------------- */

int uco_default_config(int32_t variant, uc_config* cfg) {
  if (!cfg) return -EINVAL;
  memset(cfg, 0, sizeof(*cfg));
  cfg->n = 2048;            /* receiver/Inc/main.h:97 */
  cfg->phase_deg = -90.0f;  /* receiver/Src/chirp.c:43-44 */
  cfg->snr_threshold = 2.0f;/* receiver/Inc/main.h:98 */
  cfg->mag_mean = 1.0f;
  cfg->carrier = 18000.0f;  /* iq_modulation/Inc/iq_modem.h:10 */
  cfg->variant = variant;
  cfg->device = 0;
  cfg->flags = 0;
  switch (variant) {
    case UC_RX_REAL:
    case UC_SYNC_CPLX:
      cfg->fs = 78125.0f;   /* 80 MHz / 32 / 32 / 1, receiver/Src/main.c:367-369 */
      cfg->f0 = 16000.0f; cfg->f1 = 19000.0f; /* receiver/Inc/chirp.h:18-19 */
      cfg->time_frame = 0.0205f;              /* receiver/Inc/chirp.h:16 (Q4) */
      return 0;
    case UC_COMPRESS:
    case UC_DECHIRP_DOWN:
      cfg->fs = 100000.0f;  /* Divider 25: chirp_compression_*\/Src/dfsdm.c:73 */
      cfg->f0 = 17000.0f; cfg->f1 = 18000.0f; /* chirp_compression_*\/Inc/chirp.h */
      cfg->time_frame = 0.0f;                 /* n/fs */
      return 0;
    case UC_IQ:
      cfg->fs = 100000.0f;  /* iq_modulation/Src/dfsdm.c:73 */
      cfg->f0 = 16000.0f; cfg->f1 = 19000.0f; /* iq_modulation/Inc/chirp.h */
      cfg->time_frame = 0.0205f;
      return 0;
    case UC_STREAM:
      /* the shipping receiver's band and rate, carrier at the band centre */
      cfg->fs = 78125.0f;
      cfg->f0 = 16000.0f; cfg->f1 = 19000.0f;
      cfg->time_frame = 0.0f; /* one symbol = n samples */
      cfg->carrier = 17500.0f;
      cfg->decim = 8;
      return 0;
    default:
      return -EINVAL;
  }
}

static int is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

int uco_create(const uc_config* cfg, uco_ctx** out) {
  if (!cfg || !out) return -EINVAL;
  if (!is_pow2(cfg->n) || cfg->n < 64 || cfg->n > 65536) return -EINVAL;
  if (cfg->variant < 0 || cfg->variant >= UC_NUM_VARIANTS) return -EINVAL;
  if (!(cfg->fs > 0.0f)) return -EINVAL;
  uco_ctx* c = (uco_ctx*)calloc(1, sizeof(uco_ctx));
  if (!c) return -ENOMEM;
  c->cfg = *cfg;
  uint32_t n = c->n = cfg->n;
  int libm = (cfg->flags & UC_FLAG_LIBM_TRIG) ? 1 : 0;
  float tf = cfg->time_frame > 0.0f ? cfg->time_frame : (float)n / cfg->fs;

  /* bandwidth etc.: receiver/Src/main.c:372-374 */
  c->bandwidth = (uint32_t)((cfg->f1 - cfg->f0) * (float)n / cfg->fs);
  c->bandwidth2 = c->bandwidth * 2;
  c->idx_left_zero = n - c->bandwidth2;
  if (cfg->variant == UC_DECHIRP_DOWN) {
    /* bandwidth*8 windows: chirp_compression_freq_domain/Src/main.c:144-147 */
    c->bandwidth2 = c->bandwidth * 8;
    c->idx_left_zero = n - c->bandwidth2;
  }
  if (cfg->variant == UC_IQ && (cfg->flags & UC_FLAG_IQ_BASEBAND)) {
    /* the notebook's maths (simulation/IQ_modulation.ipynb cells 28-31: the dechirped tone sits at DC):
     * windows of `bandwidth` bins either side of DC, searched as receiver/Src/main.c:205-215 does */
    c->bandwidth2 = c->bandwidth;            /* window length used by fill_history() */
    c->idx_left_zero = n - c->bandwidth;
    c->center = 0;
    c->bandwidth4 = 2 * c->bandwidth;
    if (c->bandwidth == 0 || c->bandwidth4 > n / 2) { free(c); return -EINVAL; }
  } else if (cfg->variant == UC_IQ) {
    /* iq_modulation/Src/main.c:215-219 */
    c->center = (uint32_t)((cfg->f0 + cfg->f1) * (float)n / cfg->fs);
    c->bandwidth4 = c->bandwidth * 4;
    c->idx_left_zero = c->center - c->bandwidth2;
    if (c->center + c->bandwidth2 > n / 2 || c->center < c->bandwidth2) {
      free(c);
      return -EINVAL;
    }
  } else if (c->bandwidth2 == 0 || c->bandwidth2 > n / 2) {
    free(c);
    return -EINVAL;
  }

  if (cfg->variant == UC_STREAM) {
    uint32_t d = cfg->decim ? cfg->decim : 8;
    if (d != 4 && d != 8 && d != 16) { free(c); return -EINVAL; }
    c->decim = d;
    c->tmpl_len = n / d;
  }

  c->hann = (float*)malloc(sizeof(float) * n);
  int cplx = (cfg->variant == UC_SYNC_CPLX || cfg->variant == UC_IQ);
  c->up = (float*)malloc(sizeof(float) * n * (cplx ? 2 : 1));
  c->down = (float*)malloc(sizeof(float) * n * (cplx ? 2 : 1));

  c->tw64 = (double*)malloc(sizeof(double) * n);
  c->tw32 = (float*)malloc(sizeof(float) * n);
  c->tw32h = (float*)malloc(sizeof(float) * n / 2);
  for (uint32_t k = 0; k < n / 2; k++) {
    double a = -2.0 * M_PI * (double)k / (double)n;
    c->tw64[2 * k] = cos(a); c->tw64[2 * k + 1] = sin(a);
    c->tw32[2 * k] = (float)cos(a); c->tw32[2 * k + 1] = (float)sin(a);
  }
  for (uint32_t k = 0; k < n / 4; k++) {
    double a = -2.0 * M_PI * (double)k / (double)(n / 2);
    c->tw32h[2 * k] = (float)cos(a); c->tw32h[2 * k + 1] = (float)sin(a);
  }
  c->rev = make_rev(n);
  c->revh = make_rev(n / 2);

  switch (cfg->variant) {
    case UC_RX_REAL:
      uco_hann_periodic(c->hann, n, libm);
      gen_ref_chirp_deg(c->up, n, 1, cfg->f0, cfg->f1, tf, cfg->fs, cfg->phase_deg, 0, libm);
      gen_ref_chirp_deg(c->down, n, 0, cfg->f0, cfg->f1, tf, cfg->fs, cfg->phase_deg, 0, libm);
      break;
    case UC_SYNC_CPLX:
      uco_hann_periodic(c->hann, n, libm);
      gen_ref_chirp_deg(c->up, n, 1, cfg->f0, cfg->f1, tf, cfg->fs, cfg->phase_deg, 1, libm);
      gen_ref_chirp_deg(c->down, n, 0, cfg->f0, cfg->f1, tf, cfg->fs, cfg->phase_deg, 1, libm);
      break;
    case UC_DECHIRP_DOWN:
      uco_hann_periodic(c->hann, n, libm);
      /* only the down chirp exists: chirp_compression_freq_domain/Src/chirp.c:38-40 */
      gen_ref_chirp_rad(c->up, n, 1, cfg->f0, cfg->f1, cfg->fs, 0.0f, 0, libm);
      gen_ref_chirp_rad(c->down, n, 0, cfg->f0, cfg->f1, cfg->fs, 0.0f, 0, libm);
      break;
    case UC_COMPRESS: {
      /* chirp_compression_time_domain/Src/chirp.c:52-75 */
      hann_symmetric(c->hann, n, libm);
      float phase = (float)(-3.14159265358979f / 2.0);
      gen_ref_chirp_rad(c->up, n, 1, cfg->f0, cfg->f1, cfg->fs, phase, 1, libm);
      gen_ref_chirp_rad(c->down, n, 0, cfg->f0, cfg->f1, cfg->fs, phase, 1, libm);
      c->h_up = (float*)malloc(sizeof(float) * n);
      c->h_down = (float*)malloc(sizeof(float) * n);
      float* tmp = (float*)malloc(sizeof(float) * n);
      for (uint32_t i = 0; i < n; i++) tmp[i] = c->up[i] * c->hann[i];
      rfft32_core(tmp, c->h_up, n, c->tw32, c->tw32h, c->revh);
      for (uint32_t i = 0; i < n; i++) tmp[i] = c->down[i] * c->hann[i];
      rfft32_core(tmp, c->h_down, n, c->tw32, c->tw32h, c->revh);
      free(tmp);
      break;
    }
    case UC_IQ: {
      uco_hann_periodic(c->hann, n, libm);
      /* UC_FLAG_IQ_BASEBAND: the reference chirps are BASE-BAND (IQ_modulation.ipynb cell 3: F0 = -BW/2,
       * F1 = +BW/2 around the carrier), generated by the firmware's own generator */
      float off = (cfg->flags & UC_FLAG_IQ_BASEBAND) ? cfg->carrier : 0.0f;
      gen_ref_chirp_deg(c->up, n, 1, cfg->f0 - off, cfg->f1 - off, tf, cfg->fs, cfg->phase_deg, 1, libm);
      gen_ref_chirp_deg(c->down, n, 0, cfg->f0 - off, cfg->f1 - off, tf, cfg->fs, cfg->phase_deg, 1, libm);
      c->carrier_c = (float*)malloc(sizeof(float) * n);
      c->carrier_s = (float*)malloc(sizeof(float) * n);
      gen_carrier(c->carrier_c, c->carrier_s, n, cfg->carrier, cfg->fs, tf, libm);
      memcpy(c->fir, fir_taps, sizeof(fir_taps));
      break;
    }
    case UC_STREAM: {
      /* base-band template of one n-sample symbol at the decimated rate (include/uchirp.h):
       * symmetric Hann and -pi/2 phase as chirp_compression_time_domain/Src/chirp.c:52-75,
       * sweep f1 -> f0 (DOWN) or f0 -> f1 (UP) over T = n/fs, relative to the carrier */
      uint32_t L = c->tmpl_len;
      double fsd = (double)cfg->fs, T = (double)n / fsd;
      double k = ((double)cfg->f1 - (double)cfg->f0) / T;
      int up = (cfg->flags & UC_FLAG_STREAM_UP) != 0;
      c->tmpl64 = (double*)malloc(sizeof(double) * 2 * L);
      c->tmpl32 = (float*)malloc(sizeof(float) * 2 * L);
      for (uint32_t i = 0; i < L; i++) {
        double t = (double)i * (double)c->decim / fsd;
        double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)(L - 1));
        double ph = up ? 2.0 * M_PI * (((double)cfg->f0 - (double)cfg->carrier) * t + 0.5 * k * t * t)
                       : 2.0 * M_PI * (((double)cfg->f1 - (double)cfg->carrier) * t - 0.5 * k * t * t);
        ph -= M_PI / 2.0;
        c->tmpl64[2 * i] = w * cos(ph);
        c->tmpl64[2 * i + 1] = w * sin(ph);
        c->tmpl32[2 * i] = (float)c->tmpl64[2 * i];
        c->tmpl32[2 * i + 1] = (float)c->tmpl64[2 * i + 1];
      }
      uco_hann_periodic(c->hann, n, libm);
      memcpy(c->fir, fir_taps, sizeof(fir_taps));
      break;
    }
  }
  *out = c;
  return 0;
}

void uco_destroy(uco_ctx* c) {
  if (!c) return;
  free(c->up); free(c->down); free(c->hann); free(c->h_up); free(c->h_down);
  free(c->carrier_c); free(c->carrier_s);
  free(c->tmpl64); free(c->tmpl32);
  free(c->tw64); free(c->tw32); free(c->tw32h); free(c->rev); free(c->revh);
  free(c);
}

static int iq_baseband(const uco_ctx* c) {
  return c->cfg.variant == UC_IQ && (c->cfg.flags & UC_FLAG_IQ_BASEBAND) != 0;
}

int uco_stats_per_frame(const uco_ctx* c) {
  if (!c) return -EINVAL;
  return (c->cfg.variant == UC_RX_REAL || c->cfg.variant == UC_SYNC_CPLX || iq_baseband(c)) ? 2 : 1;
}

int uco_get_windows(const uco_ctx* c, uint32_t* bw, uint32_t* bw2, uint32_t* ilz) {
  if (!c) return -EINVAL;
  if (bw) *bw = c->bandwidth;
  if (bw2) *bw2 = c->bandwidth2;
  if (ilz) *ilz = c->idx_left_zero;
  return 0;
}

int uco_get_table(const uco_ctx* c, int id, float* out, size_t cap) {
  if (!c || !out) return -EINVAL;
  const float* src = NULL;
  size_t cnt = 0;
  int cplx = (c->cfg.variant == UC_SYNC_CPLX || c->cfg.variant == UC_IQ);
  switch (id) {
    case UC_TABLE_UP: src = c->up; cnt = c->n * (cplx ? 2 : 1); break;
    case UC_TABLE_DOWN: src = c->down; cnt = c->n * (cplx ? 2 : 1); break;
    case UC_TABLE_HANN: src = c->hann; cnt = c->n; break;
    case UC_TABLE_H_UP: src = c->h_up; cnt = c->n; break;
    case UC_TABLE_H_DOWN: src = c->h_down; cnt = c->n; break;
    case UC_TABLE_CARRIER_C: src = c->carrier_c; cnt = c->n; break;
    case UC_TABLE_CARRIER_S: src = c->carrier_s; cnt = c->n; break;
    case UC_TABLE_FIR:
      if (c->cfg.variant == UC_IQ || c->cfg.variant == UC_STREAM) { src = c->fir; cnt = UCO_FIR_TAPS; }
      break;
    case UC_TABLE_TEMPLATE: src = c->tmpl32; cnt = 2 * (size_t)c->tmpl_len; break;
    default: return -EINVAL;
  }
  if (!src) return -ENOENT;
  if (cap < cnt) return -ENOSPC;
  memcpy(out, src, sizeof(float) * cnt);
  return (int)cnt;
}

/* the firmware's tables are plain globals (receiver/Src/chirp.c:13-14, main.c:99): a host may fill them itself */
int uco_set_table(uco_ctx* c, int id, const float* data, size_t count) {
  if (!c || !data) return -EINVAL;
  if (c->cfg.variant != UC_RX_REAL && c->cfg.variant != UC_SYNC_CPLX && c->cfg.variant != UC_DECHIRP_DOWN) return -ENOTSUP;
  int cplx = (c->cfg.variant == UC_SYNC_CPLX);
  float* dst = NULL;
  size_t cnt = 0;
  switch (id) {
    case UC_TABLE_UP: dst = c->up; cnt = c->n * (cplx ? 2 : 1); break;
    case UC_TABLE_DOWN: dst = c->down; cnt = c->n * (cplx ? 2 : 1); break;
    case UC_TABLE_HANN: dst = c->hann; cnt = c->n; break;
    default: return -EINVAL;
  }
  if (!dst) return -ENOENT;
  if (count != cnt) return -EINVAL;
  memcpy(dst, data, sizeof(float) * cnt);
  return 0;
}

/* idx2freq: receiver/Src/main.c:154-160 -- integer arithmetic on (int32_t)fs */
static int32_t idx2freq_n(float fs, uint32_t n, uint32_t idx) {
  uint32_t ifs = (uint32_t)(int32_t)fs;
  if (idx < n / 2) return (int32_t)(ifs * idx / n);
  return (int32_t)((ifs * (n - idx) / n) * (uint32_t)-1);
}

int32_t uco_idx2freq(const uco_ctx* c, uint32_t idx) {
  if (!c) return 0;
  if (c->cfg.variant == UC_IQ && !iq_baseband(c)) /* iq_modulation/Src/main.c:112-114: float fs * idx / n */
    return (int32_t)(uint32_t)(c->cfg.fs * (float)idx / (float)c->n);
  return idx2freq_n(c->cfg.fs, c->n, idx);
}

/* ------------------------------------------------------------------------- */
/* per-frame pipelines                                                        */
/* ------------------------------------------------------------------------- */

typedef struct {
  float* y;    /* n   : windowed real frame                     */
  float* pk;   /* n   : packed RFFT scratch (float32 path)      */
  float* c32;  /* 2n  : complex float32 scratch                 */
  double* c64; /* 2n  : complex float64 scratch                 */
  float* mag;  /* n   : float32 magnitudes, reference index space */
  double* mag64; /* n */
} scratch;

static int scratch_alloc(scratch* s, uint32_t n) {
  s->y = (float*)malloc(sizeof(float) * (n + 64));
  s->pk = (float*)malloc(sizeof(float) * n);
  s->c32 = (float*)malloc(sizeof(float) * 2 * n);
  s->c64 = (double*)malloc(sizeof(double) * 2 * n);
  s->mag = (float*)malloc(sizeof(float) * n);
  s->mag64 = (double*)malloc(sizeof(double) * n);
  return (s->y && s->pk && s->c32 && s->c64 && s->mag && s->mag64) ? 0 : -ENOMEM;
}

static void scratch_free(scratch* s) {
  free(s->y); free(s->pk); free(s->c32); free(s->c64); free(s->mag); free(s->mag64);
}

static inline float load_sample(const void* frame, int dtype, size_t i) {
  /* ISR ingest cast: fifo_queue[2N+i] = (float) buf[i], receiver/Src/main.c:664 */
  if (dtype == UC_DTYPE_I32) return (float)((const int32_t*)frame)[i];
  return ((const float*)frame)[i];
}

/* magnitudes of a REAL windowed frame y[n] in the index space of the
 * reference's time_frame[] after pipeline() (receiver/Src/main.c:163-180):
 *   i < n/2 : |X[i]| ; i = 0 is hypot(Re X0, Re X(n/2)) unless UC_FLAG_TRUE_DC (Q2)
 *   i >= n/2: Q1 fix -- Hermitian mirror |X[n-i]| (the reference reads
 *             uninitialised stack there). */
static void real_frame_mags(const uco_ctx* c, scratch* s, int precision) {
  uint32_t n = c->n, h = n / 2;
  int true_dc = (c->cfg.flags & UC_FLAG_TRUE_DC) != 0;
  if (precision == UCO_F32) {
    rfft32_core(s->y, s->pk, n, c->tw32, c->tw32h, c->revh);
    /* arm_cmplx_mag_f32: sqrt(re^2 + im^2) in float32 */
    for (uint32_t i = 1; i < h; i++) {
      float re = s->pk[2 * i], im = s->pk[2 * i + 1];
      s->mag[i] = sqrtf(re * re + im * im);
      s->mag64[i] = (double)s->mag[i];
    }
    float x0 = s->pk[0], xh = s->pk[1];
    s->mag[0] = true_dc ? fabsf(x0) : sqrtf(x0 * x0 + xh * xh);
    s->mag64[0] = (double)s->mag[0];
    s->mag[h] = fabsf(xh);
    s->mag64[h] = (double)s->mag[h];
  } else {
    for (uint32_t i = 0; i < n; i++) { s->c64[2 * i] = (double)s->y[i]; s->c64[2 * i + 1] = 0.0; }
    cfft64(s->c64, n, c->tw64, c->rev, 0);
    for (uint32_t i = 1; i <= h; i++) {
      double re = s->c64[2 * i], im = s->c64[2 * i + 1];
      s->mag64[i] = sqrt(re * re + im * im);
      s->mag[i] = (float)s->mag64[i];
    }
    double x0 = s->c64[0], xh = s->c64[2 * h];
    s->mag64[0] = true_dc ? fabs(x0) : sqrt(x0 * x0 + xh * xh);
    s->mag[0] = (float)s->mag64[0];
  }
  for (uint32_t i = h + 1; i < n; i++) { s->mag[i] = s->mag[n - i]; s->mag64[i] = s->mag64[n - i]; }
}

/* magnitudes of a COMPLEX frame in c32 (interleaved, n points), natural order:
 * arm_cfft_f32 + arm_cmplx_mag_f32, experiments/synchronization/Src/main.c:153-156 */
static void cplx_frame_mags(const uco_ctx* c, scratch* s, int precision, uint32_t nbins) {
  uint32_t n = c->n;
  if (precision == UCO_F32) {
    cfft32(s->c32, n, c->tw32, c->rev, 0);
    for (uint32_t i = 0; i < nbins; i++) {
      float re = s->c32[2 * i], im = s->c32[2 * i + 1];
      s->mag[i] = sqrtf(re * re + im * im);
      s->mag64[i] = (double)s->mag[i];
    }
  } else {
    for (uint32_t i = 0; i < 2 * n; i++) s->c64[i] = (double)s->c32[i];
    cfft64(s->c64, n, c->tw64, c->rev, 0);
    for (uint32_t i = 0; i < nbins; i++) {
      double re = s->c64[2 * i], im = s->c64[2 * i + 1];
      s->mag64[i] = sqrt(re * re + im * im);
      s->mag[i] = (float)s->mag64[i];
    }
  }
}

/* the window search + history fill of dsp(): receiver/Src/main.c:205-229 */
static void fill_history(const uco_ctx* c, const float* mag, float mag_mean, uc_stats* st) {
  float mag_max, mag_max_left, mag_max_right;
  uint32_t max_idx, max_idx_left, max_idx_right;
  uco_arm_max_f32(&mag[c->idx_left_zero], c->bandwidth2, &mag_max_left, &max_idx_left);
  uco_arm_max_f32(&mag[0], c->bandwidth2, &mag_max_right, &max_idx_right);
  if (mag_max_left > mag_max_right) {
    mag_max = mag_max_left;
    max_idx = c->idx_left_zero + max_idx_left;
  } else {
    mag_max = mag_max_right;
    max_idx = max_idx_right;
  }
  st->mag_max = mag_max;
  st->mag_max_left = mag_max_left;
  st->mag_max_right = mag_max_right;
  st->max_freq = idx2freq_n(c->cfg.fs, c->n, max_idx);
  st->max_freq_left = idx2freq_n(c->cfg.fs, c->n, c->idx_left_zero + max_idx_left);
  st->max_freq_right = idx2freq_n(c->cfg.fs, c->n, max_idx_right);
  st->mag_mean = mag_mean;
  st->snr = (mag_max - mag_mean) / mag_mean;
}

/* UC_RX_REAL, one chirp: mult_ref_chirp + Hann (receiver/Src/chirp.c:47-53,
 * receiver/Src/main.c:168-171: two float32 roundings, in this order) */
static void rx_real_one(const uco_ctx* c, scratch* s, const void* frame, int dtype,
                        int updown, int precision) {
  const float* ref = (updown == UC_UP_CHIRP) ? c->up : c->down;
  for (uint32_t i = 0; i < c->n; i++) {
    float v = load_sample(frame, dtype, i);
    v = v * ref[i];
    v = v * c->hann[i];
    s->y[i] = v;
  }
  real_frame_mags(c, s, precision);
}

/* UC_SYNC_CPLX, one chirp: (x,0) * (cos,sin) then * hann, CFFT, all magnitudes
 * experiments/synchronization/Src/main.c:144-158,175-180, Src/chirp.c:51-57 */
static void sync_cplx_one(const uco_ctx* c, scratch* s, const void* frame, int dtype,
                          int updown, int precision) {
  const float* ref = (updown == UC_UP_CHIRP) ? c->up : c->down;
  for (uint32_t i = 0; i < c->n; i++) {
    float a = load_sample(frame, dtype, i), b = 0.0f;
    float cr = ref[2 * i], ci = ref[2 * i + 1];
    float re = a * cr - b * ci; /* arm_cmplx_mult_cmplx_f32 */
    float im = a * ci + b * cr;
    s->c32[2 * i] = re * c->hann[i]; /* arm_cmplx_mult_real_f32 */
    s->c32[2 * i + 1] = im * c->hann[i];
  }
  cplx_frame_mags(c, s, precision, c->n);
}

/* UC_DECHIRP_DOWN: fft(), experiments/chirp_compression_freq_domain/Src/main.c:113-160
 * (Q6 fix: the left window is the Hermitian mirror instead of raw re/im words) */
static void dechirp_down_one(const uco_ctx* c, scratch* s, const void* frame, int dtype,
                             int precision) {
  for (uint32_t i = 0; i < c->n; i++) {
    float v = load_sample(frame, dtype, i);
    v = v * c->down[i];
    v = v * c->hann[i];
    s->y[i] = v;
  }
  real_frame_mags(c, s, precision);
}

/* UC_COMPRESS: compress_chirp(), experiments/chirp_compression_time_domain/Src/chirp.c:78-83
 * result (n signed reals) left in mag64 / mag.  Q5 fix: DC and Nyquist are
 * multiplied separately (numerically invisible). */
static void compress_one(const uco_ctx* c, scratch* s, const void* frame, int dtype,
                         int precision) {
  uint32_t n = c->n, h = n / 2;
  for (uint32_t i = 0; i < n; i++) s->y[i] = load_sample(frame, dtype, i) * c->hann[i];
  if (precision == UCO_F32) {
    rfft32_core(s->y, s->pk, n, c->tw32, c->tw32h, c->revh);
    const float* H = c->h_down;
    /* unpack to a full Hermitian spectrum, multiply, inverse complex FFT */
    s->c32[0] = s->pk[0] * H[0]; s->c32[1] = 0.0f;
    s->c32[2 * h] = s->pk[1] * H[1]; s->c32[2 * h + 1] = 0.0f;
    for (uint32_t k = 1; k < h; k++) {
      float ar = s->pk[2 * k], ai = s->pk[2 * k + 1], br = H[2 * k], bi = H[2 * k + 1];
      float re = ar * br - ai * bi, im = ar * bi + ai * br;
      s->c32[2 * k] = re; s->c32[2 * k + 1] = im;
      s->c32[2 * (n - k)] = re; s->c32[2 * (n - k) + 1] = -im;
    }
    cfft32(s->c32, n, c->tw32, c->rev, 1);
    for (uint32_t i = 0; i < n; i++) { s->mag[i] = s->c32[2 * i] / (float)n; s->mag64[i] = (double)s->mag[i]; }
  } else {
    const float* H = c->h_down;
    for (uint32_t i = 0; i < n; i++) { s->c64[2 * i] = (double)s->y[i]; s->c64[2 * i + 1] = 0.0; }
    cfft64(s->c64, n, c->tw64, c->rev, 0);
    s->c64[0] *= (double)H[0]; s->c64[1] = 0.0;
    s->c64[2 * h] *= (double)H[1]; s->c64[2 * h + 1] = 0.0;
    for (uint32_t k = 1; k < h; k++) {
      double ar = s->c64[2 * k], ai = s->c64[2 * k + 1], br = (double)H[2 * k], bi = (double)H[2 * k + 1];
      double re = ar * br - ai * bi, im = ar * bi + ai * br;
      s->c64[2 * k] = re; s->c64[2 * k + 1] = im;
      s->c64[2 * (n - k)] = re; s->c64[2 * (n - k) + 1] = -im;
    }
    cfft64(s->c64, n, c->tw64, c->rev, 1);
    for (uint32_t i = 0; i < n; i++) { s->mag64[i] = s->c64[2 * i] / (double)n; s->mag[i] = (float)s->mag64[i]; }
  }
}

/* UC_IQ: iq_demodulation() + dsp(), experiments/iq_modulation/Src/iq_modem.c:55-75,
 * Src/main.c:117-134 (Q7 fix: I + jQ interleaved as the notebook does,
 * simulation/IQ_modulation.ipynb cells 16-31).  `frame` points at sample 0;
 * 26 samples of history sit in front of it (the FIR state the firmware
 * carries across blocks, iq_modem.c:47-48); the history samples were mixed
 * with the tail of the carrier table, as a back-to-back previous block's were. */
static void iq_one(const uco_ctx* c, scratch* s, const void* frame, int dtype, int precision, int updown) {
  uint32_t n = c->n;
  const int halo = UCO_FIR_TAPS - 1;
  const int bb = iq_baseband(c);
  /* mixed samples, index j = i + halo, i in [-halo, n) */
  float* imix = (float*)malloc(sizeof(float) * (n + halo) * 2);
  float* qmix = imix + (n + halo);
  for (int i = -halo; i < (int)n; i++) {
    float x = (dtype == UC_DTYPE_I32) ? (float)((const int32_t*)frame)[i] : ((const float*)frame)[i];
    uint32_t ci = (i < 0) ? (uint32_t)((int)n + i) : (uint32_t)i;
    qmix[i + halo] = x * c->carrier_s[ci]; /* iq_modem.c:60 */
    imix[i + halo] = x * c->carrier_c[ci]; /* iq_modem.c:61 */
  }
  for (uint32_t i = 0; i < n; i++) {
    float fi, fq;
    if (precision == UCO_F32) {
      float ai = 0.0f, aq = 0.0f;
      for (int k = 0; k < UCO_FIR_TAPS; k++) { /* y[i] = sum b[k] x[i-k], arm_fir_f32 */
        ai += c->fir[k] * imix[i + halo - k];
        aq += c->fir[k] * qmix[i + halo - k];
      }
      fi = ai; fq = aq;
    } else {
      double ai = 0.0, aq = 0.0;
      for (int k = 0; k < UCO_FIR_TAPS; k++) {
        ai += (double)c->fir[k] * (double)imix[i + halo - k];
        aq += (double)c->fir[k] * (double)qmix[i + halo - k];
      }
      fi = (float)ai; fq = (float)aq;
    }
    float re, im;
    if (bb) {
      /* R * chirp.conjugate(): IQ_modulation.ipynb cells 29 (up) and 30 (down: conj(up) there, the
       * base-band down chirp of a symmetric band) */
      const float* ref = (updown == UC_UP_CHIRP) ? c->up : c->down;
      float cr = ref[2 * i], ci = ref[2 * i + 1];
      re = fi * cr + fq * ci;
      im = fq * cr - fi * ci;
    } else {
      /* (I + jQ) * down_chirp, iq_modulation/Src/chirp.c:46-48 */
      float cr = c->down[2 * i], ci = c->down[2 * i + 1];
      re = fi * cr - fq * ci;
      im = fi * ci + fq * cr;
    }
    s->c32[2 * i] = re * c->hann[i]; /* iq_modulation/Src/main.c:126,237 */
    s->c32[2 * i + 1] = im * c->hann[i];
  }
  free(imix);
  if (bb) {
    cplx_frame_mags(c, s, precision, n); /* both signs of frequency: the windows straddle DC */
    return;
  }
  cplx_frame_mags(c, s, precision, n / 2);
  for (uint32_t i = n / 2; i < n; i++) { s->mag[i] = 0.0f; s->mag64[i] = 0.0; }
}

static void iq_history(const uco_ctx* c, const float* mag, float mag_mean, uc_stats* st) {
  /* iq_modulation/Src/main.c:283-301 */
  float mag_max, mag_max_left, mag_max_right;
  uint32_t max_idx, max_idx_left, max_idx_right;
  uco_arm_max_f32(&mag[c->idx_left_zero], c->bandwidth4, &mag_max, &max_idx);
  uco_arm_max_f32(&mag[c->idx_left_zero], c->bandwidth2, &mag_max_left, &max_idx_left);
  uco_arm_max_f32(&mag[c->center], c->bandwidth2, &mag_max_right, &max_idx_right);
  st->mag_max = mag_max;
  st->mag_max_left = mag_max_left;
  st->mag_max_right = mag_max_right;
  st->max_freq = uco_idx2freq(c, c->idx_left_zero + max_idx);
  st->max_freq_left = uco_idx2freq(c, c->idx_left_zero + max_idx_left);
  st->max_freq_right = uco_idx2freq(c, c->center + max_idx_right);
  st->mag_mean = mag_mean;
  st->snr = (mag_max - mag_mean) / mag_mean;
}

static void dechirp_history(const uco_ctx* c, const float* mag, float mag_mean, uc_stats* st) {
  /* chirp_compression_freq_domain/Src/main.c:140-158: raw indices, the left one
   * reported as bandwidth*8 - max_idx_left */
  float mag_max_left, mag_max_right;
  uint32_t max_idx_left, max_idx_right;
  uco_arm_max_f32(&mag[0], c->bandwidth2, &mag_max_right, &max_idx_right);
  uco_arm_max_f32(&mag[c->idx_left_zero], c->bandwidth2, &mag_max_left, &max_idx_left);
  st->mag_max_right = mag_max_right;
  st->mag_max_left = mag_max_left;
  st->max_freq_right = (int32_t)max_idx_right;
  st->max_freq_left = (int32_t)(c->bandwidth2 - max_idx_left);
  if (mag_max_left > mag_max_right) { st->mag_max = mag_max_left; st->max_freq = st->max_freq_left; }
  else { st->mag_max = mag_max_right; st->max_freq = st->max_freq_right; }
  st->mag_mean = mag_mean;
  st->snr = (st->mag_max - mag_mean) / mag_mean;
}

static void compress_history(const uco_ctx* c, const float* v, float mag_mean, uc_stats* st) {
  /* arm_max_f32 over n SIGNED reals: chirp_compression_time_domain/Src/main.c:186-189 */
  float mx; uint32_t idx;
  uco_arm_max_f32(v, c->n, &mx, &idx);
  memset(st, 0, sizeof(*st));
  st->mag_max = mx;
  st->mag_max_right = mx;
  st->max_freq = (int32_t)idx;
  st->max_freq_right = (int32_t)idx;
  st->mag_mean = mag_mean;
  st->snr = (mx - mag_mean) / mag_mean;
}

/* symbol decision: receiver/Src/main.c:518-531 */
static uint8_t decide(float snr_up, float snr_down, float thr) {
  if ((snr_up >= thr) || (snr_down >= thr)) {
    if (snr_down > snr_up) return UC_SYM_DOWN;
    return UC_SYM_UP;
  }
  return UC_SYM_NONE;
}

static void process_one(const uco_ctx* c, scratch* s, const void* frame, int dtype,
                        const float* mm, uint8_t* sym, uc_stats* st, int precision) {
  uc_stats tmp[2];
  float mm_up = mm ? mm[0] : c->cfg.mag_mean;
  float mm_dn = mm ? mm[1] : c->cfg.mag_mean;
  switch (c->cfg.variant) {
    case UC_RX_REAL:
    case UC_SYNC_CPLX:
      for (int ud = 1; ud >= 0; ud--) { /* up first, then down: main.c:518-519 */
        if (c->cfg.variant == UC_RX_REAL) rx_real_one(c, s, frame, dtype, ud, precision);
        else sync_cplx_one(c, s, frame, dtype, ud, precision);
        fill_history(c, s->mag, ud == UC_UP_CHIRP ? mm_up : mm_dn, &tmp[ud == UC_UP_CHIRP ? 0 : 1]);
      }
      if (sym) *sym = decide(tmp[0].snr, tmp[1].snr, c->cfg.snr_threshold);
      if (st) { st[0] = tmp[0]; st[1] = tmp[1]; }
      break;
    case UC_DECHIRP_DOWN:
      dechirp_down_one(c, s, frame, dtype, precision);
      dechirp_history(c, s->mag, mm_up, &tmp[0]);
      if (sym) *sym = UC_SYM_NONE;
      if (st) st[0] = tmp[0];
      break;
    case UC_COMPRESS:
      compress_one(c, s, frame, dtype, precision);
      compress_history(c, s->mag, mm_up, &tmp[0]);
      if (sym) *sym = UC_SYM_NONE;
      if (st) st[0] = tmp[0];
      break;
    case UC_IQ:
      if (iq_baseband(c)) {
        for (int ud = 1; ud >= 0; ud--) {
          iq_one(c, s, frame, dtype, precision, ud);
          fill_history(c, s->mag, ud == UC_UP_CHIRP ? mm_up : mm_dn, &tmp[ud == UC_UP_CHIRP ? 0 : 1]);
        }
        if (sym) *sym = decide(tmp[0].snr, tmp[1].snr, c->cfg.snr_threshold);
        if (st) { st[0] = tmp[0]; st[1] = tmp[1]; }
        break;
      }
      iq_one(c, s, frame, dtype, precision, UC_DOWN_CHIRP);
      iq_history(c, s->mag, mm_up, &tmp[0]);
      if (sym) *sym = UC_SYM_NONE;
      if (st) st[0] = tmp[0];
      break;
  }
}

int uco_process_batch(uco_ctx* c, const void* frames, int dtype, size_t n_frames,
                      size_t stride_elems, const float* mag_mean, uint8_t* symbols,
                      uc_stats* stats, int precision, int threads) {
  if (!c || (!frames && n_frames)) return -EINVAL;
  if (c->cfg.variant == UC_STREAM) return -EINVAL; /* uco_process_stream */
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32) return -EINVAL;
  if (precision != UCO_F32 && precision != UCO_F64) return -EINVAL;
  if (stride_elems == 0) stride_elems = c->n;
  int spf = uco_stats_per_frame(c);
  int err = 0;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#else
  threads = 1;
#endif
#pragma omp parallel num_threads(threads)
  {
    scratch s;
    int rc = scratch_alloc(&s, c->n);
    if (rc) {
#pragma omp atomic write
      err = rc;
    }
#pragma omp barrier
    if (!err) {
#pragma omp for schedule(static)
      for (long long f = 0; f < (long long)n_frames; f++) {
        const char* base = (const char*)frames + (size_t)f * stride_elems * 4u;
        process_one(c, &s, base, dtype, mag_mean ? mag_mean + 2 * f : NULL,
                    symbols ? symbols + f : NULL, stats ? stats + (size_t)spf * f : NULL,
                    precision);
      }
    }
    scratch_free(&s);
  }
  return err;
}

int uco_spectrum(uco_ctx* c, const void* frame, int dtype, int precision, double* out) {
  if (!c || !frame || !out) return -EINVAL;
  if (c->cfg.variant == UC_STREAM) return -EINVAL;
  scratch s;
  int rc = scratch_alloc(&s, c->n);
  if (rc) { scratch_free(&s); return rc; }
  uint32_t n = c->n;
  switch (c->cfg.variant) {
    case UC_RX_REAL:
    case UC_SYNC_CPLX:
      for (int ud = 1; ud >= 0; ud--) {
        if (c->cfg.variant == UC_RX_REAL) rx_real_one(c, &s, frame, dtype, ud, precision);
        else sync_cplx_one(c, &s, frame, dtype, ud, precision);
        memcpy(out + (ud == UC_UP_CHIRP ? 0 : n), s.mag64, sizeof(double) * n);
      }
      break;
    case UC_DECHIRP_DOWN:
      dechirp_down_one(c, &s, frame, dtype, precision);
      memcpy(out, s.mag64, sizeof(double) * n);
      break;
    case UC_COMPRESS:
      compress_one(c, &s, frame, dtype, precision);
      memcpy(out, s.mag64, sizeof(double) * n);
      break;
    case UC_IQ:
      if (iq_baseband(c)) {
        for (int ud = 1; ud >= 0; ud--) {
          iq_one(c, &s, frame, dtype, precision, ud);
          memcpy(out + (ud == UC_UP_CHIRP ? 0 : n), s.mag64, sizeof(double) * n);
        }
        break;
      }
      iq_one(c, &s, frame, dtype, precision, UC_DOWN_CHIRP);
      memcpy(out, s.mag64, sizeof(double) * n);
      break;
  }
  scratch_free(&s);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* UC_STREAM: mix + FIR + decimate + linear convolution with the template     */
/* ------------------------------------------------------------------------- */

int uco_stream_geometry(const uco_ctx* c, size_t n_samples, size_t* halo, size_t* n_out,
                        size_t* n_blocks, size_t* hop) {
  if (!c || c->cfg.variant != UC_STREAM) return -EINVAL;
  size_t L = c->tmpl_len, D = c->decim;
  size_t h = (L - 1) * D + (UCO_FIR_TAPS - 1);
  size_t hp = (size_t)c->n - (L - 1);
  size_t no = n_samples > h ? (n_samples - h) / D : 0;
  if (halo) *halo = h;
  if (n_out) *n_out = no;
  if (n_blocks) *n_blocks = (no + hp - 1) / hp;
  if (hop) *hop = hp;
  return 0;
}

/* The definition of include/uchirp.h evaluated directly, in float64 on the float32 samples:
 *   z[p] = sum_k fir[k] x[r-k] exp(-j 2 pi fc (r-k) / fs), r = halo + p D
 *          (mix then low-pass: iq_demodulation(), iq_modulation/Src/iq_modem.c:55-75)
 *   y[q] = sum_i g[i] z[q-i]
 *          (what FFT x H x IFFT of compress_chirp(), chirp_compression_time_domain/Src/chirp.c:78-83,
 *          computes for a filter of support L -- here as a plain time-domain sum, so the
 *          overlap-save bookkeeping of the product is checked against something that has none)
 * The carrier phase is taken from sample 0 of the buffer; |y| does not depend on that origin. */
int uco_process_stream(uco_ctx* c, const void* samples, int dtype, size_t n_samples,
                       float* compressed, uc_peak* peaks, int threads) {
  if (!c || c->cfg.variant != UC_STREAM) return -EINVAL;
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32) return -EINVAL;
  size_t halo, n_out, n_blocks, hop;
  uco_stream_geometry(c, n_samples, &halo, &n_out, &n_blocks, &hop);
  if (n_out == 0) return 0;
  if (!samples) return -EINVAL;
  const size_t L = c->tmpl_len, D = c->decim;
  const size_t nz = n_out + (L - 1); /* z[p], p = -(L-1) .. n_out-1, stored at p + L - 1 */
  double* z = (double*)malloc(sizeof(double) * 2 * nz);
  float* mag = (float*)malloc(sizeof(float) * n_out);
  if (!z || !mag) { free(z); free(mag); return -ENOMEM; }
  const double cyc = (double)c->cfg.carrier / (double)c->cfg.fs; /* cycles per sample */
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#else
  threads = 1;
#endif
#pragma omp parallel for schedule(static) num_threads(threads)
  for (long long pi = 0; pi < (long long)nz; pi++) {
    /* p = pi - (L-1); r = halo + p D = 26 + pi D */
    size_t r = (size_t)(UCO_FIR_TAPS - 1) + (size_t)pi * D;
    double re = 0.0, im = 0.0;
    for (int k = 0; k < UCO_FIR_TAPS; k++) {
      size_t idx = r - (size_t)k;
      double x = (double)load_sample(samples, dtype, idx);
      double ph = cyc * (double)idx;
      ph -= floor(ph);
      double a = -2.0 * M_PI * ph;
      re += (double)c->fir[k] * x * cos(a);
      im += (double)c->fir[k] * x * sin(a);
    }
    z[2 * pi] = re;
    z[2 * pi + 1] = im;
  }
#pragma omp parallel for schedule(static) num_threads(threads)
  for (long long q = 0; q < (long long)n_out; q++) {
    double re = 0.0, im = 0.0;
    const double* zq = z + 2 * ((size_t)q + (L - 1)); /* z[q] */
    for (size_t i = 0; i < L; i++) {
      double gr = c->tmpl64[2 * i], gi = c->tmpl64[2 * i + 1];
      double zr = zq[-2 * (long long)i], zi = zq[-2 * (long long)i + 1];
      re += gr * zr - gi * zi;
      im += gr * zi + gi * zr;
    }
    mag[q] = (float)sqrt(re * re + im * im);
  }
  if (compressed) memcpy(compressed, mag, sizeof(float) * n_out);
  if (peaks) {
    for (size_t b = 0; b < n_blocks; b++) {
      size_t q0 = b * hop, q1 = q0 + hop < n_out ? q0 + hop : n_out;
      float m;
      uint32_t idx;
      uco_arm_max_f32(mag + q0, (uint32_t)(q1 - q0), &m, &idx); /* first maximum wins */
      peaks[b].value = m;
      peaks[b].offset = idx;
    }
  }
  free(z);
  free(mag);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* DFSDM front end: sinc^5 decimate-by-32 of the 1-bit PDM stream              */
/* ------------------------------------------------------------------------- */

/* receiver/Src/dfsdm.c:59-61 (SINC5, Oversampling 32, IntOversampling 1), :69 (clock divider 32:
 * 80 MHz / 32 = 2.5 MHz bit clock -> 78125 words/s), :78 (RightBitShift 2); the 24-bit result sits
 * in bits 31:8 of the data register (agent/ *.raw values are multiples of 256).
 * Hogenauer form: five integrators at the bit rate, decimate, five combs, in MODULAR 64-bit arithmetic: on a constant input the
 * fifth integrator grows like t^5 / 120 and passes 2^63 after ~16 k bits -- as in the hardware the combs undo the wrap exactly
 * (the true result fits 27 bits), so the sums are uint64_t (well-defined wrap; int64_t overflowed: found by tools/sanitize.sh).
 * Bit t of the stream is bit (t & 31) of word t >> 5 (LSB first), 1 -> +1, 0 -> -1.
 * The first 4 words only fill the filter: out[q] is the conversion that ends with word q + 4.
 * The hardware block itself is not in the reference (it is silicon): UNPINNED by the reference;
 * the CPU tests pin this against the direct 156-tap convolution in numpy. */
int uco_dfsdm_sinc5(const uint32_t* pdm, size_t n_words, int32_t* out) {
  if (n_words <= 4) return 0;
  if (!pdm || !out) return -EINVAL;
  uint64_t i1 = 0, i2 = 0, i3 = 0, i4 = 0, i5 = 0;
  uint64_t d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;
  for (size_t w = 0; w < n_words; w++) {
    uint32_t bits = pdm[w];
    for (int b = 0; b < 32; b++) {
      const uint64_t sgn = ((bits >> b) & 1u) ? (uint64_t)1 : ~(uint64_t)0; /* +1 / -1 mod 2^64 */
      i1 += sgn; i2 += i1; i3 += i2; i4 += i3; i5 += i4;
    }
    uint64_t u1 = i5 - d1; d1 = i5;
    uint64_t u2 = u1 - d2; d2 = u1;
    uint64_t u3 = u2 - d3; d3 = u2;
    uint64_t u4 = u3 - d4; d4 = u3;
    uint64_t u5 = u4 - d5; d5 = u4;
    const int64_t c5 = (int64_t)u5; /* |true value| <= 2^25 once the filter is full: the low 64 bits ARE the value */
    if (w >= 4) {
      /* floor shift; the 24-bit register clips the single value +2^23 (all-ones input) */
      int64_t v = c5 >> 2;
      if (v > 8388607) v = 8388607;
      if (v < -8388608) v = -8388608;
      out[w - 4] = (int32_t)(v * 256);
    }
  }
  return 0;
}

/* Test helper (not reference behaviour): second-order delta-sigma modulator, x in [-1, 1] at the bit
 * rate -> packed PDM words (LSB first), the kind of stream a MEMS microphone feeds the DFSDM. */
int uco_pdm_modulate(const float* x, size_t n_bits, uint32_t* words) {
  if (!x || !words || (n_bits & 31)) return -EINVAL;
  double s1 = 0.0, s2 = 0.0;
  for (size_t w = 0; w < n_bits / 32; w++) {
    uint32_t acc = 0;
    for (int b = 0; b < 32; b++) {
      double in = 0.5 * (double)x[32 * w + b]; /* half scale: keeps the loop stable */
      double y = (s2 >= 0.0) ? 1.0 : -1.0;
      s1 += in - y;
      s2 += s1 - y;
      if (y > 0.0) acc |= (1u << b);
    }
    words[w] = acc;
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* the receiver's main loop, literally                                        */
/* ------------------------------------------------------------------------- */

typedef struct {
  float mag_max, mag_mean, snr;
} hist_lite;

typedef struct {
  const uco_ctx* c;
  scratch s;
  float* fifo; /* 3n */
  int precision;
} rx_env;

/* dsp(): receiver/Src/main.c:183-231 on the FIFO */
static void rx_dsp(rx_env* e, uint32_t pos, hist_lite* h, float mag_mean, int updown) {
  uc_stats st;
  if (e->c->cfg.variant == UC_RX_REAL) rx_real_one(e->c, &e->s, e->fifo + pos, UC_DTYPE_F32, updown, e->precision);
  else sync_cplx_one(e->c, &e->s, e->fifo + pos, UC_DTYPE_F32, updown, e->precision);
  fill_history(e->c, e->s.mag, mag_mean, &st);
  h->mag_max = st.mag_max;
  h->mag_mean = mag_mean;
  h->snr = st.snr;
}

/* symbol_snr(): main.c:233-236 */
static float rx_symbol_snr(rx_env* e, uint32_t pos, hist_lite* h, int updown) {
  rx_dsp(e, pos, h, h->mag_mean, updown);
  return h->snr;
}

/* resync(): main.c:243-273.  Q8 fixed: an out-of-range neighbour is not evaluated
 * (the firmware evaluates it first and checks the bound afterwards). */
/* how close two compared quantities are, relative to the larger one (the decision margins of uco_receive_stream_diag) */
static float rel_gap(float a, float b) {
  const float m = fmaxf(fmaxf(fabsf(a), fabsf(b)), 1e-30f);
  return fabsf(a - b) / m;
}

static void rx_resync(rx_env* e, float snr, hist_lite* hist, uint32_t offset, uint32_t* sync_position, int updown, float* margin) {
  const uint32_t n = e->c->n;
  int32_t pos_l = (int32_t)*sync_position - (int32_t)offset;
  int32_t pos_r = (int32_t)*sync_position + (int32_t)offset;
  float snr_l = -INFINITY, snr_r = -INFINITY;
  if (pos_l >= 0) snr_l = rx_symbol_snr(e, (uint32_t)pos_l, &hist[2], updown);
  if (pos_r <= (int32_t)(2 * n)) snr_r = rx_symbol_snr(e, (uint32_t)pos_r, &hist[3], updown);
  if (margin) { /* the three compares below */
    if (isfinite(snr_l)) *margin = fminf(*margin, rel_gap(snr, snr_l));
    if (isfinite(snr_r)) *margin = fminf(*margin, rel_gap(snr, snr_r));
    if (isfinite(snr_l) && isfinite(snr_r)) *margin = fminf(*margin, rel_gap(snr_l, snr_r));
  }
  if ((snr > snr_l) && (snr > snr_r)) {
    /* keep */
  } else {
    if (snr_l >= snr_r) {
      if (pos_l >= 0) *sync_position = (uint32_t)pos_l;
    } else if (snr_l < snr_r) {
      if (pos_r <= (int32_t)(2 * n)) *sync_position = (uint32_t)pos_r;
    }
  }
}

int uco_receive_stream(uco_ctx* c, const void* samples, int dtype, size_t n_samples, int precision,
                       char* text, size_t text_cap, uc_rx_event* trace, size_t trace_cap, size_t* n_trace) {
  return uco_receive_stream_isr(c, samples, dtype, n_samples, NULL, precision, text, text_cap, trace, trace_cap, n_trace);
}

/* busy[b] != 0: the main loop had not cleared `new_pcm_data` when block b arrived, so the ISR's
 * `if (!new_pcm_data && ...)` (receiver/Src/main.c:661) skips the block: no FIFO shift, no pass of the switch. */
int uco_receive_stream_isr(uco_ctx* c, const void* samples, int dtype, size_t n_samples, const uint8_t* busy, int precision,
                           char* text, size_t text_cap, uc_rx_event* trace, size_t trace_cap, size_t* n_trace) {
  return uco_receive_stream_diag(c, samples, dtype, n_samples, busy, precision, text, text_cap, trace, trace_cap, n_trace, NULL);
}

/* The same loop; margin (nullable, trace_cap floats) receives, per processed block, how close the block's CLOSEST decision
 * was to going the other way, as a relative gap (1e30: the block took no decision): the block maximum against
 * (1 + SNR_THRESHOLD) x mag_mean and the two largest of the eight acquisition maxima (main.c:461-475); snr_up / snr_down
 * against the threshold and against each other (main.c:493-505, 521-531); resync()'s three compares (main.c:252-271).
 * Test infrastructure for the many-stream fuzz: a float32 receiver may legitimately differ from this float64 one where
 * -- and only where -- such a gap is within round-off. */
int uco_receive_stream_diag(uco_ctx* c, const void* samples, int dtype, size_t n_samples, const uint8_t* busy, int precision,
                            char* text, size_t text_cap, uc_rx_event* trace, size_t trace_cap, size_t* n_trace, float* margin) {
  if (!c || (!samples && n_samples) || !text || text_cap == 0) return -EINVAL;
  if (c->cfg.variant != UC_RX_REAL && c->cfg.variant != UC_SYNC_CPLX) return -ENOTSUP;
  const uint32_t n = c->n;
  rx_env e;
  e.c = c;
  e.precision = precision;
  if (scratch_alloc(&e.s, n)) { scratch_free(&e.s); return -ENOMEM; }
  e.fifo = (float*)calloc(3 * (size_t)n, sizeof(float));
  if (!e.fifo) { scratch_free(&e.s); return -ENOMEM; }

  /* main()'s locals: receiver/Src/main.c:314-339 (history[] starts zeroed here; the
   * firmware leaves it uninitialised, which only matters before the first full pass) */
  uint32_t max_idx = 0, turn = 0;
  const uint32_t offset = n / 8, shift = n / 4; /* main.c:406-407 */
  hist_lite history[8];
  memset(history, 0, sizeof(history));
  float mag_stat[12];
  for (int i = 0; i < 12; i++) mag_stat[i] = 1E37f;
  float mag_max, mag_mean = 0.0f, mag_max_max;
  uint32_t sync_cnt = 0, sync_position = n / 2;
  float snr, snr_up, snr_down;
  int state = UC_STATE_IDLE;
  unsigned char msg = 0;
  int msg_cnt = 0;
  size_t nt = 0, ntext = 0;
  const float thr = c->cfg.snr_threshold;

  const size_t n_blocks = n_samples / n;
  for (size_t b = 0; b < n_blocks; b++) {
    /* ISR: main.c:659-668 */
    int new_pcm_data = (busy && busy[b]) ? 1 : 0; /* still set by the previous block: the consumer is late */
    if (new_pcm_data) continue;                   /* main.c:661: the block is dropped */
    memmove(e.fifo, e.fifo + n, sizeof(float) * 2 * n);
    for (uint32_t i = 0; i < n; i++) e.fifo[2 * n + i] = load_sample(samples, dtype, b * (size_t)n + i);

    const int prev_state = state;
    int bit = -1;
    snr_up = snr_down = 0.0f;
    float mg = 1e30f;
    switch (state) {
      case UC_STATE_IDLE: {
        sync_cnt = 0;
        float sum = 0.0f; /* arm_mean_f32(&mag_stat[4], 8, &mag_mean): main.c:431 */
        for (int i = 4; i < 12; i++) sum += mag_stat[i];
        mag_mean = sum / 8.0f;
      }
        /* intentionally no break: main.c:434 */
        /* fall through */
      case UC_STATE_SYNCHRONIZING:
        for (uint32_t i = 0; i < 4; i++) { /* main.c:447-451 */
          sync_position = n / 2 + turn * offset + shift * i;
          rx_dsp(&e, sync_position, &history[i * 2 + turn], mag_mean, UC_UP_CHIRP);
        }
        turn = (turn == 0) ? 1 : 0;
        if (turn == 1) {
          for (int i = 10; i >= 0; i--) mag_stat[i + 1] = mag_stat[i];
          mag_max_max = 0.0f;
          for (int i = 0; i < 8; i++) {
            mag_max = history[i].mag_max;
            if (mag_max > mag_max_max) { mag_max_max = mag_max; max_idx = (uint32_t)i; }
          }
          mag_stat[0] = mag_max_max;
          snr = (mag_max_max - mag_mean) / mag_mean;
          {
            float second = 0.0f;
            for (int i = 0; i < 8; i++)
              if ((uint32_t)i != max_idx && history[i].mag_max > second) second = history[i].mag_max;
            mg = fminf(rel_gap(mag_max_max, (1.0f + thr) * mag_mean), rel_gap(mag_max_max, second));
          }
          if (snr >= thr) {
            state = UC_STATE_SYNCHRONIZING;
            if (++sync_cnt >= 3) {
              state = UC_STATE_SYNCHRONIZED;
              sync_position = n / 2 + max_idx * offset;
            }
          } else {
            state = UC_STATE_IDLE;
          }
        }
        break;
      case UC_STATE_SYNCHRONIZED: /* main.c:491-510 */
        snr_up = rx_symbol_snr(&e, sync_position, &history[0], UC_UP_CHIRP);
        snr_down = rx_symbol_snr(&e, sync_position, &history[1], UC_DOWN_CHIRP);
        mg = fminf(fminf(rel_gap(snr_up, thr), rel_gap(snr_down, thr)), rel_gap(snr_up, snr_down));
        if ((snr_up >= thr) || (snr_down >= thr)) {
          if (snr_down > snr_up) {
            rx_resync(&e, snr_down, history, offset, &sync_position, UC_DOWN_CHIRP, &mg);
            state = UC_STATE_DATA_RECEIVING;
          } else {
            rx_resync(&e, snr_up, history, offset, &sync_position, UC_UP_CHIRP, &mg);
          }
        } else {
          state = UC_STATE_IDLE;
        }
        break;
      case UC_STATE_DATA_RECEIVING: /* main.c:512-550 */
        snr_up = rx_symbol_snr(&e, sync_position, &history[0], UC_UP_CHIRP);
        snr_down = rx_symbol_snr(&e, sync_position, &history[1], UC_DOWN_CHIRP);
        mg = fminf(fminf(rel_gap(snr_up, thr), rel_gap(snr_down, thr)), rel_gap(snr_up, snr_down));
        if ((snr_up >= thr) || (snr_down >= thr)) {
          if (snr_down > snr_up) {
            bit = 0;
            msg = (unsigned char)((msg << 1) + 0);
            rx_resync(&e, snr_down, history, offset, &sync_position, UC_DOWN_CHIRP, &mg);
          } else {
            bit = 1;
            msg = (unsigned char)((msg << 1) + 1);
            rx_resync(&e, snr_up, history, offset, &sync_position, UC_UP_CHIRP, &mg);
          }
          if (++msg_cnt >= 8) {
            if (ntext + 1 < text_cap) text[ntext++] = (char)msg;
            msg = 0;
            msg_cnt = 0;
          }
        } else {
          if (ntext + 1 < text_cap) text[ntext++] = '\n';
          state = UC_STATE_IDLE;
          msg = 0;
          msg_cnt = 0;
        }
        break;
    }
    if (trace && nt < trace_cap) {
      uc_rx_event* ev = &trace[nt];
      ev->block = (uint32_t)b;
      ev->sync_position = sync_position;
      ev->state_before = (uint8_t)prev_state;
      ev->state_after = (uint8_t)state;
      ev->bit = (int8_t)bit;
      ev->reserved = 0;
      ev->snr_up = snr_up;
      ev->snr_down = snr_down;
      if (margin) margin[nt] = mg;
    }
    nt++;
  }
  text[ntext] = '\0';
  if (n_trace) *n_trace = nt;
  free(e.fifo);
  scratch_free(&e.s);
  return (int)ntext;
}
