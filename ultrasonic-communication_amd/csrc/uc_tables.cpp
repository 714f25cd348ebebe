// uc_tables.cpp -- see uc_tables.hpp.
#include "uc_tables.hpp"

#include <errno.h>

#include <cmath>
#include <complex>
#include <utility>

namespace uc {

namespace {

constexpr int kLutSize = 512;  // FAST_MATH_TABLE_SIZE, CMSIS/Include/arm_math.h:341
constexpr double kPi = 3.14159265358979323846;
constexpr float kPiF = 3.14159265358979f;  // the float macro PI, arm_math.h:334

// sinTable_f32: sin(2 pi k / 512), k = 0..512 (arm_common_tables.h:132)
struct SinLut {
  float v[kLutSize + 1];
  SinLut() {
    for (int k = 0; k <= kLutSize; k++) v[k] = (float)std::sin(2.0 * kPi * k / kLutSize);
  }
};
const SinLut& lut() {
  static const SinLut t;
  return t;
}

// arm_cos_f32 (radians): table lookup, linear interpolation
float lut_cos(float x) {
  const float* tab = lut().v;
  float in = x * 0.159154943092f + 0.25f;
  int32_t whole = (int32_t)in;
  if (in < 0.0f) whole--;
  in -= (float)whole;
  const float findex = (float)kLutSize * in;
  const uint16_t index = ((uint16_t)findex) & 0x1ff;
  const float fract = findex - (float)index;
  return (1.0f - fract) * tab[index] + fract * tab[index + 1];
}

// cubic Hermite segment used by arm_sin_cos_f32: f at both ends, derivative
// (scaled by the table step) from the quarter-turn-shifted table
float hermite(float f1, float f2, float d1, float d2, float fract) {
  const float dn = 0.0122718463030f;
  const float df = f2 - f1;
  float temp = dn * (d1 + d2) - 2 * df;
  temp = fract * temp + (3 * df - (d2 + 2 * d1) * dn);
  temp = fract * temp + d1 * dn;
  return fract * temp + f1;
}

// arm_sin_cos_f32 (DEGREES, arm_math.h:4627-4637)
void lut_sin_cos_deg(float theta, float& s, float& c) {
  const float* tab = lut().v;
  float in = theta * 0.00277777777778f;
  int32_t whole = (int32_t)in;
  if (in < 0.0f) whole--;
  in -= (float)whole;
  const float findex = (float)kLutSize * in;
  const uint16_t is = ((uint16_t)findex) & 0x1ff;
  const uint16_t ic = (is + kLutSize / 4) & 0x1ff;
  const float fract = findex - (float)is;
  c = hermite(tab[ic], tab[ic + 1], -tab[is], -tab[is + 1], fract);
  s = hermite(tab[is], tab[is + 1], tab[ic], tab[ic + 1], fract);
}

struct Trig {
  bool exact;
  float cos_rad(float x) const { return exact ? (float)std::cos((double)x) : lut_cos(x); }
  void sin_cos_deg(float th, float& s, float& c) const {
    if (exact) {
      const double r = (double)th * (kPi / 180.0);
      s = (float)std::sin(r);
      c = (float)std::cos(r);
    } else {
      lut_sin_cos_deg(th, s, c);
    }
  }
};

// 0.5 - 0.5 cos(i * scale); scale = 2 pi / n (periodic) or 2 pi / (n-1) (symmetric)
void hann_window(std::vector<float>& w, uint32_t n, float scale, const Trig& tr) {
  w.resize(n);
  for (uint32_t i = 0; i < n; i++) w[i] = 0.5f - 0.5f * tr.cos_rad((float)i * scale);
}

// receiver-style chirp: theta in degrees, time accumulated in float, frequency
// law f0 + k t / 2 (receiver/Src/chirp.c:16-40)
void chirp_degrees(std::vector<float>& ref, uint32_t n, bool up, float f0, float f1, float time_frame,
                   float fs, float phase, bool complex_out, const Trig& tr) {
  ref.resize(complex_out ? 2 * (size_t)n : n);
  float t = 0.0f;
  const float delta_f = (float)(f1 - f0) / time_frame;
  const float delta_t = time_frame / (time_frame * fs);
  for (uint32_t i = 0; i < n; i++) {
    const float freq = up ? (float)(f0 + delta_f * t / 2.0) : (float)(f1 - delta_f * t / 2.0);
    const float theta = (float)(360.0 * freq * t + phase);
    t = t + delta_t;
    float s, c;
    tr.sin_cos_deg(theta, s, c);
    if (complex_out) {
      ref[2 * (size_t)i] = c * 1.0f;      // AMPLITUDE 1.0f
      ref[2 * (size_t)i + 1] = s * 1.0f;
    } else {
      ref[i] = s * 1.0f;  // Q3: the cos store of chirp.c:37 is overwritten by :38
    }
  }
}

// chirp_compression-style chirp: radians, arm_cos_f32, frequency law f1 + k t
void chirp_radians(std::vector<float>& ref, uint32_t n, bool up, float f1, float f2, float fs,
                   float phase, bool use_phase, const Trig& tr) {
  ref.resize(n);
  float t = 0.0f;
  const float time_frame = (float)n / (float)fs;
  const float delta_f = (f2 - f1) / time_frame;
  const float delta_t = time_frame / (time_frame * (float)fs);
  for (uint32_t i = 0; i < n; i++) {
    const float freq = up ? f1 + delta_f * t : f2 - delta_f * t;
    const float arg = use_phase ? (float)(2.0 * kPiF * freq * t + phase) : (float)(2.0 * kPiF * freq * t);
    t = t + delta_t;
    ref[i] = tr.cos_rad(arg) * 1.0f;
  }
}

const float kFir[kFirTaps] = {
    0.01560757f, 0.02043850f, 0.02535792f, 0.03027307f, 0.03508888f, 0.03971022f, 0.04404423f,
    0.04800257f, 0.05150362f, 0.05447453f, 0.05685299f, 0.05858884f, 0.05964532f, 0.06000000f,
    0.05964532f, 0.05858884f, 0.05685299f, 0.05447453f, 0.05150362f, 0.04800257f, 0.04404423f,
    0.03971022f, 0.03508888f, 0.03027307f, 0.02535792f, 0.02043850f, 0.01560757f};

bool pow2(uint32_t v) { return v && !(v & (v - 1)); }

}  // namespace

void build_twiddles(uint32_t n, std::vector<float>& out) {
  out.resize(2 * (size_t)n);
  for (uint32_t k = 0; k < n; k++) {
    const double a = -2.0 * kPi * (double)k / (double)n;
    out[2 * (size_t)k] = (float)std::cos(a);
    out[2 * (size_t)k + 1] = (float)std::sin(a);
  }
}

// in-place forward DFT, iterative radix-2, double precision (init time only)
static void fft_double(std::vector<std::complex<double>>& a) {
  const size_t n = a.size();
  size_t bits = 0;
  while (((size_t)1 << bits) < n) bits++;
  for (size_t i = 0; i < n; i++) {
    size_t r = 0;
    for (size_t b = 0; b < bits; b++)
      if (i & ((size_t)1 << b)) r |= (size_t)1 << (bits - 1 - b);
    if (r > i) std::swap(a[i], a[r]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = -2.0 * kPi / (double)len;
    for (size_t base = 0; base < n; base += len) {
      for (size_t k = 0; k < len / 2; k++) {
        const std::complex<double> w(std::cos(ang * (double)k), std::sin(ang * (double)k));
        const std::complex<double> u = a[base + k], v = a[base + k + len / 2] * w;
        a[base + k] = u + v;
        a[base + k + len / 2] = u - v;
      }
    }
  }
}

void packed_rfft_double(const std::vector<float>& in, std::vector<float>& packed) {
  const size_t n = in.size();
  std::vector<std::complex<double>> a(n);
  for (size_t i = 0; i < n; i++) a[i] = std::complex<double>((double)in[i], 0.0);
  fft_double(a);
  packed.resize(n);
  packed[0] = (float)a[0].real();
  packed[1] = (float)a[n / 2].real();
  for (size_t k = 1; k < n / 2; k++) {
    packed[2 * k] = (float)a[k].real();
    packed[2 * k + 1] = (float)a[k].imag();
  }
}

void build_sinc5_tables(std::vector<int32_t>& t4, std::vector<int32_t>& t1) {
  std::vector<long long> h(1, 1);
  for (int rep = 0; rep < 5; rep++) {  // h <- h * boxcar(32)
    std::vector<long long> o(h.size() + 31, 0);
    for (size_t i = 0; i < h.size(); i++)
      for (int k = 0; k < 32; k++) o[i + k] += h[i];
    h.swap(o);
  }  // 156 taps, sum 2^25
  t4.assign(4 * 256 * 4, 0);
  t1.assign(4 * 256, 0);
  for (int w = 0; w < 5; w++)
    for (int b = 0; b < 4; b++)
      for (int v = 0; v < 256; v++) {
        long long acc = 0;
        for (int k = 0; k < 8; k++) {
          const int j = 32 * w + 31 - (8 * b + k);
          if (((v >> k) & 1) && j >= 0 && j < (int)h.size()) acc += h[(size_t)j];
        }
        if (w < 4) t4[(size_t)(b * 256 + v) * 4 + w] = (int32_t)acc;
        else t1[(size_t)(b * 256 + v)] = (int32_t)acc;
      }
}

int build_stream_tables(const uc_config& cfg, StreamTables& out) {
  const uint32_t n = cfg.n;
  const uint32_t D = cfg.decim ? cfg.decim : 8;
  if (D != 4 && D != 8 && D != 16) return -EINVAL;
  if (!(cfg.fs > 0.0f) || !pow2(n)) return -EINVAL;
  const uint32_t L = n / D;
  out = StreamTables();
  out.decim = D;
  out.tmpl_len = L;
  out.hop = n - (L - 1);
  out.halo = (L - 1) * D + (kFirTaps - 1);
  const double fs = (double)cfg.fs, T = (double)n / fs;
  const double k = ((double)cfg.f1 - (double)cfg.f0) / T;
  const bool up = (cfg.flags & UC_FLAG_STREAM_UP) != 0;
  // template g: one symbol at base band, decimated rate (include/uchirp.h, UC_STREAM)
  std::vector<std::complex<double>> g(n, std::complex<double>(0.0, 0.0));
  out.tmpl.resize(2 * (size_t)L);
  for (uint32_t i = 0; i < L; i++) {
    const double t = (double)i * (double)D / fs;
    const double w = 0.5 - 0.5 * std::cos(2.0 * kPi * (double)i / (double)(L - 1));
    double ph = up ? 2.0 * kPi * (((double)cfg.f0 - (double)cfg.carrier) * t + 0.5 * k * t * t)
                   : 2.0 * kPi * (((double)cfg.f1 - (double)cfg.carrier) * t - 0.5 * k * t * t);
    ph -= kPi / 2.0;
    g[i] = std::complex<double>(w * std::cos(ph), w * std::sin(ph));
    out.tmpl[2 * (size_t)i] = (float)g[i].real();
    out.tmpl[2 * (size_t)i + 1] = (float)g[i].imag();
  }
  // H/n: spectrum of the zero-padded template with the inverse transform's 1/n folded in
  fft_double(g);
  out.hn.resize(2 * (size_t)n);
  for (uint32_t i = 0; i < n; i++) {
    out.hn[2 * (size_t)i] = (float)(g[i].real() / (double)n);
    out.hn[2 * (size_t)i + 1] = (float)(g[i].imag() / (double)n);
  }
  // the carrier folded into the taps: sum_k fir[k] x[r-k] e^{-jw(r-k)} = e^{-jwr} sum_k (fir[k] e^{jwk}) x[r-k]
  const double cyc = (double)cfg.carrier / fs;  // cycles per input sample
  out.fir.assign(kFir, kFir + kFirTaps);
  out.ctap.resize(2 * (size_t)kFirTaps);
  for (int t = 0; t < kFirTaps; t++) {
    double ph = cyc * (double)t;
    ph -= std::floor(ph);
    out.ctap[2 * t] = (float)((double)kFir[t] * std::cos(2.0 * kPi * ph));
    out.ctap[2 * t + 1] = (float)((double)kFir[t] * std::sin(2.0 * kPi * ph));
  }
  // e^{-jw D i}, i < n: rotation of decimated sample i relative to the first one of its block
  out.rot.resize(2 * (size_t)n);
  for (uint32_t i = 0; i < n; i++) {
    double ph = cyc * (double)D * (double)i;
    ph -= std::floor(ph);
    out.rot[2 * (size_t)i] = (float)std::cos(-2.0 * kPi * ph);
    out.rot[2 * (size_t)i + 1] = (float)std::sin(-2.0 * kPi * ph);
  }
  return 0;
}

int build_tables(const uc_config& cfg, Tables& out) {
  if (!pow2(cfg.n) || cfg.n < 64 || cfg.n > 65536) return -EINVAL;
  if (cfg.variant < 0 || cfg.variant >= UC_NUM_VARIANTS) return -EINVAL;
  if (!(cfg.fs > 0.0f)) return -EINVAL;
  const uint32_t n = cfg.n;
  if (cfg.variant == UC_STREAM) {  // no per-frame tables: see build_stream_tables
    out = Tables();
    out.n = n;
    out.bandwidth = (uint32_t)((cfg.f1 - cfg.f0) * (float)n / cfg.fs);
    out.bandwidth2 = out.bandwidth * 2;
    out.idx_left_zero = n - out.bandwidth2;
    out.fir.assign(kFir, kFir + kFirTaps);
    return 0;
  }
  const Trig tr{(cfg.flags & UC_FLAG_LIBM_TRIG) != 0};
  const float tf = cfg.time_frame > 0.0f ? cfg.time_frame : (float)n / cfg.fs;

  out = Tables();
  out.n = n;
  // receiver/Src/main.c:372-374
  out.bandwidth = (uint32_t)((cfg.f1 - cfg.f0) * (float)n / cfg.fs);
  out.bandwidth2 = out.bandwidth * 2;
  out.idx_left_zero = n - out.bandwidth2;
  if (cfg.variant == UC_DECHIRP_DOWN) {
    // chirp_compression_freq_domain/Src/main.c:144-147 (bandwidth*8)
    out.bandwidth2 = out.bandwidth * 8;
    out.idx_left_zero = n - out.bandwidth2;
  }
  const bool iq_bb = cfg.variant == UC_IQ && (cfg.flags & UC_FLAG_IQ_BASEBAND) != 0;
  if (iq_bb) {
    // simulation/IQ_modulation.ipynb cells 28-31: the dechirped tone sits at DC; windows of `bandwidth` bins
    // either side of it, searched as receiver/Src/main.c:205-215 searches its two
    out.bandwidth2 = out.bandwidth;  // the window length
    out.idx_left_zero = n - out.bandwidth;
    out.center = 0;
    out.bandwidth4 = 2 * out.bandwidth;
    if (out.bandwidth == 0 || out.bandwidth4 > n / 2) return -EINVAL;
  } else if (cfg.variant == UC_IQ) {
    // iq_modulation/Src/main.c:215-219
    out.center = (uint32_t)((cfg.f0 + cfg.f1) * (float)n / cfg.fs);
    out.bandwidth4 = out.bandwidth * 4;
    out.idx_left_zero = out.center - out.bandwidth2;
    if (out.center < out.bandwidth2 || out.center + out.bandwidth2 > n / 2) return -EINVAL;
  } else if (out.bandwidth2 == 0 || out.bandwidth2 > n / 2) {
    return -EINVAL;
  }

  const float periodic = (float)(2.0f * kPi / (float)n);  // `const float WINDOW_SCALE = 2.0f * M_PI / (float) NN`
  out.complex_ref = (cfg.variant == UC_SYNC_CPLX || cfg.variant == UC_IQ);
  switch (cfg.variant) {
    case UC_RX_REAL:
    case UC_SYNC_CPLX:
      hann_window(out.hann, n, periodic, tr);
      chirp_degrees(out.up, n, true, cfg.f0, cfg.f1, tf, cfg.fs, cfg.phase_deg, out.complex_ref, tr);
      chirp_degrees(out.down, n, false, cfg.f0, cfg.f1, tf, cfg.fs, cfg.phase_deg, out.complex_ref, tr);
      break;
    case UC_DECHIRP_DOWN:
      hann_window(out.hann, n, periodic, tr);
      chirp_radians(out.up, n, true, cfg.f0, cfg.f1, cfg.fs, 0.0f, false, tr);
      chirp_radians(out.down, n, false, cfg.f0, cfg.f1, cfg.fs, 0.0f, false, tr);
      break;
    case UC_COMPRESS: {
      hann_window(out.hann, n, 2.0f * kPiF / (float)(n - 1), tr);
      const float phase = (float)(-kPiF / 2.0);
      chirp_radians(out.up, n, true, cfg.f0, cfg.f1, cfg.fs, phase, true, tr);
      chirp_radians(out.down, n, false, cfg.f0, cfg.f1, cfg.fs, phase, true, tr);
      std::vector<float> tmp(n);
      for (uint32_t i = 0; i < n; i++) tmp[i] = out.up[i] * out.hann[i];
      packed_rfft_double(tmp, out.h_up);
      for (uint32_t i = 0; i < n; i++) tmp[i] = out.down[i] * out.hann[i];
      packed_rfft_double(tmp, out.h_down);
      break;
    }
    case UC_IQ: {
      hann_window(out.hann, n, periodic, tr);
      // UC_FLAG_IQ_BASEBAND: BASE-BAND reference chirps (IQ_modulation.ipynb cell 3: F0 = -BW/2, F1 = +BW/2 around
      // the carrier) from the firmware's own generator
      const float off = iq_bb ? cfg.carrier : 0.0f;
      chirp_degrees(out.up, n, true, cfg.f0 - off, cfg.f1 - off, tf, cfg.fs, cfg.phase_deg, true, tr);
      chirp_degrees(out.down, n, false, cfg.f0 - off, cfg.f1 - off, tf, cfg.fs, cfg.phase_deg, true, tr);
      // init_iq_modem: iq_modulation/Src/iq_modem.c:34-45
      out.carrier_c.resize(n);
      out.carrier_s.resize(n);
      float t = 0.0f;
      const float delta_t = tf / (tf * cfg.fs);
      for (uint32_t i = 0; i < n; i++) {
        const float theta = (float)(360.0 * cfg.carrier * t);
        tr.sin_cos_deg(theta, out.carrier_s[i], out.carrier_c[i]);
        t = t + delta_t;
      }
      out.fir.assign(kFir, kFir + kFirTaps);
      break;
    }
  }
  return 0;
}

}  // namespace uc
