// uc_tables.hpp -- host-side construction of the reference tables (init only).
//
// Follows the firmware's init arithmetic rounding by rounding:
//   init_ref_chirp / generate_ref_chirp   receiver/Src/chirp.c:16-45
//   Hann window                           receiver/Src/main.c:99,390-393
//   variants: experiments/synchronization/Src/chirp.c:16-49,
//             experiments/chirp_compression_time_domain/Src/chirp.c:13-75,
//             experiments/chirp_compression_freq_domain/Src/chirp.c:15-40,
//             experiments/iq_modulation/Src/{chirp.c:16-44,iq_modem.c:16-50}
// The CMSIS-DSP V1.4.5 trig primitives those lines call (arm_cos_f32,
// arm_sin_cos_f32: 512-entry table + interpolation) are restated from the
// published algorithm; UC_FLAG_LIBM_TRIG swaps in exact sin/cos.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/uchirp.h"

namespace uc {

constexpr int kFirTaps = 27;

struct Tables {
  uint32_t n = 0;
  uint32_t bandwidth = 0, bandwidth2 = 0, idx_left_zero = 0;
  uint32_t center = 0, bandwidth4 = 0;  // IQ
  bool complex_ref = false;
  std::vector<float> up, down;  // n, or 2n interleaved (cos, sin)
  std::vector<float> hann;      // n
  std::vector<float> h_up, h_down;            // COMPRESS: packed RFFT of hann*chirp
  std::vector<float> carrier_c, carrier_s;    // IQ
  std::vector<float> fir;                     // IQ, 27 taps
};

// UC_STREAM (include/uchirp.h): everything the overlap-save kernel reads
struct StreamTables {
  uint32_t decim = 0, tmpl_len = 0, hop = 0, halo = 0;
  std::vector<float> tmpl;  // L complex: template g
  std::vector<float> hn;    // n complex: FFT_n(g zero-padded) / n
  std::vector<float> fir;   // 27 real taps (iq_modulation/Src/iq_modem.c:18)
  std::vector<float> ctap;  // 27 complex: fir[k] e^{+j 2 pi carrier k / fs}
  std::vector<float> rot;   // n complex: e^{-j 2 pi carrier D i / fs}
};
int build_stream_tables(const uc_config& cfg, StreamTables& out);

// sinc^5 / 32 of a +-1 bit stream as per-byte lookup tables (uc_cic_kernel.hip):
// t4[(b*256+v)*4 + w] (w < 4) and t1[b*256+v] (w = 4) = sum_k bit_k(v) * h[32 w + 31 - (8 b + k)],
// h = the 156-tap five-fold convolution of a 32-sample boxcar
void build_sinc5_tables(std::vector<int32_t>& t4, std::vector<int32_t>& t1);

// returns 0 or a negative errno
int build_tables(const uc_config& cfg, Tables& out);

// exp(-2 pi i k / n) for k < n, interleaved (re, im), computed in double
void build_twiddles(uint32_t n, std::vector<float>& out);

// arm_rfft_fast_f32-layout forward real FFT evaluated in double (init only)
void packed_rfft_double(const std::vector<float>& in, std::vector<float>& packed);

}  // namespace uc
