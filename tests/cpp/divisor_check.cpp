// divisor_check.cpp -- uc::rows_divisor (csrc/uc_kernels.hpp): the (magic, shift) pair the ROWS build of the band kernel and
// the multi-stream DFSDM kernel use for frame -> (stream, block) and tile -> (stream, tile): floor(u / d) must equal
// (mulhi(u, magic) + u) >> shift for EVERY u < 2^31.  Checked exhaustively near every multiple boundary and on a stride sweep.
#include <cstdint>
#include <cstdio>

#include "../../ultrasonic-communication_amd/csrc/uc_kernels.hpp"

static uint32_t divide(uint32_t u, uint32_t magic, uint32_t shift) {
  const uint32_t hi = (uint32_t)(((uint64_t)u * magic) >> 32);
  return (hi + u) >> shift;
}

int main() {
  const uint32_t top = 0x7fffffffu;
  uint64_t checked = 0;
  uint32_t ds[4200];
  int nd = 0;
  for (uint32_t d = 1; d <= 4100; d++) ds[nd++] = d;
  const uint32_t big[] = {65535u, 65536u, 65537u, 1000003u, (1u << 24) - 1, 1u << 24, (1u << 24) + 1, 123456789u, (1u << 28) - 1, 1u << 28};
  for (uint32_t d : big) ds[nd++] = d;
  for (int i = 0; i < nd; i++) {
    const uint32_t d = ds[i];
    uint32_t magic = 0, shift = 0;
    uc::rows_divisor(d, &magic, &shift);
    // around every multiple of d (up to 2000 of them, spread over the range) and at the top of the range
    const uint64_t n_mult = (uint64_t)top / d;
    const uint64_t step = n_mult > 2000 ? n_mult / 2000 : 1;
    for (uint64_t k = 0; k <= n_mult; k += step) {
      for (int64_t off = -2; off <= 2; off++) {
        const int64_t u = (int64_t)(k * d) + off;
        if (u < 0 || u > (int64_t)top) continue;
        if (divide((uint32_t)u, magic, shift) != (uint32_t)u / d) {
          printf("FAIL d=%u u=%lld: %u != %u\n", d, (long long)u, divide((uint32_t)u, magic, shift), (uint32_t)u / d);
          return 1;
        }
        checked++;
      }
    }
    for (uint32_t u = top - 4; u >= top - 4 && u <= top; u++) {
      if (divide(u, magic, shift) != u / d) { printf("FAIL d=%u u=%u\n", d, u); return 1; }
      checked++;
      if (u == top) break;
    }
    for (uint64_t u = 0; u <= top; u += 7919 * 13) {
      if (divide((uint32_t)u, magic, shift) != (uint32_t)u / d) { printf("FAIL d=%u u=%llu\n", d, (unsigned long long)u); return 1; }
      checked++;
    }
  }
  printf("rows_divisor ok: %d divisors, %llu quotients\n", nd, (unsigned long long)checked);
  return 0;
}
