#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
// simplified: reads of fixed length RL back to back in `bases`; stream = read, sep, read, sep ...; pos0 = 0
template <int VAR>
__attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi2,popcnt"))) void pack(const uint8_t *bases, uint64_t nreads, uint64_t RL, uint64_t p_hi, uint32_t *code, uint16_t *inv)
{
  const __m512i m3 = _mm512_set1_epi8(3), mdf = _mm512_set1_epi8((char)0xDF), m0f = _mm512_set1_epi8(0x0F), nl = _mm512_set1_epi8('\n');
  const __m512i w1 = _mm512_set1_epi16(0x0104), w2 = _mm512_set1_epi32(0x00010010);
  const __m128i bswap = _mm_setr_epi8(3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 15, 14, 13, 12);
  const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8((char)0xFF, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0));
  const __m512i rev = _mm512_broadcast_i32x4(_mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0));
  uint64_t i = 0, next_sep = RL;
  const uint8_t *src = bases;
  for (uint64_t P = 0; P < p_hi; P += 64) {
    uint64_t sepm = 0;
    if (VAR != 5) while (next_sep < P + 64) { sepm |= 1ULL << (next_sep - P); i++; next_sep = i < nreads ? i * (RL + 1) + RL : ~0ULL; }
    __m512i v;
    if (VAR >= 1 && sepm == 0) {
      v = _mm512_loadu_si512((const void *)src);
      src += 64;
    } else {
      const unsigned nsrc = 64u - (unsigned)__builtin_popcountll(sepm);
      const __m512i x = VAR >= 2 ? _mm512_loadu_si512((const void *)src) : _mm512_maskz_loadu_epi8(nsrc == 64 ? ~0ULL : (1ULL << nsrc) - 1, (const void *)src);
      v = _mm512_mask_expand_epi8(nl, (__mmask64)~sepm, x);
      src += nsrc;
    }
    if (VAR != 4) {
    const __m512i t = _mm512_ternarylogic_epi64(_mm512_srli_epi16(v, 1), _mm512_srli_epi16(v, 2), m3, 0x28);
    const __m512i byt = _mm512_madd_epi16(_mm512_maddubs_epi16(t, w1), w2);
    _mm_storeu_si128((__m128i *)(code + P / 16), _mm_shuffle_epi8(_mm512_cvtepi32_epi8(byt), bswap));
    }
    if (VAR != 3) {
    const __m512i u = _mm512_shuffle_epi8(_mm512_and_si512(v, mdf), rev);
    const uint64_t bad = ~(uint64_t)_mm512_cmpeq_epi8_mask(_mm512_shuffle_epi8(lut, _mm512_and_si512(u, m0f)), u);
    memcpy(inv + P / 16, &bad, 8);
    }
  }
}
int main()
{
  const uint64_t n = 2000000, RL = 150, total = n * (RL + 1) / 64 * 64;
  std::vector<uint8_t> b(n * RL + 256);
  for (size_t i = 0; i < b.size(); i++) b[i] = "ACGT"[(i * 2654435761u >> 13) & 3];
  std::vector<uint32_t> code(total / 16 + 16), c2(total / 16 + 16);
  std::vector<uint16_t> inv(total / 16 + 16), i2(total / 16 + 16);
  for (int var = 0; var < 6; var++)
    for (int rep = 0; rep < 3; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      if (var == 0) pack<0>(b.data(), n, RL, total, code.data(), inv.data());
      if (var == 1) pack<1>(b.data(), n, RL, total, c2.data(), i2.data());
      if (var == 2) pack<2>(b.data(), n, RL, total, c2.data(), i2.data());
      if (var == 3) pack<3>(b.data(), n, RL, total, c2.data(), i2.data());
      if (var == 4) pack<4>(b.data(), n, RL, total, c2.data(), i2.data());
      if (var == 5) pack<5>(b.data(), n, RL, total, c2.data(), i2.data());
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (rep == 2) printf("variant %d: %.2f GB/s of bases%s\n", var, n * RL / dt / 1e9, var && var < 3 && (memcmp(code.data(), c2.data(), total / 16 * 4) || memcmp(inv.data(), i2.data(), total / 16 * 2)) ? "  MISMATCH" : "");
    }
  return 0;
}
