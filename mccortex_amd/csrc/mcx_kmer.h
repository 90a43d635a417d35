// mcx_kmer.h -- k-mer primitives shared by the gfx950 kernels and the host
// helpers of the C ABI.  W = number of 64-bit words (1 for k<=31, 2 for
// 33<=k<=63, 3 for 65<=k<=95, 4 for 97<=k<=127); w[0] is the most significant (partial) word, first base at the
// top, exactly the BinaryKmer layout of the reference
// (src/graph/binary_kmer.h:10-18, :39-45).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MCX_HD __host__ __device__ __forceinline__
#else
#define MCX_HD inline
#endif

namespace mcx {

constexpr uint64_t kFlag = 1ULL << 63;     // slot occupied (BKMER_SET_FLAG, hash_table.h:41-46)
constexpr uint64_t kPending = 1ULL << 62;  // W=2 only: low word not yet published
constexpr uint64_t kKeyMask = ~(kFlag | kPending);

MCX_HD int words_for_k(int k) { return (2 * k + 63) / 64; }

// ASCII -> 2-bit code A=0 C=1 G=2 T=3 (src/basic/dna.c:8-25), branch free:
// bits 2:1 of 'A','C','G','T' are 00,01,11,10; x^(x>>1) maps them to 0,1,2,3.
MCX_HD uint32_t base_code(uint32_t c) { return ((c >> 1) ^ (c >> 2)) & 3u; }
MCX_HD bool base_valid(uint32_t c)
{
  const uint32_t u = c & 0xDFu;  // fold case; bytes >= 0x80 keep bit 7 and fail
  return (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
}

MCX_HD uint64_t bitrev64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bitreverse64(x);  // 2x v_bfrev_b32 + swap
#else
  x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  return __builtin_bswap64(x);
#endif
}

// Reverse the order of the 32 two-bit groups of a word and complement them
// (what binary_kmer.c:112-121 does with bswap + nibble masks): full bit
// reversal, then swap the two bits of every pair back, then NOT.
MCX_HD uint64_t revcomp_word(uint64_t x)
{
  uint64_t r = bitrev64(x);
  r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
  return ~r;
}

template <int W> struct Kmer { uint64_t w[W]; };

// binary_kmer.c:102-133
template <int W> MCX_HD Kmer<W> revcomp(const Kmer<W> &x, int k)
{
  Kmer<W> r;
  const int shift = 64 * W - 2 * k;  // unused high bits of w[0]: 2..62 for the odd k of a W-word build
  if (W == 1) {
    r.w[0] = revcomp_word(x.w[0]) >> shift;
  } else if (W == 2) {
    const uint64_t hi = revcomp_word(x.w[W - 1]);  // becomes most significant
    const uint64_t lo = revcomp_word(x.w[0]);
    r.w[0] = hi >> shift;
    r.w[W - 1] = (lo >> shift) | (hi << (64 - shift));  // 2 <= shift <= 62 for odd k in 33..63
  } else {
    // the words in reverse order, each reverse-complemented, then the 64 W-bit number moved down by `shift`
    uint64_t t[W];
    for (int i = 0; i < W; i++) t[i] = revcomp_word(x.w[W - 1 - i]);
    r.w[0] = t[0] >> shift;
    for (int i = 1; i < W; i++) r.w[i] = (t[i] >> shift) | (t[i - 1] << (64 - shift));
  }
  return r;
}

template <int W> MCX_HD bool kmer_less(const Kmer<W> &a, const Kmer<W> &b)
{
  if (W == 1) return a.w[0] < b.w[0];
  if (W == 2) return a.w[0] != b.w[0] ? a.w[0] < b.w[0] : a.w[W - 1] < b.w[W - 1];
  for (int i = 0; i < W - 1; i++)
    if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
  return a.w[W - 1] < b.w[W - 1];
}
template <int W> MCX_HD bool kmer_equal(const Kmer<W> &a, const Kmer<W> &b)
{
  bool e = true;
  for (int i = 0; i < W; i++) e = e && a.w[i] == b.w[i];
  return e;
}

// Roll a k-mer one base on (binary_kmer.h:139-167, binary_kmer_left_shift_add): the first base leaves, nuc is appended
template <int W> MCX_HD void kmer_push(Kmer<W> &x, uint32_t nuc, int k)
{
  for (int i = 0; i < W - 1; i++) x.w[i] = (x.w[i] << 2) | (x.w[i + 1] >> 62);
  x.w[W - 1] = (x.w[W - 1] << 2) | nuc;
  x.w[0] &= ~0ULL >> (64 * W - 2 * k);
}
template <int W> MCX_HD uint32_t kmer_first_base(const Kmer<W> &x, int k)
{
  return (uint32_t)(x.w[0] >> (2 * k - 2 - 64 * (W - 1))) & 3u;
}

// binary_kmer.c:43-57 + db_node.h:109-110: key = min(kmer, revcomp), orient =
// 0 (FORWARD) iff key == kmer.  k is odd, so kmer != revcomp always.
template <int W> MCX_HD Kmer<W> canonical(const Kmer<W> &fw, const Kmer<W> &rc, uint32_t &orient)
{
  const bool f = kmer_less<W>(fw, rc);
  orient = f ? 0u : 1u;
  return f ? fw : rc;
}

MCX_HD uint32_t rotl32(uint32_t x, int n)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_rotateleft32(x, n);
#else
  return (x << n) | (x >> (32 - n));
#endif
}

// lookup3 mix/final (src/kmer/kmer_hash.h:89-97, :124-133)
MCX_HD void lk3_mix(uint32_t &a, uint32_t &b, uint32_t &c)
{
  a -= c; a ^= rotl32(c, 4);  c += b;
  b -= a; b ^= rotl32(a, 6);  a += c;
  c -= b; c ^= rotl32(b, 8);  b += a;
  a -= c; a ^= rotl32(c, 16); c += b;
  b -= a; b ^= rotl32(a, 19); a += c;
  c -= b; c ^= rotl32(b, 4);  b += a;
}
MCX_HD void lk3_final(uint32_t &a, uint32_t &b, uint32_t &c)
{
  c ^= b; c -= rotl32(b, 14);
  a ^= c; a -= rotl32(c, 11);
  b ^= a; b -= rotl32(a, 25);
  c ^= b; c -= rotl32(b, 16);
  a ^= c; a -= rotl32(c, 4);
  b ^= a; b -= rotl32(a, 14);
  c ^= b; c -= rotl32(b, 24);
}

// bklk3_hashlittle (src/kmer/kmer_hash.h:162-211) over the 8*W key bytes.
// Returns c (the reference's hash value); *second receives b, the other
// well-mixed word of lookup3's final(), used as the shard-owner hash.
template <int W> MCX_HD uint32_t kmer_hash(const Kmer<W> &key, uint32_t initval, uint32_t *second)
{
  uint32_t a, b, c;
  a = b = c = 0xdeadbeefu + 8u * W + initval;
  if (W <= 2) {
    a += (uint32_t)key.w[0];
    b += (uint32_t)(key.w[0] >> 32);
    if (W == 2) {
      c += (uint32_t)key.w[W - 1];
      lk3_mix(a, b, c);
      a += (uint32_t)(key.w[W - 1] >> 32);
    }
  } else {
    // 2 W 32-bit halves in memory order (low half of w[0] first): blocks of three are mixed while more than three
    // are left ("while (length > 12)" of hashlittle), the last one to three go into a, b, c before final()
    uint32_t h[2 * W];
    for (int i = 0; i < W; i++) { h[2 * i] = (uint32_t)key.w[i]; h[2 * i + 1] = (uint32_t)(key.w[i] >> 32); }
    int p = 0, n = 2 * W;
    while (n > 3) { a += h[p]; b += h[p + 1]; c += h[p + 2]; lk3_mix(a, b, c); p += 3; n -= 3; }
    if (n >= 1) a += h[p];
    if (n >= 2) b += h[p + 1];
    if (n >= 3) c += h[p + 2];
  }
  lk3_final(a, b, c);
  if (second) *second = b;
  return c;
}

}  // namespace mcx
