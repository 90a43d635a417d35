/*
 * ref_revcmp_wrap.c -- OUR wrapper that exposes the four BinaryKmer reverse-complement
 * implementations of the reference's standalone benchmark dev/bkmer_revcmp/revcmp.c
 * (libc only; it carries its own BinaryKmer typedef and BKMER_TOP_BITS macros, the same
 * as src/graph/binary_kmer.h:10-28) as a shared library.  The file is included from where
 * it lies under /root/reference (-I $(REF)), unmodified and never copied; its main() is
 * renamed away.  Built twice, -DNUM_BKMER_WORDS=1 and =2, into
 * oracle/_ref/librevcmp{1,2}.so (git-ignored).  binary_kmer_reverse_complement2 is, line
 * for line, what src/graph/binary_kmer.c:102-133 does; 1, 3 and 4 are the generic-loop and
 * table-driven variants the benchmark compares with it.
 * TEST INFRASTRUCTURE ONLY: pins orc_kmer_revcomp (hence the b[0]-is-top-word layout and
 * the shift across words) and the product's mcx_kmer_canonical against reference code.
 */
#define main ref_revcmp_benchmark_main
#include "dev/bkmer_revcmp/revcmp.c"
#undef main

int ref_revcmp_words(void) { return NUM_BKMER_WORDS; }

/* method 1..4; in/out: NUM_BKMER_WORDS words, b[0] first */
void ref_revcmp(int method, const uint64_t *in, size_t kmer_size, uint64_t *out)
{
  BinaryKmer x, y;
  size_t i;
  for(i = 0; i < NUM_BKMER_WORDS; i++) x.b[i] = in[i];
  switch(method) {
    case 1: y = binary_kmer_reverse_complement1(x, kmer_size); break;
    case 2: y = binary_kmer_reverse_complement2(x, kmer_size); break;
    case 3: y = binary_kmer_reverse_complement3(x, kmer_size); break;
    default: y = binary_kmer_reverse_complement4(x, kmer_size); break;
  }
  for(i = 0; i < NUM_BKMER_WORDS; i++) out[i] = y.b[i];
}

/*
 * The reference's own fixed-length hash of a BinaryKmer, src/kmer/kmer_hash.h:162-211 (bklk3_hashlittle), compiled
 * from where it lies.  It needs <string.h>, the BinaryKmer typedef (revcmp.c brings it, above) and BKMER_BYTES as a
 * number the preprocessor can compare: revcmp.c spells it with sizeof, which #if cannot evaluate -- the reference's
 * src/graph/binary_kmer.h:17-18 keeps that spelling commented out for the same reason and uses the line below.
 * That one macro value is the only thing written here; no header is stood in for.
 */
#include <string.h>
#undef BKMER_BYTES
#define BKMER_BYTES (NUM_BKMER_WORDS * 8) /* src/graph/binary_kmer.h:18 */
#include "src/kmer/kmer_hash.h"

uint32_t ref_bklk3_hashlittle(const uint64_t *in, uint32_t initval)
{
  BinaryKmer x;
  size_t i;
  for(i = 0; i < NUM_BKMER_WORDS; i++) x.b[i] = in[i];
  return bklk3_hashlittle(x, initval);
}
