/*
 * ref_lookup3_wrap.c -- OUR thin wrapper that exposes the reference's vendored,
 * self-contained libs/misc/lookup3.h (included from where it lies under
 * /root/reference, never copied) as a shared library, so tests can pin the
 * oracle's orc_kmer_hash against the real reference hash code.
 * src/basic/hash.h:18-21 selects lk3_hashlittle as ctx_hash32; src/kmer/kmer_hash.h
 * is the reference's own fixed-length specialisation of the same function.
 * TEST INFRASTRUCTURE ONLY.  Output: oracle/_ref/liblk3ref.so (git-ignored).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "misc/lookup3.h"

uint32_t ref_lk3_hashlittle(const void *key, size_t nbytes, uint32_t initval)
{
  return lk3_hashlittle(key, nbytes, initval);
}
