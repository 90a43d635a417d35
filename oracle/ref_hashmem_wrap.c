/*
 * ref_hashmem_wrap.c -- OUR wrapper that exposes what the reference's self-contained header src/basic/hash_mem.h
 * defines -- the table constants REHASH_LIMIT, MAX_BUCKET_SIZE, IDEAL_OCCUPANCY, WARN_OCCUPANCY and the memory formula
 * ht_mem() -- as a shared library.  The header is included from where it lies under /root/reference (-I $(REF)),
 * unmodified and never copied; it needs nothing but size_t / uint8_t.  (src/basic/hash_mem.c, which holds
 * hash_table_cap(), includes global.h and with it headers from empty submodules: not buildable here, and no stand-in
 * is written for it -- that function stays pinned by the capacity table in tests/golden/reference_kats.json.)
 * TEST INFRASTRUCTURE ONLY: pins the oracle's ORC_REHASH_LIMIT / ORC_MAX_BUCKET and the product's table sizing
 * (mccortex_amd/host/host_util.c) against reference code.  Output: oracle/_ref/libhashmemref.so (git-ignored).
 */
#include <stddef.h>
#include <stdint.h>
#include "src/basic/hash_mem.h"

size_t ref_ht_mem(size_t bktsize, size_t nbkts, size_t nbits) { return ht_mem(bktsize, nbkts, nbits); }
int ref_rehash_limit(void) { return REHASH_LIMIT; }
int ref_max_bucket_size(void) { return MAX_BUCKET_SIZE; }
float ref_ideal_occupancy(void) { return IDEAL_OCCUPANCY; }
float ref_warn_occupancy(void) { return WARN_OCCUPANCY; }
