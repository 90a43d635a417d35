/*
 * mcx_oracle.c -- CPU restatement of the McCortex `build` hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see mcx_oracle.h).  Written from scratch in plain
 * C; each function names the reference file:line (under /root/reference) whose
 * arithmetic it follows.  The reference itself cannot be built in this image
 * (its libs/ submodules are empty) so this file is pinned against the
 * reference's own known-answer/unit tests, the vendored libs/misc/lookup3.h
 * compiled from where it lies (oracle/_ref), and the KATs recorded in
 * SURVEY.md 8(c) -- see oracle/README.md.
 */
#define _GNU_SOURCE
#include "mcx_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sched.h>

#define ORC_REHASH_LIMIT 20    /* src/basic/hash_mem.h:4 */
#define ORC_MAX_BUCKET   48    /* src/basic/hash_mem.h:8 */
#define ORC_FLAG (1ULL << 63)  /* BKMER_SET_FLAG, src/graph/hash_table.h:41-46 */

/* ------------------------------------------------------------------------ */
/* Row A-C: k-mer primitives                                                 */
/* ------------------------------------------------------------------------ */

/* binary_kmer.h:10: NUM_BKMER_WORDS64(k) = (2k+63)/64.  For a legal k of a
 * given MAXK build (MAXK-30 <= k <= MAXK) this equals the compile-time W. */
int orc_words_for_k(int k) { return (2 * k + 63) / 64; }

/* dna.c:8-25: A/a=0 C/c=1 G/g=2 T/t=3 N/n=4 other=8 */
int orc_char_to_nuc(unsigned char c)
{
  switch(c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case 'N': case 'n': return 4;
    default: return 8;
  }
}
static inline int is_acgt(char c) { return orc_char_to_nuc((unsigned char)c) < 4; }

static inline int top_bits(int k) { return 2 * (k & 31); } /* binary_kmer.h:40-42 */

/* binary_kmer.c:156-186: top word takes the first k&31 bases, each later word 32 */
orc_bkmer orc_kmer_from_str(const char *seq, int k)
{
  orc_bkmer x; memset(&x, 0, sizeof(x));
  const int W = orc_words_for_k(k);
  int pos = 0, w, end = k & 31;
  for(; pos < end; pos++) x.b[0] = (x.b[0] << 2) | (uint64_t)orc_char_to_nuc((unsigned char)seq[pos]);
  for(w = 1; w < W; w++)
    for(end += 32; pos < end; pos++)
      x.b[w] = (x.b[w] << 2) | (uint64_t)orc_char_to_nuc((unsigned char)seq[pos]);
  return x;
}

/* binary_kmer.h:139-146,160-167 and binary_kmer.c:80-97: shift every word left
 * one base carrying from the word below, mask the top word, add nuc at the end */
orc_bkmer orc_kmer_shift_add(orc_bkmer x, int k, int nuc)
{
  const int W = orc_words_for_k(k);
  int w;
  for(w = 0; w + 1 < W; w++) x.b[w] = (x.b[w] << 2) | (x.b[w + 1] >> 62);
  x.b[W - 1] <<= 2;
  x.b[0] &= UINT64_MAX >> (64 - top_bits(k));
  x.b[W - 1] |= (uint64_t)nuc;
  return x;
}

/* binary_kmer.c:102-133 */
orc_bkmer orc_kmer_revcomp(orc_bkmer x, int k)
{
  const int W = orc_words_for_k(k);
  const int tb = top_bits(k), unused = 64 - tb;
  orc_bkmer r; memset(&r, 0, sizeof(r));
  int i, j;
  for(i = 0, j = W - 1; i < W; i++, j--) {
    uint64_t w = __builtin_bswap64(x.b[i]);
    w = ((w & 0x0303030303030303ULL) << 6) | ((w & 0x0c0c0c0c0c0c0c0cULL) << 2) |
        ((w & 0x3030303030303030ULL) >> 2) | ((w & 0xc0c0c0c0c0c0c0c0ULL) >> 6);
    r.b[j] = ~w;
  }
  for(i = W - 1; i > 0; i--) r.b[i] = (r.b[i] >> unused) | (r.b[i - 1] << tb);
  r.b[0] >>= unused;
  return r;
}

static inline int bkmer_lt(const orc_bkmer *a, const orc_bkmer *b, int W)
{ /* binary_kmer.h:79-94: word 0 first, unsigned */
  int i;
  for(i = 0; i < W; i++) if(a->b[i] != b->b[i]) return a->b[i] < b->b[i];
  return 0;
}
static inline int bkmer_eq(const orc_bkmer *a, const orc_bkmer *b, int W)
{
  int i;
  for(i = 0; i < W; i++) if(a->b[i] != b->b[i]) return 0;
  return 1;
}

/* binary_kmer.c:43-57 */
orc_bkmer orc_kmer_get_key(orc_bkmer x, int k)
{
  const int W = orc_words_for_k(k);
  unsigned first = (unsigned)(x.b[0] >> (top_bits(k) - 2)) & 3u;
  unsigned last = (unsigned)x.b[W - 1] & 3u;
  unsigned rev_last = ~last & 3u; /* dna.h:22 */
  if(first < rev_last) return x;
  orc_bkmer rc = orc_kmer_revcomp(x, k);
  return bkmer_lt(&x, &rc, W) ? x : rc;
}

#define ROT(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
/* kmer_hash.h:89-97 */
#define LK3_MIX(a, b, c) do { \
  a -= c; a ^= ROT(c, 4);  c += b; b -= a; b ^= ROT(a, 6);  a += c; \
  c -= b; c ^= ROT(b, 8);  b += a; a -= c; a ^= ROT(c, 16); c += b; \
  b -= a; b ^= ROT(a, 19); a += c; c -= b; c ^= ROT(b, 4);  b += a; } while(0)
/* kmer_hash.h:124-133 */
#define LK3_FINAL(a, b, c) do { \
  c ^= b; c -= ROT(b, 14); a ^= c; a -= ROT(c, 11); b ^= a; b -= ROT(a, 25); \
  c ^= b; c -= ROT(b, 16); a ^= c; a -= ROT(c, 4);  b ^= a; b -= ROT(a, 14); \
  c ^= b; c -= ROT(b, 24); } while(0)

/* kmer_hash.h:162-211: lookup3 hashlittle over the W*8 key bytes in memory
 * order (little-endian 32-bit halves of b[0], b[1], ...). */
uint32_t orc_kmer_hash(orc_bkmer key, int k, uint32_t initval)
{
  const int W = orc_words_for_k(k);
  const uint32_t nbytes = (uint32_t)W * 8u;
  uint32_t w32[2 * ORC_MAX_W];
  int i, n = 2 * W;
  for(i = 0; i < W; i++) { w32[2 * i] = (uint32_t)key.b[i]; w32[2 * i + 1] = (uint32_t)(key.b[i] >> 32); }
  uint32_t a, b, c;
  a = b = c = 0xdeadbeefu + nbytes + initval;
  const uint32_t *p = w32;
  while(n > 3) { /* "i+12 < BKMER_BYTES" blocks */
    a += p[0]; b += p[1]; c += p[2];
    LK3_MIX(a, b, c);
    p += 3; n -= 3;
  }
  if(n == 3) { a += p[0]; b += p[1]; c += p[2]; }
  else if(n == 2) { a += p[0]; b += p[1]; }
  else { a += p[0]; }
  LK3_FINAL(a, b, c);
  return c;
}

/* ctx_exp_hashtest.c:61-66 (hash_loop, no graph: `bkmer.b[0] = i; hash ^= binary_kmer_hash(bkmer, 0);`) and
 * :160-175 (one job per thread over [start, end), `hash += jobs[i].hash`) */
uint64_t orc_hashtest_func(int k, uint64_t n, uint32_t nparts)
{
  uint64_t sum = 0, i;
  uint32_t p;
  for(p = 0; p < nparts; p++) {
    const uint64_t start = (uint64_t)p * (n / nparts);
    const uint64_t end = (p + 1 == nparts) ? n : start + (n / nparts);
    uint32_t hash = 0;
    orc_bkmer bkmer;
    memset(&bkmer, 0, sizeof(bkmer));
    for(i = start; i < end; i++) {
      bkmer.b[0] = i;
      hash ^= orc_kmer_hash(bkmer, k, 0);
    }
    sum += hash;
  }
  return sum;
}

void orc_kmer_to_str(orc_bkmer x, int k, char *out)
{
  static const char nuc2c[4] = {'A', 'C', 'G', 'T'};
  const int W = orc_words_for_k(k);
  int i;
  for(i = k - 1; i >= 0; i--) {
    out[i] = nuc2c[x.b[W - 1] & 3];
    int w; /* shift right one base across words */
    for(w = W - 1; w > 0; w--) x.b[w] = (x.b[w] >> 2) | (x.b[w - 1] << 62);
    x.b[0] >>= 2;
  }
  out[k] = '\0';
}

/* hash_mem.c:5-15 */
int orc_rehash_limit(void) { return ORC_REHASH_LIMIT; }
int orc_max_bucket_size(void) { return ORC_MAX_BUCKET; }

uint64_t orc_hash_table_cap(uint64_t nkmers, uint64_t *nbuckets, uint8_t *bucket_size)
{
  uint64_t nbits = 10;
  while(nkmers / (1ULL << nbits) > ORC_MAX_BUCKET) nbits++;
  uint64_t nb = 1ULL << nbits;
  uint64_t bs = (nkmers + nb - 1) / nb;
  if(bs < 1) bs = 1;
  if(nbuckets) *nbuckets = nb;
  if(bucket_size) *bucket_size = (uint8_t)bs;
  return nb * bs;
}

/* ------------------------------------------------------------------------ */
/* Row G: contig splitting                                                   */
/* ------------------------------------------------------------------------ */

/* seq_reader.c:61-117 */
size_t orc_contig_start(const char *seq, size_t seqlen, const char *qual, size_t quallen,
                        size_t offset, size_t k, uint8_t qual_cutoff, uint8_t hp_cutoff)
{
  if(!qual || !quallen) { qual = NULL; quallen = 0; }
  size_t pos = offset, kend;
  while((kend = pos + k) <= seqlen) {
    size_t i = kend;
    while(i > pos && is_acgt(seq[i - 1])) i--;
    if(i > pos) { pos = i; continue; }

    if(qual && qual_cutoff > 0) {
      i = kend < quallen ? kend : quallen;
      while(i > pos && qual[i - 1] > qual_cutoff) i--;
      if(i > pos) { pos = i; continue; }
    }

    if(hp_cutoff > 0) {
      size_t run = 1;
      for(i = kend - 1; i > pos; i--) {
        if(seq[i - 1] == seq[i]) { run++; if(run == (size_t)hp_cutoff) break; }
        else run = 1;
      }
      if(i > pos) { pos = i; continue; }
    }
    return pos;
  }
  return seqlen;
}

/* seq_reader.c:127-172 */
size_t orc_contig_end(const char *seq, size_t seqlen, const char *qual, size_t quallen,
                      size_t contig_start, size_t k, uint8_t qual_cutoff, uint8_t hp_cutoff,
                      size_t *search_start)
{
  if(!qual || !quallen) { qual = NULL; quallen = 0; }
  size_t end = contig_start + k;
  size_t hp_run = 1;
  if(hp_cutoff > 0)
    while(hp_run < end && seq[end - 1 - hp_run] == seq[end - 1]) hp_run++;

  for(; end < seqlen; end++) {
    if(!is_acgt(seq[end]) || (end < quallen && qual[end] < qual_cutoff)) break;
    if(hp_cutoff > 0) {
      if(seq[end] == seq[end - 1]) { hp_run++; if(hp_run >= (size_t)hp_cutoff) break; }
      else hp_run = 1;
    }
  }
  if(hp_cutoff > 0 && hp_run >= (size_t)hp_cutoff) *search_start = end - (size_t)hp_cutoff + 1;
  else *search_start = end;
  return end;
}

/* ------------------------------------------------------------------------ */
/* Rows D-F: bucketed hash table + node arrays                               */
/* ------------------------------------------------------------------------ */

struct orc_graph {
  int k, W, ncols;
  /* HashTable: hash_table.h:18-31 */
  uint64_t *table;         /* capacity * W words; slot word0 carries ORC_FLAG */
  uint8_t *bsize;          /* items appended per bucket (== bitems: no deletes on build) */
  volatile uint8_t *locks; /* one lock byte per bucket (reference: one bit) */
  uint64_t nbuckets, capacity, hash_mask;
  uint8_t bucket_size;
  uint32_t seed;
  volatile uint64_t num_kmers;
  /* dBGraph arrays: db_graph.h, db_node.h:240-241 */
  uint32_t *covgs; /* [capacity][ncols] */
  uint8_t *edges;  /* [capacity][ncols] */
  /* GraphInfo per colour: graph_info.h:20-27 */
  uint32_t *mean_read_length;
  uint64_t *total_sequence;
  char (*sample)[256];
  volatile int full;
  int tuned;            /* orc_graph_tune: prefaulted arrays, per-worker node tallies (timed baseline only) */
  int force_generic; /* tests: run the multi-word code even when W == 1 */
  /* build --intersect (ctx_build.c:341-363,384-413) */
  uint8_t *isec_edges;  /* [capacity]: union of the intersection graphs' edges, NULL when not intersecting */
  int must_exist;       /* BuildGraphTask.prefs.must_exist_in_graph */
  /* build --remove-pcr: dBGraph.readstrt, one bit per (slot, orientation) (db_graph.c:62-63) */
  uint8_t *readstrt;
};

orc_graph *orc_graph_new(int k, int ncols, uint64_t capacity_kmers, uint32_t seed)
{
  if(k < 3 || !(k & 1) || orc_words_for_k(k) > ORC_MAX_W || ncols < 1) return NULL;
  orc_graph *g = calloc(1, sizeof(*g));
  g->k = k; g->W = orc_words_for_k(k); g->ncols = ncols;
  g->capacity = orc_hash_table_cap(capacity_kmers, &g->nbuckets, &g->bucket_size);
  g->hash_mask = g->nbuckets - 1;
  g->seed = seed; /* reference: rand(), hash_table.c:49 -- any value gives the same sorted graph */
  g->table = calloc(g->capacity * (size_t)g->W, sizeof(uint64_t));
  g->bsize = calloc(g->nbuckets, 1);
  g->locks = calloc(g->nbuckets, 1);
  g->covgs = calloc(g->capacity * (size_t)ncols, sizeof(uint32_t));
  g->edges = calloc(g->capacity * (size_t)ncols, 1);
  g->mean_read_length = calloc((size_t)ncols, sizeof(uint32_t));
  g->total_sequence = calloc((size_t)ncols, sizeof(uint64_t));
  g->sample = calloc((size_t)ncols, sizeof(*g->sample));
  int c;
  for(c = 0; c < ncols; c++) strcpy(g->sample[c], "undefined"); /* graph_info.c:62 */
  if(!g->table || !g->bsize || !g->locks || !g->covgs || !g->edges) { orc_graph_free(g); return NULL; }
  return g;
}

void orc_graph_free(orc_graph *g)
{
  if(!g) return;
  free(g->table); free(g->bsize); free((void *)g->locks); free(g->covgs); free(g->edges);
  free(g->mean_read_length); free(g->total_sequence); free(g->sample); free(g->isec_edges); free(g->readstrt);
  free(g);
}

int orc_graph_set_sample(orc_graph *g, int col, const char *name)
{
  if(col < 0 || col >= g->ncols || strlen(name) > 255) return -1;
  strcpy(g->sample[col], name);
  return 0;
}
void orc_graph_force_generic(orc_graph *g, int on) { g->force_generic = on; }
uint64_t orc_graph_nkmers(const orc_graph *g) { return g->num_kmers; }
uint64_t orc_graph_capacity(const orc_graph *g) { return g->capacity; }

/* num_kmers: the reference adds to ONE shared counter from every worker (hash_table.c:113, an
 * atomic add per new node).  The timed baseline (orc_graph_tune) tallies per worker and merges at
 * the end of the batch instead, so that the port is not slowed by a cache line all threads write. */
static __thread uint64_t *tl_new_nodes = NULL;
static inline void count_new_node(orc_graph *g)
{
  if(tl_new_nodes) (*tl_new_nodes)++;
  else __sync_add_and_fetch(&g->num_kmers, 1);
}

static inline void bkt_lock(volatile uint8_t *l)
{ /* bitlock_yield_acquire: hash_table.c:260 */
  while(__sync_lock_test_and_set(l, 1)) sched_yield();
}
static inline void bkt_unlock(volatile uint8_t *l) { __sync_lock_release(l); }

#define ORC_NOT_FOUND (UINT64_MAX >> 1) /* hash_table.h:13 */

/* hash_table.c:250-281 (find_or_insert_mt) with :78-91 (find_in_bucket) and
 * :94-117 (insert_in_bucket).  Returns slot index or ORC_NOT_FOUND when all
 * REHASH_LIMIT buckets are full (the reference dies: "Hash table is full"). */
static uint64_t find_or_insert(orc_graph *g, const orc_bkmer *key, int *found)
{
  const int W = g->W;
  int i, w;
  for(i = 0; i < ORC_REHASH_LIMIT; i++) {
    uint64_t h = orc_kmer_hash(*key, g->k, g->seed + (uint32_t)i) & g->hash_mask;
    bkt_lock(&g->locks[h]);
    uint64_t *slot = g->table + h * g->bucket_size * (uint64_t)W;
    unsigned n = g->bsize[h], j;
    for(j = 0; j < n; j++, slot += W) {
      if(slot[0] != (key->b[0] | ORC_FLAG)) continue;
      for(w = 1; w < W && slot[w] == key->b[w]; w++) {}
      if(w == W) {
        *found = 1;
        bkt_unlock(&g->locks[h]);
        return h * g->bucket_size + j;
      }
    }
    if(n < g->bucket_size) {
      slot[0] = key->b[0] | ORC_FLAG;
      for(w = 1; w < W; w++) slot[w] = key->b[w];
      g->bsize[h] = (uint8_t)(n + 1);
      count_new_node(g);
      *found = 0;
      bkt_unlock(&g->locks[h]);
      return h * g->bucket_size + n;
    }
    bkt_unlock(&g->locks[h]);
  }
  return ORC_NOT_FOUND;
}

/* hash_table_find: hash_table.c:125-154 (a bucket that is not full ends the search) */
static uint64_t find_only(const orc_graph *g, const orc_bkmer *key)
{
  const int W = g->W;
  int i, w;
  for(i = 0; i < ORC_REHASH_LIMIT; i++) {
    uint64_t h = orc_kmer_hash(*key, g->k, g->seed + (uint32_t)i) & g->hash_mask;
    const uint64_t *slot = g->table + h * g->bucket_size * (uint64_t)W;
    unsigned n = g->bsize[h], j;
    for(j = 0; j < n; j++, slot += W) {
      if(slot[0] != (key->b[0] | ORC_FLAG)) continue;
      for(w = 1; w < W && slot[w] == key->b[w]; w++) {}
      if(w == W) return h * g->bucket_size + j;
    }
    if(n < g->bucket_size) break;
  }
  return ORC_NOT_FOUND;
}

/* db_node.c:139-144: saturating thread-safe +1 */
static inline void covg_inc(uint32_t *p)
{
  uint32_t v;
  while((v = *(volatile uint32_t *)p) < UINT32_MAX && !__sync_bool_compare_and_swap(p, v, v + 1)) {}
}

typedef struct { uint64_t hkey; int orient; } orc_node;

/* db_graph.c:126-135 + :101-105 */
static inline orc_node find_or_add_node(orc_graph *g, const orc_bkmer *bk, int colour, int *found)
{
  orc_bkmer key = orc_kmer_get_key(*bk, g->k);
  orc_node n;
  if(g->must_exist) { /* _find_or_insert with must_exist_in_graph: build_graph.c:99-117 */
    n.hkey = find_only(g, &key);
    *found = n.hkey != ORC_NOT_FOUND;
  } else {
    n.hkey = find_or_insert(g, &key, found);
  }
  n.orient = bkmer_eq(&key, bk, g->W) ? 0 : 1; /* db_node.h:109-110 */
  if(n.hkey != ORC_NOT_FOUND) covg_inc(&g->covgs[n.hkey * (uint64_t)g->ncols + (uint64_t)colour]);
  return n;
}

/* db_graph.c:152-166; nuc_orient_to_edge: db_node.h:180; set_col_edge_mt: db_node.h:273-274.
 * The oriented first/last bases are read back from the stored key exactly as the
 * reference does (bkmer_get_first_nuc / bkmer_get_last_nuc, db_node.h:115-121). */
static inline void add_edge(orc_graph *g, int colour, orc_node src, orc_node tgt)
{
  const int W = g->W, tb = top_bits(g->k);
  const uint64_t *ks = g->table + src.hkey * (uint64_t)W, *kt = g->table + tgt.hkey * (uint64_t)W;
  unsigned s_first = (unsigned)((ks[0] & ~ORC_FLAG) >> (tb - 2)) & 3u, s_last = (unsigned)ks[W - 1] & 3u;
  unsigned t_first = (unsigned)((kt[0] & ~ORC_FLAG) >> (tb - 2)) & 3u, t_last = (unsigned)kt[W - 1] & 3u;
  unsigned lhs = src.orient == 0 ? s_first : (~s_last & 3u);
  unsigned rhs = tgt.orient == 0 ? t_last : (~t_first & 3u);
  unsigned lhs_rev = ~lhs & 3u;
  uint8_t e_src = (uint8_t)(1u << (rhs + 4u * (unsigned)src.orient));
  uint8_t e_tgt = (uint8_t)(1u << (lhs_rev + 4u * (unsigned)!tgt.orient));
  __sync_fetch_and_or(&g->edges[src.hkey * (uint64_t)g->ncols + (uint64_t)colour], e_src);
  __sync_fetch_and_or(&g->edges[tgt.hkey * (uint64_t)g->ncols + (uint64_t)colour], e_tgt);
}

/* ---- W == 1 fast path -------------------------------------------------------------
 * Same algorithm with the key held in one uint64_t, as the reference's MAXK=31 build has it
 * at compile time (NUM_BKMER_WORDS == 1): used for the timed CPU baseline so the port is not
 * handicapped by the generic multi-word code above.  Results are identical (tests compare). */
static inline uint32_t hash_w1(uint64_t key, uint32_t initval)
{ /* kmer_hash.h:162-211 with BKMER_BYTES == 8 */
  uint32_t a, b, c;
  a = b = c = 0xdeadbeefu + 8u + initval;
  a += (uint32_t)key; b += (uint32_t)(key >> 32);
  LK3_FINAL(a, b, c);
  return c;
}

static inline uint64_t find_or_insert_w1(orc_graph *g, uint64_t key, int *found)
{ /* hash_table.c:250-281 */
  const uint64_t want = key | ORC_FLAG;
  int i;
  for(i = 0; i < ORC_REHASH_LIMIT; i++) {
    uint64_t h = hash_w1(key, g->seed + (uint32_t)i) & g->hash_mask;
    bkt_lock(&g->locks[h]);
    uint64_t *slot = g->table + h * g->bucket_size;
    unsigned n = g->bsize[h], j;
    for(j = 0; j < n; j++)
      if(slot[j] == want) { *found = 1; bkt_unlock(&g->locks[h]); return h * g->bucket_size + j; }
    if(n < g->bucket_size) {
      slot[n] = want;
      g->bsize[h] = (uint8_t)(n + 1);
      count_new_node(g);
      *found = 0;
      bkt_unlock(&g->locks[h]);
      return h * g->bucket_size + n;
    }
    bkt_unlock(&g->locks[h]);
  }
  return ORC_NOT_FOUND;
}

static inline orc_node node_w1(orc_graph *g, uint64_t bk, int colour, int *found)
{ /* db_graph.c:126-135 + :101-105; binary_kmer.c:43-57 */
  const int k = g->k;
  unsigned first = (unsigned)(bk >> (2 * k - 2)) & 3u, last = (unsigned)bk & 3u;
  uint64_t key = bk;
  if(!(first < (~last & 3u))) {
    orc_bkmer t; t.b[0] = bk;
    uint64_t rc = orc_kmer_revcomp(t, k).b[0];
    if(rc < bk) key = rc;
  }
  orc_node n;
  n.hkey = find_or_insert_w1(g, key, found);
  n.orient = key == bk ? 0 : 1;
  if(n.hkey != ORC_NOT_FOUND) covg_inc(&g->covgs[n.hkey * (uint64_t)g->ncols + (uint64_t)colour]);
  return n;
}

static size_t build_from_str_w1(orc_graph *g, int colour, const char *seq, size_t len)
{ /* build_graph.c:122-150 */
  const int k = g->k;
  const uint64_t mask = UINT64_MAX >> (64 - 2 * k);
  size_t i, nonnovel = 0;
  int found = 0;
  uint64_t bk = 0;
  for(i = 0; i < (size_t)k; i++) bk = (bk << 2) | (uint64_t)orc_char_to_nuc((unsigned char)seq[i]);
  orc_node prev = node_w1(g, bk, colour, &found), curr;
  if(prev.hkey == ORC_NOT_FOUND) { g->full = 1; return 0; }
  nonnovel += (size_t)found;
  for(i = (size_t)k; i < len; i++, prev = curr) {
    bk = ((bk << 2) | (uint64_t)orc_char_to_nuc((unsigned char)seq[i])) & mask;
    curr = node_w1(g, bk, colour, &found);
    if(curr.hkey == ORC_NOT_FOUND) { g->full = 1; return nonnovel; }
    add_edge(g, colour, prev, curr);
    nonnovel += (size_t)found;
  }
  return nonnovel;
}

/* build_graph.c:122-150 */
static size_t build_from_str(orc_graph *g, int colour, const char *seq, size_t len)
{
  if(g->W == 1 && !g->force_generic && !g->must_exist) return build_from_str_w1(g, colour, seq, len);
  const int k = g->k;
  size_t i, nonnovel = 0;
  int found = 0;
  orc_bkmer bk = orc_kmer_from_str(seq, k);
  orc_node prev = find_or_add_node(g, &bk, colour, &found), curr;
  if(prev.hkey == ORC_NOT_FOUND && !g->must_exist) { g->full = 1; return 0; }
  nonnovel += (size_t)found;
  for(i = (size_t)k; i < len; i++, prev = curr) {
    bk = orc_kmer_shift_add(bk, k, orc_char_to_nuc((unsigned char)seq[i]));
    curr = find_or_add_node(g, &bk, colour, &found);
    if(curr.hkey == ORC_NOT_FOUND && !g->must_exist) { g->full = 1; return nonnovel; }
    if(prev.hkey != ORC_NOT_FOUND && curr.hkey != ORC_NOT_FOUND) /* build_graph.c:143-144 */
      add_edge(g, colour, prev, curr);
    nonnovel += (size_t)found;
  }
  return nonnovel;
}

/* build_graph.c:154-189 (load_read) + :192-231 (SE path of build_graph_from_reads_mt,
 * no PCR-duplicate removal, must_exist_in_graph=false) */
static void load_read(orc_graph *g, int colour, const char *seq, size_t len, const char *qual,
                      uint8_t fq_cutoff, uint8_t hp_cutoff, orc_stats *st)
{
  const size_t k = (size_t)g->k;
  size_t cs, ce, search = 0, ncontigs = 0;
  const size_t qlen = qual ? len : 0;
  st->total_bases_read += len;
  st->num_se_reads += 1;
  while((cs = orc_contig_start(seq, len, qual, qlen, search, k, fq_cutoff, hp_cutoff)) < len) {
    ce = orc_contig_end(seq, len, qual, qlen, cs, k, fq_cutoff, hp_cutoff, &search);
    size_t clen = ce - cs;
    size_t nonnovel = build_from_str(g, colour, seq + cs, clen);
    if(g->full) return;
    size_t ckmers = clen + 1 - k;
    st->total_bases_loaded += clen;
    if(g->must_exist) {  /* build_graph.c:175-177: only the k-mers that were found count as loaded */
      st->num_kmers_loaded += nonnovel;
    } else {
      st->num_kmers_loaded += ckmers;
      st->num_kmers_novel += ckmers - nonnovel;
    }
    ncontigs++;
  }
  st->contigs_parsed += ncontigs;
  st->num_good_reads += (ncontigs > 0);
  st->num_bad_reads += (ncontigs == 0);
}

typedef struct {
  orc_graph *g; int colour; const char *bases, *quals; const uint64_t *off;
  uint64_t lo, hi; uint8_t fq, hp; orc_stats st;
  uint64_t new_nodes; /* tuned mode: this worker's share of num_kmers */
} orc_job;

static void *job_run(void *p)
{
  orc_job *j = p;
  uint64_t r;
  if(j->g->tuned) tl_new_nodes = &j->new_nodes;
  for(r = j->lo; r < j->hi && !j->g->full; r++) {
    size_t len = (size_t)(j->off[r + 1] - j->off[r]);
    load_read(j->g, j->colour, j->bases + j->off[r], len, j->quals ? j->quals + j->off[r] : NULL,
              j->fq, j->hp, &j->st);
  }
  tl_new_nodes = NULL;
  return NULL;
}

static void stats_merge(orc_stats *d, const orc_stats *s)
{ /* seq_loading_stats.c merge */
  d->num_se_reads += s->num_se_reads; d->num_good_reads += s->num_good_reads;
  d->num_bad_reads += s->num_bad_reads; d->total_bases_read += s->total_bases_read;
  d->total_bases_loaded += s->total_bases_loaded; d->contigs_parsed += s->contigs_parsed;
  d->num_kmers_loaded += s->num_kmers_loaded; d->num_kmers_novel += s->num_kmers_novel;
}

int orc_graph_add_reads(orc_graph *g, int colour, const char *bases, const char *quals,
                        const uint64_t *offsets, uint64_t nreads,
                        uint8_t fq_cutoff, uint8_t hp_cutoff, int nthreads, orc_stats *stats_accum)
{
  if(colour < 0 || colour >= g->ncols) return -2;
  if(nthreads < 1) nthreads = 1;
  if((uint64_t)nthreads > nreads) nthreads = nreads ? (int)nreads : 1;
  orc_job *jobs = calloc((size_t)nthreads, sizeof(orc_job));
  pthread_t *th = calloc((size_t)nthreads, sizeof(pthread_t));
  int t;
  for(t = 0; t < nthreads; t++) {
    orc_job *j = &jobs[t];
    j->g = g; j->colour = colour; j->bases = bases; j->quals = quals; j->off = offsets;
    j->lo = nreads * (uint64_t)t / (uint64_t)nthreads;
    j->hi = nreads * (uint64_t)(t + 1) / (uint64_t)nthreads;
    j->fq = fq_cutoff; j->hp = hp_cutoff;
  }
  if(nthreads == 1) job_run(&jobs[0]);
  else {
    for(t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, job_run, &jobs[t]);
    for(t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  }
  if(stats_accum) for(t = 0; t < nthreads; t++) stats_merge(stats_accum, &jobs[t].st);
  for(t = 0; t < nthreads; t++) g->num_kmers += jobs[t].new_nodes;
  free(jobs); free(th);
  return g->full ? -1 : 0;
}

/* ---- the timed CPU baseline (bench.py's cpu_baseline leg) ---------------------------------------
 * orc_graph_tune: what a careful user of the reference would do before timing it, without touching
 * the algorithm: the table and node arrays are touched once up front, by `nthreads` threads in
 * interleaved slices (so that the first-touch page faults -- which serialise on the process's
 * mmap lock and made 4 threads slower than 1 on the 256-CPU host -- and the NUMA placement are out
 * of the timed region), transparent huge pages are requested for them, and new nodes are counted
 * per worker (see count_new_node). */
#include <sys/mman.h>
typedef struct { uint8_t *p; size_t n; int t, nt; } touch_job;
static void *touch_run(void *a)
{
  touch_job *j = a;
  const size_t slice = 2u << 20;
  size_t o;
  for(o = (size_t)j->t * slice; o < j->n; o += (size_t)j->nt * slice)
    memset(j->p + o, 0, j->n - o < slice ? j->n - o : slice);
  return NULL;
}
static void touch_array(void *p, size_t n, int nthreads)
{
  if(!p || !n) return;
#ifdef MADV_HUGEPAGE
  { uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095; if(a < (uintptr_t)p + n) madvise((void *)a, (uintptr_t)p + n - a, MADV_HUGEPAGE); }
#endif
  pthread_t th[256];
  touch_job jb[256];
  int t;
  if(nthreads > 256) nthreads = 256;
  for(t = 0; t < nthreads; t++) { jb[t] = (touch_job){p, n, t, nthreads}; pthread_create(&th[t], NULL, touch_run, &jb[t]); }
  for(t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}
void orc_graph_tune(orc_graph *g, int nthreads)
{ /* only on an empty graph: the arrays are zeroed again */
  if(g->num_kmers) return;
  if(nthreads < 1) nthreads = 1;
  g->tuned = 1;
  touch_array(g->table, g->capacity * (size_t)g->W * 8, nthreads);
  touch_array(g->bsize, g->nbuckets, nthreads);
  touch_array((void *)g->locks, g->nbuckets, nthreads);
  touch_array(g->covgs, g->capacity * (size_t)g->ncols * 4, nthreads);
  touch_array(g->edges, g->capacity * (size_t)g->ncols, nthreads);
}

/* Reference-shaped end-to-end build of ONE 4-line FASTQ / one-line FASTA file into colour 0: one
 * reader thread parses the file into a pool of read slots (asyncio_run_pool + async_io_reader,
 * src/basic/async_read_io.c:145-175,283-310: one reader per file, a 2048-slot message pool) and
 * `nthreads` workers take reads from it and load them (add_reads_to_graph, build_graph.c:233-254).
 * Returns the seconds spent (*insert_s: until the workers are done); the sorted .ctx is left in the
 * graph for orc_graph_write_ctx.  Test infrastructure: the baseline of bench.py's e2e figure. */
#include <time.h>
#define POOL_SLOTS 2048
typedef struct { char *seq; size_t len, cap; } pool_read;
typedef struct {
  orc_graph *g;
  pool_read slot[POOL_SLOTS];
  size_t head, tail;  /* ring: [tail, head) filled */
  int done;
  pthread_mutex_t mu;
  pthread_cond_t not_empty, not_full;
  orc_stats st;
} read_pool;
typedef struct { read_pool *pool; orc_stats st; uint64_t new_nodes; } pool_worker;

static void *pool_worker_run(void *a)
{
  pool_worker *w = a;
  read_pool *P = w->pool;
  pool_read mine = {NULL, 0, 0};
  if(P->g->tuned) tl_new_nodes = &w->new_nodes;
  for(;;) {
    pthread_mutex_lock(&P->mu);
    while(P->head == P->tail && !P->done) pthread_cond_wait(&P->not_empty, &P->mu);
    if(P->head == P->tail) { pthread_mutex_unlock(&P->mu); break; }
    pool_read *r = &P->slot[P->tail % POOL_SLOTS];
    pool_read tmp = *r; *r = mine; mine = tmp;  /* swap buffers: the slot is free again at once */
    P->tail++;
    pthread_cond_signal(&P->not_full);
    pthread_mutex_unlock(&P->mu);
    load_read(P->g, 0, mine.seq, mine.len, NULL, 0, 0, &w->st);
  }
  free(mine.seq);
  tl_new_nodes = NULL;
  return NULL;
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

double orc_build_file(orc_graph *g, const char *path, int nthreads, double *insert_s, orc_stats *stats_accum)
{
  FILE *f = fopen(path, "r");
  if(!f) return -1.0;
  if(nthreads < 1) nthreads = 1;
  read_pool *P = calloc(1, sizeof(*P));
  P->g = g;
  pthread_mutex_init(&P->mu, NULL);
  pthread_cond_init(&P->not_empty, NULL);
  pthread_cond_init(&P->not_full, NULL);
  pthread_t *th = calloc((size_t)nthreads, sizeof(*th));
  pool_worker *w = calloc((size_t)nthreads, sizeof(*w));
  const double t0 = now_s();
  int t;
  for(t = 0; t < nthreads; t++) { w[t].pool = P; pthread_create(&th[t], NULL, pool_worker_run, &w[t]); }
  /* the reader (this thread): header line, sequence line [, '+' line, quality line] */
  char *line = NULL;
  size_t cap = 0;
  ssize_t n;
  int fastq = -1;
  while((n = getline(&line, &cap, f)) > 0) {
    if(fastq < 0) fastq = line[0] == '@';
    if((n = getline(&line, &cap, f)) <= 0) break;
    while(n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) n--;
    pthread_mutex_lock(&P->mu);
    while(P->head - P->tail == POOL_SLOTS) pthread_cond_wait(&P->not_full, &P->mu);
    pool_read *r = &P->slot[P->head % POOL_SLOTS];
    if(r->cap < (size_t)n + 1) { r->cap = (size_t)n + 64; r->seq = realloc(r->seq, r->cap); }
    memcpy(r->seq, line, (size_t)n);
    r->len = (size_t)n;
    P->head++;
    pthread_cond_signal(&P->not_empty);
    pthread_mutex_unlock(&P->mu);
    if(fastq) { if(getline(&line, &cap, f) <= 0 || getline(&line, &cap, f) <= 0) break; }
  }
  free(line);
  fclose(f);
  pthread_mutex_lock(&P->mu);
  P->done = 1;
  pthread_cond_broadcast(&P->not_empty);
  pthread_mutex_unlock(&P->mu);
  for(t = 0; t < nthreads; t++) {
    pthread_join(th[t], NULL);
    if(stats_accum) stats_merge(stats_accum, &w[t].st);
    g->num_kmers += w[t].new_nodes;
  }
  if(insert_s) *insert_s = now_s() - t0;
  for(t = 0; t < POOL_SLOTS; t++) free(P->slot[t].seq);
  free(th); free(w); free(P);
  return now_s() - t0;
}

/* ---- build --remove-pcr -------------------------------------------------- */
/* seq_read_reverse_complement lives in the third-party seq_file library (libs/seq_file, an empty
 * submodule in the checkout): reverse the bases and qualities, complement A<->T, C<->G in either
 * case; every other byte splits contigs whatever it becomes, so it is left as it is. */
static void read_revcomp(char *seq, char *qual, size_t len)
{
  static const char from[] = "ACGTacgt", to[] = "TGCAtgca";
  size_t i;
  for(i = 0; i < len; i++) { const char *p = seq[i] ? strchr(from, seq[i]) : NULL; if(p) seq[i] = to[p - from]; }
  for(i = 0; i + 1 < len - i; i++) {
    char t = seq[i]; seq[i] = seq[len - 1 - i]; seq[len - 1 - i] = t;
    if(qual) { t = qual[i]; qual[i] = qual[len - 1 - i]; qual[len - 1 - i] = t; }
  }
}

/* ctx_build.c:392-395: the read-start bits are wiped when the colour being loaded changes */
void orc_graph_pcr_reset(orc_graph *g)
{
  if(g->readstrt) memset(g->readstrt, 0, (size_t)((2 * g->capacity + 7) / 8));
}

static inline int readstrt_get(const orc_graph *g, orc_node n)
{ uint64_t b = 2 * n.hkey + (uint64_t)n.orient; return (g->readstrt[b >> 3] >> (b & 7)) & 1; }
static inline void readstrt_set(orc_graph *g, orc_node n)
{ uint64_t b = 2 * n.hkey + (uint64_t)n.orient; g->readstrt[b >> 3] |= (uint8_t)(1u << (b & 7)); }

/* the contigs of one read, without the per-read counters of the caller (build_graph.c:154-189) */
static void load_read_contigs(orc_graph *g, int colour, const char *seq, size_t len, const char *qual,
                              uint8_t fq_cutoff, uint8_t hp_cutoff, orc_stats *st)
{
  orc_stats tmp; memset(&tmp, 0, sizeof(tmp));
  load_read(g, colour, seq, len, qual, fq_cutoff, hp_cutoff, &tmp);
  tmp.total_bases_read = 0; tmp.num_se_reads = 0;
  stats_merge(st, &tmp);
}

/* build_graph_from_reads_mt with prefs.remove_pcr_dups (build_graph.c:192-231) over a batch, in
 * read order on one thread (the reference's filter depends on the order in which its workers
 * reach the reads).  paired: reads 2i and 2i + 1 are mates.  matedir: 0 FF, 1 FR, 2 RF, 3 RR
 * (cortex_types.h:18-25).  The mates are turned to FF first and are LOADED in that orientation
 * (seq_reader_orient_mp_FF, seq_reader.c:506-510, called at build_graph.c:49).
 * counts[0] += duplicate SE reads, counts[1] += duplicate pairs, counts[2] += reads seen as pairs. */
int orc_graph_add_reads_pcr(orc_graph *g, int colour, const char *bases, const char *quals,
                            const uint64_t *offsets, uint64_t nreads, uint8_t fq_cutoff1, uint8_t fq_cutoff2,
                            uint8_t hp_cutoff, int paired, int matedir, orc_stats *st, uint64_t *counts)
{
  if(colour < 0 || colour >= g->ncols || (paired && (nreads & 1))) return -2;
  if(!g->readstrt) g->readstrt = calloc((size_t)((2 * g->capacity + 7) / 8), 1);
  const size_t k = (size_t)g->k;
  const uint64_t step = paired ? 2 : 1;
  uint64_t r;
  int m;
  for(r = 0; r < nreads && !g->full; r += step) {
    char *seq[2] = {NULL, NULL}, *qual[2] = {NULL, NULL};
    size_t len[2] = {0, 0};
    const uint8_t fq[2] = {fq_cutoff1, fq_cutoff2};
    const int nm = paired ? 2 : 1;
    for(m = 0; m < nm; m++) {
      len[m] = (size_t)(offsets[r + m + 1] - offsets[r + m]);
      seq[m] = malloc(len[m] + 1); memcpy(seq[m], bases + offsets[r + m], len[m]); seq[m][len[m]] = 0;
      if(quals) { qual[m] = malloc(len[m] + 1); memcpy(qual[m], quals + offsets[r + m], len[m]); qual[m][len[m]] = 0; }
      st->total_bases_read += len[m];
    }
    if(paired) counts[2] += 2; else st->num_se_reads += 1; /* build_graph.c:211-212 */
    /* seq_reads_are_novel, build_graph.c:35-92 */
    if(matedir & 2) read_revcomp(seq[0], qual[0], len[0]);            /* read_mate_r1 */
    if(paired && (matedir & 1)) read_revcomp(seq[1], qual[1], len[1]); /* read_mate_r2 */
    int got[2] = {0, 0}, found;
    orc_node node[2];
    uint64_t novel = 0;
    for(m = 0; m < nm; m++) {
      size_t start = orc_contig_start(seq[m], len[m], qual[m], qual[m] ? len[m] : 0, 0, k, fq[m], hp_cutoff);
      got[m] = start < len[m];
      if(got[m]) { /* db_graph_find_or_add_node_mt: the node is created, its coverage is not touched */
        orc_bkmer bk = orc_kmer_from_str(seq[m] + start, g->k), key = orc_kmer_get_key(bk, g->k);
        node[m].hkey = find_or_insert(g, &key, &found);
        node[m].orient = bkmer_eq(&key, &bk, g->W) ? 0 : 1;
        if(node[m].hkey == ORC_NOT_FOUND) { g->full = 1; got[m] = 0; break; }
        novel += !found;
      }
    }
    st->num_kmers_novel += novel;
    int dup = (!got[0] || readstrt_get(g, node[0])) && (!got[1] || readstrt_get(g, node[1]));
    if(dup) counts[paired ? 1 : 0] += 1;
    else {
      for(m = 0; m < nm; m++) if(got[m]) readstrt_set(g, node[m]);
      for(m = 0; m < nm; m++) load_read_contigs(g, colour, seq[m], len[m], qual[m], fq[m], hp_cutoff, st);
    }
    for(m = 0; m < nm; m++) { free(seq[m]); free(qual[m]); }
  }
  return g->full ? -1 : 0;
}

/* ------------------------------------------------------------------------ */
/* Row H: GraphInfo + .ctx v6 image                                          */
/* ------------------------------------------------------------------------ */

/* graph_info.c:116-133 */
static void ginfo_update_contigs(uint32_t *mean, uint64_t *total, uint64_t added, uint64_t ncontigs)
{
  if(!added && !ncontigs) return;
  size_t have = 0;
  if(*total && *mean) have = (size_t)(((double)*total / *mean) + 0.5);
  if(have + ncontigs > 0) *mean = (uint32_t)((double)(*total + added) / (double)(have + ncontigs));
  *total += added;
}

/* graph_info.c:172-175 */
void orc_graph_update_stats(orc_graph *g, int colour, const orc_stats *st)
{
  ginfo_update_contigs(&g->mean_read_length[colour], &g->total_sequence[colour],
                       st->total_bases_loaded, st->contigs_parsed);
}

/* Header GraphInfo after graph_writer_mkhdr (graph_writer.c:11-30) merges the
 * graph's ginfo into a freshly initialised one (graph_info.c:135-170). */
static void hdr_ginfo(const orc_graph *g, int c, uint32_t *mean, uint64_t *total, long double *seq_err)
{
  /* dst: graph_info_init -> total 0, mean 0, seq_err 0.01 (double constant) */
  uint32_t dmean = 0; uint64_t dtotal = 0; long double derr = 0.01;
  /* src: the graph's ginfo; its seq_err was also initialised to (double)0.01 */
  const long double serr = 0.01;
  const uint64_t stotal = g->total_sequence[c];
  const uint32_t smean = g->mean_read_length[c];
  uint64_t tot = dtotal + stotal;
  if(tot > 0) {
    derr = (derr * dtotal + serr * stotal) / tot;
    size_t src_contigs = 0;
    if(stotal && smean) src_contigs = (size_t)(((double)stotal / smean) + 0.5);
    ginfo_update_contigs(&dmean, &dtotal, stotal, src_contigs);
  }
  dtotal = tot;
  *mean = dmean; *total = dtotal; *seq_err = derr;
}

/* graph_load's per-record body (src/graph/graphs_load.c:117-186) once graph_file_read_reset
 * (graph_file_reader.c:392-420) has mapped the file's colours onto the graph's: covgs/edges hold
 * g->ncols entries.  Returns 1 = loaded, 0 = skipped (no coverage / absent with must_exist),
 * -1 = hash table full. */
int orc_graph_add_record(orc_graph *g, const uint64_t *key_words, const uint32_t *covgs, const uint8_t *edges,
                         int must_exist)
{
  int c, found = 0;
  uint32_t keep = 0;
  for(c = 0; c < g->ncols; c++) keep |= covgs[c];
  if(keep == 0) return 0; /* "If kmer has no covg -> don't load" */
  orc_bkmer key;
  memset(&key, 0, sizeof(key));
  for(c = 0; c < g->W; c++) key.b[c] = key_words[c];
  uint64_t hkey;
  uint8_t edge_mask = 0xff;
  if(must_exist) {
    hkey = find_only(g, &key);
    if(hkey == ORC_NOT_FOUND) return 0;
    if(g->isec_edges) edge_mask = g->isec_edges[hkey]; /* prefs.must_exist_in_edges: graphs_load.c:166-167 */
  } else {
    hkey = find_or_insert(g, &key, &found);
    if(hkey == ORC_NOT_FOUND) { g->full = 1; return -1; }
  }
  for(c = 0; c < g->ncols; c++) { /* db_node_add_col_covg: SAFE_SUM_COVG, cortex_types.h:10-11 */
    uint32_t *cv = &g->covgs[hkey * (uint64_t)g->ncols + (uint64_t)c];
    *cv = ((uint64_t)*cv + covgs[c] > UINT32_MAX) ? UINT32_MAX : *cv + covgs[c];
    g->edges[hkey * (uint64_t)g->ncols + (uint64_t)c] |= edges[c] & edge_mask;
  }
  return 1;
}

/* ---- build --intersect ------------------------------------------------------------------ */
void orc_graph_set_must_exist(orc_graph *g, int on) { g->must_exist = on; }

/* One record of an intersection graph: graph_load with col_covgs == NULL and col_edges ==
 * isec_edges, one edge column, filter flattened into colour 0 (ctx_build.c:197-203,347-361;
 * graphs_load.c:117-186).  covg_sum / edges_or are the flattened values of the record. */
int orc_graph_add_isec_record(orc_graph *g, const uint64_t *key_words, uint32_t covg_sum, uint8_t edges_or)
{
  int c, found = 0;
  if(!g->isec_edges) g->isec_edges = calloc(g->capacity, 1); /* ctx_build.c:341-343 */
  if(covg_sum == 0) return 0;
  orc_bkmer key;
  memset(&key, 0, sizeof(key));
  for(c = 0; c < g->W; c++) key.b[c] = key_words[c];
  uint64_t hkey = find_or_insert(g, &key, &found);
  if(hkey == ORC_NOT_FOUND) { g->full = 1; return -1; }
  g->isec_edges[hkey] |= edges_or;
  return 1;
}

/* db_graph_remove_no_covg_kmers + db_graph_intersect_edges (db_graph.c:630-673, ctx_build.c:409-413) */
void orc_graph_isec_finish(orc_graph *g)
{
  uint64_t s;
  int c;
  if(!g->isec_edges) return;
  for(s = 0; s < g->capacity; s++) {
    if(!(g->table[s * (uint64_t)g->W] & ORC_FLAG)) continue;
    uint32_t covg = 0;
    for(c = 0; c < g->ncols; c++) covg |= g->covgs[s * (uint64_t)g->ncols + (uint64_t)c];
    if(!covg) { /* hash_table_delete: hash_table.c:285-299 */
      memset(g->table + s * (uint64_t)g->W, 0, sizeof(uint64_t) * (size_t)g->W);
      g->num_kmers--;
    }
  }
  for(s = 0; s < g->capacity; s++)
    for(c = 0; c < g->ncols; c++) g->edges[s * (uint64_t)g->ncols + (uint64_t)c] &= g->isec_edges[s];
}

size_t orc_graph_header_size(const orc_graph *g)
{
  size_t n = 6 + 16, c;
  n += (size_t)g->ncols * 12;
  for(c = 0; c < (size_t)g->ncols; c++) n += 4 + strlen(g->sample[c]);
  n += (size_t)g->ncols * 16;
  n += (size_t)g->ncols * (4 + 8 + 4 + 9);
  return n + 6;
}

size_t orc_graph_ctx_size(const orc_graph *g)
{
  return orc_graph_header_size(g) + g->num_kmers * ((size_t)g->W * 8 + 5 * (size_t)g->ncols);
}

static int g_sort_W;
static const uint64_t *g_sort_table;
static int cmp_slots(const void *a, const void *b)
{ /* binary_kmers_qcmp_ptrs: hash_table.c:371 (flag set on all, order unaffected) */
  const uint64_t *x = g_sort_table + *(const uint64_t *)a * (uint64_t)g_sort_W;
  const uint64_t *y = g_sort_table + *(const uint64_t *)b * (uint64_t)g_sort_W;
  int i;
  for(i = 0; i < g_sort_W; i++) if(x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
  return 0;
}

#define PUT(ptr, val, type) do { type _v = (type)(val); memcpy(ptr, &_v, sizeof(type)); ptr += sizeof(type); } while(0)

/* graph_writer.c:62-110 (header), :116-127 (record), :182-268 (iteration) */
size_t orc_graph_write_ctx(const orc_graph *g, int sorted, uint8_t *out)
{
  uint8_t *p = out;
  int c;
  memcpy(p, "CORTEX", 6); p += 6;
  PUT(p, 6, uint32_t); PUT(p, g->k, uint32_t); PUT(p, g->W, uint32_t); PUT(p, g->ncols, uint32_t);
  uint32_t mean[64]; uint64_t total[64]; long double err[64];
  if(g->ncols > 64) return 0;
  for(c = 0; c < g->ncols; c++) hdr_ginfo(g, c, &mean[c], &total[c], &err[c]);
  for(c = 0; c < g->ncols; c++) PUT(p, mean[c], uint32_t);
  for(c = 0; c < g->ncols; c++) PUT(p, total[c], uint64_t);
  for(c = 0; c < g->ncols; c++) {
    uint32_t len = (uint32_t)strlen(g->sample[c]);
    PUT(p, len, uint32_t); memcpy(p, g->sample[c], len); p += len;
  }
  for(c = 0; c < g->ncols; c++) { /* 10-byte x87 value + 6 zero bytes (calloc'd header) */
    memset(p, 0, 16); memcpy(p, &err[c], 10); p += 16;
  }
  for(c = 0; c < g->ncols; c++) { /* write_error_cleaning_object: graph_writer.c:33-59 */
    memset(p, 0, 4 + 8); p += 12;
    PUT(p, 9, uint32_t); memcpy(p, "undefined", 9); p += 9;
  }
  memcpy(p, "CORTEX", 6); p += 6;

  /* occupied slots in table order (HASH_ITERATE, hash_table.h:102-111) */
  uint64_t n = 0, s, *order = malloc((g->num_kmers ? g->num_kmers : 1) * sizeof(uint64_t));
  for(s = 0; s < g->capacity; s++) if(g->table[s * (uint64_t)g->W] & ORC_FLAG) order[n++] = s;
  if(sorted) { g_sort_W = g->W; g_sort_table = g->table; qsort(order, n, sizeof(uint64_t), cmp_slots); }
  for(s = 0; s < n; s++) {
    uint64_t slot = order[s];
    int w;
    PUT(p, g->table[slot * (uint64_t)g->W] & ~ORC_FLAG & ~(1ULL << 62), uint64_t); /* hash_table_fetch */
    for(w = 1; w < g->W; w++) PUT(p, g->table[slot * (uint64_t)g->W + (uint64_t)w], uint64_t);
    memcpy(p, &g->covgs[slot * (uint64_t)g->ncols], 4 * (size_t)g->ncols); p += 4 * (size_t)g->ncols;
    memcpy(p, &g->edges[slot * (uint64_t)g->ncols], (size_t)g->ncols); p += g->ncols;
  }
  free(order);
  return (size_t)(p - out);
}

int orc_graph_lookup(const orc_graph *g, const char *kmer, uint32_t *covgs, uint8_t *edges)
{
  orc_bkmer key = orc_kmer_get_key(orc_kmer_from_str(kmer, g->k), g->k);
  int i, w;
  for(i = 0; i < ORC_REHASH_LIMIT; i++) { /* hash_table_find: hash_table.c:125-.. */
    uint64_t h = orc_kmer_hash(key, g->k, g->seed + (uint32_t)i) & g->hash_mask;
    const uint64_t *slot = g->table + h * g->bucket_size * (uint64_t)g->W;
    unsigned j;
    for(j = 0; j < g->bsize[h]; j++, slot += g->W) {
      if(slot[0] != (key.b[0] | ORC_FLAG)) continue;
      for(w = 1; w < g->W && slot[w] == key.b[w]; w++) {}
      if(w < g->W) continue;
      uint64_t hk = h * g->bucket_size + j;
      if(covgs) memcpy(covgs, &g->covgs[hk * (uint64_t)g->ncols], 4 * (size_t)g->ncols);
      if(edges) memcpy(edges, &g->edges[hk * (uint64_t)g->ncols], (size_t)g->ncols);
      return 1;
    }
    if(g->bsize[h] < g->bucket_size) return 0;
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Tuple stream: the closed form of SURVEY 0.3, derived from db_graph.c:152-166 */
/* ------------------------------------------------------------------------ */
uint64_t orc_tuples(int k, const char *bases, const char *quals, const uint64_t *offsets,
                    uint64_t nreads, uint8_t fq_cutoff, uint8_t hp_cutoff,
                    uint64_t *keys, uint8_t *edges)
{
  const int W = orc_words_for_k(k);
  uint64_t n = 0, r;
  for(r = 0; r < nreads; r++) {
    const char *seq = bases + offsets[r];
    const char *qual = quals ? quals + offsets[r] : NULL;
    size_t len = (size_t)(offsets[r + 1] - offsets[r]), qlen = qual ? len : 0;
    size_t cs, ce, search = 0;
    while((cs = orc_contig_start(seq, len, qual, qlen, search, (size_t)k, fq_cutoff, hp_cutoff)) < len) {
      ce = orc_contig_end(seq, len, qual, qlen, cs, (size_t)k, fq_cutoff, hp_cutoff, &search);
      orc_bkmer bk = orc_kmer_from_str(seq + cs, k);
      size_t i;
      for(i = cs; i + (size_t)k <= ce; i++) {
        if(i > cs) bk = orc_kmer_shift_add(bk, k, orc_char_to_nuc((unsigned char)seq[i + (size_t)k - 1]));
        orc_bkmer key = orc_kmer_get_key(bk, k);
        unsigned o = bkmer_eq(&key, &bk, W) ? 0u : 1u;
        unsigned e = 0;
        if(i + (size_t)k < ce) e |= 1u << ((unsigned)orc_char_to_nuc((unsigned char)seq[i + (size_t)k]) + 4u * o);
        if(i > cs) e |= 1u << ((3u - (unsigned)orc_char_to_nuc((unsigned char)seq[i - 1])) + 4u * (1u - o));
        if(keys) { int w; for(w = 0; w < W; w++) keys[n * (uint64_t)W + (uint64_t)w] = key.b[w]; }
        if(edges) edges[n] = (uint8_t)e;
        n++;
      }
    }
  }
  return n;
}
