/*
 * mcx_oracle.h -- CPU restatement of the McCortex `build` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (mccortex_amd/, include/,
 * the host CLI) may include, link or call this.  Only tests/, smoke() and
 * bench.py's cpu_baseline leg use it, as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.  Parity pinning status: see oracle/README.md.
 */
#ifndef MCX_ORACLE_H_
#define MCX_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_W 4 /* up to MAXK=127 */

typedef struct { uint64_t b[ORC_MAX_W]; } orc_bkmer; /* b[0] = most significant */

/* src/basic/seq_loading_stats.h:5-14 (fields on the build path) */
typedef struct {
  uint64_t num_se_reads, num_good_reads, num_bad_reads;
  uint64_t total_bases_read, total_bases_loaded, contigs_parsed;
  uint64_t num_kmers_loaded, num_kmers_novel;
} orc_stats;

typedef struct orc_graph orc_graph;

/* ---- primitives (rows A-C of SURVEY 8a) ---- */
int       orc_words_for_k(int k);                                  /* binary_kmer.h:10 */
int       orc_char_to_nuc(unsigned char c);                        /* dna.c:8-25 */
orc_bkmer orc_kmer_from_str(const char *seq, int k);               /* binary_kmer.c:156-186 */
orc_bkmer orc_kmer_shift_add(orc_bkmer x, int k, int nuc);         /* binary_kmer.h:139-167, .c:80-97 */
orc_bkmer orc_kmer_revcomp(orc_bkmer x, int k);                    /* binary_kmer.c:102-133 */
orc_bkmer orc_kmer_get_key(orc_bkmer x, int k);                    /* binary_kmer.c:43-57 */
uint32_t  orc_kmer_hash(orc_bkmer key, int k, uint32_t initval);   /* kmer_hash.h:162-211 */
/* `hashtest -F` (src/commands/ctx_exp_hashtest.c:61-66,160-175): [0, n) in nparts ranges (range i starts at
 * i * (n / nparts), the last one ends at n); per range the XOR of binary_kmer_hash(bkmer with b[0] = i, 0); the sum. */
uint64_t  orc_hashtest_func(int k, uint64_t n, uint32_t nparts);
void      orc_kmer_to_str(orc_bkmer x, int k, char *out);          /* binary_kmer.c:190-.. */
int       orc_rehash_limit(void);     /* REHASH_LIMIT, src/basic/hash_mem.h:4 */
int       orc_max_bucket_size(void);  /* MAX_BUCKET_SIZE, src/basic/hash_mem.h:8 */
uint64_t  orc_hash_table_cap(uint64_t nkmers, uint64_t *nbuckets, uint8_t *bucket_size); /* hash_mem.c:5-15 */

/* contig splitting (row G): seq_reader.c:61-117 and :127-172 */
size_t orc_contig_start(const char *seq, size_t seqlen, const char *qual, size_t quallen,
                        size_t offset, size_t k, uint8_t qual_cutoff, uint8_t hp_cutoff);
size_t orc_contig_end(const char *seq, size_t seqlen, const char *qual, size_t quallen,
                      size_t contig_start, size_t k, uint8_t qual_cutoff, uint8_t hp_cutoff,
                      size_t *search_start);

/* ---- graph (rows D-F) ---- */
orc_graph *orc_graph_new(int k, int ncols, uint64_t capacity_kmers, uint32_t seed);
void       orc_graph_free(orc_graph *g);
int        orc_graph_set_sample(orc_graph *g, int col, const char *name);
void       orc_graph_force_generic(orc_graph *g, int on); /* W==1: use the multi-word code path */
uint64_t   orc_graph_nkmers(const orc_graph *g);
uint64_t   orc_graph_capacity(const orc_graph *g);

/* build_graph.c:122-231.  Reads are bases[offsets[i] .. offsets[i+1]) (no
 * separators), quals NULL or same layout.  fq_cutoff is the absolute cutoff
 * (prefs.fq_cutoff + fq offset).  nthreads>1 splits the reads over pthreads
 * (the reference's worker pool, async_read_io.c:283-310).
 * Returns 0, or -1 if the table filled up (hash_table.c:119-123). */
int orc_graph_add_reads(orc_graph *g, int colour, const char *bases, const char *quals,
                        const uint64_t *offsets, uint64_t nreads,
                        uint8_t fq_cutoff, uint8_t hp_cutoff, int nthreads,
                        orc_stats *stats_accum);

/* build --remove-pcr: build_graph_from_reads_mt with prefs.remove_pcr_dups (build_graph.c:28-92,
 * 192-231), in read order on one thread.  paired: reads 2i, 2i+1 are mates; matedir 0 FF, 1 FR,
 * 2 RF, 3 RR.  counts[0] += duplicate SE reads, [1] += duplicate pairs, [2] += reads seen as pairs. */
int orc_graph_add_reads_pcr(orc_graph *g, int colour, const char *bases, const char *quals,
                            const uint64_t *offsets, uint64_t nreads, uint8_t fq_cutoff1, uint8_t fq_cutoff2,
                            uint8_t hp_cutoff, int paired, int matedir, orc_stats *stats_accum, uint64_t *counts);
void orc_graph_pcr_reset(orc_graph *g); /* ctx_build.c:392-395 */

/* graph_info.c:172-175 (called once per input file, in task order) */
void orc_graph_update_stats(orc_graph *g, int colour, const orc_stats *stats);

/* .ctx v6 image (row H): graph_writer.c:11-30,62-127,182-268.
 * sorted!=0 -> records ordered by key (hash_table.c:362-374). */
/* one record of graph_load (graphs_load.c:117-186), colours already mapped onto the graph's */
int orc_graph_add_record(orc_graph *g, const uint64_t *key_words, const uint32_t *covgs, const uint8_t *edges,
                         int must_exist);
/* build --intersect (ctx_build.c:341-363,384-413): intersection-graph records, must-exist reads,
 * final removal of k-mers without coverage + edge intersection */
void orc_graph_set_must_exist(orc_graph *g, int on);
int  orc_graph_add_isec_record(orc_graph *g, const uint64_t *key_words, uint32_t covg_sum, uint8_t edges_or);
void orc_graph_isec_finish(orc_graph *g);
/* timed-baseline helpers (bench.py's cpu_baseline leg): see mcx_oracle.c */
void orc_graph_tune(orc_graph *g, int nthreads);
double orc_build_file(orc_graph *g, const char *path, int nthreads, double *insert_s, orc_stats *stats_accum);
size_t orc_graph_ctx_size(const orc_graph *g);
size_t orc_graph_write_ctx(const orc_graph *g, int sorted, uint8_t *out);
size_t orc_graph_header_size(const orc_graph *g);

/* Look up one k-mer given as a string; returns 1 if present. */
int orc_graph_lookup(const orc_graph *g, const char *kmer, uint32_t *covgs, uint8_t *edges);

/* Per-occurrence tuple stream: for every k-mer occurrence in read
 * order emits W key words + edge byte.  keys: n*W u64, edges: n bytes.  Returns
 * the number of tuples (call with NULL outputs to count). */
uint64_t orc_tuples(int k, const char *bases, const char *quals, const uint64_t *offsets,
                    uint64_t nreads, uint8_t fq_cutoff, uint8_t hp_cutoff,
                    uint64_t *keys, uint8_t *edges);

#ifdef __cplusplus
}
#endif
#endif
