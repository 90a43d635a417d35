/* host.h -- C host side of `mccortex<K> build` over the MI355X backend (include/mcx_gpu.h).
 * Mirrors the reference's command surface: src/main/mccortex.c (dispatcher),
 * src/commands/ctx_build.c (options, batching, sizing), src/graph/graph_writer.c (.ctx v6). */
#ifndef MCX_HOST_H_
#define MCX_HOST_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifndef MAX_KMER_SIZE
#define MAX_KMER_SIZE 31
#endif
/* MIN_KMER_SIZE = MAXK - 30, and 3 for MAXK = 31 (reference Makefile:33-48); literals, because the usage texts print them */
#if MAX_KMER_SIZE == 31
#define MIN_KMER_SIZE 3
#elif MAX_KMER_SIZE == 63
#define MIN_KMER_SIZE 33
#elif MAX_KMER_SIZE == 95
#define MIN_KMER_SIZE 65
#elif MAX_KMER_SIZE == 127
#define MIN_KMER_SIZE 97
#else
#error "MAX_KMER_SIZE must be 31, 63, 95 or 127 (three- and four-word keys are the widest the backend builds)"
#endif
#define MCX_STR_(x) #x
#define MCX_STR(x) MCX_STR_(x)
#define CMD_NAME "mccortex" MCX_STR(MAX_KMER_SIZE)

/* ---- logging (src/global/ctx_output.h:14-34) ---- */
extern FILE *msg_out; /* NULL = quiet */
extern int host_fast_exit_ok; /* set by a `build` that finished and saw its device idle: main() may _exit */
void status(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void warn(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void die(const char *fmt, ...) __attribute__((format(printf, 1, 2), noreturn));
void print_usage(const char *usage, const char *errfmt, ...) __attribute__((noreturn));
void host_set_cmdline(int argc, char **argv);

/* ---- number parsing / formatting (src/global/util.c:108-117,206-222,251-264,343) ---- */
bool parse_entire_size(const char *s, size_t *out);
bool parse_entire_uint(const char *s, unsigned *out);
bool mem_to_integer(const char *s, size_t *bytes);
char *ulong_to_str(unsigned long n, char *out);
char *bytes_to_str(unsigned long n, int decimals, char *out);
/* binary_kmer_to_str (binary_kmer.c:190-210) for the key of a .ctx record: word 0 (most significant) first */
void kmer_words_to_str(const unsigned char *rec, unsigned kmer_size, char *out);

/* ---- table sizing: entry counts and memory figures of -n / -m as the reference computes them
 * (src/basic/hash_mem.c:5-51, src/graph/cmd_mem.c:38-130) ---- */
typedef struct { uint64_t nbuckets, bucket_size, capacity; size_t bytes; } table_plan;
table_plan table_plan_for_kmers(uint64_t nkmers, size_t entry_bits);
table_plan table_plan_for_memory(size_t mem, size_t entry_bits);
const char *table_plan_for_build(size_t mem_to_use, bool mem_set, size_t num_kmers, bool nkmers_set, size_t entry_bits,
                                 int64_t max_kmers, table_plan *out, char *errbuf, size_t errlen);

/* ---- sequence input (replaces seq_file + src/basic/async_read_io.c for FASTA/FASTQ/plain, .gz) ---- */
typedef enum { SEQ_FMT_UNKNOWN = 0, SEQ_FMT_FASTA, SEQ_FMT_FASTQ, SEQ_FMT_PLAIN, SEQ_FMT_SAM } seq_fmt;
typedef struct seq_in seq_in;
seq_in *seq_in_open(const char *path); /* "-" = stdin; NULL on failure */
void seq_in_close(seq_in *s);
seq_fmt seq_in_format(seq_in *s);
const char *seq_in_path(const seq_in *s);
/* Appends reads to the batch until >= max_bases are held or EOF.  Returns reads appended,
 * 0 at EOF.  bases/quals are concatenated, offsets has nreads+1 entries. */
typedef struct {
  uint8_t *bases, *quals;
  uint64_t *offsets;
  size_t nreads, nbases, cap_bases, cap_reads;
  bool want_quals;
} read_batch;
void read_batch_init(read_batch *b, bool want_quals);
void read_batch_clear(read_batch *b);
void read_batch_free(read_batch *b);
size_t seq_in_fill(seq_in *s, read_batch *b, size_t max_bases);
void read_batch_append(read_batch *dst, const read_batch *src, size_t i); /* read i of src */
/* FASTQ offset guess from the qualities seen so far (33 or 64); 0 if no qualities */
int seq_in_guess_fq_offset(const seq_in *s);
int fq_offset_from_range(int qmin, int qmax);
/* the offset of a file from its first 1000 records; 0 = no qualities, -1 = stdin (cannot look ahead) */
int fq_offset_probe(const char *path);

/* Multi-threaded parse of an uncompressed regular file (par_ingest.c): `submit` is called on the
 * calling thread for every batch; `started` (may be NULL) once, on the calling thread, after the
 * parser threads have been started and before the first batch is waited for.  Returns 0 = done,
 * 1 = not suitable (nothing submitted: use the sequential parser), 2 = irregular record met after
 * submission began. */
int par_ingest(const char *path, seq_fmt fmt, int nthreads, bool want_quals, size_t batch_bases,
               void (*submit)(void *arg, read_batch *b, int fq_offset_guess), void (*started)(void *arg), void *arg);

/* ---- .ctx header: GraphInfo arithmetic, writer, reader
 * (src/basic/graph_info.c, src/graph/graph_writer.c:11-110, src/graph/graph_file_reader.c:78-340,
 *  colour filters src/basic/file_filter.c) ---- */
typedef struct { /* ErrorCleaning, graph_info.h */
  uint8_t cleaned_tips, cleaned_unitigs, cleaned_kmers, is_graph_intersection;
  uint32_t clean_unitigs_thresh, clean_kmers_thresh;
  char *intersection_name;
} err_cleaning;
typedef struct { /* GraphInfo */
  uint32_t mean_read_length;
  uint64_t total_sequence;
  long double seq_err;
  char *name;
  err_cleaning cleaning;
} col_info;
void col_info_init(col_info *c);
void col_info_free(col_info *c);
void col_info_set_name(col_info *c, const char *name);
void col_info_update(col_info *c, uint64_t bases_loaded, uint64_t contigs);
void col_info_merge(col_info *dst, const col_info *src); /* graph_info_merge */
size_t ctx_write_header(FILE *fh, uint32_t kmer_size, uint32_t ncols, const col_info *cols);

typedef struct { uint32_t from, into; } col_filter;
typedef struct {
  char *input, *path;        /* "0,1:in.ctx:2-3" and "in.ctx" */
  FILE *fh;
  uint32_t version, kmer_size, num_words, num_cols;
  col_info *ginfo;           /* [num_cols] */
  size_t hdr_size;
  long long file_size, num_kmers; /* -1 when reading a stream */
  col_filter *filter;        /* sorted by `into` */
  size_t nfilter, into_ncols;
} ctx_reader;
/* graph_file_open2: parse "<into>:path:<from>", read and check the header; dies on error */
void ctx_reader_open(ctx_reader *r, const char *input, size_t into_offset, size_t min_k, size_t max_k);
void ctx_reader_open_mode(ctx_reader *r, const char *input, const char *mode, size_t into_offset, size_t min_k, size_t max_k);
bool ctx_reader_from_direct(const ctx_reader *r); /* file_filter_from_direct: no colour filter in the path */
/* graph_write_header (graph_writer.c:62-110): the parsed header as it is, no merging */
size_t ctx_write_header_raw(FILE *fh, const ctx_reader *r);
void ctx_reader_close(ctx_reader *r);

/* ---- commands ---- */
int ctx_build(int argc, char **argv);
int ctx_sort(int argc, char **argv);
int ctx_index(int argc, char **argv);
int ctx_hashtest(int argc, char **argv); /* src/commands/ctx_exp_hashtest.c */

#endif
