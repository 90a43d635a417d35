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
#if MAX_KMER_SIZE == 31
#define MIN_KMER_SIZE 3 /* reference Makefile:48 */
#else
#define MIN_KMER_SIZE (MAX_KMER_SIZE - 30)
#endif
#define MCX_STR_(x) #x
#define MCX_STR(x) MCX_STR_(x)
#define CMD_NAME "mccortex" MCX_STR(MAX_KMER_SIZE)

/* ---- logging (src/global/ctx_output.h:14-34) ---- */
extern FILE *msg_out; /* NULL = quiet */
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

/* ---- table sizing (src/basic/hash_mem.c:5-51, src/graph/cmd_mem.c:38-130) ---- */
uint64_t hash_table_cap(uint64_t nkmers, uint64_t *nbuckets, uint8_t *bucket_size);
size_t hash_table_mem(uint64_t nkmers, size_t entrybits, uint64_t *nkmers_out);
size_t hash_table_mem_limit(size_t memlimit, size_t entrybits, uint64_t *nkmers_out);

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
/* FASTQ offset guess from the qualities seen so far (33 or 64); 0 if no qualities */
int seq_in_guess_fq_offset(const seq_in *s);

/* Multi-threaded parse of an uncompressed regular file (par_ingest.c): `submit` is called on the
 * calling thread for every batch.  Returns 0 = done, 1 = not suitable (nothing submitted: use the
 * sequential parser), 2 = irregular record met after submission began. */
int par_ingest(const char *path, seq_fmt fmt, int nthreads, bool want_quals, size_t batch_bases,
               void (*submit)(void *arg, read_batch *b, int fq_offset_guess), void *arg);

/* ---- .ctx v6 header (src/graph/graph_writer.c:11-110, src/basic/graph_info.c:116-175) ---- */
typedef struct {
  uint32_t mean_read_length;
  uint64_t total_sequence;
  char name[256];
} col_info;
void col_info_init(col_info *c);
void col_info_update(col_info *c, uint64_t bases_loaded, uint64_t contigs);
size_t ctx_write_header(FILE *fh, uint32_t kmer_size, uint32_t ncols, const col_info *cols);

/* ---- commands ---- */
int ctx_build(int argc, char **argv);

#endif
