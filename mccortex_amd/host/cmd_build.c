/* cmd_build.c -- `mccortex<K> build`: the reference's command surface
 * (src/commands/ctx_build.c:13-77 options, :133-242 parsing rules, :245-436 flow) driving the
 * MI355X backend through include/mcx_gpu.h instead of build_graph() / graph_writer. */
#define _GNU_SOURCE
#include "host.h"
#include <time.h>

#include <ctype.h>
#include <errno.h>
#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/mcx_gpu.h"

#define DEFAULT_NTHREADS 2
#define DEFAULT_MEM (1UL << 29)
#define IDEAL_OCCUPANCY 0.75f
#define WARN_OCCUPANCY 0.9f
#define BATCH_BASES (48u << 20)
/* batches of the parallel parser: small enough that submitting (one thread) overlaps with parsing */
#define PAR_BATCH_BASES (32u << 20)

static const char build_usage[] =
"usage: " CMD_NAME " build [options] <out.ctx>\n"
"\n"
"  Build a cortex graph.  \n"
"\n"
"  -h, --help               This help message\n"
"  -q, --quiet              Silence status output normally printed to STDERR\n"
"  -f, --force              Overwrite output files\n"
"  -m, --memory <mem>       Memory to use\n"
"  -n, --nkmers <kmers>     Number of hash table entries (e.g. 1G ~ 1 billion)\n"
"  -t, --threads <T>        Number of threads to use [default: " MCX_STR(DEFAULT_NTHREADS) "]\n"
"  -k, --kmer <kmer>        Kmer size must be odd (" MCX_STR(MAX_KMER_SIZE) " >= k >= " MCX_STR(MIN_KMER_SIZE) ")\n"
"  -s, --sample <name>      Sample name (required before any seq args)\n"
"  -1, --seq <in.fa>        Load sequence data\n"
"  -2, --seq2 <in1:in2>     Load paired end sequence data\n"
"  -i, --seqi <in.bam>      Load paired end sequence from a single file\n"
"  -Q, --fq-cutoff <Q>      Filter quality scores [default: 0 (off)]\n"
"  -O, --fq-offset <N>      FASTQ ASCII offset    [default: 0 (auto-detect)]\n"
"  -H, --cut-hp <bp>        Breaks reads at homopolymers >= <bp> [default: off]\n"
"  -p, --remove-pcr         Remove (or keep) PCR duplicate reads\n"
"  -P, --keep-pcr           Don't do PCR duplicate removal [default]\n"
"  -M, --matepair <orient>  Mate pair orientation: FF,FR,RF,RR [default: FR]\n"
"                           (for --keep_pcr only)\n"
"  -g, --graph <in.ctx>     Load samples from a graph file (.ctx)\n"
"  -I, --intersect <i.ctx>  Only load kmers that appear in i.ctx. Multiple -I\n"
"                           graphs will be merged, not intersected. Treated as\n"
"                           single colour graphs.\n"
"  -S, --sort               Output a graph file ordered by kmer\n"
"  -D, --device <N[,N..]>   GPU(s) to build on: one table split over 1, 2, 4, 8.. devices [default: 0]\n"
"\n"
"  Note: Argument must come before input file\n"
"  --sample <name> is required before sequence input can be loaded.\n"
"  Consecutive sequence options are loaded into the same colour.\n"
"  This build runs the graph construction on an MI355X: SAM/BAM/CRAM input is not\n"
"  available; -m/-n size the table in HBM.\n"
"\n";

static struct option longopts[] = {
  {"help", no_argument, NULL, 'h'},       {"memory", required_argument, NULL, 'm'},
  {"nkmers", required_argument, NULL, 'n'}, {"threads", required_argument, NULL, 't'},
  {"force", no_argument, NULL, 'f'},      {"kmer", required_argument, NULL, 'k'},
  {"sample", required_argument, NULL, 's'}, {"sort", no_argument, NULL, 'S'},
  {"seq", required_argument, NULL, '1'},  {"seq2", required_argument, NULL, '2'},
  {"seqi", required_argument, NULL, 'i'}, {"matepair", required_argument, NULL, 'M'},
  {"fq-cutoff", required_argument, NULL, 'Q'}, {"fq-offset", required_argument, NULL, 'O'},
  {"cut-hp", required_argument, NULL, 'H'}, {"remove-pcr", no_argument, NULL, 'p'},
  {"keep-pcr", no_argument, NULL, 'P'},   {"graph", required_argument, NULL, 'g'},
  {"intersect", required_argument, NULL, 'I'}, {"device", required_argument, NULL, 'D'},
  {NULL, 0, NULL, 0}};

typedef struct {
  char *path, *path2;   /* path2: second mate file of a --seq2 task kept paired (--remove-pcr) */
  int colour;
  uint8_t fq_cutoff, fq_offset, hp_cutoff;
  int fq_detected, fq_detected2; /* offset of path / path2 from the head of the file (fq_offset_probe); 0 = not yet known */
  bool remove_pcr, interleaved; /* interleaved: --seqi with --remove-pcr, reads 2i / 2i+1 are mates */
  int matedir;          /* 0 FF, 1 FR, 2 RF, 3 RR (cortex_types.h:18-21) */
  mcx_load_stats stats;
  seq_fmt fmt;
} build_task;

static build_task *tasks = NULL;
static size_t ntasks = 0, tasks_cap = 0;
static char **sample_names = NULL;
static int *sample_cols = NULL;   /* SampleName.colour: ctx_build.c:83-86,159-164 */
static size_t nsamples = 0;
static ctx_reader *gfiles = NULL; /* --graph inputs, in command-line order (gfilebuf) */
static size_t ngfiles = 0;
static ctx_reader *gisec = NULL;  /* --intersect inputs (gisecbuf) */
static size_t ngisec = 0;

#define usage_die(...) print_usage(build_usage, __VA_ARGS__)

static void optname(char c, char *out)
{ /* "-k, --kmer" style, cmd_get_longopt_str (cmd.c:66-84) */
  sprintf(out, "-%c, --Unknown", c);
  for (int i = 0; longopts[i].name; i++)
    if (longopts[i].val == c) sprintf(out, "-%c, --%s", c, longopts[i].name);
}

/* ctx_build.c:120-131 */
static void check_sample_name(const char *s)
{
  if (strlen(s) < 1) die("Sample name is too short: '%s'", s);
  if (!strcmp(s, "undefined")) die("Bad sample name: '%s'", s);
  if (!strcmp(s, "noname")) die("Bad sample name: '%s'", s);
  if (s[0] == '.') die("Sample name should start with a dot: '%s'", s);
  if (strlen(s) > 255) die("Sample name too long: '%s'", s);
  for (const char *p = s; *p; p++) {
    if (isspace((unsigned char)*p)) die("Sample name should not contain whitespace: '%s'", s);
    if (!isgraph((unsigned char)*p)) die("Bad character in sample name: '%s'", s);
  }
}

static build_task *add_task(const char *path, int colour, uint8_t fq_cutoff, uint8_t fq_offset, uint8_t hp)
{ /* add_task: ctx_build.c:99-117 */
  if (fq_offset >= 128) die("fq-offset too big: %i", (int)fq_offset);
  if (fq_offset + fq_cutoff >= 128) die("fq-cutoff too big: %i", fq_offset + fq_cutoff);
  if (ntasks == tasks_cap) { tasks_cap = tasks_cap ? tasks_cap * 2 : 16; tasks = realloc(tasks, tasks_cap * sizeof(*tasks)); }
  build_task *t = &tasks[ntasks++];
  memset(t, 0, sizeof(*t));
  t->path = strdup(path); t->colour = colour;
  t->fq_cutoff = fq_cutoff; t->fq_offset = fq_offset; t->hp_cutoff = hp;
  if (strcmp(path, "-") != 0 && access(path, R_OK) != 0) die("Cannot open -1 file: %s", path);
  return t;
}

static long file_size(const char *path)
{
  struct stat st;
  if (!strcmp(path, "-") || stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return -1;
  return (long)st.st_size;
}

static int write_sink(void *ctx, const void *recs, size_t n)
{
  return fwrite(recs, 1, n, (FILE *)ctx) == n ? 0 : 1;
}

/* Sink for a regular output file: every chunk is split over a few threads that pwrite() their part
 * (copying into the page cache is what bounds the dump: ~6 GB/s from one thread). */
#include <pthread.h>
typedef struct { int fd; off_t off; int nthreads; bool failed; } pwrite_sink_ctx;
typedef struct { int fd; const unsigned char *p; size_t n; off_t off; bool failed; } pwrite_job;
static void *pwrite_main(void *arg)
{
  pwrite_job *j = arg;
  while (j->n) {
    ssize_t w = pwrite(j->fd, j->p, j->n, j->off);
    if (w <= 0) { j->failed = true; break; }
    j->p += w; j->n -= (size_t)w; j->off += w;
  }
  return NULL;
}
static int pwrite_sink(void *ctx, const void *recs, size_t n)
{
  pwrite_sink_ctx *c = ctx;
  const int nt = n < (8u << 20) ? 1 : c->nthreads;
  pthread_t th[16];
  pwrite_job job[16];
  const size_t per = (n + (size_t)nt - 1) / (size_t)nt;
  for (int i = 0; i < nt; i++) {
    const size_t lo = (size_t)i * per, hi = lo + per < n ? lo + per : n;
    job[i] = (pwrite_job){c->fd, (const unsigned char *)recs + lo, hi > lo ? hi - lo : 0, c->off + (off_t)lo, false};
    if (i + 1 < nt) pthread_create(&th[i], NULL, pwrite_main, &job[i]);
  }
  pwrite_main(&job[nt - 1]);
  bool bad = job[nt - 1].failed;
  for (int i = 0; i + 1 < nt; i++) { pthread_join(th[i], NULL); bad |= job[i].failed; }
  c->off += (off_t)n;
  return bad ? 1 : 0;
}

/* MCX_TIMING=1: elapsed milliseconds at every stage (stderr), for the end-to-end breakdown */
#include <time.h>
static void stage_time(const char *what)
{
  static int on = -1;
  static struct timespec t0;
  if (on < 0) { on = getenv("MCX_TIMING") != NULL; clock_gettime(CLOCK_MONOTONIC, &t0); }
  if (!on) return;
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  fprintf(stderr, "[timing] %8.1f ms  %s\n", (t.tv_sec - t0.tv_sec) * 1e3 + (t.tv_nsec - t0.tv_nsec) * 1e-6, what);
}

static void mcx_check(int rc, const char *what)
{
  if (rc == MCX_ERR_FULL) die("Hash table is full");
  if (rc != MCX_OK) die("%s: %s", what, mcx_last_error());
}

/* one batch of parsed reads -> the GPU (callback of the parallel parser and body of the sequential loop) */
typedef struct { mcx_graph *g; build_task *bt; bool use_q; uint8_t fq_abs; } submit_ctx;
static double submit_ms = 0; /* time inside mcx_graph_add_reads (MCX_TIMING) */
static unsigned long submit_calls = 0;
static void submit_batch(void *arg, read_batch *b, int fq_offset_guess)
{
  submit_ctx *sc = arg;
  if (!b->nreads) return;
  static int parse_only = -1;  /* MCX_PARSE_ONLY=1: measure the parser alone (nothing reaches the GPU) */
  if (parse_only < 0) parse_only = getenv("MCX_PARSE_ONLY") != NULL;
  if (parse_only) return;
  if (sc->use_q && !sc->fq_abs) { /* build_graph.c:203-206: cutoff + offset; offset auto-detected when 0 */
    /* decided once per file from its head (task_detect_fq_offset); stdin: from its first batch */
    if (!sc->bt->fq_offset && sc->bt->fq_detected <= 0) sc->bt->fq_detected = fq_offset_guess;
    int off = sc->bt->fq_offset ? sc->bt->fq_offset : sc->bt->fq_detected;
    if (off <= 0) off = 33;
    sc->fq_abs = (uint8_t)(sc->bt->fq_cutoff + off);
  }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  mcx_check(mcx_graph_add_reads(sc->g, sc->bt->colour, b->bases, sc->use_q ? b->quals : NULL, b->offsets, b->nreads,
                                sc->fq_abs, sc->bt->hp_cutoff, &sc->bt->stats), "add reads");
  clock_gettime(CLOCK_MONOTONIC, &t1);
  submit_ms += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
  submit_calls++;
}

/* While the parser threads work on their first batches: let the library allocate what the first
 * mcx_graph_add_reads would (pinned staging, partition workspace: ~0.1 s). */
static void prepare_graph(void *arg)
{
  submit_ctx *sc = arg;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  if (!getenv("MCX_PARSE_ONLY")) mcx_check(mcx_graph_configure(sc->g, "prepare", 1), "prepare");
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (getenv("MCX_TIMING")) fprintf(stderr, "[timing] %8.1f ms  inside prepare (overlaps the first parse)\n", (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
}

/* ---- read-ahead for gzip'd inputs -------------------------------------------------------------
 * One thread inflates and parses ONE file (as the reference's reader thread per file does,
 * src/basic/async_read_io.c:145-175); zlib gives ~0.3 GB/s per stream, so with several .gz inputs
 * the files after the current one are inflated ahead, up to -t of them at a time.  Batches still
 * reach the GPU file by file in task order (the per-file statistics are counter deltas around a file). */
#define RA_DEPTH 3
typedef struct {
  build_task *bt;
  pthread_t th;
  bool started, joined;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  struct { read_batch *b; int fq_guess; } q[RA_DEPTH];
  int qh, qn;
  bool done, open_failed, use_q;
} reader_job;

static bool is_gzip_file(const char *path)
{
  if (file_size(path) < 2) return false;
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  unsigned char m[2] = {0, 0};
  const bool gz = fread(m, 1, 2, f) == 2 && m[0] == 0x1f && m[1] == 0x8b;
  fclose(f);
  return gz;
}

static void *reader_main(void *arg)
{
  reader_job *j = arg;
  seq_in *in = seq_in_open(j->bt->path);
  if (!in) {
    pthread_mutex_lock(&j->mu); j->open_failed = j->done = true; pthread_cond_broadcast(&j->cv); pthread_mutex_unlock(&j->mu);
    return NULL;
  }
  j->use_q = j->bt->fq_cutoff > 0 && seq_in_format(in) == SEQ_FMT_FASTQ; /* (read by the consumer after its first pop) */
  for (;;) {
    read_batch *b = malloc(sizeof(*b));
    if (!b) die("Out of memory");
    read_batch_init(b, j->use_q);
    const size_t got = seq_in_fill(in, b, BATCH_BASES);
    if (!got && !b->nreads) { read_batch_free(b); free(b); break; }
    pthread_mutex_lock(&j->mu);
    while (j->qn == RA_DEPTH) pthread_cond_wait(&j->cv, &j->mu);
    const int at = (j->qh + j->qn) % RA_DEPTH;
    j->q[at].b = b; j->q[at].fq_guess = seq_in_guess_fq_offset(in);
    j->qn++;
    pthread_cond_broadcast(&j->cv);
    pthread_mutex_unlock(&j->mu);
  }
  seq_in_close(in);
  pthread_mutex_lock(&j->mu); j->done = true; pthread_cond_broadcast(&j->cv); pthread_mutex_unlock(&j->mu);
  return NULL;
}

/* next batch of the file, NULL at its end */
static read_batch *reader_pop(reader_job *j, int *fq_guess)
{
  read_batch *b = NULL;
  pthread_mutex_lock(&j->mu);
  while (j->qn == 0 && !j->done) pthread_cond_wait(&j->cv, &j->mu);
  if (j->qn) {
    b = j->q[j->qh].b; *fq_guess = j->q[j->qh].fq_guess;
    j->qh = (j->qh + 1) % RA_DEPTH; j->qn--;
    pthread_cond_broadcast(&j->cv);
  }
  pthread_mutex_unlock(&j->mu);
  return b;
}

/* --remove-pcr task: reads go to the GPU in input order, mates side by side
 * (build_graph_from_reads_mt with prefs.remove_pcr_dups, build_graph.c:192-231) */
static uint8_t fq_abs_of(build_task *bt, seq_in *in, int which)
{ /* build_graph.c:203-206: cutoff + offset; offset auto-detected when 0 (once per file) */
  if (!bt->fq_cutoff || seq_in_format(in) != SEQ_FMT_FASTQ) return 0;
  int *det = which ? &bt->fq_detected2 : &bt->fq_detected;
  if (!bt->fq_offset && *det <= 0) *det = seq_in_guess_fq_offset(in); /* stdin: latched at the first batch */
  int off = bt->fq_offset ? bt->fq_offset : *det;
  if (off <= 0) off = 33;
  return (uint8_t)(bt->fq_cutoff + off);
}

/* FASTQ offset of a task's file(s), once, before any of it is loaded (seq_reader.c:304,376-381) */
static void task_detect_fq_offset(build_task *bt)
{
  if (!bt->fq_cutoff || bt->fq_offset) return;
  bt->fq_detected = fq_offset_probe(bt->path);
  if (bt->path2) bt->fq_detected2 = fq_offset_probe(bt->path2);
}

static void load_task_pcr(mcx_graph *g, build_task *bt)
{
  seq_in *in1 = seq_in_open(bt->path), *in2 = NULL;
  if (!in1) die("Cannot open -1 file: %s", bt->path);
  if (bt->path2 && !(in2 = seq_in_open(bt->path2))) die("Cannot open -2 file: %s", bt->path2);
  const bool paired = in2 != NULL || bt->interleaved;
  const bool use_q = bt->fq_cutoff > 0 && (seq_in_format(in1) == SEQ_FMT_FASTQ || (in2 && seq_in_format(in2) == SEQ_FMT_FASTQ));
  read_batch b1, b2, both;
  read_batch_init(&b1, use_q); read_batch_init(&b2, use_q); read_batch_init(&both, use_q);
  for (;;) {
    const size_t got1 = seq_in_fill(in1, &b1, BATCH_BASES);
    read_batch *send = &b1;
    if (in2) { /* zip the two files: pair i = read i of each */
      const size_t got2 = seq_in_fill(in2, &b2, BATCH_BASES);
      const size_t n = b1.nreads < b2.nreads ? b1.nreads : b2.nreads;
      if (!got1 && !got2 && n == 0) {
        if (b1.nreads != b2.nreads) die("Different number of reads in paired files: %s, %s", bt->path, bt->path2);
        break;
      }
      read_batch_clear(&both);
      for (size_t i = 0; i < n; i++) { read_batch_append(&both, &b1, i); read_batch_append(&both, &b2, i); }
      /* keep the unpaired tail of the longer batch for the next round */
      read_batch *bs[2] = {&b1, &b2};
      for (int m = 0; m < 2; m++) {
        read_batch *b = bs[m], tmp;
        read_batch_init(&tmp, use_q);
        for (size_t i = n; i < b->nreads; i++) read_batch_append(&tmp, b, i);
        read_batch_free(b); *b = tmp;
      }
      send = &both;
      if (!n) continue;
    } else if (!got1 && !b1.nreads) {
      break;
    } else if (bt->interleaved && (b1.nreads & 1)) { /* hold the odd read back for its mate */
      if (!got1) die("Odd number of reads in interleaved file: %s", bt->path);
      read_batch_clear(&both);
      for (size_t i = 0; i + 1 < b1.nreads; i++) read_batch_append(&both, &b1, i);
      read_batch tmp;
      read_batch_init(&tmp, use_q);
      read_batch_append(&tmp, &b1, b1.nreads - 1);
      read_batch_free(&b1); b1 = tmp;
      send = &both;
      if (!send->nreads) continue;
    }
    const uint8_t fq1 = use_q ? fq_abs_of(bt, in1, 0) : 0, fq2 = use_q ? (in2 ? fq_abs_of(bt, in2, 1) : fq1) : 0;
    mcx_check(mcx_graph_add_reads_pcr(g, bt->colour, send->bases, use_q ? send->quals : NULL, send->offsets, send->nreads,
                                      fq1, fq2, bt->hp_cutoff, paired ? 1 : 0, bt->matedir, &bt->stats), "add reads");
    if (send == &b1) read_batch_clear(&b1);
  }
  read_batch_free(&b1); read_batch_free(&b2); read_batch_free(&both);
  seq_in_close(in1);
  if (in2) seq_in_close(in2);
}

/* file_filter_status: file_filter.c:170-192 */
static void filter_status(const ctx_reader *r)
{
  char line[1024];
  int n = snprintf(line, sizeof(line), "[FileFilter] Reading file %s [%u src colour%s]", r->path, r->num_cols, r->num_cols == 1 ? "" : "s");
  bool direct = true;
  for (size_t i = 0; i < r->nfilter; i++) direct &= (r->filter[i].from == i && r->filter[i].into == i);
  if (!direct) {
    for (size_t i = 0; i < r->nfilter && n < (int)sizeof(line) - 32; i++)
      n += snprintf(line + n, sizeof(line) - (size_t)n, "%s%u->%u", i ? "," : " with filter: ", r->filter[i].from, r->filter[i].into);
  }
  status("%s", line);
}

/* graph_load (graphs_load.c:86-214) with the hash table on the GPU: merge the header's GraphInfo
 * into the colours it loads into, then stream the records through mcx_graph_add_records */
/* isec_col >= 0: the file is an intersection graph, every colour of it goes to that (hidden) colour
 * and its header is not merged (ctx_build.c:347-363); rec_flags: MCX_RECORDS_* */
static void load_graph_file(mcx_graph *g, ctx_reader *r, col_info *cols, size_t ncols, int isec_col, uint32_t rec_flags)
{
  char a[64], b[64];
  filter_status(r);
  if (r->file_size >= 0)
    status("[GReader] %s kmers, %s filesize", ulong_to_str((uint64_t)r->num_kmers, a), bytes_to_str((uint64_t)r->file_size, 1, b));
  else status("  reading from a stream.");
  /* graph_load_ginfo: graphs_load.c:48-77 */
  if (r->into_ncols > ncols)
    die("Program has not assigned enough colours! [colours in graph: %zu vs file: %zu; path: %s]", ncols, r->into_ncols, r->path);
  if (isec_col < 0)
    for (size_t i = 0; i < r->nfilter; i++) col_info_merge(&cols[r->filter[i].into], &r->ginfo[r->filter[i].from]);

  const size_t rec_bytes = 8 * (size_t)r->num_words + 5 * (size_t)r->num_cols;
  const size_t chunk_recs = (64u << 20) / rec_bytes;
  unsigned char *buf = malloc(chunk_recs * rec_bytes);
  int32_t *from = malloc(r->nfilter * sizeof(int32_t)), *into = malloc(r->nfilter * sizeof(int32_t));
  if (!buf || !from || !into) die("Out of memory");
  for (size_t i = 0; i < r->nfilter; i++) {
    from[i] = (int32_t)r->filter[i].from;
    into[i] = isec_col >= 0 ? isec_col : (int32_t)r->filter[i].into;
  }
  mcx_records_stats st = {0, 0, 0, -1, -1, -1};
  bool warned_zero = false, warned_edges = false;
  for (;;) {
    const size_t got = fread(buf, 1, chunk_recs * rec_bytes, r->fh);
    if (got == 0) break;
    if (got % rec_bytes) {
      /* graph_file_read_raw: a partial key is "Unexpected end of file", a partial tail an _gfread error */
      die("Unexpected end of file: %s", r->path);
    }
    const uint64_t base = st.nkmers_read;
    int rc = mcx_graph_add_records(g, buf, got / rec_bytes, (int)r->num_cols, from, into, (int)r->nfilter, rec_flags, &st);
    if (rc != MCX_OK && st.first_oversized >= 0) die("Oversized kmer in path [kmer: %u]: %s", r->kmer_size, r->path);
    mcx_check(rc, "load graph records");
    if (st.first_zero_covg >= 0 && !warned_zero) {
      char kstr[2 * MAX_KMER_SIZE + 8];
      kmer_words_to_str(buf + ((uint64_t)st.first_zero_covg - base) * rec_bytes, r->kmer_size, kstr);
      warn("Kmer has zero covg in all colours [kmer: %s; path: %s]", kstr, r->path);
      warned_zero = true;
    }
    if (st.first_edges_no_covg >= 0 && !warned_edges) {
      char kstr[2 * MAX_KMER_SIZE + 8];
      kmer_words_to_str(buf + ((uint64_t)st.first_edges_no_covg - base) * rec_bytes, r->kmer_size, kstr);
      warn("Kmer has edges but no coverage [kmer: %s; path: %s]", kstr, r->path);
      warned_edges = true;
    }
  }
  if (r->num_kmers >= 0 && st.nkmers_read != (uint64_t)r->num_kmers)
    warn("%s kmers in the graph file than expected [exp: %zu; act: %zu; path: %s]",
         st.nkmers_read > (uint64_t)r->num_kmers ? "More" : "Fewer", (size_t)r->num_kmers, (size_t)st.nkmers_read, r->path);
  status("[GReader] Loaded %s / %s (%.2f%%) of kmers parsed", ulong_to_str(st.nkmers_loaded, a), ulong_to_str(st.nkmers_read, b),
         st.nkmers_read ? 100.0 * (double)st.nkmers_loaded / (double)st.nkmers_read : 0.0);
  free(buf); free(from); free(into);
}

int ctx_build(int argc, char **argv)
{
  size_t nthreads = 0, kmer_size = 0, mem_to_use = DEFAULT_MEM, num_kmers = 0;
  bool mem_set = false, nkmers_set = false, force = false, sort_kmers = false;
  bool sample_named = false, pref_unused = false, remove_pcr = false;
  uint8_t fq_offset = 0, fq_cutoff = 0, hp_cutoff = 0;
  int intocolour = -1, device = 0, c, matedir = 1 /* FR: build_graph.h:38-42 */;
  int devices[32] = {0}, ndevices = 1;
  char cmd[100];

  /* '+': stop at the first non-option; single-dash long options accepted (cmd.c:87-102, ctx_build.c:149) */
  optind = 1;
  while ((c = getopt_long_only(argc, argv, "+hm:n:t:fk:s:S1:2:i:M:Q:O:H:pPg:I:D:", longopts, NULL)) != -1) {
    optname((char)c, cmd);
    unsigned u;
    switch (c) {
      case 'h': print_usage(build_usage, NULL);
      case 't':
        if (nthreads) usage_die("%s given twice", cmd);
        if (!parse_entire_uint(optarg, &u)) usage_die("%s requires an int x >= 0: %s", cmd, optarg);
        if (!u) usage_die("%s <N> must be > 0: %s", cmd, optarg);
        nthreads = u; break;
      case 'm':
        if (mem_set) usage_die("-m, --memory <M> specifed more than once");
        if (!mem_to_integer(optarg, &mem_to_use) || !mem_to_use) usage_die("Invalid memory argument: %s", optarg);
        mem_set = true; break;
      case 'n':
        if (nkmers_set) usage_die("-n, --nkmers <N> specifed more than once");
        if (!mem_to_integer(optarg, &num_kmers) || !num_kmers) usage_die("Invalid hash size: %s", optarg);
        nkmers_set = true; break;
      case 'f': if (force) usage_die("%s given twice", cmd); force = true; break;
      case 'k': {
        if (kmer_size) usage_die("%s given twice", cmd);
        size_t k;
        if (!parse_entire_size(optarg, &k)) usage_die("%s requires an int x >= 0: %s", cmd, optarg);
        if (!k) usage_die("%s <N> must be > 0: %s", cmd, optarg);
        if (k < MIN_KMER_SIZE || k > MAX_KMER_SIZE) die("Please recompile with correct kmer size (%zu)", k);
        if (!(k & 1)) die("Invalid kmer-size (%zu): requires odd number %d <= k <= %d", k, MIN_KMER_SIZE, MAX_KMER_SIZE);
        kmer_size = k; break;
      }
      case 's':
        intocolour++;
        check_sample_name(optarg);
        sample_names = realloc(sample_names, (nsamples + 1) * sizeof(char *));
        sample_cols = realloc(sample_cols, (nsamples + 1) * sizeof(int));
        sample_cols[nsamples] = intocolour;
        sample_names[nsamples++] = optarg;
        sample_named = true; break;
      case 'S': if (sort_kmers) usage_die("%s given twice", cmd); sort_kmers = true; break;
      case '1': case '2': case 'i':
        pref_unused = false;
        if (!sample_named) usage_die("Please give sample name first [-s,--sample <name>]");
        if (c == '2') { /* <in1>:<in2> (or a comma) */
          char *sep = strchr(optarg, ':');
          if (!sep) sep = strchr(optarg, ',');
          if (!sep || strchr(sep + 1, ':')) die("Expected -2 <in1>:<in2>");
          *sep = '\0';
          build_task *bt = add_task(optarg, intocolour, fq_cutoff, fq_offset, hp_cutoff);
          if (remove_pcr) { /* mates stay together (ctx_build.c:105-107) */
            if (strcmp(sep + 1, "-") != 0 && access(sep + 1, R_OK) != 0) die("Cannot open -2 file: %s", sep + 1);
            bt->path2 = strdup(sep + 1);
          } else { /* loaded as two single-ended files (ctx_build.c:108-116) */
            add_task(sep + 1, intocolour, fq_cutoff, fq_offset, hp_cutoff);
          }
          bt->remove_pcr = remove_pcr; bt->matedir = matedir;
        } else {
          build_task *bt = add_task(optarg, intocolour, fq_cutoff, fq_offset, hp_cutoff);
          bt->remove_pcr = remove_pcr; bt->matedir = matedir;
          bt->interleaved = remove_pcr && c == 'i';
        }
        break;
      case 'M':
        if (strcmp(optarg, "FF") && strcmp(optarg, "FR") && strcmp(optarg, "RF") && strcmp(optarg, "RR"))
          die("-M,--matepair <orient> must be one of: FF,FR,RF,RR");
        matedir = (optarg[0] == 'R' ? 2 : 0) | (optarg[1] == 'R' ? 1 : 0); /* READPAIR_*: cortex_types.h:18-21 */
        pref_unused = true; break;
      case 'O': if (!parse_entire_uint(optarg, &u)) usage_die("%s requires an int 0 <= x < 255: %s", cmd, optarg);
        fq_offset = (uint8_t)u; pref_unused = true; break;
      case 'Q': if (!parse_entire_uint(optarg, &u)) usage_die("%s requires an int 0 <= x < 255: %s", cmd, optarg);
        fq_cutoff = (uint8_t)u; pref_unused = true; break;
      case 'H': if (!parse_entire_uint(optarg, &u)) usage_die("%s requires an int 0 <= x < 255: %s", cmd, optarg);
        hp_cutoff = (uint8_t)u; pref_unused = true; break;
      case 'p': remove_pcr = true; pref_unused = true; break;
      case 'P': remove_pcr = false; pref_unused = true; break;
      case 'g': /* ctx_build.c:189-196 */
        if (intocolour == -1) intocolour = 0;
        gfiles = realloc(gfiles, (ngfiles + 1) * sizeof(ctx_reader));
        ctx_reader_open(&gfiles[ngfiles], optarg, (size_t)intocolour, MIN_KMER_SIZE, MAX_KMER_SIZE);
        if ((int)gfiles[ngfiles].into_ncols - 1 > intocolour) intocolour = (int)gfiles[ngfiles].into_ncols - 1;
        ngfiles++;
        sample_named = false; break;
      case 'I': /* ctx_build.c:197-203 */
        gisec = realloc(gisec, (ngisec + 1) * sizeof(ctx_reader));
        ctx_reader_open(&gisec[ngisec], optarg, 0, MIN_KMER_SIZE, MAX_KMER_SIZE);
        if (gisec[ngisec].into_ncols > 1) warn("Flattening intersection graph into colour 0: %s", optarg);
        for (size_t i = 0; i < gisec[ngisec].nfilter; i++) gisec[ngisec].filter[i].into = 0; /* file_filter_flatten */
        gisec[ngisec].into_ncols = 1;
        ngisec++;
        break;
      case 'D': { /* one device, or a comma-separated list: the table is split over them (mcx_graph_create_multi) */
        ndevices = 0;
        char *list = strdup(optarg), *save = NULL;
        for (char *tok = strtok_r(list, ",", &save); tok; tok = strtok_r(NULL, ",", &save)) {
          if (!parse_entire_uint(tok, &u)) usage_die("%s requires ints x >= 0, comma separated: %s", cmd, optarg);
          if (ndevices == 32) usage_die("%s takes at most 32 devices: %s", cmd, optarg);
          devices[ndevices++] = (int)u;
        }
        free(list);
        if (ndevices == 0 || (ndevices & (ndevices - 1))) usage_die("%s: the number of devices must be a power of two: %s", cmd, optarg);
        device = devices[0];
        break;
      }
      case ':': case '?':
        die("`" CMD_NAME " build -h` for help. Bad option: %s", argv[optind - 1]);
      default: die("Bad option: %s", cmd);
    }
  }
  if (!nthreads) nthreads = DEFAULT_NTHREADS;
  if (optind + 1 > argc) usage_die("Expected exactly one graph file");
  else if (optind + 1 < argc) usage_die("Expected only one graph file. What is this: '%s'", argv[optind]);
  const char *out_path = argv[optind];
  status("Saving graph to: %s", strcmp(out_path, "-") ? out_path : "STDOUT");
  if (nsamples == 0) usage_die("No inputs given");
  if (pref_unused) usage_die("Arguments not given BEFORE sequence file");
  if (!kmer_size) die("kmer size not set with -k <K>");
  for (size_t i = 0; i < ngfiles; i++) /* ctx_build.c:231-239 */
    if (gfiles[i].kmer_size != kmer_size)
      usage_die("Input graph kmer_size doesn't match [%u vs %zu]: %s", gfiles[i].kmer_size, kmer_size, gfiles[i].input);
  const size_t ncols = (size_t)intocolour + (sample_named ? 1 : 0);

  /* print inputs in sample/task order (ctx_build.c:270-279) and estimate k-mers from file
   * sizes (asyncio_input_nkmers: bytes, halved for FASTQ, times 5; async_read_io.c:313-334) */
  size_t max_kmers = 0;
  uint64_t seq_bytes_est = 0; /* sequence characters the inputs hold at most (an upper bound of the k-mer occurrences) */
  bool seq_bytes_known = true;
  bool size_unknown = false;
  for (size_t i = 0; i < ngfiles; i++) { /* ctx_build.c:268-272 (a stream counts as -1 there too) */
    filter_status(&gfiles[i]);
    max_kmers += (size_t)gfiles[i].num_kmers;
  }
  for (size_t s = 0, t = 0; s < nsamples || t < ntasks;) {
    if (t == ntasks || (s < nsamples && sample_cols[s] <= tasks[t].colour)) { status("[sample] %zu: %s", s, sample_names[s]); s++; }
    else {
      build_task *bt = &tasks[t];
      bt->fmt = SEQ_FMT_UNKNOWN;
      if (file_size(bt->path) >= 0) { /* stdin, pipes and <(...) can only be read once: no format probe */
        seq_in *probe = seq_in_open(bt->path);
        if (!probe) die("Cannot open -1 file: %s", bt->path);
        bt->fmt = seq_in_format(probe);
        seq_in_close(probe);
      }
      char fqo[30] = "auto-detect", fqc[30] = "off", hpc[30] = "off"; /* build_graph_task_print: build_graph.c:332-350 */
      if (bt->fq_offset > 0) sprintf(fqo, "%u", bt->fq_offset);
      if (bt->fq_cutoff > 0) sprintf(fqc, "%u", bt->fq_cutoff);
      if (bt->hp_cutoff > 0) sprintf(hpc, "%u", bt->hp_cutoff);
      status("[task] %s%s%s; FASTQ offset: %s, threshold: %s; cut homopolymers: %s; remove PCR duplicates: %s; colour: %i\n",
             bt->path, bt->path2 ? ", " : "", bt->path2 ? bt->path2 : "", fqo, fqc, hpc, bt->remove_pcr ? "yes" : "no", bt->colour);
      long fs = file_size(bt->path), fs2 = bt->path2 ? file_size(bt->path2) : 0;
      if (fs < 0 || fs2 < 0) size_unknown = true;
      else max_kmers += (size_t)(bt->fmt == SEQ_FMT_FASTQ ? (fs + fs2) / 2 : fs + fs2) * 5;
      if (fs < 0 || fs2 < 0 || is_gzip_file(bt->path) || (bt->path2 && is_gzip_file(bt->path2))) seq_bytes_known = false;
      else seq_bytes_est += (uint64_t)(bt->fmt == SEQ_FMT_FASTQ ? (fs + fs2) / 2 + 1024 : fs + fs2);
      t++;
    }
  }
  if (size_unknown) max_kmers = SIZE_MAX;
  if (ngisec > 0) { /* ctx_build.c:291-303: reads and graphs can only touch k-mers of the intersection */
    for (size_t t = 0; t < ntasks; t++) if (tasks[t].remove_pcr) usage_die("Cannot use --remove-pcr and --intersect");
    max_kmers = 0;
    for (size_t i = 0; i < ngisec; i++) max_kmers += (size_t)gisec[i].num_kmers;
  }

  /* ---- decide on memory (ctx_build.c:305-322, cmd_mem.c:38-130) ---- */
  const size_t W = (2 * kmer_size + 63) / 64;
  bool remove_pcr_used = false; /* ctx_build.c:260-261 */
  for (size_t t = 0; t < ntasks; t++) remove_pcr_used |= tasks[t].remove_pcr;
  /* remove_pcr_dups requires a fw and rv bit per kmer (ctx_build.c:310-315) */
  size_t bits_per_kmer = W * 64 + (4 + 1) * 8 * ncols + (ngisec > 0 ? 8 : 0) + (remove_pcr_used ? 2 : 0) + (sort_kmers ? 64 : 0);
  uint64_t kmers_in_hash = 0;
  size_t graph_mem = 0;
  char s1[64], s2[64];
  status("[memory] %zu bits per kmer", bits_per_kmer);
  {
    table_plan plan;
    char ebuf[256];
    const char *err = table_plan_for_build(mem_to_use, mem_set, num_kmers, nkmers_set, bits_per_kmer,
                                           max_kmers == SIZE_MAX ? -1 : (int64_t)max_kmers, &plan, ebuf, sizeof(ebuf));
    if (err) die("%s", err);
    graph_mem = plan.bytes;
    kmers_in_hash = plan.capacity;
  }
  status("[memory] graph: %s", bytes_to_str(graph_mem, 1, s1));

  stage_time("arguments parsed");
  if (mcx_device_count() < 1) die("No MI355X / HIP device found: %s has no CPU build path", CMD_NAME);
  stage_time("HIP runtime up");
  const size_t dev_cols = ncols + (ngisec > 0 ? 1 : 0); /* + the hidden colour of the intersection edges */
  /* per device: its share of the table (+ the read-start table of --remove-pcr, + the overflow area) */
  const uint64_t dev_bytes = (kmers_in_hash / (uint64_t)ndevices + kmers_in_hash / (uint64_t)ndevices / 32) * 8 * (W + dev_cols + (remove_pcr_used ? 1 : 0));
  uint64_t xchg_bytes = 0; /* exchange buffers between the devices of a split table (0 on one device) */
  mcx_check(mcx_multi_exchange_bytes((int)kmer_size, ndevices, kmers_in_hash, &xchg_bytes), "exchange sizing");
  for (int d = 0; d < ndevices; d++) {
    uint64_t hbm_free = 0, hbm_total = 0;
    mcx_check(mcx_device_memory(devices[d], &hbm_free, &hbm_total), "device query");
    if (dev_bytes > hbm_free)
      die("Requesting more memory than is available [ Reqeusted: %s HBM free: %s ]",
          bytes_to_str(dev_bytes, 1, s1), bytes_to_str(hbm_free, 1, s2));
    if (xchg_bytes && dev_bytes + xchg_bytes > hbm_free)
      warn("device %d: table %s + exchange buffers %s exceed the free HBM: the exchange will run in smaller pieces",
           devices[d], bytes_to_str(dev_bytes, 1, s1), bytes_to_str(xchg_bytes, 1, s2));
    status("[memory] device %d: table %s of %s HBM\n", devices[d], bytes_to_str(dev_bytes, 1, s1), bytes_to_str(hbm_total, 1, s2));
    if (xchg_bytes) status("[memory] device %d: exchange buffers %s\n", devices[d], bytes_to_str(xchg_bytes, 1, s1));
  }

  /* ---- output path (futil_create_output: file_util.c:164-186) ---- */
  FILE *fout = stdout;
  if (strcmp(out_path, "-") != 0) {
    if (!force && access(out_path, F_OK) == 0) die("File already exists: %s", out_path);
    fout = fopen(out_path, "wb");
    if (!fout) die("Cannot open output file: %s [%s]", out_path, strerror(errno));
  }
  status("Writing %zu colour graph to %s\n", ncols, strcmp(out_path, "-") ? out_path : "STDOUT");

  mcx_graph *g = NULL;
  if (ndevices == 1) devices[0] = device;
  mcx_check(mcx_graph_create_multi(&g, (int)kmer_size, (int)dev_cols, kmers_in_hash, devices, ndevices), "Cannot allocate graph");
  if (ngisec > 0) mcx_check(mcx_graph_configure(g, "intersect", 1), "intersect mode");
  /* The partition workspace is sized for the occurrences buffered per flush (the library's default: as much as
   * 40 % of the free HBM allows, up to 8.6 G occurrences = 65 GB): when the inputs cannot hold that many, size it for
   * the inputs -- and never beyond 2.1 G occurrences (18 GB) here: a build fed from files is bound by the host (the
   * device is busy for 145 ms of a 1 s build, round 5), so the extra table passes of a smaller window run in time the
   * device would otherwise idle, while a 65 GB hipMalloc costs 0.6-2.2 s whenever the device's memory was freed by
   * another process moments before (the driver hands it over at ~30 GB/s then; 0.8 ms when it is clean). */
  uint64_t slots = 0, tbytes = 0;
  mcx_graph_capacity(g, &slots, &tbytes);
  if (ngisec == 0 && !getenv("MCX_DEFER_TUPLES")) {
    uint64_t window = 1ull << 31;
    const uint64_t by_table = 64 * (slots / (uint64_t)ndevices);  /* (the library's own rule: 64 occurrences per slot of a device's table) */
    if (by_table < window) window = by_table;
    if (seq_bytes_known && seq_bytes_est + (1u << 20) < window) window = seq_bytes_est + (1u << 20);
    mcx_check(mcx_graph_configure(g, "defer_tuples", window), "flush size");
  }
  status("[hasht] Allocated table in HBM with %s entries, using %s", ulong_to_str(slots, s1), bytes_to_str(tbytes, 1, s2));
  stage_time("table allocated");

  col_info *cols = calloc(ncols, sizeof(col_info));
  for (size_t i = 0; i < ncols; i++) col_info_init(&cols[i]);

  /* ---- load graphs (graph_load: graphs_load.c:86-214), then name the samples (ctx_build.c:365-382) ---- */
  for (size_t i = 0; i < ngisec; i++) { /* intersection graphs: keys + their edges (ctx_build.c:347-363) */
    if (gisec[i].kmer_size != kmer_size)
      die("Graph has different kmer size [kmer_size: %u vs %zu; path: %s]", gisec[i].kmer_size, kmer_size, gisec[i].path);
    load_graph_file(g, &gisec[i], cols, ncols, (int)ncols, 0);
    ctx_reader_close(&gisec[i]);
  }
  for (size_t i = 0; i < ngfiles; i++) {
    load_graph_file(g, &gfiles[i], cols, ncols, -1, ngisec > 0 ? (MCX_RECORDS_MUST_EXIST | MCX_RECORDS_MASK_EDGES) : 0);
    uint64_t nk0 = 0;
    mcx_check(mcx_graph_nkmers(g, &nk0), "nkmers");
    status("[hash] occupancy: %s / %s (%.2f%%)", ulong_to_str(nk0, s1), ulong_to_str(slots, s2), 100.0 * (double)nk0 / (double)slots);
    ctx_reader_close(&gfiles[i]);
  }
  for (size_t i = 0; i < nsamples; i++) col_info_set_name(&cols[sample_cols[i]], sample_names[i]);
  if (ngisec > 0) mcx_check(mcx_graph_configure(g, "must_exist", 1), "must-exist mode"); /* ctx_build.c:297-298 */

  /* ---- load every input in task order (build_graph(): build_graph.c:257-300) ---- */
  read_batch batch;
  /* gzip'd files not under --remove-pcr get a read-ahead thread each, at most -t running at a time */
  reader_job *readers = calloc(ntasks ? ntasks : 1, sizeof(reader_job));
  size_t next_reader = 0, readers_live = 0;
  for (size_t t = 0; t < ntasks; t++) {
    readers[t].bt = &tasks[t];
    pthread_mutex_init(&readers[t].mu, NULL);
    pthread_cond_init(&readers[t].cv, NULL);
  }
  mcx_load_stats prev;
  mcx_check(mcx_graph_device_stats(g, &prev), "device stats"); /* k-mers created by --graph are not a file's */
  for (size_t t = 0; t < ntasks; t++) {
    build_task *bt = &tasks[t];
    task_detect_fq_offset(bt);
    /* -t threads parse an uncompressed regular file in parallel; gzip, stdin and files the fast
     * path declines go through the sequential parser */
    submit_ctx sc = {g, bt, bt->fq_cutoff > 0 && bt->fmt == SEQ_FMT_FASTQ, 0};
    int prc = 1;
    /* the read-start bits are wiped when the colour changes (ctx_build.c:389-395) */
    if (remove_pcr_used && t > 0 && bt->colour != tasks[t - 1].colour) mcx_check(mcx_graph_pcr_reset(g), "pcr reset");
    /* start read-ahead threads for this and the following gzip'd files */
    if (next_reader < t) next_reader = t;
    while (next_reader < ntasks && readers_live < nthreads && nthreads > 1) {
      reader_job *rj = &readers[next_reader];
      if (!rj->bt->remove_pcr && is_gzip_file(rj->bt->path)) {
        if (pthread_create(&rj->th, NULL, reader_main, rj) != 0) die("Cannot start a reader thread");
        rj->started = true;
        readers_live++;
      }
      next_reader++;   /* (files that take another path do not hold the look-ahead up) */
    }
    if (readers[t].started) {
      reader_job *rj = &readers[t];
      int guess = 0;
      read_batch *b;
      bool first = true;
      while ((b = reader_pop(rj, &guess)) != NULL) {
        if (first) { sc.use_q = rj->use_q; first = false; }
        submit_batch(&sc, b, guess);
        read_batch_free(b); free(b);
      }
      pthread_join(rj->th, NULL);
      rj->joined = true;
      readers_live--;
      if (rj->open_failed) die("Cannot open -1 file: %s", bt->path);
      prc = 0;
    }
    else if (bt->remove_pcr) { load_task_pcr(g, bt); prc = 0; }
    else if (nthreads > 1 && strcmp(bt->path, "-") != 0)
    {
      /* batch size of the range parsers: every batch costs its mcx_graph_add_reads call about half a millisecond of
       * HIP calls whatever its size (rocprofv3 --hip-trace of a 12 GB build, round 5), so large files get 64 MB
       * batches -- as long as the 2 x threads buffers stay a small part of the file */
      size_t pb = PAR_BATCH_BASES;
      if (file_size(bt->path) >= (off_t)4 * (off_t)nthreads * (off_t)(64u << 20)) pb = 64u << 20;
      if (getenv("MCX_PAR_BATCH")) pb = (size_t)atol(getenv("MCX_PAR_BATCH"));
      prc = par_ingest(bt->path, bt->fmt, (int)nthreads, sc.use_q, pb, submit_batch, t == 0 ? prepare_graph : NULL, &sc);
    }
    if (prc == 2) {
      /* A record the range parsers do not handle (multi-line FASTQ) beyond the part of the file that was
       * probed: batches of this file are in the graph already.  When nothing else is (first input, no
       * --graph), empty the graph and read the file again with the sequential parser; otherwise there
       * is no way back. */
      if (t == 0 && ngfiles == 0 && ngisec == 0) {
        warn("Irregular %s record in %s: reading the file again with one parser thread", bt->fmt == SEQ_FMT_FASTQ ? "FASTQ" : "sequence", bt->path);
        mcx_check(mcx_graph_reset(g), "reset");
        memset(&bt->stats, 0, sizeof(bt->stats));
        memset(&prev, 0, sizeof(prev));
        sc.fq_abs = 0;
        prc = 1;
      } else {
        die("Irregular %s record in %s (multi-line FASTQ?): rerun with -t 1", bt->fmt == SEQ_FMT_FASTQ ? "FASTQ" : "sequence", bt->path);
      }
    }
    if (prc == 1) {
      seq_in *in = seq_in_open(bt->path);
      if (!in) die("Cannot open -1 file: %s", bt->path);
      sc.use_q = bt->fq_cutoff > 0 && seq_in_format(in) == SEQ_FMT_FASTQ;
      read_batch_init(&batch, sc.use_q);
      while (seq_in_fill(in, &batch, BATCH_BASES) > 0 || batch.nreads) {
        submit_batch(&sc, &batch, seq_in_guess_fq_offset(in));
        read_batch_clear(&batch);
      }
      read_batch_free(&batch);
      seq_in_close(in);
    }
    /* per-file contig statistics = device counter delta around the file */
    mcx_load_stats cur;
    mcx_check(mcx_graph_device_stats(g, &cur), "device stats");
    bt->stats.num_good_reads = cur.num_good_reads - prev.num_good_reads;
    bt->stats.num_bad_reads = cur.num_bad_reads - prev.num_bad_reads;
    bt->stats.total_bases_loaded = cur.total_bases_loaded - prev.total_bases_loaded;
    bt->stats.contigs_parsed = cur.contigs_parsed - prev.contigs_parsed;
    bt->stats.num_kmers_loaded = cur.num_kmers_loaded - prev.num_kmers_loaded;
    bt->stats.num_kmers_novel = cur.num_kmers_novel - prev.num_kmers_novel;
    prev = cur;
    col_info_update(&cols[bt->colour], bt->stats.total_bases_loaded, bt->stats.contigs_parsed);
  }
  if (ngisec > 0) { /* ctx_build.c:409-413 */
    uint64_t removed = 0;
    mcx_check(mcx_graph_intersect_finish(g, &removed), "intersect");
  }
  stage_time("inputs submitted");
  if (getenv("MCX_TIMING")) fprintf(stderr, "[timing] %8.1f ms  inside mcx_graph_add_reads (%lu calls)\n", submit_ms, submit_calls);
  mcx_check(mcx_graph_sync(g), "sync");
  stage_time("graph built (device idle)");
  if (getenv("MCX_TIMING")) { /* the slow-but-correct insert paths, made visible (the reference: hash_table_print_stats) */
    mcx_insert_stats is;
    mcx_check(mcx_graph_insert_stats(g, &is), "insert stats");
    fprintf(stderr, "[timing] insert paths: %lu table passes, %lu fallback inserts (bin overflow), %lu foreign inserts, %lu spilled between devices\n",
            (unsigned long)is.flushes, (unsigned long)is.fallback_inserts, (unsigned long)is.foreign_inserts, (unsigned long)is.spilled);
  }

  uint64_t nk = 0;
  mcx_check(mcx_graph_nkmers(g, &nk), "nkmers");
  status("[hash] occupancy: %s / %s (%.2f%%)", ulong_to_str(nk, s1), ulong_to_str(slots, s2), 100.0 * (double)nk / (double)slots);

  /* per-file statistics (build_graph_task_print_stats: build_graph.c:352-386) */
  for (size_t t = 0; t < ntasks; t++) {
    const mcx_load_stats *st = &tasks[t].stats;
    char a[64], b[64];
    status("[task] input: %s%s%s colour: %i", tasks[t].path, tasks[t].path2 ? ", " : "", tasks[t].path2 ? tasks[t].path2 : "", tasks[t].colour);
    status("  SE reads: %s  PE reads: %s", ulong_to_str(st->num_se_reads, a), ulong_to_str(st->num_pe_reads, b));
    status("  good reads: %s  bad reads: %s", ulong_to_str(st->num_good_reads, a), ulong_to_str(st->num_bad_reads, b));
    status("  dup SE reads: %s  dup PE pairs: %s", ulong_to_str(st->num_dup_se_reads, a), ulong_to_str(st->num_dup_pe_pairs, b));
    status("  bases read: %s  bases loaded: %s", ulong_to_str(st->total_bases_read, a), ulong_to_str(st->total_bases_loaded, b));
    status("  num contigs: %s  num kmers: %s novel kmers: %s", ulong_to_str(st->contigs_parsed, a),
           ulong_to_str(st->num_kmers_loaded, b), ulong_to_str(st->num_kmers_novel, s1));
  }

  status("Dumping graph...\n");
  size_t hdr = ctx_write_header(fout, (uint32_t)kmer_size, (uint32_t)ncols, cols);
  if (fflush(fout) != 0) die("Cannot write to file");
  struct stat ost;
  if (fout != stdout && fstat(fileno(fout), &ost) == 0 && S_ISREG(ost.st_mode)) {
    pwrite_sink_ctx pc = {fileno(fout), (off_t)hdr, (int)(nthreads < 2 ? 2 : nthreads > 6 ? 6 : nthreads), false};
    mcx_check(mcx_graph_export(g, sort_kmers ? 1 : 0, pwrite_sink, &pc), "export");
    if (fseeko(fout, pc.off, SEEK_SET) != 0) die("Cannot write to file");
  } else {
    mcx_check(mcx_graph_export(g, sort_kmers ? 1 : 0, write_sink, fout), "export");
  }
  if (fflush(fout) != 0) die("Cannot write to file");
  stage_time("graph written");
  const size_t recsz = 8 * W + 5 * ncols;
  status("Dumped %s kmers in %zu colour%s into: %s (format version: 6; %s)", ulong_to_str(nk, s1), ncols,
         ncols == 1 ? "" : "s", strcmp(out_path, "-") ? out_path : "STDOUT", bytes_to_str(hdr + nk * recsz, 1, s2));
  if (fout != stdout) fclose(fout);
  /* The process ends here: the table and the workspace (tens of GB) go back with it.  Releasing
   * them one hipFree at a time first took 0.3 s of a 1.5 s run (MCX_KEEP_DESTROY=1 does it anyway). */
  if (getenv("MCX_KEEP_DESTROY")) mcx_graph_destroy(g);
  else {
    /* ... but only after the device has confirmed that nothing is in flight or went wrong late: a
     * fault that would surface at destroy time must not be lost with the fast exit */
    mcx_check(mcx_graph_sync(g), "final sync");
    host_fast_exit_ok = 1;
  }
  stage_time("device released");
  for (size_t t = 0; t < ntasks; t++) { pthread_mutex_destroy(&readers[t].mu); pthread_cond_destroy(&readers[t].cv); }
  free(readers);
  for (size_t t = 0; t < ntasks; t++) { free(tasks[t].path); free(tasks[t].path2); }
  for (size_t i = 0; i < ncols; i++) col_info_free(&cols[i]);
  free(tasks); free(cols); free(sample_names); free(sample_cols); free(gfiles); free(gisec);
  return EXIT_SUCCESS;
}
