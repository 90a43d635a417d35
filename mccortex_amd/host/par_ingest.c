/* par_ingest.c -- multi-threaded parsing of uncompressed FASTA / FASTQ / plain files.
 *
 * The reference parses every input file with ONE reader thread (src/basic/async_read_io.c:204-213)
 * and `-t` worker threads insert; with the insert on the GPU the parser is what bounds `build`,
 * so here `-t` threads parse disjoint byte ranges of the memory-mapped file into flat read batches
 * (bases, quals, offsets) and the submitting thread hands them to mcx_graph_add_reads().
 * Record boundaries inside a byte range:
 *   FASTQ  a line starting with '@' whose line+2 starts with '+' (a quality line may start with
 *          '@', but then line+2 is a sequence line and cannot start with '+'); 4-line records only
 *          -- anything else makes the caller fall back to the sequential parser (seq_in.c)
 *   FASTA  a line starting with '>'
 *   plain  any line start
 * gzip'd input and stdin keep the sequential parser. */
#define _GNU_SOURCE
#include "host.h"

#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

static double now_ms(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

typedef struct {
  const unsigned char *p, *end; /* byte range, starts at a record boundary */
  seq_fmt fmt;
  bool want_quals;
  size_t max_bases;
  int qmin, qmax;
  bool bad; /* structure the fast path does not handle */
  unsigned nsampled; /* records whose qualities went into qmin/qmax (a sample is enough for the offset guess) */
} range_parser;

static inline const unsigned char *line_end(const unsigned char *p, const unsigned char *end)
{
  const unsigned char *nl = memchr(p, '\n', (size_t)(end - p));
  return nl ? nl : end;
}

static void batch_append(read_batch *b, const unsigned char *seq, size_t n, const unsigned char *qual)
{
  if (b->nbases + n > b->cap_bases) {
    while (b->nbases + n > b->cap_bases) b->cap_bases *= 2;
    b->bases = realloc(b->bases, b->cap_bases);
    if (b->want_quals) b->quals = realloc(b->quals, b->cap_bases);
    if (!b->bases || (b->want_quals && !b->quals)) die("Out of memory");
  }
  memcpy(b->bases + b->nbases, seq, n);
  if (b->want_quals) {
    if (qual) memcpy(b->quals + b->nbases, qual, n);
    else memset(b->quals + b->nbases, 0, n);
  }
  b->nbases += n;
}

static void batch_close_read(read_batch *b)
{
  if (b->nreads + 1 > b->cap_reads) {
    b->cap_reads *= 2;
    b->offsets = realloc(b->offsets, (b->cap_reads + 1) * sizeof(uint64_t));
    if (!b->offsets) die("Out of memory");
  }
  b->offsets[++b->nreads] = b->nbases;
}

static inline size_t strip_cr(const unsigned char *s, size_t n) { return (n && s[n - 1] == '\r') ? n - 1 : n; }

/* Parse records until the batch holds max_bases or the range ends.  Returns reads appended. */
static size_t range_fill(range_parser *rp, read_batch *b)
{
  size_t added = 0;
  const unsigned char *p = rp->p, *end = rp->end;
  while (p < end && b->nbases < rp->max_bases && !rp->bad) {
    if (rp->fmt == SEQ_FMT_PLAIN) {
      const unsigned char *e = line_end(p, end);
      size_t n = strip_cr(p, (size_t)(e - p));
      if (n) { batch_append(b, p, n, NULL); batch_close_read(b); added++; }
      p = e < end ? e + 1 : end;
    } else if (rp->fmt == SEQ_FMT_FASTA) {
      if (*p == '\n' || *p == '\r') { p++; continue; }
      if (*p != '>') { rp->bad = true; break; }
      const unsigned char *e = line_end(p, end);
      p = e < end ? e + 1 : end;
      while (p < end && *p != '>') {
        e = line_end(p, end);
        size_t n = strip_cr(p, (size_t)(e - p));
        if (n) batch_append(b, p, n, NULL);
        p = e < end ? e + 1 : end;
      }
      batch_close_read(b); added++;
    } else { /* FASTQ, 4-line records */
      if (*p == '\n' || *p == '\r') { p++; continue; }
      if (*p != '@') { rp->bad = true; break; }
      const unsigned char *h = line_end(p, end);
      if (h >= end) { rp->bad = true; break; }
      const unsigned char *s = h + 1, *se = line_end(s, end);
      if (se >= end) { rp->bad = true; break; }
      const unsigned char *pl = se + 1;
      if (pl >= end || *pl != '+') { rp->bad = true; break; }
      const unsigned char *ple = line_end(pl, end);
      if (ple >= end) { rp->bad = true; break; }
      const unsigned char *q = ple + 1, *qe = line_end(q, end);
      size_t n = strip_cr(s, (size_t)(se - s)), qn = strip_cr(q, (size_t)(qe - q));
      if (qn != n) { rp->bad = true; break; }
      if (rp->nsampled < 4096) { /* sample for the offset guess */
        rp->nsampled++;
        for (size_t i = 0; i < (qn < 64 ? qn : 64); i++) {
          if (q[i] < rp->qmin) rp->qmin = q[i];
          if (q[i] > rp->qmax) rp->qmax = q[i];
        }
      }
      batch_append(b, s, n, rp->want_quals ? q : NULL);
      batch_close_read(b); added++;
      p = qe < end ? qe + 1 : end;
    }
  }
  rp->p = p;
  return added;
}

/* first record boundary at or after `pos` */
static const unsigned char *find_record(const unsigned char *base, size_t size, size_t pos, seq_fmt fmt)
{
  const unsigned char *end = base + size, *p = base + pos;
  if (pos == 0) return base;
  /* move to the start of the next line */
  p = line_end(p - 1, end);
  if (p >= end) return end;
  p++;
  if (fmt == SEQ_FMT_PLAIN) return p;
  for (; p < end;) {
    const unsigned char *e1 = line_end(p, end);
    if (fmt == SEQ_FMT_FASTA) { if (*p == '>') return p; }
    else if (*p == '@' && e1 < end) {
      const unsigned char *l2 = e1 + 1, *e2 = line_end(l2, end);
      if (e2 < end && e2 + 1 < end && e2[1] == '+') return p;
    }
    if (e1 >= end) return end;
    p = e1 + 1;
  }
  return end;
}

static unsigned char *big_alloc(size_t n)
{
  const size_t huge = (size_t)2 << 20;
  n = (n + huge - 1) / huge * huge;
  void *p = aligned_alloc(huge, n);
  if (p) madvise(p, n, MADV_HUGEPAGE);
  return p;
}

/* ---- worker pool ---- */
typedef struct par_ctx par_ctx;
typedef struct {
  par_ctx *ctx;
  range_parser rp;
  read_batch batch[2]; /* double buffered: one being filled, one with the submitter */
  int ready;           /* index of a full batch waiting for the submitter, or -1 */
  bool done;
  pthread_t th;
} worker;

struct par_ctx {
  pthread_mutex_t mu;
  pthread_cond_t cv_ready, cv_free;
  worker *w;
  int nw;
};

static void *worker_main(void *arg)
{
  worker *w = arg;
  par_ctx *c = w->ctx;
  int cur = 0;
  for (;;) {
    read_batch_clear(&w->batch[cur]);
    size_t n = range_fill(&w->rp, &w->batch[cur]);
    pthread_mutex_lock(&c->mu);
    while (w->ready >= 0) pthread_cond_wait(&c->cv_free, &c->mu); /* previous batch not taken yet */
    if (n) w->ready = cur;
    const bool fin = w->rp.bad || w->rp.p >= w->rp.end;
    if (fin) w->done = true;
    pthread_cond_signal(&c->cv_ready);
    pthread_mutex_unlock(&c->mu);
    if (fin) break;
    cur ^= 1;
  }
  return NULL;
}

/* What is left to do when the last batch has been handed over: give back 2 x nthreads batch buffers (33 MB each, huge
 * pages) and unmap the file.  For a 12 GB FASTQ that took 150-190 ms on the build's critical path (round 5,
 * tools/exp_parse.sh: parse alone 80 ms, teardown 150 ms) -- a third of the whole ingest.  A detached thread does it
 * while the caller goes on to flush, sort and write the graph. */
typedef struct { worker *w; int nw; void *base; size_t size; } teardown_job;
/* (One teardown at a time: the next file's par_ingest joins the previous one before it allocates its own buffers --
 * with several --seq inputs the two sets, 2 x nthreads x up to 64 MB each, would otherwise overlap for a moment,
 * outside what -m accounts for.  The last input's teardown is simply left behind at exit.) */
static pthread_t g_teardown_th;
static bool g_teardown_live = false;
static void join_teardown(void)
{
  if (g_teardown_live) { pthread_join(g_teardown_th, NULL); g_teardown_live = false; }
}
static void *teardown_main(void *arg)
{
  teardown_job *j = arg;
  for (int t = 0; t < j->nw; t++) { read_batch_free(&j->w[t].batch[0]); read_batch_free(&j->w[t].batch[1]); }
  free(j->w);
  munmap(j->base, j->size);
  free(j);
  return NULL;
}

/* Parse `path` with nthreads threads, calling submit(batch) for every batch on the calling thread.
 * Returns 0 on success, 1 if the file is not suitable (caller uses the sequential parser; nothing
 * has been submitted), 2 if an irregular record was met after submission began. */
int par_ingest(const char *path, seq_fmt fmt, int nthreads, bool want_quals, size_t batch_bases,
               void (*submit)(void *arg, read_batch *b, int fq_offset_guess), void (*started)(void *arg), void *arg)
{
  if (nthreads < 2 || (fmt != SEQ_FMT_FASTA && fmt != SEQ_FMT_FASTQ && fmt != SEQ_FMT_PLAIN)) return 1;
  const bool timing = getenv("MCX_TIMING") != NULL;
  const double t_start = now_ms();
  double t_submit = 0;
  int fd = open(path, O_RDONLY);
  if (fd < 0) return 1;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < (off_t)(1 << 20)) { close(fd); return 1; }
  const size_t size = (size_t)st.st_size;
  const unsigned char *base = mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (base == MAP_FAILED) return 1;
  if (size >= 2 && base[0] == 0x1f && base[1] == 0x8b) { munmap((void *)base, size); return 1; } /* gzip */
  madvise((void *)base, size, MADV_SEQUENTIAL | MADV_WILLNEED);

  /* probe: the first records must have the regular structure */
  {
    range_parser rp = {base, base + (size < (4u << 20) ? size : (4u << 20)), fmt, false, 1u << 20, 255, 0, false, 0};
    read_batch b;
    read_batch_init(&b, false);
    range_fill(&rp, &b);
    read_batch_free(&b);
    if (rp.bad && rp.p < rp.end - 65536) { munmap((void *)base, size); return 1; }
  }

  par_ctx c;
  pthread_mutex_init(&c.mu, NULL);
  pthread_cond_init(&c.cv_ready, NULL);
  pthread_cond_init(&c.cv_free, NULL);
  c.nw = nthreads;
  join_teardown(); /* the previous input's buffers are back before this one's are taken */
  c.w = calloc((size_t)nthreads, sizeof(worker));
  const unsigned char *prev = base;
  for (int t = 0; t < nthreads; t++) {
    worker *w = &c.w[t];
    const unsigned char *next = t + 1 == nthreads ? base + size : find_record(base, size, size / (size_t)nthreads * (size_t)(t + 1), fmt);
    if (next < prev) next = prev;
    w->ctx = &c;
    w->rp = (range_parser){prev, next, fmt, want_quals, batch_bases, 255, 0, false, 0};
    w->ready = -1;
    for (int i = 0; i < 2; i++) { /* sized for a whole batch up front: no realloc growth while parsing */
      read_batch *b = &w->batch[i];
      read_batch_init(b, want_quals);
      b->cap_bases = batch_bases + (1u << 20);
      b->cap_reads = batch_bases / 32 + 1024;
      /* 2 MiB-aligned and advised for huge pages: 2 x nthreads buffers of 33 MB are first touched
       * while parsing, 4 KiB at a time that was half a million page faults on one address space */
      free(b->bases); free(b->quals);
      b->bases = big_alloc(b->cap_bases);
      b->quals = want_quals ? big_alloc(b->cap_bases) : NULL;
      b->offsets = realloc(b->offsets, (b->cap_reads + 1) * sizeof(uint64_t));
      if (!b->bases || !b->offsets || (want_quals && !b->quals)) die("Out of memory");
    }
    prev = next;
  }
  const double t_setup = now_ms();
  for (int t = 0; t < nthreads; t++) pthread_create(&c.w[t].th, NULL, worker_main, &c.w[t]);
  if (started) started(arg); /* the first batches are tens of milliseconds away */

  int rc = 0, live = nthreads;
  pthread_mutex_lock(&c.mu);
  while (live > 0) {
    bool progressed = false;
    for (int t = 0; t < nthreads; t++) {
      worker *w = &c.w[t];
      if (w->ready >= 0) {
        const int idx = w->ready;
        pthread_mutex_unlock(&c.mu);
        int guess = w->rp.qmax ? (w->rp.qmin >= 59 ? 64 : 33) : 0;
        const double ts = timing ? now_ms() : 0;
        submit(arg, &w->batch[idx], guess);
        if (timing) t_submit += now_ms() - ts;
        pthread_mutex_lock(&c.mu);
        w->ready = -1;
        pthread_cond_broadcast(&c.cv_free);
        progressed = true;
      }
      if (w->done && w->ready < 0 && w->th) {
        pthread_mutex_unlock(&c.mu);
        pthread_join(w->th, NULL);
        pthread_mutex_lock(&c.mu);
        w->th = 0;
        if (w->rp.bad) rc = 2;
        live--;
        progressed = true;
      }
    }
    if (!progressed && live > 0) pthread_cond_wait(&c.cv_ready, &c.mu);
  }
  pthread_mutex_unlock(&c.mu);
  const double t_parsed = now_ms();
  pthread_mutex_destroy(&c.mu);
  pthread_cond_destroy(&c.cv_ready);
  pthread_cond_destroy(&c.cv_free);
  {
    teardown_job *j = malloc(sizeof(*j));
    if (!j) die("Out of memory");
    *j = (teardown_job){c.w, nthreads, (void *)base, size};
    if (getenv("MCX_SYNC_TEARDOWN") || pthread_create(&g_teardown_th, NULL, teardown_main, j) != 0) teardown_main(j);
    else g_teardown_live = true;
  }
  if (timing)
    fprintf(stderr, "[timing] par_ingest %s: %d threads, %.1f MB: set-up %.1f ms, parse + submit %.1f ms (of which inside submit %.1f), teardown %.1f ms\n",
            path, nthreads, size / 1e6, t_setup - t_start, t_parsed - t_setup, t_submit, now_ms() - t_parsed);
  return rc;
}
