/* seq_in.c -- FASTA / FASTQ / plain sequence input (optionally gzip'd, "-" = stdin).
 * Stands in for the reference's seq_file library + the per-file reader thread of
 * src/basic/async_read_io.c:145-175 for the formats its tests feed `build`.  Reads are
 * delivered as flat batches (bases, quals, offsets) ready for mcx_graph_add_reads().
 * SAM/BAM/CRAM (htslib in the reference) are not handled on this path. */
#include "host.h"

#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <zlib.h>

#define IN_BUF (4u << 20)

struct seq_in {
  gzFile gz;
  char *path;
  seq_fmt fmt;
  unsigned char *buf;
  size_t pos, end;
  bool eof;
  int pending;            /* first byte of the next record header already consumed (FASTA) */
  int qmin, qmax;         /* quality range seen (for the offset guess) */
  char *line; size_t line_cap;
};

static bool refill(seq_in *s)
{
  if (s->eof) return false;
  int n = gzread(s->gz, s->buf, IN_BUF);
  if (n <= 0) { s->eof = true; s->pos = s->end = 0; return false; }
  s->pos = 0; s->end = (size_t)n;
  return true;
}

static int peek_byte(seq_in *s)
{
  if (s->pos == s->end && !refill(s)) return -1;
  return s->buf[s->pos];
}

/* Read one line (without the terminator, '\r' stripped) into s->line; returns length or -1 at EOF */
static long read_line(seq_in *s)
{
  size_t len = 0;
  bool got = false;
  for (;;) {
    if (s->pos == s->end && !refill(s)) break;
    got = true;
    unsigned char *p = s->buf + s->pos, *e = s->buf + s->end;
    unsigned char *nl = memchr(p, '\n', (size_t)(e - p));
    size_t take = nl ? (size_t)(nl - p) : (size_t)(e - p);
    if (len + take + 1 > s->line_cap) {
      s->line_cap = (len + take + 1) * 2;
      s->line = realloc(s->line, s->line_cap);
      if (!s->line) die("Out of memory");
    }
    memcpy(s->line + len, p, take);
    len += take;
    s->pos += take + (nl ? 1 : 0);
    if (nl) break;
  }
  if (!got) return -1;
  while (len && (s->line[len - 1] == '\r')) len--;
  s->line[len] = '\0';
  return (long)len;
}

seq_in *seq_in_open(const char *path)
{
  seq_in *s = calloc(1, sizeof(*s));
  if (!s) die("Out of memory");
  s->gz = strcmp(path, "-") == 0 ? gzdopen(0, "rb") : gzopen(path, "rb");
  if (!s->gz) { free(s); return NULL; }
  gzbuffer(s->gz, 1u << 20);
  s->path = strdup(path);
  s->buf = malloc(IN_BUF);
  s->line_cap = 1 << 16; s->line = malloc(s->line_cap);
  s->qmin = 255; s->qmax = 0;
  s->pending = -1;
  /* format from the first non-blank byte */
  int c;
  while ((c = peek_byte(s)) == '\n' || c == '\r' || c == ' ' || c == '\t') s->pos++;
  if (c < 0) s->fmt = SEQ_FMT_PLAIN; /* empty file: no reads */
  else if (c == '>') s->fmt = SEQ_FMT_FASTA;
  else if (c == '@') {
    s->fmt = SEQ_FMT_FASTQ;
    if (s->end - s->pos >= 4 && (!memcmp(s->buf + s->pos, "@HD\t", 4) || !memcmp(s->buf + s->pos, "@SQ\t", 4)))
      s->fmt = SEQ_FMT_SAM;
  } else if (s->end - s->pos >= 4 && !memcmp(s->buf + s->pos, "BAM\1", 4)) s->fmt = SEQ_FMT_SAM;
  else s->fmt = SEQ_FMT_PLAIN;
  return s;
}

void seq_in_close(seq_in *s)
{
  if (!s) return;
  gzclose(s->gz);
  free(s->path); free(s->buf); free(s->line); free(s);
}

seq_fmt seq_in_format(seq_in *s) { return s->fmt; }
const char *seq_in_path(const seq_in *s) { return s->path; }

/* FASTQ offset from the range of quality characters seen.  The reference asks seq_file
 * (seq_guess_fastq_format, third-party library github.com/noporpoise/seq_file, an empty submodule
 * in this checkout: no pinned version) through guess_fastq_format (src/basic/seq_reader.c:252-288,
 * which prints the table's FASTQ_OFFSET / FASTQ_MIN / FASTQ_MAX).  seq_file's published rule: the
 * Sanger range is tested FIRST -- min >= 33 and max <= 73 is "Sanger (Phred+33)" even when every
 * quality is high (simulated reads, binned qualities: all 'I') --, a wider range from 33 is Sanger /
 * Illumina 1.8+, then the Phred+64 families by their minimum (67: Illumina 1.5+, 64: Illumina 1.3+,
 * 59: Solexa) with max <= 104 (one above the nominal top is tolerated); anything else is read as
 * offset 33. */
int fq_offset_from_range(int qmin, int qmax)
{
  if (qmax == 0 || qmin > qmax) return 0;          /* no qualities seen */
  if (qmin >= 33 && qmax <= 73) return 33;          /* Sanger */
  if (qmin >= 33 && qmin < 59) return 33;           /* Sanger / Illumina 1.8+, range up to 126 */
  if (qmin >= 59 && qmax <= 105) return 64;         /* Solexa (59), Illumina 1.3+ (64), Illumina 1.5+ (66/67) */
  return 33;                                        /* unknown: offset 33, max 126 */
}

int seq_in_guess_fq_offset(const seq_in *s) { return fq_offset_from_range(s->qmin, s->qmax); }

/* The offset of a file, decided ONCE from the head of the file (the first 1000 records with
 * qualities) before anything of it is loaded, so that it depends neither on -t nor on which batch
 * is parsed first.  0 = not a FASTQ file / no qualities; -1 = cannot look ahead: stdin, a FIFO,
 * a process substitution (/dev/fd/N) -- anything that is not a regular file can only be read once,
 * so probing it would swallow the head of the stream; the offset is then latched from the first
 * batch of the real load (fq_abs_of / the ingest callback in cmd_build.c). */
int fq_offset_probe(const char *path)
{
  struct stat st;
  if (strcmp(path, "-") == 0 || stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return -1;
  seq_in *s = seq_in_open(path);
  if (!s) return 0;
  int off = 0;
  if (s->fmt == SEQ_FMT_FASTQ) {
    read_batch b;
    read_batch_init(&b, false);
    while (b.nreads < 1000 && seq_in_fill(s, &b, b.nbases + 1) > 0) {}
    off = fq_offset_from_range(s->qmin, s->qmax);
    read_batch_free(&b);
  }
  seq_in_close(s);
  return off;
}

void read_batch_init(read_batch *b, bool want_quals)
{
  memset(b, 0, sizeof(*b));
  b->want_quals = want_quals;
  b->cap_bases = 1 << 20; b->cap_reads = 1 << 12;
  b->bases = malloc(b->cap_bases);
  b->quals = want_quals ? malloc(b->cap_bases) : NULL;
  b->offsets = malloc((b->cap_reads + 1) * sizeof(uint64_t));
  if (!b->bases || !b->offsets || (want_quals && !b->quals)) die("Out of memory");
  b->offsets[0] = 0;
}

void read_batch_clear(read_batch *b) { b->nreads = 0; b->nbases = 0; b->offsets[0] = 0; }

void read_batch_free(read_batch *b) { free(b->bases); free(b->quals); free(b->offsets); memset(b, 0, sizeof(*b)); }

static void batch_reserve(read_batch *b, size_t extra)
{
  if (b->nbases + extra > b->cap_bases) {
    while (b->nbases + extra > b->cap_bases) b->cap_bases *= 2;
    b->bases = realloc(b->bases, b->cap_bases);
    if (b->want_quals) b->quals = realloc(b->quals, b->cap_bases);
    if (!b->bases || (b->want_quals && !b->quals)) die("Out of memory");
  }
}

static void batch_end_read(read_batch *b)
{
  if (b->nreads + 1 > b->cap_reads) {
    b->cap_reads *= 2;
    b->offsets = realloc(b->offsets, (b->cap_reads + 1) * sizeof(uint64_t));
    if (!b->offsets) die("Out of memory");
  }
  b->offsets[++b->nreads] = b->nbases;
}

void read_batch_append(read_batch *dst, const read_batch *src, size_t i)
{
  const size_t at = (size_t)src->offsets[i], n = (size_t)(src->offsets[i + 1] - src->offsets[i]);
  batch_reserve(dst, n);
  memcpy(dst->bases + dst->nbases, src->bases + at, n);
  if (dst->want_quals) {
    if (src->want_quals) memcpy(dst->quals + dst->nbases, src->quals + at, n);
    else memset(dst->quals + dst->nbases, 0, n);
  }
  dst->nbases += n;
  batch_end_read(dst);
}

size_t seq_in_fill(seq_in *s, read_batch *b, size_t max_bases)
{
  size_t added = 0;
  long n;
  if (s->fmt == SEQ_FMT_SAM) die("SAM/BAM/CRAM input is not supported by this build: %s", s->path);
  while (b->nbases < max_bases) {
    if (s->fmt == SEQ_FMT_PLAIN) {
      if ((n = read_line(s)) < 0) break;
      if (n == 0) continue;
      batch_reserve(b, (size_t)n);
      memcpy(b->bases + b->nbases, s->line, (size_t)n);
      if (b->want_quals) memset(b->quals + b->nbases, 0, (size_t)n);
      b->nbases += (size_t)n;
      batch_end_read(b); added++;
    } else if (s->fmt == SEQ_FMT_FASTA) {
      /* header line */
      if (s->pending < 0) {
        if ((n = read_line(s)) < 0) break;
        if (n == 0) continue;
        if (s->line[0] != '>') die("Expected '>' in FASTA file %s, got: %.20s", s->path, s->line);
      }
      s->pending = -1;
      int c;
      while ((c = peek_byte(s)) >= 0 && c != '>') {
        n = read_line(s);
        if (n <= 0) continue;
        batch_reserve(b, (size_t)n);
        memcpy(b->bases + b->nbases, s->line, (size_t)n);
        if (b->want_quals) memset(b->quals + b->nbases, 0, (size_t)n);
        b->nbases += (size_t)n;
      }
      batch_end_read(b); added++;
    } else { /* FASTQ */
      if ((n = read_line(s)) < 0) break;
      if (n == 0) continue;
      if (s->line[0] != '@') die("Expected '@' in FASTQ file %s, got: %.20s", s->path, s->line);
      size_t start = b->nbases, slen = 0;
      while ((n = read_line(s)) >= 0 && !(n > 0 && s->line[0] == '+')) {
        batch_reserve(b, (size_t)n);
        memcpy(b->bases + b->nbases, s->line, (size_t)n);
        b->nbases += (size_t)n; slen += (size_t)n;
      }
      if (n < 0) die("Truncated FASTQ record in %s", s->path);
      size_t qlen = 0;
      while (qlen < slen && (n = read_line(s)) >= 0) {
        size_t take = (size_t)n;
        if (qlen + take > slen) take = slen - qlen;
        for (size_t i = 0; i < take; i++) {
          int q = (unsigned char)s->line[i];
          if (q < s->qmin) s->qmin = q;
          if (q > s->qmax) s->qmax = q;
        }
        if (b->want_quals) memcpy(b->quals + start + qlen, s->line, take);
        qlen += take;
      }
      if (qlen < slen) {
        warn("FASTQ record with fewer qualities than bases in %s", s->path);
        if (b->want_quals) memset(b->quals + start + qlen, 0, slen - qlen);
      }
      batch_end_read(b); added++;
    }
  }
  return added;
}
