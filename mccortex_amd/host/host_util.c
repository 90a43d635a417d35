/* host_util.c -- logging, number parsing, table sizing for the `build` host program. */
#define _GNU_SOURCE
#include "host.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

FILE *msg_out = NULL;
int host_fast_exit_ok = 0;
static char g_cmdline[4096] = "";
static char g_runcode[4] = "xxx";

void host_set_cmdline(int argc, char **argv)
{
  size_t n = 0;
  for (int i = 0; i < argc && n + strlen(argv[i]) + 2 < sizeof(g_cmdline); i++)
    n += (size_t)sprintf(g_cmdline + n, "%s%s", i ? " " : "", argv[i]);
  /* three-letter run tag, like the reference's status lines (ctx_output.h:26-34) */
  unsigned seed = (unsigned)time(NULL) ^ ((unsigned)getpid() << 7);
  static const char ab[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789";
  for (int i = 0; i < 3; i++) { seed = seed * 1103515245u + 12345u; g_runcode[i] = ab[(seed >> 16) % 36]; }
}

static void stamp(FILE *fh)
{
  char buf[64];
  time_t t = time(NULL);
  strftime(buf, sizeof(buf), "%d %b %Y %H:%M:%S", localtime(&t));
  fprintf(fh, "[%s-%s] ", buf, g_runcode);
}

void status(const char *fmt, ...)
{
  if (!msg_out) return;
  va_list ap;
  va_start(ap, fmt);
  stamp(msg_out);
  vfprintf(msg_out, fmt, ap);
  va_end(ap);
  if (!*fmt || fmt[strlen(fmt) - 1] != '\n') fputc('\n', msg_out);
}

void warn(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  stamp(stderr);
  fputs("Warning: ", stderr);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  if (!*fmt || fmt[strlen(fmt) - 1] != '\n') fputc('\n', stderr);
}

void die(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  fflush(stdout);
  stamp(stderr);
  fputs("Fatal Error: ", stderr);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  if (!*fmt || fmt[strlen(fmt) - 1] != '\n') fputc('\n', stderr);
  exit(EXIT_FAILURE);
}

/* cmd_print_usage (src/basic/cmd.c:279-296): header, optional error, usage, EXIT_FAILURE */
void print_usage(const char *usage, const char *errfmt, ...)
{
  msg_out = stderr;
  status("[cmd] %s", g_cmdline);
  if (errfmt) {
    fputs("\nError: ", stderr);
    va_list ap;
    va_start(ap, errfmt);
    vfprintf(stderr, errfmt, ap);
    va_end(ap);
    if (errfmt[strlen(errfmt) - 1] != '\n') fputc('\n', stderr);
    fputc('\n', stderr);
  }
  fputs(usage, stderr);
  exit(EXIT_FAILURE);
}

/* util.c:108-117 */
bool parse_entire_size(const char *s, size_t *out)
{
  char *end = NULL;
  if (*s < '0' || *s > '9') return false;
  unsigned long v = strtoul(s, &end, 10);
  if (!end || *end != '\0') return false;
  *out = (size_t)v;
  return true;
}

bool parse_entire_uint(const char *s, unsigned *out)
{
  size_t v;
  if (!parse_entire_size(s, &v) || v > 0xFFFFFFFFul) return false;
  *out = (unsigned)v;
  return true;
}

/* util.c:206-222: K/M/G/T (optionally with B) are binary units */
bool mem_to_integer(const char *arg, size_t *bytes)
{
  char *end;
  unsigned long num = strtoul(arg, &end, 10);
  if (end == arg) return false;
  static const struct { const char *a, *b; int shift; } units[] = {
    {"T", "TB", 40}, {"G", "GB", 30}, {"M", "MB", 20}, {"K", "KB", 10}};
  for (size_t i = 0; i < 4; i++)
    if (!strcasecmp(end, units[i].a) || !strcasecmp(end, units[i].b)) { *bytes = num << units[i].shift; return true; }
  if (*end != '\0') return false;
  *bytes = num;
  return true;
}

/* util.c:251-264: thousands separators */
char *ulong_to_str(unsigned long num, char *out)
{
  char tmp[32];
  int n = sprintf(tmp, "%lu", num), o = 0;
  for (int i = 0; i < n; i++) {
    out[o++] = tmp[i];
    if ((n - 1 - i) % 3 == 0 && i != n - 1) out[o++] = ',';
  }
  out[o] = '\0';
  return out;
}

/* util.c:310-347: 1024-based units, trailing zeros trimmed */
char *bytes_to_str(unsigned long num, int decimals, char *out)
{
  static const char *units[] = {"B", "KB", "MB", "GB", "TB", "PB", "EB"};
  int u = 0;
  double v = (double)num;
  while (v >= 1024 && u + 1 < 7) { v /= 1024; u++; }
  char buf[64];
  sprintf(buf, "%.*f", decimals, v);
  if (strchr(buf, '.')) {
    char *p = buf + strlen(buf) - 1;
    while (*p == '0') *p-- = '\0';
    if (*p == '.') *p = '\0';
  }
  sprintf(out, "%s%s", buf, units[u]);
  return out;
}

/* ---- table sizing ------------------------------------------------------------------------------
 * `-n <kmers>` and `-m <mem>` must size the table as the reference does, or a command line that
 * fits there does not fit here: the reference's table is 2^b buckets of s <= 48 entries plus two
 * bytes per bucket (src/basic/hash_mem.c:5-51).  The HBM table has no buckets of that kind; only the
 * resulting entry count (and the memory figure the status lines print) is taken from this plan. */
#define PLAN_MAX_BUCKET 48
static size_t plan_bytes(const table_plan *p, size_t entry_bits) { return (size_t)(p->bucket_size * p->nbuckets * entry_bits) / 8 + (size_t)p->nbuckets * 2; }

table_plan table_plan_for_kmers(uint64_t nkmers, size_t entry_bits)
{ /* smallest power-of-two bucket count (>= 1024) whose buckets need at most 48 entries each */
  table_plan p;
  unsigned b = 10;
  while (nkmers / (1UL << b) > PLAN_MAX_BUCKET) b++;
  p.nbuckets = 1UL << b;
  p.bucket_size = (nkmers + p.nbuckets - 1) / p.nbuckets;
  if (p.bucket_size < 1) p.bucket_size = 1;
  p.capacity = p.nbuckets * p.bucket_size;
  p.bytes = plan_bytes(&p, entry_bits);
  return p;
}

table_plan table_plan_for_memory(size_t mem, size_t entry_bits)
{ /* the most entries that fit `mem`: first bucket count whose full-size table reaches it, filled as far as it goes */
  table_plan p;
  unsigned b = 10;
  p.nbuckets = 1UL << b; p.bucket_size = PLAN_MAX_BUCKET;
  while (plan_bytes(&p, entry_bits) < mem) p.nbuckets = 1UL << ++b;
  p.bucket_size = (mem - p.nbuckets * 2) / ((p.nbuckets * entry_bits) / 8);
  if (p.bucket_size == 0) { p.nbuckets = 1UL << --b; p.bucket_size = 1; }
  if (p.bucket_size > PLAN_MAX_BUCKET) p.bucket_size = PLAN_MAX_BUCKET;
  p.capacity = p.nbuckets * p.bucket_size;
  p.bytes = plan_bytes(&p, entry_bits);
  return p;
}

/* cmd_get_kmers_in_hash as `build` calls it (src/graph/cmd_mem.c:38-130 with min_num_kmer_req = 0,
 * use_mem_limit = true; src/commands/ctx_build.c:317-322): -n wins over -m; with neither, the 512 MB
 * default of -m is filled; an estimate of the k-mers the inputs can hold (max_kmers > 0, at the
 * ideal occupancy 0.75) caps the table unless -n was given; never fewer than 1024 entries.
 * Returns NULL or the reference's error message (the caller dies with it). */
const char *table_plan_for_build(size_t mem_to_use, bool mem_set, size_t num_kmers, bool nkmers_set, size_t entry_bits,
                                 int64_t max_kmers, table_plan *out, char *errbuf, size_t errlen)
{
  table_plan p = nkmers_set ? table_plan_for_kmers(num_kmers, entry_bits) : table_plan_for_memory(mem_to_use, entry_bits);
  if (max_kmers > 0 && !nkmers_set) {
    /* (single precision, as `max_num_kmers_req/IDEAL_OCCUPANCY` with IDEAL_OCCUPANCY 0.75f is: cmd_mem.c:66, hash_mem.h:5) */
    const table_plan q = table_plan_for_kmers((uint64_t)((float)max_kmers / 0.75f), entry_bits);
    if (q.bytes < p.bytes) p = q;
  }
  if (p.capacity < 1024) p = table_plan_for_kmers(1024, entry_bits);
  *out = p;
  char s1[64], s2[64];
  if (mem_set && nkmers_set && num_kmers > p.capacity) {
    snprintf(errbuf, errlen, "-n <kmers> requires more memory than given with -m <mem> [%s > %s]",
             bytes_to_str(p.bytes, 1, s1), bytes_to_str(mem_to_use, 1, s2));
    return errbuf;
  }
  /* As the reference (cmd_mem.c:120-123): graph_mem is checked against -m whether or not -m was
   * given (its default is 512 MB), so `-n 1G` alone dies here as it dies there, with the same
   * message -- a command line behaves alike in both.  (Until round 3 the check was only made for an
   * explicit -m.)  The HBM the table really takes is checked apart. */
  (void)mem_set;
  if (p.bytes > mem_to_use) {
    snprintf(errbuf, errlen, "Not enough memory for requested graph: require at least %s [>%s]",
             bytes_to_str(p.bytes, 1, s1), bytes_to_str(mem_to_use, 1, s2));
    return errbuf;
  }
  return NULL;
}

void kmer_words_to_str(const unsigned char *rec, unsigned kmer_size, char *out)
{
  const unsigned W = (2 * kmer_size + 63) / 64;
  for (unsigned i = 0; i < kmer_size; i++) {
    /* base i (first base = most significant): bit offset from the low end of the W-word integer */
    const unsigned bit = 2 * (kmer_size - 1 - i);
    const unsigned word = W - 1 - bit / 64, sh = bit % 64;
    uint64_t w;
    memcpy(&w, rec + 8 * word, 8);
    out[i] = "ACGT"[(w >> sh) & 3];
  }
  out[kmer_size] = '\0';
}
