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

/* hash_mem.h:4-13, hash_mem.c:5-51 */
#define MAX_BUCKET_SIZE 48
static size_t ht_mem(size_t bktsize, size_t nbkts, size_t nbits) { return (bktsize * nbkts * nbits) / 8 + nbkts * 2; }

uint64_t hash_table_cap(uint64_t nkmers, uint64_t *nbuckets, uint8_t *bucket_size)
{
  uint64_t bits = 10;
  while (nkmers / (1UL << bits) > MAX_BUCKET_SIZE) bits++;
  uint64_t nb = 1UL << bits, bs = (nkmers + nb - 1) / nb;
  if (bs < 1) bs = 1;
  if (nbuckets) *nbuckets = nb;
  if (bucket_size) *bucket_size = (uint8_t)bs;
  return nb * bs;
}

size_t hash_table_mem(uint64_t nkmers, size_t entrybits, uint64_t *nkmers_out)
{
  uint64_t nb; uint8_t bs;
  uint64_t cap = hash_table_cap(nkmers, &nb, &bs);
  if (nkmers_out) *nkmers_out = cap;
  return ht_mem(bs, nb, entrybits);
}

size_t hash_table_mem_limit(size_t memlimit, size_t entrybits, uint64_t *nkmers_out)
{
  size_t bits = 10, nb = 1UL << bits, bs;
  while (ht_mem(MAX_BUCKET_SIZE, nb, entrybits) < memlimit) { bits++; nb = 1UL << bits; }
  bs = (memlimit - nb * 2) / ((nb * entrybits) / 8);
  if (bs == 0) { bits--; nb = 1UL << bits; bs = 1; }
  if (bs > MAX_BUCKET_SIZE) bs = MAX_BUCKET_SIZE;
  if (nkmers_out) *nkmers_out = nb * bs;
  return ht_mem(bs, nb, entrybits);
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
