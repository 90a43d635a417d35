/* ctx_file.c -- .ctx header writer and reader, colour filters and the per-colour GraphInfo
 * arithmetic (docs/file_formats/graph_file_format.txt; src/graph/graph_writer.c:11-110,
 * src/graph/graph_file_reader.c:78-340, src/basic/file_filter.c, src/basic/range.c,
 * src/basic/graph_info.c).  The reader is written from the format document: a table of header
 * fields and one scanner for colour lists; the reference's checks and messages are kept.
 * x86-64 only: the header stores `long double seq_err` as its 16
 * in-memory bytes (10-byte x87 value + padding). */
#include "host.h"

#include <errno.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

static char *dupstr(const char *s)
{
  char *d = malloc(strlen(s) + 1);
  if (!d) die("Out of memory");
  strcpy(d, s);
  return d;
}

static void append_str(char **dst, char sep, const char *s)
{
  const size_t n = strlen(*dst);
  *dst = realloc(*dst, n + 1 + strlen(s) + 1);
  if (!*dst) die("Out of memory");
  (*dst)[n] = sep;
  strcpy(*dst + n + 1, s);
}

void col_info_init(col_info *c)
{ /* graph_info_alloc + graph_info_init: graph_info.c:60-75 */
  c->mean_read_length = 0;
  c->total_sequence = 0;
  c->seq_err = 0.01;
  c->name = dupstr("undefined");
  memset(&c->cleaning, 0, sizeof(c->cleaning));
  c->cleaning.intersection_name = dupstr("undefined");
}

void col_info_free(col_info *c)
{
  free(c->name); free(c->cleaning.intersection_name);
  c->name = c->cleaning.intersection_name = NULL;
}

void col_info_set_name(col_info *c, const char *name)
{
  free(c->name);
  c->name = dupstr(name);
}

/* graph_info_update_contigs: graph_info.c:116-133 */
static void update_contigs(uint32_t *mean, uint64_t *total, uint64_t added, uint64_t ncontigs)
{
  if (!added && !ncontigs) return;
  size_t have = 0;
  if (*total && *mean) have = (size_t)(((double)*total / *mean) + 0.5);
  if (have + ncontigs > 0) *mean = (uint32_t)((double)(*total + added) / (double)(have + ncontigs));
  *total += added;
}

/* graph_info_update_stats: graph_info.c:172-175 -- call once per input file, in task order */
void col_info_update(col_info *c, uint64_t bases_loaded, uint64_t contigs)
{
  update_contigs(&c->mean_read_length, &c->total_sequence, bases_loaded, contigs);
}

/* error_cleaning_merge + graph_info_append_intersect: graph_info.c:35-59,91-103 */
static void cleaning_merge(err_cleaning *dst, const err_cleaning *src)
{
  dst->cleaned_tips |= src->cleaned_tips;
  dst->cleaned_unitigs |= src->cleaned_unitigs;
  dst->cleaned_kmers |= src->cleaned_kmers;
  if (src->clean_unitigs_thresh > 0 && (dst->clean_unitigs_thresh == 0 || src->clean_unitigs_thresh < dst->clean_unitigs_thresh))
    dst->clean_unitigs_thresh = src->clean_unitigs_thresh;
  if (src->clean_kmers_thresh > 0 && (dst->clean_kmers_thresh == 0 || src->clean_kmers_thresh < dst->clean_kmers_thresh))
    dst->clean_kmers_thresh = src->clean_kmers_thresh;
  if (src->is_graph_intersection) {
    if (!dst->is_graph_intersection) { free(dst->intersection_name); dst->intersection_name = dupstr(src->intersection_name); }
    else append_str(&dst->intersection_name, ',', src->intersection_name);
    dst->is_graph_intersection = 1;
  }
  dst->is_graph_intersection |= src->is_graph_intersection;
}

/* graph_info_merge: graph_info.c:135-170 */
void col_info_merge(col_info *dst, const col_info *src)
{
  if (strcmp(src->name, "undefined") != 0) {
    if (strcmp(dst->name, "undefined") == 0) col_info_set_name(dst, src->name);
    else append_str(&dst->name, ',', src->name);
  }
  const uint64_t total = dst->total_sequence + src->total_sequence;
  if (total > 0) {
    dst->seq_err = (dst->seq_err * dst->total_sequence + src->seq_err * src->total_sequence) / total;
    size_t src_contigs = 0;
    if (src->total_sequence && src->mean_read_length)
      src_contigs = (size_t)(((double)src->total_sequence / src->mean_read_length) + 0.5);
    update_contigs(&dst->mean_read_length, &dst->total_sequence, src->total_sequence, src_contigs);
  }
  cleaning_merge(&dst->cleaning, &src->cleaning);
  dst->total_sequence = total;
}

static size_t put(FILE *fh, const void *p, size_t n) { return fwrite(p, 1, n, fh); }

static size_t put_cleaning(FILE *fh, const err_cleaning *ec)
{ /* write_error_cleaning_object: graph_writer.c:33-60 */
  size_t n = 0;
  n += put(fh, &ec->cleaned_tips, 1); n += put(fh, &ec->cleaned_unitigs, 1);
  n += put(fh, &ec->cleaned_kmers, 1); n += put(fh, &ec->is_graph_intersection, 1);
  const uint32_t tu = ec->cleaned_unitigs ? ec->clean_unitigs_thresh : 0, tk = ec->cleaned_kmers ? ec->clean_kmers_thresh : 0;
  n += put(fh, &tu, 4); n += put(fh, &tk, 4);
  const uint32_t len = (uint32_t)strlen(ec->intersection_name);
  n += put(fh, &len, 4); n += put(fh, ec->intersection_name, len);
  return n;
}

size_t ctx_write_header(FILE *fh, uint32_t kmer_size, uint32_t ncols, const col_info *cols)
{
  _Static_assert(sizeof(long double) == 16, "x87 long double layout expected");
  size_t n = 0;
  const uint32_t version = 6, W = (2 * kmer_size + 63) / 64;
  n += put(fh, "CORTEX", 6);
  n += put(fh, &version, 4); n += put(fh, &kmer_size, 4); n += put(fh, &W, 4); n += put(fh, &ncols, 4);

  /* graph_writer_mkhdr merges every colour into a fresh GraphInfo (graph_writer.c:11-30): the
   * mean is re-derived from total/contigs and may differ from the graph's */
  col_info *h = calloc(ncols, sizeof(col_info));
  for (uint32_t c = 0; c < ncols; c++) { col_info_init(&h[c]); col_info_merge(&h[c], &cols[c]); }
  for (uint32_t c = 0; c < ncols; c++) n += put(fh, &h[c].mean_read_length, 4);
  for (uint32_t c = 0; c < ncols; c++) n += put(fh, &h[c].total_sequence, 8);
  for (uint32_t c = 0; c < ncols; c++) {
    uint32_t len = (uint32_t)strlen(h[c].name);
    n += put(fh, &len, 4); n += put(fh, h[c].name, len);
  }
  for (uint32_t c = 0; c < ncols; c++) {
    unsigned char b[16] = {0};
    memcpy(b, &h[c].seq_err, 10);
    n += put(fh, b, 16);
  }
  for (uint32_t c = 0; c < ncols; c++) n += put_cleaning(fh, &h[c].cleaning);
  n += put(fh, "CORTEX", 6);
  for (uint32_t c = 0; c < ncols; c++) col_info_free(&h[c]);
  free(h);
  return n;
}

/* ---- colour lists: "3", "0-2", "4-1" (descending), "0,2-3,7"; the syntax of the CLI's
 * "<into>:in.ctx:<from>" graph arguments (semantics of src/basic/range.c) ----
 * One scanner over a (pointer, length) slice serves both the sizing pass (out == NULL) and the
 * filling pass.  An empty slice stands for "every colour 0..max". */
#define COL_LIST_CHARS "0123456789,-"

typedef struct { const char *s; size_t n; } slice;

static long long col_list_expand(slice txt, uint64_t max, uint32_t *out)
{
  if (txt.n == 0) { /* nothing named: all of them */
    if (out) for (uint64_t c = 0; c <= max; c++) out[c] = (uint32_t)c;
    return (long long)(max + 1);
  }
  long long count = 0;
  uint64_t num[2] = {0, 0};  /* the item being read: num[0], or num[0]-num[1] */
  int which = 0, digits = 0; /* which number of the item, digits seen in it */
  for (size_t i = 0; i <= txt.n; i++) {
    const char ch = i < txt.n ? txt.s[i] : ','; /* the end of the text closes the last item */
    if (ch >= '0' && ch <= '9') {
      num[which] = num[which] * 10 + (uint64_t)(ch - '0');
      if (num[which] > max) return -1;
      digits++;
    } else if (ch == '-') {
      if (which == 1 || digits == 0) return -1;
      which = 1; digits = 0;
    } else if (ch == ',') {
      if (digits == 0) return -1; /* empty item, ",," or a comma at either end */
      const uint64_t first = num[0], last = which ? num[1] : num[0];
      const uint64_t span = (first <= last ? last - first : first - last) + 1;
      if (out)
        for (uint64_t j = 0; j < span; j++) out[count + (long long)j] = (uint32_t)(first <= last ? first + j : first - j);
      count += (long long)span;
      num[0] = num[1] = 0; which = 0; digits = 0;
    } else return -1;
  }
  return count;
}

/* ---- graph argument "<into>:path:<from>" (semantics of src/basic/file_filter.c:7-143) ----
 * <into> is a leading colour list closed by ':', <from> a trailing one opened by ':'; the path
 * between them keeps at least one character. */
typedef struct { slice into, path, from; bool has_into, has_from; } graph_arg;

static graph_arg graph_arg_split(const char *arg)
{
  graph_arg ga;
  memset(&ga, 0, sizeof(ga));
  const size_t len = strlen(arg);
  size_t lo = 0, hi = len;
  const size_t lead = strspn(arg, COL_LIST_CHARS);
  if (lead > 0 && arg[lead] == ':') {
    ga.has_into = true; ga.into = (slice){arg, lead};
    lo = lead + 1;
  }
  size_t t = len;
  while (t > lo + 1 && memchr(COL_LIST_CHARS, arg[t - 1], sizeof(COL_LIST_CHARS) - 1)) t--;
  if (t > lo + 1 && arg[t - 1] == ':') {
    ga.has_from = true; ga.from = (slice){arg + t, len - t};
    hi = t - 1;
  }
  ga.path = (slice){arg + lo, hi - lo};
  return ga;
}

static int filter_order(const void *a, const void *b)
{
  const col_filter *x = a, *y = b;
  const uint64_t kx = (uint64_t)x->into << 32 | x->from, ky = (uint64_t)y->into << 32 | y->from;
  return (kx > ky) - (kx < ky);
}

/* Colours of the file (`from`) and where they go in the graph (`into`).  Without <from>: every colour
 * of the file; without <into>: into_offset, into_offset + 1, ...; one <into> colour takes them all. */
static void reader_set_filter(ctx_reader *r, size_t into_offset)
{
  const graph_arg ga = graph_arg_split(r->input);
  const uint64_t file_max = (uint64_t)r->num_cols - 1, into_max = UINT32_MAX - 1;

  long long nfrom = ga.has_from ? col_list_expand(ga.from, file_max, NULL) : (long long)r->num_cols;
  if (nfrom < 0) die("Invalid filter path: %s (from size: %zu)", r->input, (size_t)r->num_cols);
  long long ninto = ga.has_into ? col_list_expand(ga.into, into_max, NULL) : nfrom;
  if (ninto < 0 || (ninto != 1 && ninto != nfrom))
    die("Invalid filter path: %s (s:%i ncols:%zu)", r->input, (int)ninto, (size_t)nfrom);

  const size_t n = (size_t)nfrom;
  uint32_t *from = calloc(n + 1, sizeof(uint32_t)), *into = calloc(n + 1, sizeof(uint32_t));
  r->filter = calloc(n + 1, sizeof(col_filter));
  if (!from || !into || !r->filter) die("Out of memory");
  col_list_expand(ga.has_from ? ga.from : (slice){NULL, 0}, file_max, from);
  if (ga.has_into) {
    col_list_expand(ga.into, into_max, into);
    if (ninto == 1) for (size_t i = 1; i < n; i++) into[i] = into[0];
  } else for (size_t i = 0; i < n; i++) into[i] = (uint32_t)(into_offset + i);

  r->nfilter = n;
  r->into_ncols = 0;
  for (size_t i = 0; i < n; i++) {
    r->filter[i] = (col_filter){from[i], into[i]};
    if ((size_t)into[i] + 1 > r->into_ncols) r->into_ncols = (size_t)into[i] + 1;
  }
  qsort(r->filter, n, sizeof(col_filter), filter_order);
  free(from); free(into);
}

/* ---- header reader, driven by a description of docs/file_formats/graph_file_format.txt
 * (checks and messages of graph_file_read_header: graph_file_reader.c:78-260) ---- */
typedef enum { FT_U8, FT_U32, FT_U64, FT_LDBL, FT_TEXT } field_type;
typedef struct {
  const char *what;   /* how a short read names the field */
  field_type type;
  size_t offset;      /* in the record the section fills */
  uint32_t since;     /* first format version that has it */
} hdr_field;

static const hdr_field hdr_dims[] = { /* into ctx_reader */
  {"graph version", FT_U32, offsetof(ctx_reader, version), 4},
  {"kmer size", FT_U32, offsetof(ctx_reader, kmer_size), 4},
  {"num of bitfields", FT_U32, offsetof(ctx_reader, num_words), 4},
  {"number of colours", FT_U32, offsetof(ctx_reader, num_cols), 4},
};
/* one array of <cols> entries each, in this order; into col_info */
static const hdr_field hdr_colour_arrays[] = {
  {"mean read length for each colour", FT_U32, offsetof(col_info, mean_read_length), 4},
  {"total sequence loaded for each colour", FT_U64, offsetof(col_info, total_sequence), 4},
  {"sample name", FT_TEXT, offsetof(col_info, name), 6},
  {"seq error rates", FT_LDBL, offsetof(col_info, seq_err), 6},
};
/* one block per colour; into err_cleaning */
static const hdr_field hdr_cleaning[] = {
  {"tip cleaning", FT_U8, offsetof(err_cleaning, cleaned_tips), 6},
  {"remove low covg unitig", FT_U8, offsetof(err_cleaning, cleaned_unitigs), 6},
  {"remove low covg kmers", FT_U8, offsetof(err_cleaning, cleaned_kmers), 6},
  {"cleaned against graph", FT_U8, offsetof(err_cleaning, is_graph_intersection), 6},
  {"remove low covg unitig threshold", FT_U32, offsetof(err_cleaning, clean_unitigs_thresh), 6},
  {"remove low covg kmer threshold", FT_U32, offsetof(err_cleaning, clean_kmers_thresh), 6},
  {"cleaned against graph name", FT_TEXT, offsetof(err_cleaning, intersection_name), 6},
};
#define NFIELDS(t) (sizeof(t) / sizeof((t)[0]))

static size_t hdr_take(ctx_reader *r, void *dst, size_t n, const char *what)
{
  const size_t got = fread(dst, 1, n, r->fh);
  if (got != n) die("Unexpected end of file: %s [%s; read %zu of %zu bytes]", r->path, what, got, n);
  return n;
}

/* reads one field into record + f->offset; returns the bytes it took from the file */
static size_t hdr_field_read(ctx_reader *r, const hdr_field *f, void *record, size_t colour)
{
  static const size_t width[] = {[FT_U8] = 1, [FT_U32] = 4, [FT_U64] = 8, [FT_LDBL] = 16};
  char *dst = (char *)record + f->offset;
  if (f->type != FT_TEXT) return hdr_take(r, dst, width[f->type], f->what);
  uint32_t len;
  size_t n = hdr_take(r, &len, 4, f->what);
  if (len > 10000) die("Very big sample name. Length: %u", len);
  char *text = calloc((size_t)len + 1, 1);
  if (!text) die("Out of memory");
  n += hdr_take(r, text, len, f->what);
  if (strlen(text) != len)
    warn("Sample %zu name has length %u but is only %zu chars long (premature '\\0') [path: %s]\n", colour, len, strlen(text), r->path);
  free(*(char **)dst);
  *(char **)dst = text;
  return n;
}

static size_t hdr_magic(ctx_reader *r, const char *what, bool at_end)
{
  char word[7] = {0};
  hdr_take(r, word, 6, what);
  if (memcmp(word, "CORTEX", 6) == 0) return 6;
  if (at_end) die("Magic word doesn't match '%s' (end): '%s' [path: %s]\n", "CORTEX", word, r->path);
  die("Magic word doesn't match '%s' (start): %s", "CORTEX", r->path);
}

static void hdr_check_dims(const ctx_reader *r)
{
  const uint32_t v = r->version, k = r->kmer_size, w = r->num_words;
  if (v < 4 || v > 7) die("Sorry, we only support graph file versions 4, 5, 6 & 7 [version: %u; path: %s]\n", v, r->path);
  if (!(k & 1)) die("kmer size is not an odd number [kmer_size: %u; path: %s]\n", k, r->path);
  if (k < 3) die("kmer size is less than three [kmer_size: %u; path: %s]\n", k, r->path);
  /* W*32 >= kmer_size > (W-1)*32 */
  if ((uint64_t)w * 32 < k) die("Not enough bitfields for kmer size [kmer_size: %u; bitfields: %u; path: %s]\n", k, w, r->path);
  if (((uint64_t)w - 1) * 32 >= k) die("using more than the minimum number of bitfields [path: %s]\n", r->path);
  if (r->num_cols == 0) die("number of colours is zero [path: %s]\n", r->path);
  if (r->num_cols > 10000) die("Very high number of colours: %zu [path: %s]", (size_t)r->num_cols, r->path);
}

/* "0 if not used": files up to version 6 may hold -1 there, and a threshold without its flag is dropped */
static void hdr_settle_threshold(const ctx_reader *r, uint8_t used, uint32_t *thresh, const char *of)
{
  if (used) return;
  if (r->version <= 6 && *thresh == UINT32_MAX) *thresh = 0;
  if (*thresh == 0) return;
  warn("Graph header gives cleaning threshold for %s when no cleaning was performed [path: %s]", of, r->path);
  *thresh = 0;
}

static size_t read_header(ctx_reader *r)
{
  size_t bytes = hdr_magic(r, "Magic word", false);
  for (size_t f = 0; f < NFIELDS(hdr_dims); f++) bytes += hdr_field_read(r, &hdr_dims[f], r, 0);
  hdr_check_dims(r);

  r->ginfo = calloc(r->num_cols, sizeof(col_info));
  if (!r->ginfo) die("Out of memory");
  for (uint32_t c = 0; c < r->num_cols; c++) col_info_init(&r->ginfo[c]);

  for (size_t f = 0; f < NFIELDS(hdr_colour_arrays); f++) {
    if (r->version < hdr_colour_arrays[f].since) continue;
    for (uint32_t c = 0; c < r->num_cols; c++) bytes += hdr_field_read(r, &hdr_colour_arrays[f], &r->ginfo[c], c);
  }
  for (uint32_t c = 0; c < r->num_cols; c++) {
    err_cleaning *ec = &r->ginfo[c].cleaning;
    for (size_t f = 0; f < NFIELDS(hdr_cleaning); f++)
      if (r->version >= hdr_cleaning[f].since) bytes += hdr_field_read(r, &hdr_cleaning[f], ec, c);
    hdr_settle_threshold(r, ec->cleaned_unitigs, &ec->clean_unitigs_thresh, "unitig");
    hdr_settle_threshold(r, ec->cleaned_kmers, &ec->clean_kmers_thresh, "nodes");
  }
  return bytes + hdr_magic(r, "magic word (end)", true);
}

void ctx_reader_open(ctx_reader *r, const char *input, size_t into_offset, size_t min_k, size_t max_k)
{
  ctx_reader_open_mode(r, input, "r", into_offset, min_k, max_k);
}

bool ctx_reader_from_direct(const ctx_reader *r)
{
  for (size_t i = 0; i < r->nfilter; i++)
    if (r->filter[i].from != i || r->filter[i].into != i) return false;
  return r->nfilter == r->num_cols;
}

size_t ctx_write_header_raw(FILE *fh, const ctx_reader *r)
{
  size_t n = 0;
  n += fwrite("CORTEX", 1, 6, fh);
  n += fwrite(&r->version, 1, 4, fh); n += fwrite(&r->kmer_size, 1, 4, fh);
  n += fwrite(&r->num_words, 1, 4, fh); n += fwrite(&r->num_cols, 1, 4, fh);
  for (uint32_t c = 0; c < r->num_cols; c++) n += fwrite(&r->ginfo[c].mean_read_length, 1, 4, fh);
  for (uint32_t c = 0; c < r->num_cols; c++) n += fwrite(&r->ginfo[c].total_sequence, 1, 8, fh);
  if (r->version >= 6) {
    for (uint32_t c = 0; c < r->num_cols; c++) {
      const uint32_t len = (uint32_t)strlen(r->ginfo[c].name);
      n += fwrite(&len, 1, 4, fh); n += fwrite(r->ginfo[c].name, 1, len, fh);
    }
    for (uint32_t c = 0; c < r->num_cols; c++) n += fwrite(&r->ginfo[c].seq_err, 1, 16, fh);
    for (uint32_t c = 0; c < r->num_cols; c++) n += put_cleaning(fh, &r->ginfo[c].cleaning);
  }
  n += fwrite("CORTEX", 1, 6, fh);
  return n;
}

void ctx_reader_open_mode(ctx_reader *r, const char *input, const char *mode, size_t into_offset, size_t min_k, size_t max_k)
{
  memset(r, 0, sizeof(*r));
  r->input = dupstr(input);
  const graph_arg ga = graph_arg_split(input);
  r->path = calloc(ga.path.n + 1, 1);
  memcpy(r->path, ga.path.s, ga.path.n);
  r->file_size = r->num_kmers = -1;
  if (strcmp(input, "-") != 0) {
    struct stat st;
    if (stat(r->path, &st) == 0) r->file_size = (long long)st.st_size;
    else warn("Couldn't get file size: %s", r->path);
  }
  if (!strcmp(r->path, "-")) {
    if (strcmp(mode, "r") != 0) die("Cannot open pipe with mode: %s", mode);
    r->fh = stdin;
  } else if (!(r->fh = fopen(r->path, mode))) die("Cannot open file: %s [%s]", r->path, strerror(errno));
  setvbuf(r->fh, NULL, _IOFBF, 1 << 20);
  r->hdr_size = read_header(r);
  reader_set_filter(r, into_offset);
  /* db_graph_check_kmer_size: db_graph.c:386-393 */
  if (r->kmer_size < min_k || r->kmer_size > max_k)
    die("Cannot handle kmer size %zu [%zu-%zu; %s]", (size_t)r->kmer_size, min_k, max_k, r->path);
  if (r->file_size != -1) {
    const size_t bytes_per_kmer = 8 * (size_t)r->num_words + 5 * (size_t)r->num_cols;
    const size_t remaining = (size_t)(r->file_size - (long long)r->hdr_size);
    r->num_kmers = (long long)(remaining / bytes_per_kmer);
    if (remaining % bytes_per_kmer != 0)
      warn("Truncated graph file: %s [bytes per kmer: %zu remaining: %zu; fsize: %zu; header: %zu; nkmers: %zu]", r->path,
           bytes_per_kmer, remaining, (size_t)r->file_size, r->hdr_size, (size_t)r->num_kmers);
  }
}

void ctx_reader_close(ctx_reader *r)
{
  if (r->fh && r->fh != stdin) fclose(r->fh);
  for (uint32_t i = 0; r->ginfo && i < r->num_cols; i++) col_info_free(&r->ginfo[i]);
  free(r->ginfo); free(r->filter); free(r->input); free(r->path);
  memset(r, 0, sizeof(*r));
}
