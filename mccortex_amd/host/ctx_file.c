/* ctx_file.c -- .ctx header writer and reader, colour filters and the per-colour GraphInfo
 * arithmetic (docs/file_formats/graph_file_format.txt; src/graph/graph_writer.c:11-110,
 * src/graph/graph_file_reader.c:78-340, src/basic/file_filter.c, src/basic/range.c,
 * src/basic/graph_info.c).  x86-64 only: the header stores `long double seq_err` as its 16
 * in-memory bytes (10-byte x87 value + padding). */
#include "host.h"

#include <errno.h>
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

/* ---- colour ranges (src/basic/range.c) ---- */
static int range_parse(const char *str, size_t *start, size_t *end, size_t range_max)
{
  char *endp;
  if (*str == '*') { *start = 0; *end = range_max; return 1; }
  unsigned long from = strtoul(str, &endp, 10), to = from;
  if (endp == str) return -1;
  if (*endp == '-') {
    const char *s2 = endp + 1;
    to = strtoul(s2, &endp, 10);
    if (endp == s2) return -1;
  }
  if (from > range_max || to > range_max) return -1;
  *start = from; *end = to;
  return (int)(endp - str);
}

static int range_get_num(const char *str, size_t range_max)
{
  size_t start, end, num = 0;
  int bytes;
  for (const char *p = str; *p;) {
    if ((bytes = range_parse(p, &start, &end, range_max)) == -1) return -1;
    p += bytes;
    num += (start > end ? start - end : end - start) + 1;
    if (*p == ',') p++;
  }
  return num == 0 ? (int)(range_max + 1) : (int)num;
}

static int range_parse_array(const char *str, size_t *arr, size_t range_max)
{
  size_t num = 0, start, end;
  int bytes;
  const char *p = str;
  while (*p) {
    if ((bytes = range_parse(p, &start, &end, range_max)) == -1) return -1;
    p += bytes;
    if (*p == ',') p++;
    if (start <= end) for (size_t j = start; j <= end; j++) arr[num++] = j;
    /* descending range a-b = a, a-1, .. b.  (The reference's loop, range.c:71-72, has no lower
     * bound: it runs on to 0 and past the array range_get_num() sized.) */
    else for (size_t j = start; j <= start; j--) { arr[num++] = j; if (j == end) break; }
  }
  if (p > str && *(p - 1) == ',') return -1;
  if (num == 0) for (num = 0; num <= range_max; num++) arr[num] = num;
  return (int)num;
}

static int range_parse_array_fill(const char *str, size_t *arr, size_t range_max, size_t num_entries)
{
  const int r = range_parse_array(str, arr, range_max);
  if (r < 0) return -1;
  if (r == 0) for (size_t i = 0; i < num_entries; i++) arr[i] = i;
  else if (r == 1) for (size_t i = 1; i < num_entries; i++) arr[i] = arr[0];
  else if ((size_t)r != num_entries) return -1;
  return (int)num_entries;
}

/* ---- colour filter "<into>:path:<from>" (src/basic/file_filter.c:7-143) ---- */
static int is_range_char(char c) { return (c >= '0' && c <= '9') || c == '-' || c == ','; }

static void deconstruct_path(const char *path, const char **start, const char **end)
{
  const char *p = path;
  *start = path;
  while (is_range_char(*p)) p++;
  if (p > path && *p == ':') { p++; *start = p; }
  p = *end = path + strlen(path);
  while (p > (*start) + 1) {
    p--;
    if (*p == ':') { *end = p; break; }
    else if (!is_range_char(*p)) break;
  }
}

static int cmp_into(const void *a, const void *b)
{
  const col_filter *x = a, *y = b;
  if (x->into != y->into) return x->into < y->into ? -1 : 1;
  return x->from < y->from ? -1 : (x->from > y->from);
}

/* file_filter_set_cols */
static void filter_set_cols(ctx_reader *r, size_t srcncols, size_t into_offset)
{
  const char *ps, *pe;
  deconstruct_path(r->input, &ps, &pe);
  char *path_start = r->input + (ps - r->input), *path_end = r->input + (pe - r->input);
  char *from_fltr = (*path_end == ':' ? path_end + 1 : NULL);
  char *into_fltr = (path_start > r->input ? r->input : NULL);
  size_t ncols;
  if (from_fltr) {
    int s = range_get_num(from_fltr, srcncols - 1);
    if (s < 0) die("Invalid filter path: %s (from size: %zu)", r->input, srcncols);
    ncols = (size_t)s;
  } else ncols = srcncols;
  if (into_fltr) {
    *(path_start - 1) = '\0';
    int s = range_get_num(into_fltr, SIZE_MAX - 1);
    *(path_start - 1) = ':';
    if (s < 0 || (s != 1 && (size_t)s != ncols)) die("Invalid filter path: %s (s:%i ncols:%zu)", r->input, s, ncols);
  }
  r->filter = calloc(ncols ? ncols : 1, sizeof(col_filter));
  r->nfilter = ncols;
  /* a range may name up to range_max+1 entries; ranges are bounded by the checks above */
  size_t *tmp = calloc((ncols > srcncols ? ncols : srcncols) + 1, sizeof(size_t));
  if (from_fltr) {
    if (range_parse_array(from_fltr, tmp, srcncols - 1) == -1) die("Invalid filter path: %s", r->input);
    for (size_t i = 0; i < ncols; i++) r->filter[i].from = (uint32_t)tmp[i];
  } else for (size_t i = 0; i < ncols; i++) r->filter[i].from = (uint32_t)i;
  if (into_fltr) {
    *(path_start - 1) = '\0';
    int s = range_parse_array_fill(into_fltr, tmp, SIZE_MAX - 1, ncols);
    *(path_start - 1) = ':';
    if (s < 0 || (size_t)s != ncols) die("Invalid filter path: %s (s:%i ncols:%zu)", r->input, s, ncols);
    for (size_t i = 0; i < ncols; i++) r->filter[i].into = (uint32_t)tmp[i];
  } else for (size_t i = 0; i < ncols; i++) r->filter[i].into = (uint32_t)(into_offset + i);
  free(tmp);
  qsort(r->filter, r->nfilter, sizeof(col_filter), cmp_into);
  r->into_ncols = 0;
  for (size_t i = 0; i < r->nfilter; i++)
    if ((size_t)r->filter[i].into + 1 > r->into_ncols) r->into_ncols = (size_t)r->filter[i].into + 1;
}

/* ---- header reader (graph_file_read_header: graph_file_reader.c:78-260) ---- */
static void gfread(ctx_reader *r, void *ptr, size_t n, const char *entry)
{
  const size_t got = fread(ptr, 1, n, r->fh);
  if (got != n) die("Unexpected end of file: %s [%s; read %zu of %zu bytes]", r->path, entry, got, n);
}

static char *read_name(ctx_reader *r, const char *what, size_t colour, size_t *bytes_read)
{
  uint32_t len;
  gfread(r, &len, 4, what);
  if (len > 10000) die("Very big sample name. Length: %u", len);
  char *s = calloc((size_t)len + 1, 1);
  gfread(r, s, len, what);
  *bytes_read += 4 + len;
  if (strlen(s) != len)
    warn("Sample %zu name has length %u but is only %zu chars long (premature '\\0') [path: %s]\n", colour, len, strlen(s), r->path);
  return s;
}

static size_t read_header(ctx_reader *r)
{
  size_t bytes = 0;
  char magic[7] = {0};
  gfread(r, magic, 6, "Magic word");
  if (strcmp(magic, "CORTEX") != 0) die("Magic word doesn't match '%s' (start): %s", "CORTEX", r->path);
  bytes += 6;
  gfread(r, &r->version, 4, "graph version");
  gfread(r, &r->kmer_size, 4, "kmer size");
  gfread(r, &r->num_words, 4, "num of bitfields");
  gfread(r, &r->num_cols, 4, "number of colours");
  bytes += 16;
  if (r->version > 7 || r->version < 4)
    die("Sorry, we only support graph file versions 4, 5, 6 & 7 [version: %u; path: %s]\n", r->version, r->path);
  if (r->kmer_size % 2 == 0) die("kmer size is not an odd number [kmer_size: %u; path: %s]\n", r->kmer_size, r->path);
  if (r->kmer_size < 3) die("kmer size is less than three [kmer_size: %u; path: %s]\n", r->kmer_size, r->path);
  if (r->num_words * 32 < r->kmer_size)
    die("Not enough bitfields for kmer size [kmer_size: %u; bitfields: %u; path: %s]\n", r->kmer_size, r->num_words, r->path);
  if ((r->num_words - 1) * 32 >= r->kmer_size) die("using more than the minimum number of bitfields [path: %s]\n", r->path);
  if (r->num_cols == 0) die("number of colours is zero [path: %s]\n", r->path);
  if (r->num_cols > 10000) die("Very high number of colours: %zu [path: %s]", (size_t)r->num_cols, r->path);

  r->ginfo = calloc(r->num_cols, sizeof(col_info));
  for (uint32_t i = 0; i < r->num_cols; i++) col_info_init(&r->ginfo[i]);
  for (uint32_t i = 0; i < r->num_cols; i++) gfread(r, &r->ginfo[i].mean_read_length, 4, "mean read length for each colour");
  for (uint32_t i = 0; i < r->num_cols; i++) gfread(r, &r->ginfo[i].total_sequence, 8, "total sequance loaded for each colour");
  bytes += (size_t)r->num_cols * 12;
  if (r->version >= 6) {
    for (uint32_t i = 0; i < r->num_cols; i++) {
      free(r->ginfo[i].name);
      r->ginfo[i].name = read_name(r, "sample name", i, &bytes);
    }
    for (uint32_t i = 0; i < r->num_cols; i++) {
      unsigned char b[16];
      gfread(r, b, 16, "seq error rates");
      memcpy(&r->ginfo[i].seq_err, b, 16);
    }
    bytes += 16 * (size_t)r->num_cols;
    for (uint32_t i = 0; i < r->num_cols; i++) {
      err_cleaning *ec = &r->ginfo[i].cleaning;
      gfread(r, &ec->cleaned_tips, 1, "tip cleaning");
      gfread(r, &ec->cleaned_unitigs, 1, "remove low covg unitig");
      gfread(r, &ec->cleaned_kmers, 1, "remove low covg kmers");
      gfread(r, &ec->is_graph_intersection, 1, "cleaned against graph");
      uint32_t tu = 0, tk = 0;
      gfread(r, &tu, 4, "remove low covg unitig threshold");
      gfread(r, &tk, 4, "remove low covg kmer threshold");
      bytes += 12;
      if (r->version <= 6) { /* old versions wrote -1 for "no threshold" */
        if (!ec->cleaned_unitigs && tu == (uint32_t)-1) tu = 0;
        if (!ec->cleaned_kmers && tk == (uint32_t)-1) tk = 0;
      }
      if (!ec->cleaned_unitigs && tu > 0) {
        warn("Graph header gives cleaning threshold for unitig when no cleaning was performed [path: %s]", r->path);
        tu = 0;
      }
      if (!ec->cleaned_kmers && tk > 0) {
        warn("Graph header gives cleaning threshold for nodes when no cleaning was performed [path: %s]", r->path);
        tk = 0;
      }
      ec->clean_unitigs_thresh = tu; ec->clean_kmers_thresh = tk;
      free(ec->intersection_name);
      ec->intersection_name = read_name(r, "cleaned against graph name", i, &bytes);
    }
  }
  gfread(r, magic, 6, "magic word (end)");
  if (strcmp(magic, "CORTEX") != 0) die("Magic word doesn't match '%s' (end): '%s' [path: %s]\n", "CORTEX", magic, r->path);
  bytes += 6;
  return bytes;
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
  const char *ps, *pe;
  deconstruct_path(input, &ps, &pe);
  r->path = calloc((size_t)(pe - ps) + 1, 1);
  memcpy(r->path, ps, (size_t)(pe - ps));
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
  filter_set_cols(r, r->num_cols, into_offset);
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
