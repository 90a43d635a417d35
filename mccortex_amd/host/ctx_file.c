/* ctx_file.c -- .ctx v6 header (docs/file_formats/graph_file_format.txt; src/graph/graph_writer.c:11-110)
 * and the per-colour GraphInfo arithmetic (src/basic/graph_info.c:60-175).  x86-64 only: the
 * header stores `long double seq_err` as its 16 in-memory bytes (10-byte x87 value + padding). */
#include "host.h"

#include <string.h>

void col_info_init(col_info *c)
{ /* graph_info_init: graph_info.c:60-67 */
  c->mean_read_length = 0;
  c->total_sequence = 0;
  strcpy(c->name, "undefined");
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

static size_t put(FILE *fh, const void *p, size_t n) { return fwrite(p, 1, n, fh); }

size_t ctx_write_header(FILE *fh, uint32_t kmer_size, uint32_t ncols, const col_info *cols)
{
  _Static_assert(sizeof(long double) == 16, "x87 long double layout expected");
  size_t n = 0;
  const uint32_t version = 6, W = (2 * kmer_size + 63) / 64;
  n += put(fh, "CORTEX", 6);
  n += put(fh, &version, 4); n += put(fh, &kmer_size, 4); n += put(fh, &W, 4); n += put(fh, &ncols, 4);

  /* graph_writer_mkhdr merges every colour into a fresh GraphInfo (graph_info_merge,
   * graph_info.c:135-170): mean is re-derived from total/contigs and may differ from the input */
  uint32_t mean[ncols]; long double err[ncols];
  for (uint32_t c = 0; c < ncols; c++) {
    uint32_t dmean = 0; uint64_t dtotal = 0; long double derr = 0.01;
    const long double serr = 0.01;
    const uint64_t stotal = cols[c].total_sequence; const uint32_t smean = cols[c].mean_read_length;
    const uint64_t tot = dtotal + stotal;
    if (tot > 0) {
      derr = (derr * dtotal + serr * stotal) / tot;
      size_t src_contigs = 0;
      if (stotal && smean) src_contigs = (size_t)(((double)stotal / smean) + 0.5);
      update_contigs(&dmean, &dtotal, stotal, src_contigs);
    }
    mean[c] = dmean; err[c] = derr;
  }
  for (uint32_t c = 0; c < ncols; c++) n += put(fh, &mean[c], 4);
  for (uint32_t c = 0; c < ncols; c++) n += put(fh, &cols[c].total_sequence, 8);
  for (uint32_t c = 0; c < ncols; c++) {
    uint32_t len = (uint32_t)strlen(cols[c].name);
    n += put(fh, &len, 4); n += put(fh, cols[c].name, len);
  }
  for (uint32_t c = 0; c < ncols; c++) {
    unsigned char b[16] = {0};
    memcpy(b, &err[c], 10);
    n += put(fh, b, 16);
  }
  for (uint32_t c = 0; c < ncols; c++) { /* ErrorCleaning: 4 flags, 2 thresholds, intersection name "undefined" */
    unsigned char z[12] = {0};
    const uint32_t len = 9;
    n += put(fh, z, 12); n += put(fh, &len, 4); n += put(fh, "undefined", 9);
  }
  n += put(fh, "CORTEX", 6);
  return n;
}
