/* Test helper: opens a graph argument ("<into>:in.ctx:<from>") with the host program's reader
 * (mccortex_amd/host/ctx_file.c) and prints what it understood, one line per field, so that
 * tests/test_ctxio.py can compare it with the checker's parser without a GPU. */
#include "host.h"

#include <stdlib.h>

int main(int argc, char **argv)
{
  if (argc < 3) return 2;
  ctx_reader r;
  ctx_reader_open(&r, argv[1], (size_t)strtoul(argv[2], NULL, 10), 3, 63);
  printf("path %s\n", r.path);
  printf("dims %u %u %u %u %zu %lld\n", r.version, r.kmer_size, r.num_words, r.num_cols, r.hdr_size, r.num_kmers);
  printf("into_ncols %zu\n", r.into_ncols);
  for (size_t i = 0; i < r.nfilter; i++) printf("filter %u %u\n", r.filter[i].from, r.filter[i].into);
  for (uint32_t c = 0; c < r.num_cols; c++) {
    const col_info *g = &r.ginfo[c];
    printf("colour %u|%s|%u|%llu|%.12Lg|%u%u%u%u|%u|%u|%s\n", c, g->name, g->mean_read_length,
           (unsigned long long)g->total_sequence, g->seq_err, g->cleaning.cleaned_tips, g->cleaning.cleaned_unitigs,
           g->cleaning.cleaned_kmers, g->cleaning.is_graph_intersection, g->cleaning.clean_unitigs_thresh,
           g->cleaning.clean_kmers_thresh, g->cleaning.intersection_name);
  }
  ctx_reader_close(&r);
  return 0;
}
