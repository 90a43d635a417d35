/* cmd_sort.c -- `mccortex<K> sort` and `mccortex<K> index` (src/commands/ctx_sort.c,
 * src/commands/ctx_index.c): same options, messages and output; the sort itself and the
 * sortedness check run on the MI355X (mcx_sort_records / mcx_records_sorted). */
#define _GNU_SOURCE
#include "host.h"

#include <errno.h>
#include <fcntl.h>
#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/mcx_gpu.h"

#define DEFAULT_MEM (1UL << 29) /* cmd.h:13 */

static const char sort_usage[] =
"usage: " CMD_NAME " sort [options] <in.ctx>\n"
"\n"
"  Sort a cortex graph file. Loads entire graph into memory then sorts.\n"
"\n"
"  -h, --help              This help message\n"
"  -q, --quiet             Silence status output normally printed to STDERR\n"
"  -f, --force             Overwrite output files\n"
"  -m, --memory <mem>      Memory to use\n"
"  -n, --nkmers <kmers>    Number of hash table entries (e.g. 1G ~ 1 billion)\n"
"  -o, --out <out.ctx>     Output file [default: overwrite input]\n"
"\n";

static struct option sort_opts[] = {
  {"help", no_argument, NULL, 'h'},          {"force", no_argument, NULL, 'f'},
  {"memory", required_argument, NULL, 'm'},  {"nkmers", required_argument, NULL, 'n'},
  {"out", required_argument, NULL, 'o'},     {"device", required_argument, NULL, 'D'},
  {NULL, 0, NULL, 0}};

static void optname(const struct option *opts, char c, char *out)
{
  sprintf(out, "-%c, --Unknown", c);
  for (int i = 0; opts[i].name; i++)
    if (opts[i].val == c) sprintf(out, "-%c, --%s", c, opts[i].name);
}

/* futil_create_output (file_util.c:139-174) */
static void create_output(const char *path, bool force)
{
  if (!strcmp(path, "-")) return;
  int fd = open(path, O_CREAT | (force ? 0 : O_EXCL) | O_WRONLY | O_APPEND, 0666);
  if (fd < 0) {
    if (errno == EEXIST) die("File already exists: %s", path);
    die("Cannot write to file: %s [%s]", path, strerror(errno));
  }
  close(fd);
}

int ctx_sort(int argc, char **argv)
{
  const char *out_path = NULL;
  size_t mem_to_use = DEFAULT_MEM, num_kmers_arg = 0;
  bool mem_set = false, nkmers_set = false, force = false;
  unsigned device = 0;
  char cmd[100];
  int c;
  optind = 1;
  while ((c = getopt_long_only(argc, argv, "hfm:n:o:D:", sort_opts, NULL)) != -1) {
    optname(sort_opts, (char)c, cmd);
    switch (c) {
      case 'h': print_usage(sort_usage, NULL);
      case 'f': if (force) print_usage(sort_usage, "%s given twice", cmd); force = true; break;
      case 'm':
        if (mem_set) print_usage(sort_usage, "-m, --memory <M> specifed more than once");
        if (!mem_to_integer(optarg, &mem_to_use) || !mem_to_use) print_usage(sort_usage, "Invalid memory argument: %s", optarg);
        mem_set = true; break;
      case 'n':
        if (nkmers_set) print_usage(sort_usage, "-n, --nkmers <N> specifed more than once");
        if (!mem_to_integer(optarg, &num_kmers_arg) || !num_kmers_arg) print_usage(sort_usage, "Invalid hash size: %s", optarg);
        nkmers_set = true; break;
      case 'o': if (out_path) print_usage(sort_usage, "%s given twice", cmd); out_path = optarg; break;
      case 'D': if (!parse_entire_uint(optarg, &device)) print_usage(sort_usage, "%s requires an int x >= 0: %s", cmd, optarg); break;
      case ':': case '?': die("`" CMD_NAME " sort -h` for help. Bad option: %s", argv[optind - 1]);
      default: die("Bad option: [%c]: %s", c, cmd);
    }
  }
  if (optind + 1 != argc) print_usage(sort_usage, "Require exactly one input graph file (.ctx)");
  const char *ctx_path = argv[optind];

  ctx_reader r;
  ctx_reader_open_mode(&r, ctx_path, out_path ? "r" : "r+", 0, MIN_KMER_SIZE, MAX_KMER_SIZE);
  if (!ctx_reader_from_direct(&r)) die("Cannot open graph file with a filter ('in.ctx:blah' syntax)");

  size_t num_kmers;
  if (r.num_kmers < 0) {
    if (!nkmers_set) die("If reading from a stream, must give -n <num_kmers>");
    num_kmers = num_kmers_arg;
  } else num_kmers = (size_t)r.num_kmers;

  FILE *fout = NULL;
  if (out_path) {
    create_output(out_path, force);
    fout = !strcmp(out_path, "-") ? stdout : fopen(out_path, "w");
    if (!fout) die("Cannot open file: %s [%s]", out_path, strerror(errno));
  }

  const size_t ncols = r.num_cols;
  const size_t kmer_mem = 8 * (size_t)r.num_words + 5 * ncols;
  const size_t memory = (sizeof(char *) + kmer_mem) * num_kmers;
  char mem_str[64];
  bytes_to_str(memory, 1, mem_str);
  if (memory > mem_to_use) die("Require at least %s memory", mem_str);
  status("[memory] Total: %s", mem_str);

  unsigned char *mem = malloc(kmer_mem * num_kmers + 1);
  if (!mem) die("Out of memory");
  const size_t nkread = fread(mem, 1, num_kmers * kmer_mem, r.fh);
  if (nkread != num_kmers * kmer_mem) die("Could only read %zu bytes [<%zu]", nkread, num_kmers * kmer_mem);
  char tmpc;
  if (fread(&tmpc, 1, 1, r.fh) != 0) die("More kmers in file than believed (kmers: %zu ncols: %zu).", num_kmers, ncols);
  status("Read %zu kmers with %zu colour%s", num_kmers, ncols, ncols == 1 ? "" : "s");

  if (mcx_device_count() < 1) die("No MI355X / HIP device found: %s has no CPU build path", CMD_NAME);
  int rc = mcx_sort_records(mem, num_kmers, (int)r.kmer_size, (int)ncols, (int)device);
  if (rc != MCX_OK) die("sort: %s", mcx_last_error());

  if (out_path) ctx_write_header_raw(fout, &r);
  else {
    if (fseek(r.fh, (long)r.hdr_size, SEEK_SET) != 0) die("fseek failed");
    fout = r.fh;
  }
  if (fwrite(mem, 1, num_kmers * kmer_mem, fout) != num_kmers * kmer_mem) die("Cannot write to file");
  if (out_path) { if (fout != stdout) fclose(fout); else fflush(fout); }
  ctx_reader_close(&r);
  free(mem);
  return EXIT_SUCCESS;
}

static const char index_usage[] =
"usage: " CMD_NAME " index [options] <in.ctx>\n"
"\n"
"  Index a sorted cortex graph file (sort with `" CMD_NAME " sort` first).\n"
"\n"
"  -h, --help               This help message\n"
"  -q, --quiet              Silence status output normally printed to STDERR\n"
"  -f, --force              Overwrite output files\n"
"  -o, --out <out.ctx.idx>  Output file [default: STDOUT]\n"
"  -s, --block-size <S>     Block of <S> bytes [default: 4MB]\n"
"  -b, --block-kmers <B>    Block of <B> kmers\n"
"\n";

static struct option index_opts[] = {
  {"help", no_argument, NULL, 'h'},               {"force", no_argument, NULL, 'f'},
  {"out", required_argument, NULL, 'o'},          {"block-size", required_argument, NULL, 's'},
  {"block-kmers", required_argument, NULL, 'b'},  {"device", required_argument, NULL, 'D'},
  {NULL, 0, NULL, 0}};

/* ctx_index.c:36-177.  The table is produced on the host exactly as the reference does (including
 * its block arithmetic: bl_kmers = 1 + bl_bytes / kmer_mem counts the first k-mer twice); the
 * "File is not sorted" check compares whole blocks on the device first, so an unsorted file is
 * rejected even when only k-mers inside a block are out of order -- stricter than the reference,
 * which only compares the first k-mers of consecutive blocks. */
int ctx_index(int argc, char **argv)
{
  const char *out_path = NULL;
  size_t block_size = 0, block_kmers = 0;
  bool force = false;
  unsigned device = 0;
  char cmd[100];
  int c;
  optind = 1;
  while ((c = getopt_long_only(argc, argv, "hfo:s:b:D:", index_opts, NULL)) != -1) {
    optname(index_opts, (char)c, cmd);
    switch (c) {
      case 'h': print_usage(index_usage, NULL);
      case 'f': force = true; break;
      case 'o': if (out_path) print_usage(index_usage, "%s given twice", cmd); out_path = optarg; break;
      case 'b':
        if (block_kmers) print_usage(index_usage, "%s given twice", cmd);
        if (!parse_entire_size(optarg, &block_kmers)) print_usage(index_usage, "%s requires an int x >= 0: %s", cmd, optarg);
        if (!block_kmers) print_usage(index_usage, "%s <N> must be > 0: %s", cmd, optarg);
        break;
      case 's':
        if (block_size) print_usage(index_usage, "%s given twice", cmd);
        if (!parse_entire_size(optarg, &block_size)) print_usage(index_usage, "%s requires an int x >= 0: %s", cmd, optarg);
        if (!block_size) print_usage(index_usage, "%s <N> must be > 0: %s", cmd, optarg);
        break;
      case 'D': if (!parse_entire_uint(optarg, &device)) print_usage(index_usage, "%s requires an int x >= 0: %s", cmd, optarg); break;
      case ':': case '?': die("`" CMD_NAME " index -h` for help. Bad option: %s", argv[optind - 1]);
      default: abort();
    }
  }
  if (optind + 1 != argc) print_usage(index_usage, "Require exactly one input graph file (.ctx)");
  if (block_size && block_kmers) print_usage(index_usage, "Cannot use --block-kmers and --block-size together");
  const char *ctx_path = argv[optind];

  ctx_reader r;
  ctx_reader_open_mode(&r, ctx_path, "r+", 0, MIN_KMER_SIZE, MAX_KMER_SIZE);
  if (!ctx_reader_from_direct(&r)) die("Cannot open graph file with a filter ('in.ctx:blah' syntax)");

  FILE *fout = stdout;
  if (out_path) {
    create_output(out_path, force);
    fout = fopen(out_path, "w");
    if (!fout) die("Cannot open file: %s [%s]", out_path, strerror(errno));
  }

  const size_t kmer_size = r.kmer_size;
  const size_t kmer_mem = 8 * (size_t)r.num_words + 5 * (size_t)r.num_cols;
  if (block_size) block_kmers = block_size / kmer_mem;
  else if (!block_size && !block_kmers) { block_size = 4u << 20; block_kmers = block_size / kmer_mem; }
  block_size = block_kmers * kmer_mem;
  status("[index] block bytes: %zu kmers: %zu; kmer bytes: %zu, hdr: %zu", block_size, block_kmers, kmer_mem, r.hdr_size);
  if (block_kmers == 0) die("Cannot set block_kmers to zero");

  fputs("#block_start\tnext_block\tfirst_kmer\tkmer_idx\tnext_kmer_idx\n", fout);
  unsigned char *blk = malloc(block_size);
  unsigned char prev[8 * ((2 * MAX_KMER_SIZE + 63) / 64)] = {0}; /* the previous block's first k-mer */
  char kstr[2 * MAX_KMER_SIZE + 8];
  if (!blk) die("Out of memory");
  size_t nblocks = 0, bl_bytes = 0, bl_kmers = 0, bl_byte_offset = r.hdr_size, bl_kmer_offset = 0;
  const bool have_gpu = mcx_device_count() > 0;
  for (;;) {
    /* graph_file_read: one whole record or end of file */
    const size_t got1 = fread(blk, 1, kmer_mem, r.fh);
    if (got1 == 0) { status("Read kmer failed"); break; }
    if (got1 != kmer_mem) die("Unexpected end of file: %s", r.path);
    kmer_words_to_str(blk, (unsigned)kmer_size, kstr);
    bool ge = false; /* binary_kmer_ge(prev, bkmer): prev >= this block's first k-mer */
    if (nblocks > 0) {
      int cmp = 0;
      for (size_t w = 0; w < r.num_words && !cmp; w++) {
        uint64_t a, b;
        memcpy(&a, prev + 8 * w, 8); memcpy(&b, blk + 8 * w, 8);
        cmp = a < b ? -1 : (a > b);
      }
      ge = cmp >= 0;
    }
    if (ge) die("File is not sorted: %s [%s]", kstr, r.path);
    memcpy(prev, blk, 8 * (size_t)r.num_words);
    bl_bytes = kmer_mem + fread(blk + kmer_mem, 1, block_size - kmer_mem, r.fh);
    if (have_gpu && bl_bytes >= 2 * kmer_mem) {
      int64_t bad = -1;
      if (mcx_records_sorted(blk, bl_bytes / kmer_mem, (int)kmer_size, (int)r.num_cols, (int)device, &bad) != MCX_OK)
        die("index: %s", mcx_last_error());
      if (bad >= 0) {
        kmer_words_to_str(blk + (size_t)bad * kmer_mem, (unsigned)kmer_size, kstr);
        die("File is not sorted: %s [%s]", kstr, r.path);
      }
    }
    bl_kmers = 1 + bl_bytes / kmer_mem;
    fprintf(fout, "%zu\t%zu\t%s\t%zu\t%zu\n", bl_byte_offset, bl_byte_offset + bl_bytes, kstr, bl_kmer_offset, bl_kmer_offset + bl_kmers);
    bl_byte_offset += bl_bytes;
    bl_kmer_offset += bl_kmers;
    nblocks++;
    if (bl_kmers < block_kmers) { status("last block %zu < %zu; %zu vs %zu", bl_kmers, block_kmers, bl_bytes, block_size); break; }
  }
  free(blk);
  char a[64], b[64], c2[64], d[64];
  status("Read %s kmers in %s block%s (block size %s / %s kmers)", ulong_to_str(bl_kmer_offset, a), ulong_to_str(nblocks, b),
         nblocks == 1 ? "" : "s", bytes_to_str(block_size, 1, c2), ulong_to_str(block_kmers, d));
  if (fout != stdout) status("Saved to %s", out_path);
  ctx_reader_close(&r);
  if (fout != stdout) fclose(fout); else fflush(fout);
  return EXIT_SUCCESS;
}
