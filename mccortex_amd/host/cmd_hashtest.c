/* cmd_hashtest.c -- `mccortex<K> hashtest` (src/commands/ctx_exp_hashtest.c): the reference's own benchmark of
 * this path's table.  Same options, same messages; find-or-insert of the integer keys runs on the MI355X
 * (mcx_graph_hashtest), the -F mode (hash function only) likewise (mcx_hashtest_func).  `-t` keeps its meaning for
 * -F (the reference adds up one XOR per thread range, so the printed hash depends on it) and is otherwise only
 * reported: the device does not run host threads. */
#define _GNU_SOURCE
#include "host.h"

#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "../../include/mcx_gpu.h"

#define DEFAULT_MEM (1UL << 30)   /* cmd.h:12: -m default 1GB */
#define DEFAULT_NTHREADS 2        /* cmd.h:9 */

static const char hashtest_usage[] =
"usage: " CMD_NAME " hashtest [options] <num_ops>\n"
"\n"
"  Test hash table speed. If threads is set to 0, use single-threaded code.\n"
"\n"
"  -h, --help        This help message\n"
"  -m, --memory <M>  Memory to use\n"
"  -n, --nkmers <N>  Number of hash table entries (e.g. 1G ~ 1 billion)\n"
"  -t, --threads <T> Number of threads to use [default: " MCX_STR(DEFAULT_NTHREADS) "]\n"
"  -k, --kmer <K>    Kmer size must be odd (" MCX_STR(MAX_KMER_SIZE) " >= k >= " MCX_STR(MIN_KMER_SIZE) ")\n"
"  -F, --func-only   Only use the hash function, do not store kmers\n"
"\n";

static struct option hashtest_opts[] = {
  {"help", no_argument, NULL, 'h'},          {"memory", required_argument, NULL, 'm'},
  {"nkmers", required_argument, NULL, 'n'},  {"threads", required_argument, NULL, 't'},
  {"kmer", required_argument, NULL, 'k'},    {"func-only", no_argument, NULL, 'F'},
  {"device", required_argument, NULL, 'D'},  {NULL, 0, NULL, 0}};

static void optname(char c, char *out)
{
  sprintf(out, "-%c, --Unknown", c);
  for (int i = 0; hashtest_opts[i].name; i++)
    if (hashtest_opts[i].val == c)
      sprintf(out, "-%c, --%s%s", c, hashtest_opts[i].name, hashtest_opts[i].has_arg == required_argument ? " <arg>" : "");
}

/* hash_table_print_stats_brief (hash_table.c:301-319) for the table in HBM: its slots and the bytes they take */
static void print_table_stats(mcx_graph *g)
{
  uint64_t slots = 0, bytes = 0, nk = 0;
  char s1[64], s2[64], s3[64];
  mcx_graph_capacity(g, &slots, &bytes);
  int rc = mcx_graph_nkmers(g, &nk);
  if (rc == MCX_ERR_FULL) die("Hash table is full");
  if (rc != MCX_OK) die("nkmers: %s", mcx_last_error());
  status("[hasht] memory: %s; filled: %s / %s (%.2f%%)\n", bytes_to_str(bytes, 1, s1), ulong_to_str(nk, s2),
         ulong_to_str(slots, s3), slots ? 100.0 * (double)nk / (double)slots : 0.0);
}

int ctx_hashtest(int argc, char **argv)
{
  size_t nthreads = 0, kmer_size = 0, mem_to_use = DEFAULT_MEM, num_kmers = 0, num_ops = 0;
  bool threads_set = false, mem_set = false, nkmers_set = false, store_kmers = true;
  unsigned device = 0, u;
  char cmd[100];
  int c;
  if (argc == 1) print_usage(hashtest_usage, NULL); /* "Type a command with no arguments to see help" (mccortex.c) */
  optind = 1;
  while ((c = getopt_long_only(argc, argv, "hm:n:t:k:FD:", hashtest_opts, NULL)) != -1) {
    optname((char)c, cmd);
    switch (c) {
      case 'h': print_usage(hashtest_usage, NULL);
      case 't':
        if (threads_set) print_usage(hashtest_usage, "%s given twice", cmd);
        if (!parse_entire_uint(optarg, &u)) print_usage(hashtest_usage, "%s requires an int x >= 0: %s", cmd, optarg);
        nthreads = u; threads_set = true; break;
      case 'm':
        if (mem_set) print_usage(hashtest_usage, "-m, --memory <M> specifed more than once");
        if (!mem_to_integer(optarg, &mem_to_use) || !mem_to_use) print_usage(hashtest_usage, "Invalid memory argument: %s", optarg);
        mem_set = true; break;
      case 'n':
        if (nkmers_set) print_usage(hashtest_usage, "-n, --nkmers <N> specifed more than once");
        if (!mem_to_integer(optarg, &num_kmers) || !num_kmers) print_usage(hashtest_usage, "Invalid hash size: %s", optarg);
        nkmers_set = true; break;
      case 'k':
        if (kmer_size) print_usage(hashtest_usage, "%s given twice", cmd);
        if (!parse_entire_uint(optarg, &u) || !u) print_usage(hashtest_usage, "%s requires an int x > 0: %s", cmd, optarg);
        kmer_size = u; break;
      case 'F': if (!store_kmers) print_usage(hashtest_usage, "%s given twice", cmd); store_kmers = false; break;
      case 'D': if (!parse_entire_uint(optarg, &device)) print_usage(hashtest_usage, "%s requires an int x >= 0: %s", cmd, optarg); break;
      case ':': case '?': die("`" CMD_NAME " hashtest -h` for help. Bad option: %s", argv[optind - 1]);
      default: abort();
    }
  }
  bool single_threaded = false;
  if (nthreads == 0) { single_threaded = true; nthreads = 1; }

  if (!kmer_size) die("kmer size not set with -k <K>");
  if (kmer_size < MIN_KMER_SIZE || kmer_size > MAX_KMER_SIZE) die("Please recompile with correct kmer size (%zu)", kmer_size);
  if (!(kmer_size & 1))
    die("Invalid kmer-size (%zu): requires odd number %i <= k <= %i", kmer_size, MIN_KMER_SIZE, MAX_KMER_SIZE);
  if (optind + 1 != argc) print_usage(hashtest_usage, NULL);
  if (!parse_entire_size(argv[optind], &num_ops)) print_usage(hashtest_usage, "Invalid <num_ops>");

  if (mcx_device_count() < 1) die("No MI355X / HIP device found: %s has no CPU build path", CMD_NAME);
  mcx_graph *g = NULL;
  char s1[64], s2[64];
  if (store_kmers) {
    /* every operation adds a (probably unique) kmer: min and max number of kmers are both num_ops */
    const size_t W = (2 * kmer_size + 63) / 64, bits_per_kmer = W * 64;
    table_plan plan;
    char ebuf[256];
    const char *err = table_plan_for_build(mem_to_use, mem_set, num_kmers, nkmers_set, bits_per_kmer, (int64_t)num_ops,
                                           &plan, ebuf, sizeof(ebuf));
    if (err) die("%s", err);
    status("[memory] graph: %s", bytes_to_str(plan.bytes, 1, s1));
    uint64_t hbm_free = 0, hbm_total = 0;
    if (mcx_device_memory((int)device, &hbm_free, &hbm_total) != MCX_OK) die("device query: %s", mcx_last_error());
    const uint64_t dev_bytes = (plan.capacity + plan.capacity / 32) * 8 * (W + 1);
    if (dev_bytes > hbm_free)
      die("Requesting more memory than is available [ Reqeusted: %s HBM free: %s ]", bytes_to_str(dev_bytes, 1, s1), bytes_to_str(hbm_free, 1, s2));
    if (mcx_graph_create(&g, (int)kmer_size, 1, plan.capacity, (int)device) != MCX_OK) die("Cannot allocate graph: %s", mcx_last_error());
    print_table_stats(g);
  }

  status("[threads] using %zu thread%s (%s-threaded code)", nthreads, nthreads == 1 ? "" : "s", single_threaded ? "single" : "multi");

  struct timeval t0, t1;
  gettimeofday(&t0, NULL);
  uint64_t hash = 0;
  if (store_kmers) {
    const int rc = mcx_graph_hashtest(g, 0, num_ops);
    if (rc == MCX_ERR_FULL) die("Hash table is full");
    if (rc != MCX_OK) die("hashtest: %s", mcx_last_error());
  } else if (mcx_hashtest_func((int)device, (int)kmer_size, num_ops, (uint32_t)nthreads, &hash) != MCX_OK) {
    die("hashtest: %s", mcx_last_error());
  }
  gettimeofday(&t1, NULL);
  const double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-6 * (double)(t1.tv_usec - t0.tv_usec);
  status("[device] %s operations in %.3f seconds: %.1f M per second", ulong_to_str(num_ops, s1), secs,
         secs > 0 ? 1e-6 * (double)num_ops / secs : 0.0);

  if (store_kmers) {
    print_table_stats(g);
    mcx_graph_destroy(g);
  }
  status("Output hash: %zu", (size_t)hash);
  return EXIT_SUCCESS;
}
