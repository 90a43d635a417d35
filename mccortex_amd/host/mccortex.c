/* mccortex.c -- `mccortex<K> <command>` dispatcher (src/main/mccortex.c:15-28,279-332).
 * `build` (the hot path this repository replaces) and the two commands next to it that run on the
 * same device code: `sort` and `index`, and the reference's benchmark of this path's table, `hashtest`. */
#include "host.h"

#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

static const char usage[] =
"usage: " CMD_NAME " <command> [options] <args>\n"
"version: mccortex_amd (MI355X build backend) k=" MCX_STR(MIN_KMER_SIZE) ".." MCX_STR(MAX_KMER_SIZE) "\n"
"\n"
"Commands:   build       construct cortex graph from FASTA/FASTQ\n"
"            sort        sort the kmers in a graph file\n"
"            index       index a sorted cortex graph file\n"
"            hashtest    test hash table speed\n"
"\n"
"  Type a command with no arguments to see help.\n"
"\n"
"Common Options:\n"
"  -h, --help            Help message\n"
"  -q, --quiet           Silence status output normally printed to STDERR\n"
"  -f, --force           Overwrite output files if they already exist\n"
"  -m, --memory <M>      Memory e.g. 1GB [default: 1GB]\n"
"  -n, --nkmers <H>      Hash entries [default: 4M, ~4 million]\n"
"  -t, --threads <T>     Limit on proccessing threads [default: 2]\n"
"\n";

int main(int argc, char **argv)
{
  struct timeval t0, t1;
  gettimeofday(&t0, NULL);
  const int timing = getenv("MCX_TIMING") != NULL; /* wall-clock stamps of main()'s entry and exit, for end-to-end breakdowns */
  if (timing) fprintf(stderr, "[timing] epoch_main_entry %ld.%06ld\n", (long)t0.tv_sec, (long)t0.tv_usec);
  msg_out = stderr;
  host_set_cmdline(argc, argv);
  if (argc == 1) { fputs(usage, stderr); return EXIT_FAILURE; }
  /* -q anywhere silences status output (mccortex.c:255-277) */
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--")) break;
    if (!strcmp(argv[i], "-q") || !strcmp(argv[i], "--quiet")) {
      msg_out = NULL;
      memmove(argv + i, argv + i + 1, (size_t)(argc - i) * sizeof(char *));
      argc--; i--;
    }
  }
  if (argc < 2 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) { fputs(usage, stderr); return EXIT_FAILURE; }
  int (*func)(int, char **) = NULL;
  if (!strcmp(argv[1], "build")) func = ctx_build;
  else if (!strcmp(argv[1], "sort")) func = ctx_sort;
  else if (!strcmp(argv[1], "index")) func = ctx_index;
  else if (!strcmp(argv[1], "hashtest")) func = ctx_hashtest;
  if (!func) {
    fprintf(stderr, "%s: command '%s' is not part of this build (build, sort, index and hashtest are)\n\n", CMD_NAME, argv[1]);
    fputs(usage, stderr);
    return EXIT_FAILURE;
  }
  int rc = func(argc - 1, argv + 1);
  gettimeofday(&t1, NULL);
  double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-6 * (double)(t1.tv_usec - t0.tv_usec);
  if (timing) fprintf(stderr, "[timing] epoch_main_exit %ld.%06ld\n", (long)t1.tv_sec, (long)t1.tv_usec);
  if (rc == 0) status("[time] %.2f seconds", secs);
  status(rc == 0 ? "  Done." : "  Fail.");
  /* Everything is written and closed.  Leaving through exit() would run the HIP runtime's teardown
   * over tens of GB of device allocations (0.1-0.3 s); the kernel driver reclaims them anyway.
   * Only a `build` that succeeded and has seen its device idle after the last call takes the
   * short way out (it sets host_fast_exit_ok).  Every other command and every failure returns
   * normally: atexit handlers, the HIP runtime's teardown and whatever profilers / sanitizers flush
   * at exit (rocprofv3, ASan/LSan, gcov) run as usual.  MCX_KEEP_DESTROY=1 forces the normal path. */
  fflush(NULL);
  if (rc == 0 && host_fast_exit_ok && !getenv("MCX_KEEP_DESTROY")) _exit(0);
  return rc;
}
