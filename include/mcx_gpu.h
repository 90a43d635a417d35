/*
 * mcx_gpu.h -- C ABI of the MI355X (gfx950) backend for McCortex `build`.
 *
 * The reference has no plugin/FFI layer: `ctx_build` (src/commands/ctx_build.c:245)
 * calls statically linked C functions.  This header is the set of entry points
 * a C host (the reference's own `ctx_build`, or our `mccortex<K> build`) binds
 * instead of those functions; every entry names the reference call it replaces
 * (paths relative to the reference root).  Plain pointers and sizes only.
 *
 * Conventions: every function returns MCX_OK (0) or a negative mcx_status and
 * sets a thread-local message readable with mcx_last_error().  No callbacks
 * into the host except the export sink.  A handle is NOT thread-safe: one
 * submitting host thread per handle (kernels run asynchronously on the
 * handle's HIP stream).  The library never falls back to a CPU path: without
 * a gfx950 device mcx_graph_create() fails with MCX_ERR_NODEVICE.
 */
#ifndef MCX_GPU_H_
#define MCX_GPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MCX_OK = 0,
  MCX_ERR_ARG = -1,       /* bad argument (even k, k out of range, colour >= ncols ...) */
  MCX_ERR_NODEVICE = -2,  /* no usable HIP device */
  MCX_ERR_NOMEM = -3,     /* device/host allocation failed */
  MCX_ERR_FULL = -4,      /* "Hash table is full" (src/graph/hash_table.c:119-123) */
  MCX_ERR_HIP = -5,       /* HIP runtime error, see mcx_last_error() */
  MCX_ERR_SINK = -6       /* export sink returned non-zero */
} mcx_status;

/* Subset of SeqLoadingStats (src/basic/seq_loading_stats.h:5-14) that the
 * build path fills (src/tools/build_graph.c:173-188,209-213). */
typedef struct {
  uint64_t num_se_reads;
  uint64_t num_good_reads;      /* reads with >= 1 contig */
  uint64_t num_bad_reads;       /* reads with no contig   */
  uint64_t total_bases_read;
  uint64_t total_bases_loaded;  /* sum of contig lengths  */
  uint64_t contigs_parsed;
  uint64_t num_kmers_loaded;    /* k-mer occurrences inserted */
  uint64_t num_kmers_novel;     /* occurrences that created a new node */
  /* --remove-pcr only (mcx_graph_add_reads_pcr) */
  uint64_t num_pe_reads;        /* reads loaded as mates of a pair */
  uint64_t num_dup_se_reads;    /* single reads dropped as PCR duplicates */
  uint64_t num_dup_pe_pairs;    /* pairs dropped as PCR duplicates */
} mcx_load_stats;

typedef struct mcx_graph mcx_graph;

const char *mcx_last_error(void);
const char *mcx_version(void);
int mcx_device_count(void);
/* Free / total HBM of a device in bytes (the build command checks -m/-n against
 * HBM where the reference checks host RAM, src/graph/cmd_mem.c:133-151). */
int mcx_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* Replaces db_graph_alloc (src/graph/db_graph.c:23) + hash_table_alloc
 * (src/graph/hash_table.c:16-52) for the build path.
 *   kmer_size       odd, 3..127 (W = 1 word for k<=31, 2 words for 33..63, 3 for
 *                   65..95, 4 for 97..127: the reference's MAXK = 31 / 63 / 95 / 127
 *                   builds, src/graph/binary_kmer.h:10-18, Makefile:33-48).  k > 63
 *                   builds on one device with the fused insert kernel (one HBM
 *                   atomic per occurrence); the partitioned insert, the exchange
 *                   formats and mcx_graph_create_multi are for k <= 63.
 *   ncols           colours (>=1); coverage and edges are kept per colour
 *   capacity_kmers  minimum number of k-mer slots (the reference's -n); the
 *                   slot layout, probe sequence and seed are free because
 *                   parity is on the sorted record set
 *   device          HIP device ordinal */
int mcx_graph_create(mcx_graph **g, int kmer_size, int ncols,
                     uint64_t capacity_kmers, int device);
/* One shard of a graph whose table is split over `nparts` GPUs (a power of two, <= 32) by hash
 * prefix: the key's owner is the top bits of the quotient-hash region index (mcx_graph_key_owner).
 * mcx_graph_create == nparts 1.  Every shard must be created with the same k / ncols / capacity. */
int mcx_graph_create_shard(mcx_graph **g, int kmer_size, int ncols, uint64_t capacity_kmers,
                           int device, int nparts, int part);
void mcx_graph_destroy(mcx_graph *g);

/* Empty the table and zero the statistics (graph stays allocated). */
int mcx_graph_reset(mcx_graph *g);

/* Tuning knobs (no reference equivalent).  Keys:
 *   "defer"        1 (default): k-mer occurrences are radix-partitioned by table region into
 *                  HBM bins and applied one sub-table at a time from LDS when the bins fill
 *                  up or the graph is read (sync / nkmers / stats / export); 0: every
 *                  occurrence is inserted straight into the HBM table with device atomics.
 *                  Both give the same graph.
 *   "defer_tuples" occurrences buffered per flush (sizes the bin workspace in HBM); default: 64 per
 *                  table slot within 30 % of the HBM free after the table was allocated.  A graph with
 *                  several colours shares this workspace between its colours (a pool of bin sets, each
 *                  bound to the colour that first writes to it): colours may alternate from call to
 *                  call without a flush, and a flush makes one pass over the table per colour.
 *                  Reads handed over in host memory while the device is idle are flushed in the
 *                  background, one group of table regions at a time (MCX_IDLE_FLUSH=0: off).
 *   "flush_regions" table regions split + applied per step of a flush (0 = automatic: 16 K sub-tables
 *                  per step); bounds the sub-table bin workspace to that share of the table
 *   "place_bins"   1 (default) .. 32: the sub-table bin workspace is the best of up to n allocations -- each half of the
 *                  flush overlap chosen by itself -- by a probe that writes the split kernel's pattern (16 K fronts,
 *                  128-byte runs) into them: where 8 GB of bins happen to lie in HBM decides 13 % of that kernel
 *                  (37.9-38.6 against 43.8-44.3 ms per 12 G occurrences at C2).  Costs n allocations and ~3 ms of probes
 *                  each at the first use of the workspace; candidates are held while HBM has room for them, and the search ends
 *                  as soon as one is 30 % faster than the slowest seen (about one place in five is).
 *   "intersect"    1: `build --intersect` (ctx_build.c:341-363,384-413).  The graph must have been
 *                  created with ONE colour more than the output: the last colour becomes hidden (not
 *                  exported, not scanned) and holds the union of the intersection graphs' edges,
 *                  loaded with mcx_graph_add_records(... into = that colour).  Implies "defer" 0.
 *   "must_exist"   1: mcx_graph_add_reads only updates k-mers that are already in the graph and adds
 *                  an edge only between consecutive k-mers that were both found
 *                  (BuildGraphTask.prefs.must_exist_in_graph, src/tools/build_graph.c:99-150)
 *   "prepare"      allocate now what the first mcx_graph_add_reads otherwise allocates (pinned staging
 *                  buffers, the partition workspace): lets a host that parses with other threads hide
 *                  ~0.1 s behind the parse of its first batch
 *   "profile"      1: time every kernel launch with HIP events (see mcx_graph_profile) */
int mcx_graph_configure(mcx_graph *g, const char *key, uint64_t value);
/* "kernel calls total_ms" per line for the launches recorded since "profile" was set. */
int mcx_graph_profile(mcx_graph *g, char *buf, size_t buflen);

/* One table over several GPUs of a node, driven by one host process: what the reference's batch
 * loop (src/commands/ctx_build.c:384-407) becomes when the hash table it feeds is split across
 * `ndevices` devices (a power of two <= 32; a device may be named more than once): by the hash of a
 * k-mer's canonical minimizer for odd k in 29..63 (exchange format v3: reads travel as super-k-mer
 * records, every device holds an ordinary table), by hash prefix otherwise (format v2: occurrences
 * travel; MCX_MULTI_EXCHANGE=v2 forces it).  mcx_graph_key_owner tells which device holds a key.
 * SURVEY.md 8(b) sketched this as the devices / ndevices arguments of mcx_graph_create.  The handle
 * that comes back is used like any other: mcx_graph_add_reads (the batch is dealt out to the
 * shards, each k-merises its piece and sends every other shard its occurrences with peer copies
 * over xGMI), mcx_graph_add_reads_pcr, mcx_graph_add_records, sync / statistics / scans / checksum,
 * mcx_graph_export (sorted: one device sort merges the shards).  mcx_graph_add_stream_dev takes a
 * stream on any of the devices.  Intersect / must-exist mode works as on one device (records are
 * dealt with by the shard that owns their key; reads are walked by every shard, which exchange what
 * they found before any of them updates a node: an edge needs both of its k-mers).  Hot k-mers and
 * low-complexity input (poly-G reads, satellite repeats) do not overflow the exchange: what fits neither
 * its segment nor the owner's overflow bin is spilled on the sender and routed by the host.  Format v2's
 * spill area holds a whole piece, so nothing can be lost; format v3 has room for 3 records per 16
 * positions in the segments (random reads make 2.3, low-complexity input fewer: long runs) and its spill
 * area holds one record per start position of a piece -- a record is a run of >= 1 k-mers, so no input
 * can overflow it.  Start-up runs a peer self-test: for every ordered pair of distinct devices one copy
 * kernel writes a pattern into the other device's memory through the peer mapping and the result is read
 * back; a pair that fails switches the group to hipMemcpyPeerAsync copies (what MCX_MULTI_COPY=memcpy
 * selects by hand) with one line on stderr.  Not available on such a handle: the
 * device-pointer exchange calls below (those take a single shard).  ndevices == 1 is mcx_graph_create. */
int mcx_graph_create_multi(mcx_graph **out, int kmer_size, int ncols, uint64_t capacity_kmers,
                           const int *devices, int ndevices);
/* HBM per device that the exchange buffers of such a table take on first use (send / receive sets of
 * one piece, spill areas), for the caller's memory check next to the table itself: the build command
 * adds it when it checks -m / -n against the free HBM (src/graph/cmd_mem.c:133-151).  0 for one device.
 * When the buffers do not fit after all, the library halves the piece instead of failing. */
int mcx_multi_exchange_bytes(int kmer_size, int ndevices, uint64_t capacity_kmers, uint64_t *bytes_per_device);
/* Devices the handle spans (1 for an ordinary graph). */
int mcx_graph_ndevices(const mcx_graph *g);

/* Slots actually allocated (>= capacity_kmers) and bytes of HBM held. */
int mcx_graph_capacity(const mcx_graph *g, uint64_t *slots, uint64_t *bytes);

/* Replaces build_graph_from_reads_mt (src/tools/build_graph.c:192-231, SE path
 * without --remove-pcr) for a whole batch of reads held in host memory:
 * contig split (src/basic/seq_reader.c:61-172), rolling k-mers, canonical key,
 * find-or-insert, coverage +1, edge OR (src/tools/build_graph.c:122-150).
 *   bases          concatenated read bases (ASCII, any case), no separators
 *   quals          NULL, or quality bytes in the same layout (only read when
 *                  fq_cutoff_abs > 0)
 *   read_offsets   nreads+1 offsets into bases/quals
 *   fq_cutoff_abs  prefs.fq_cutoff + FASTQ offset, 0 = off (build_graph.c:203-206)
 *   hp_cutoff      homopolymer cutoff, 0 = off
 *   stats_accum    optional; num_se_reads and total_bases_read are added here.
 *                  Contig / k-mer / good-read counters live on the device: read
 *                  them with mcx_graph_device_stats() (the host takes the delta
 *                  around each input file, as graph_info_update_stats needs
 *                  per-file totals, src/tools/build_graph.c:294-298).
 * The call returns once the batch is staged; use mcx_graph_sync() to drain. */
int mcx_graph_add_reads(mcx_graph *g, int colour,
                        const uint8_t *bases, const uint8_t *quals,
                        const uint64_t *read_offsets, uint64_t nreads,
                        uint8_t fq_cutoff_abs, uint8_t hp_cutoff,
                        mcx_load_stats *stats_accum);

/* The same step with prefs.remove_pcr_dups (`build --remove-pcr`): replaces
 * build_graph_from_reads_mt (src/tools/build_graph.c:192-231) + seq_reads_are_novel (:28-92).
 *   paired         != 0: reads 2i and 2i + 1 are the two mates of pair i (a --seq2 / --seqi
 *                  task); 0: single reads (--seq)
 *   matedir        0 FF, 1 FR, 2 RF, 3 RR (cortex_types.h:18-25): mates are turned to FF and
 *                  loaded that way (seq_reader_orient_mp_FF, src/basic/seq_reader.c:506-510);
 *                  a single read is reverse-complemented for RF / RR
 *   fq_cutoff_abs1/2  the cutoff of the first / second mate's file (single reads: the first)
 * A read (pair) is dropped when the first k-mer of (each of) its read(s) is the start, in the same
 * orientation, of a read (pair) loaded before it -- earlier in this batch or in an earlier call.
 * The reference's outcome depends on the order in which its worker threads reach the reads; this
 * gives the outcome of walking the reads in input order on one thread.  The start k-mers of
 * dropped reads stay in the graph, as in the reference.  num_pe_reads / num_se_reads,
 * total_bases_read and the duplicate counts are added to stats_accum; the call returns when the
 * filter has run (it is synchronous).  Costs 8 bytes of HBM per table slot while in use. */
int mcx_graph_add_reads_pcr(mcx_graph *g, int colour,
                            const uint8_t *bases, const uint8_t *quals,
                            const uint64_t *read_offsets, uint64_t nreads,
                            uint8_t fq_cutoff_abs1, uint8_t fq_cutoff_abs2, uint8_t hp_cutoff,
                            int paired, int matedir, mcx_load_stats *stats_accum);
/* Forget all read starts: the reference wipes dBGraph.readstrt whenever the colour being loaded
 * changes (src/commands/ctx_build.c:392-395). */
int mcx_graph_pcr_reset(mcx_graph *g);

/* Device-resident variant of the same step: `d_stream` is a byte stream in HBM
 * in which reads are separated by at least one byte that is not one of
 * ACGTacgt (e.g. '\n'); every maximal ACGT run of length >= k is one contig
 * (what seq_contig_start/end yield with -Q/-H off).  Must be 16-byte aligned.
 * Asynchronous on the handle's stream. */
int mcx_graph_add_stream_dev(mcx_graph *g, int colour,
                             const void *d_stream, uint64_t nbytes);

/* The same stream in packed form: per 16 positions one code word (2 bits per base, A=0 C=1 G=2 T=3,
 * first base on top) and 16 invalid flags (first base = bit 15; set for every position that does
 * not hold one of ACGTacgt, and for positions >= npos of the last word) -- 3 bits per position
 * instead of 8.  This is what mcx_graph_add_reads stages over PCIe (mcx_pack_bases on the host);
 * mcx_pack_stream_dev converts an ASCII stream that is already in HBM (d_code: (npos + 15) / 16
 * u32, d_inv: as many u16; asynchronous on `hip_stream`, a hipStream_t or NULL). */
int mcx_pack_stream_dev(const void *d_stream, uint64_t nbytes, void *d_code, void *d_inv, void *hip_stream);
int mcx_graph_add_packed_dev(mcx_graph *g, int colour, const void *d_code, const void *d_inv, uint64_t npos);

/* Sharded build.  Step 1 on every rank: k-merise a device
 * stream and bin the per-occurrence tuples (canonical key words, edge byte) by
 * owner = (second result word of lookup3(key) * nparts) >> 32 (= mcx_key_owner()).  d_keys holds nparts bins of
 * bin_capacity tuples, W words each; d_edges likewise one byte per tuple;
 * d_counts[nparts] (uint64) receives the fill of every bin (must be zeroed by
 * the caller).  Overflowing a bin sets MCX_ERR_FULL at the next sync. */
int mcx_graph_partition_stream_dev(mcx_graph *g, const void *d_stream, uint64_t nbytes,
                                   int nparts, uint64_t bin_capacity,
                                   void *d_keys, void *d_edges, void *d_counts);
/* Step 2 on the owner, after the exchange: insert n tuples. */
int mcx_graph_insert_tuples_dev(mcx_graph *g, int colour, const void *d_keys,
                                const void *d_edges, uint64_t n);
/* Owner of a canonical key in the exchange above (host-side helper for tests). */
uint32_t mcx_key_owner(const uint64_t *key_words, int kmer_size, int nparts);

/* `mccortex<K> hashtest` (src/commands/ctx_exp_hashtest.c:40-69: `bkmer.b[0] = i; hash_table_find_or_insert(...)`):
 * find-or-insert of the n BinaryKmers whose most significant word is i, i in [first, first + n), all other words 0 --
 * generated on the device, no edges, colour 0.  Ends with the table drained (mcx_graph_sync); MCX_ERR_FULL when the
 * table cannot hold them. */
int mcx_graph_hashtest(mcx_graph *g, uint64_t first, uint64_t n);
/* The command's -F mode (hash function only, nothing stored): the reference splits [0, n) into `nparts` ranges
 * (one per thread: range i starts at i * (n / nparts), the last one ends at n), XORs binary_kmer_hash(bkmer, 0) =
 * bklk3_hashlittle over each range and ADDS the ranges' results (ctx_exp_hashtest.c:61-66,160-175). */
int mcx_hashtest_func(int device, int kmer_size, uint64_t n, uint32_t nparts, uint64_t *hash_out);

/* Diagnostics (tools/exp_hbm_map.py): the split kernel's write pattern -- nregions x spb fronts `cap_words` apart,
 * 128-byte runs, `iters` x 64 runs per front -- on any device buffer, timed (the second of two runs, ms). */
int mcx_debug_probe(void *d_buf, uint64_t cap_words, uint32_t nregions, uint32_t spb, uint32_t iters, uint32_t jit_mask, float *ms_out);

/* Like mcx_graph_insert_tuples_dev for tuples that sit in `nseg` segments of `seg_cap` slots with
 * the fills in device memory (d_counts[nseg], u64; a fill above seg_cap is read as seg_cap), so the
 * caller needs no host round trip to learn how many arrived: the overflow bins of the sharded
 * exchange are consumed this way. */
int mcx_graph_insert_tuple_segments_dev(mcx_graph *g, int colour, const void *d_keys, const void *d_edges,
                                        const void *d_counts, uint32_t nseg, uint64_t seg_cap);

/* Sharded build, compact exchange: the sender bins packed occurrences (one 64-bit word per key
 * word) by (owner, region) of the sharded table, so every owner's block is one fixed-size message
 * and the receiver consumes it without re-binning.
 *   mcx_graph_shard_layout      segments per owner, tuples per segment and overflow capacity for
 *                               calls of at most `tuples_per_call` occurrences
 *   mcx_graph_shard_bins_dev    k-merise a stream into d_keys[nparts][segs][seg_cap][W] with fills
 *                               d_counts[nparts][segs] (zeroed by the caller); occurrences beyond a
 *                               segment go to the owner's overflow bin (full keys + edge bytes,
 *                               d_ov_counts[nparts] zeroed by the caller), beyond that: MCX_ERR_FULL
 *   mcx_graph_add_segments_dev  owner side: split received blocks (nseg segments of packed tuples
 *                               of THIS shard, `ntuples` in total) by sub-table; applied at the
 *                               next flush.  Overflow bins go through mcx_graph_insert_tuples_dev.
 *   mcx_graph_key_owner         owner of a canonical key under this graph's geometry */
int mcx_graph_shard_layout(mcx_graph *g, uint64_t tuples_per_call, uint32_t *segs_per_owner,
                           uint64_t *seg_cap, uint64_t *ov_cap);
int mcx_graph_shard_bins_dev(mcx_graph *g, const void *d_stream, uint64_t nbytes, void *d_keys,
                             void *d_counts, uint64_t seg_cap, void *d_ov_keys, void *d_ov_edges,
                             void *d_ov_counts, uint64_t ov_cap);
int mcx_graph_add_segments_dev(mcx_graph *g, int colour, const void *d_keys, const void *d_counts,
                               uint32_t nseg, uint64_t seg_cap, uint64_t ntuples);
uint32_t mcx_graph_key_owner(const mcx_graph *g, const uint64_t *key_words);

/* Bulk load of `.ctx` records: what graph_load() does for `build --graph <in.ctx>`
 * (src/graph/graphs_load.c:86-214, reader src/graph/graph_file_reader.c:347-420).
 * `recs` = nrecs records in the .ctx body layout (W x u64 key, file_ncols x u32 coverage,
 * file_ncols x u8 edges, byte-packed, host memory).  The colour filter is the reference's list of
 * (from, into) pairs (src/basic/file_filter.c): file colour from_col[i] is added to colour
 * into_col[i] of this graph (coverage +=, saturating at 2^32-1 on export; edges |=; targets and
 * sources may repeat).  Records whose loaded colours all have zero coverage are skipped.  MCX_RECORDS_MUST_EXIST: only update k-mers already in the graph
 * (GraphLoadingPrefs.must_exist_in_graph).  The per-record checks of graph_file_read_raw are
 * reported through the stats: index (counted from the first record ever passed with this stats
 * object) of the first record with zero coverage in every file colour / with edges but no coverage
 * in some colour (the reference warns once for each), or -1; an oversized k-mer fails the call
 * with MCX_ERR_ARG (the reference dies).  Synchronous. */
enum { MCX_RECORDS_MUST_EXIST = 1, MCX_RECORDS_MASK_EDGES = 2 };
typedef struct {
  uint64_t nkmers_read, nkmers_loaded, nkmers_novel;
  int64_t first_oversized, first_zero_covg, first_edges_no_covg; /* initialise to -1 */
} mcx_records_stats;
/* MCX_RECORDS_MASK_EDGES (intersect mode): only OR in edges that the hidden colour has for that
 * k-mer (GraphLoadingPrefs.must_exist_in_edges, graphs_load.c:166-167). */
int mcx_graph_add_records(mcx_graph *g, const void *recs, uint64_t nrecs, int file_ncols,
                          const int32_t *from_col, const int32_t *into_col, int nmap, uint32_t flags,
                          mcx_records_stats *stats_accum);

/* End of an intersect build: remove k-mers with no coverage in any visible colour and AND every
 * colour's edges with the intersection graphs' edges (db_graph_remove_no_covg_kmers +
 * db_graph_intersect_edges, src/graph/db_graph.c:630-673).  *removed = k-mers dropped. */
int mcx_graph_intersect_finish(mcx_graph *g, uint64_t *removed);

/* Table scans (imply a sync).
 * mcx_graph_kmer_covg       db_graph_get_kmer_covg (src/graph/db_graph.c:490-534): per colour, the
 *                           number of k-mers with coverage and their summed coverage (ncols entries)
 * mcx_graph_covg_histogram  the k-mer coverage histogram of `clean`'s first pass
 *                           (src/tools/clean_graph.c:365-377): hist[min(c, nbins-1)]++ for every
 *                           k-mer, c = its coverage summed over colours (saturating, db_node.h:302) */
int mcx_graph_kmer_covg(mcx_graph *g, uint64_t *nkmers, uint64_t *sumcov);
int mcx_graph_covg_histogram(mcx_graph *g, uint64_t *hist, uint32_t nbins);

/* Order-independent checksum of the graph: the sum, over its k-mers, of a 64-bit mix of the
 * exported record (key words, coverages clamped to 2^32-1, edges).  mcx_records_checksum computes
 * the same sum on the host from `.ctx` body records, so that two builds -- or a build and a file --
 * can be compared at sizes where byte-wise comparison of sorted exports is impractical. */
int mcx_graph_checksum(mcx_graph *g, uint64_t *checksum, uint64_t *nkmers);
uint64_t mcx_records_checksum(const void *recs, uint64_t nrecs, int kmer_size, int ncols);

/* The two ceilings of the device the build's roofline figures are quoted against ("to be
 * measured on the box by an in-repo microbenchmark and quoted next to the nominal figure"; the reference
 * times its table and the competitor in one script as well, results/hash_table_benchmark/benchmark-tables.sh:14-59).
 * mcx_ubench_stream      streaming rates in GB/s over two buffers of `bytes` each (far beyond the 256 MiB Infinity
 *                        Cache): copy (read + write bytes counted), read only, write only; 16-byte accesses
 * mcx_ubench_random_rmw  random 64-byte-sector rates per second over a `table_bytes` working set of 16-byte
 *                        records: agent-scope atomic add (what an occurrence costs the direct insert), 16-byte
 *                        load, load + atomic.  Any out pointer may be NULL. */
int mcx_ubench_stream(int device, uint64_t bytes, double *copy_gbs, double *read_gbs, double *write_gbs);
int mcx_ubench_random_rmw(int device, uint64_t table_bytes, uint64_t nupdates, double *rmw_per_s,
                          double *load16_per_s, double *load_rmw_per_s);

/* `.ctx` records on the device without a graph handle (host buffers, .ctx body layout).
 * mcx_sort_records    sort in place by k-mer, most significant word first -- what `sort` does with
 *                     qsort (src/commands/ctx_sort.c:133-152; binary_kmers_qcmp_unaligned_ptrs)
 * mcx_records_sorted  *first_unsorted = index of the first record whose k-mer is not greater than
 *                     its predecessor's, or -1 (the check `index` makes, src/commands/ctx_index.c:136) */
int mcx_sort_records(void *recs, uint64_t nrecs, int kmer_size, int ncols, int device);
int mcx_records_sorted(const void *recs, uint64_t nrecs, int kmer_size, int ncols, int device,
                       int64_t *first_unsorted);

/* Sharded build, exchange format v3 ("reads travel, not occurrences"; odd k in 29..63,
 * mcx_superk_supported(); records are mcx_superk_record_bytes(k) = 16 bytes for one-word keys, 32
 * for two-word keys: an 80-base window).  The owner of a k-mer is a function of its canonical minimizer
 * (smallest hashed canonical 13-mer), so consecutive k-mers of a read mostly share an owner and
 * are sent as one 16-byte record: the 48-base window of 16 consecutive positions plus the start and
 * length of the run -- about 2.3 bytes per occurrence instead of 8.5.  Every rank holds an ordinary
 * (unsharded) table with the k-mers it owns.
 *   mcx_superk_owner           owner of a canonical key (host; tests)
 *   mcx_graph_superk_layout    records per (owner, replica) segment for calls of at most
 *                              `positions_per_call` stream bytes; segs_per_owner replicas per owner
 *   mcx_graph_superk_bins_dev  sender: d_recs[nparts][segs][seg_cap] 16-byte records with fills
 *                              d_counts[segs][nparts] (u64, replica-major, zeroed by the caller); a
 *                              full segment drops records and raises MCX_ERR_FULL at the next sync
 *   mcx_graph_add_superk_dev   owner: k-merise nseg received segments d_recs[nseg][seg_cap] (fills
 *                              d_counts[nseg] in device memory) into the region bins; applied at the
 *                              next flush */
int mcx_superk_supported(int kmer_size);
int mcx_superk_record_bytes(int kmer_size);
uint32_t mcx_superk_owner(const uint64_t *key_words, int kmer_size, int nparts);
int mcx_graph_superk_layout(mcx_graph *g, int nparts, uint64_t positions_per_call, uint32_t *segs_per_owner,
                            uint64_t *seg_cap);
int mcx_graph_superk_bins_dev(mcx_graph *g, const void *d_stream, uint64_t nbytes, int nparts, void *d_recs,
                              void *d_counts, uint64_t seg_cap);
int mcx_graph_add_superk_dev(mcx_graph *g, int colour, const void *d_recs, const void *d_counts, uint32_t nseg,
                             uint64_t seg_cap, uint64_t kmers_upper_bound);

/* Host-side packer of the staging path of mcx_graph_add_reads, exported for its test: n (a
 * multiple of 32) ASCII characters -> n / 16 code words (2 bits per base, A=0 C=1 G=2 T=3 as
 * src/basic/dna.c:8-25, first base on top) and n / 16 x 16 invalid flags (first base = bit 15; set
 * for every character that is not one of ACGTacgt). */
void mcx_pack_bases(const uint8_t *src, uint64_t n, uint32_t *code, uint16_t *inv);

/* The one-pass packer of the same staging path (round 4), exported for its test: the reads
 * bases[off[i] .. off[i+1]) as the stream "128 separators, read 0, separator, read 1, separator, ...,
 * separators up to a multiple of 64 positions" -> code[p / 16], inv[p / 16]; returns the number of
 * positions (0: cap_pos too small, or fused != 0 on a host without AVX-512).  fused == 0 assembles
 * the ASCII stream and packs it with mcx_pack_bases: both must give the same words.  Replaces the
 * per-read copy of the reference's workers (src/basic/async_read_io.c:283-310 hands whole reads over). */
uint64_t mcx_pack_reads_host(const uint8_t *bases, const uint64_t *off, uint64_t nreads, uint32_t *code,
                             uint16_t *inv, uint64_t cap_pos, int fused);

/* Wait for all submitted work; reports MCX_ERR_FULL if any insert ran out of
 * slots (the reference dies with "Hash table is full"). */
int mcx_graph_sync(mcx_graph *g);

/* hash_table_nkmers (src/graph/hash_table.h:37). Implies a sync. */
int mcx_graph_nkmers(mcx_graph *g, uint64_t *n);
/* Contig/k-mer counters accumulated by the device since create/reset. */
int mcx_graph_device_stats(mcx_graph *g, mcx_load_stats *out);

/* How the inserts went: the slow-but-correct paths ("cliffs") of the partitioned build, made visible.  The
 * reference prints its own insert diagnostics, the collision histogram of hash_table_print_stats
 * (src/graph/hash_table.c:301-333); these are the ones this table has.  All zero on well-spread input (bench
 * C2: asserted in tests/test_gpu_fullsize.py); printed by `mccortex<K> build` under MCX_TIMING=1.
 *   fallback_inserts  occurrences that did not fit their partition bin (hot k-mers, an owner far above its share)
 *                     and took the per-occurrence lock-free insert in HBM instead of the LDS insert
 *   foreign_inserts   occurrences whose key another shard owns, handed to this shard through an entry that does
 *                     not route (mcx_graph_add_reads on a shard handle): inserted on the spot
 *   spilled           multi-GPU table: occurrences (exchange v2) / super-k-mer records (v3) that did not fit a
 *                     send segment and went through the sender's spill area, routed by the host
 *   flushes           passes over the table made by the partitioned insert (per colour with buffered tuples)
 * Implies a sync; a multi-GPU handle reports sums over its shards. */
typedef struct {
  uint64_t fallback_inserts;
  uint64_t foreign_inserts;
  uint64_t spilled;
  uint64_t flushes;
} mcx_insert_stats;
int mcx_graph_insert_stats(mcx_graph *g, mcx_insert_stats *out);

/* HIP stream the handle submits on (hipStream_t as void*), so callers can
 * bracket it with their own events. */
void *mcx_graph_stream(mcx_graph *g);

/* Replaces graph_write_all_kmers_direct (src/graph/graph_writer.c:182-268):
 * streams the records in .ctx v6 body layout (W x u64 key, ncols x u32 covg,
 * ncols x u8 edges; src/graph/graph_writer.c:116-127) to `sink` in chunks.
 * sorted != 0 orders records by key (hash_table_sorted, hash_table.c:362-374),
 * which is the only order in which the reference output is reproducible. */
typedef int (*mcx_sink_fn)(void *ctx, const void *records, size_t nbytes);
int mcx_graph_export(mcx_graph *g, int sorted, mcx_sink_fn sink, void *ctx);

/* Host-side primitives exported for parity tests of rows A-C (no device). */
void mcx_kmer_from_str(const char *seq, int kmer_size, uint64_t *words_out);
void mcx_kmer_canonical(const uint64_t *words_in, int kmer_size, uint64_t *key_out, int *orient_out);
uint32_t mcx_kmer_hash(const uint64_t *key_words, int kmer_size, uint32_t initval);

#ifdef __cplusplus
}
#endif
#endif /* MCX_GPU_H_ */
