// mcx_api.hip -- C ABI (include/mcx_gpu.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC mcx_api.hip -o libmcxgpu.so
#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <array>
#include <cmath>
#include <memory>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <functional>

#include "../../include/mcx_gpu.h"
#include "mcx_kernels.h"
#include "mcx_superk.h"
#include "mcx_streamfc.h"
#include "mcx_ubench.h"

using namespace mcx;

static thread_local char g_err[512] = "";
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int fail(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(_e == hipErrorOutOfMemory ? MCX_ERR_NOMEM : MCX_ERR_HIP, "%s: %s (%s:%d)",   \
                  #expr, hipGetErrorString(_e), __FILE__, __LINE__);                           \
  } while (0)

// device scratch that is released on every way out of a function (HIP_TRY returns early)
template <class T> struct DevBuf {
  T *p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void **)&p, n * sizeof(T)); }
  operator T *() const { return p; }
};

// Persistent staging threads.  A chunk of 128 Mi positions is packed in ~2 ms: creating and joining
// fifteen threads for each chunk was a tenth of that.  run(T, fn) calls fn(0) on the caller and
// fn(1..T-1) on the pool, and returns when all have finished; concurrent callers take turns.
class StagePool {
 public:
  static StagePool &get() { static StagePool *p = new StagePool(); return *p; }  // (never destroyed: its threads sleep until the process ends)
  // The pool's turn (call_mu_) is held from begin() to finish(): a run() or begin() reached on the SAME thread in
  // between would wait for itself.  No such path exists (submit() of add_reads_packed never packs), and owner_
  // turns one into an abort with a message instead of a hang.
  void run(int T, const std::function<void(int)> &fn)
  {
    if (T <= 1) { fn(0); return; }
    check_not_owner();
    std::lock_guard<std::mutex> turn(call_mu_);
    {
      std::unique_lock<std::mutex> lk(mu_);
      while ((int)th_.size() < T - 1) th_.emplace_back([this] { worker(); }), th_.back().detach();
      fn_ = &fn; T_ = T; next_ = 1; pending_ = T - 1;
    }
    cv_work_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr; T_ = 0; next_ = 0;
  }
  // Asynchronous variant: fn(0) .. fn(T - 1) all run on pool threads while the caller does something else;
  // finish() returns when they are done.  One job at a time: begin() takes the pool's turn, finish() gives it back
  // (both from the same thread).  `fn` must stay alive until finish().
  void begin(int T, const std::function<void(int)> &fn)
  {
    check_not_owner();
    call_mu_.lock();
    owner_.store(std::this_thread::get_id());
    {
      std::unique_lock<std::mutex> lk(mu_);
      while ((int)th_.size() < T) th_.emplace_back([this] { worker(); }), th_.back().detach();
      fn_ = &fn; T_ = T; next_ = 0; pending_ = T;
    }
    cv_work_.notify_all();
  }
  void finish()
  {
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_done_.wait(lk, [&] { return pending_ == 0; });
      fn_ = nullptr; T_ = 0; next_ = 0;
    }
    owner_.store(std::thread::id());
    call_mu_.unlock();
  }
 private:
  void check_not_owner() const
  {
    if (owner_.load() == std::this_thread::get_id()) {
      fprintf(stderr, "mcx: internal error: the staging pool was re-entered between begin() and finish()\n");
      abort();
    }
  }
  std::atomic<std::thread::id> owner_{std::thread::id()};
  void worker()
  {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_work_.wait(lk, [&] { return next_ < T_; });
      const int ti = next_++;
      const std::function<void(int)> *f = fn_;
      lk.unlock();
      (*f)(ti);
      lk.lock();
      if (--pending_ == 0) cv_done_.notify_one();
    }
  }
  std::mutex mu_, call_mu_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<std::thread> th_;
  const std::function<void(int)> *fn_ = nullptr;
  int T_ = 0, next_ = 0, pending_ = 0;
};

constexpr uint64_t kCarry = 128;  // context bytes carried between chunks
constexpr uint32_t kL1Sets = 32;  // L1 bin sets of a graph with several colours (mcx_graph::nsets; <= TupleIn::set_map)
// New stream bytes per staged chunk (MCX_STAGE_BYTES overrides, for tests of the chunk seams).
static uint64_t stage_bytes()
{
  static uint64_t v = 0;
  if (!v) {
    const char *e = getenv("MCX_STAGE_BYTES");
    v = e ? strtoull(e, nullptr, 10) : (128ull << 20);
    if (v < 1024) v = 1024;
    v = (v + 63) / 64 * 64;
  }
  return v;
}
#define kStageBytes stage_bytes()

struct mcx_group;
struct mcx_graph {
  int k = 0, W = 0, ncols = 0, device = 0;
  hipStream_t stream = nullptr;
  TableView t{};
  uint64_t table_bytes = 0;
  uint64_t touch_bytes = 0;  // TableView::touch
  Counters *d_ctr = nullptr;
  Counters *h_ctr = nullptr;  // pinned
  uint32_t *h_full = nullptr;  // pinned: the table's "full" flag (TableView::touch[-2]) as of a recent chunk of the host entry
  // host staging (double buffered)
  // (three staging pairs since round 4: with two, the host waited 20 ms per 6 G occurrences for the device to release one)
  static constexpr int kStageBufs = 3;
  uint8_t *h_stage[kStageBufs] = {nullptr, nullptr, nullptr};
  uint8_t *d_stage[kStageBufs] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[kStageBufs] = {nullptr, nullptr, nullptr};
  // the packed host entry copies on a stream of its own (cstream), so that the graph's stream only holds compute:
  // ev_copy[b] = chunk b has arrived; ev_wait0 / ev_wait1[b] bracket the compute stream's wait for it (timing events:
  // their distance is how long the device sat idle waiting for PCIe -- add_reads_packed)
  hipStream_t cstream = nullptr;
  hipEvent_t ev_copy[kStageBufs] = {nullptr, nullptr, nullptr}, ev_wait0[kStageBufs] = {nullptr, nullptr, nullptr}, ev_wait1[kStageBufs] = {nullptr, nullptr, nullptr};
  bool ev_wait_used[kStageBufs] = {false, false, false};
  uint64_t stage_alloc = 0;
  int cur = 0;
  int grid = 0;
  int grid_stream = 0, grid_split = 0, grid_insert = 0;  // 0 = default; blocks of the three build kernels (experiments with concurrent launches)
  // ---- deferred (partition -> LDS insert) path, mcx_defer.h ----
  bool defer = true;
  uint64_t defer_tuples = 0;    // tuples buffered per flush (0 = pick from the table size)
  uint32_t nsub = 0;            // sub-tables
  uint32_t b1 = 0, subs_per_bin = 0;
  uint64_t cap1 = 0, cap2 = 0;  // tuples per L1 (replica, bin) segment / per L2 (sub-table) bin
  // Replicas of every L1 bin.  A block appends to replica blockIdx % rep1; all blocks of a replica reserve from the same 64
  // counter lines once per tile.  8 (one per XCD) until round 6; for a one-colour graph with a large flush window 32 or 64
  // since (segments of >= 64 K tuples): k_stream_bin 23.3 -> 21.2 (32) -> 20.4 ms (64) per 6 G occurrences at C2.  The split
  // paid for many short segments from 64 on (21.9 / 22.3 / 22.9 ms at 32 / 64 / 128) until its output was placed
  // (place_bins: 19.4 at 32 and at 64, 20.2 at 128); profiles/r06_experiments.md.  Chosen in ensure_defer; MCX_REP1 forces it.
  uint32_t rep1 = 8;
  bool rep1_forced = false;
  uint64_t *l1_keys = nullptr, *l2_keys = nullptr;  // packed tuples: W words each
  uint64_t *l2_hi = nullptr;    // placed bins (place_bins > 1): the second half of the flush overlap is an allocation of its own;
  uint64_t l2_hi_off = 0;       // bins >= l2_hi_off live there (l2_ptr)
  uint64_t *l2_base = nullptr;  // the allocation l2_keys lies in
  int place_tries = 1;          // > 1: the sub-table bins are the best of this many allocations by the split's write-pattern probe (ensure_l2)
  uint64_t l2_off = 0;            // first sub-table bin the next split / insert launch uses (flush overlap: two halves)
  hipStream_t stream2 = nullptr;  // flush overlap: the LDS insert of group g runs beside the split of group g + 1
  hipEvent_t ev_split[2] = {nullptr, nullptr}, ev_ins[2] = {nullptr, nullptr};
  unsigned long long *l1_cnt = nullptr, *l2_cnt = nullptr;
  // L1 bin sets.  A one-colour graph has one set: the L1 bins.  A graph with several colours cuts the
  // same workspace into kL1Sets smaller sets, each [rep1][b1][cap1]; a set is bound to the colour that
  // first writes to it and takes set_cap occurrences (upper bound), a colour takes as many sets as it
  // needs, and nothing is flushed until the pool is exhausted -- so the samples of a population build
  // may alternate (db_node.h:240-241: coverage and edges are per colour) without a table pass per
  // switch.  A flush then makes ONE pass per colour: all sets of a colour are split in one launch
  // (TupleIn::set_map) and applied by one LDS insert.
  uint32_t nsets = 1;
  uint64_t set_cap = 0;                 // occurrences (upper bound) one set takes
  std::vector<int> set_colour;          // colour a set is bound to, -1 = free
  std::vector<uint64_t> set_pending;    // upper bound of the tuples in the set
  uint64_t pending = 0;         // upper bound of tuples sitting in the L1 bins (all sets)
  uint64_t pending_l2 = 0;      // tuples already split into the sub-table bins (sharded receive path)
  uint32_t l2_regions = 0;      // regions the L2 (sub-table) bins cover: a flush splits and applies
                                // the L1 bins in groups of this many regions, reusing the same bins
  uint32_t flush_regions = 0;   // configured group size (0 = automatic)
  uint32_t idle_next = 0;       // next region group the idle-device flush takes (flush_if_device_idle)
  // idle flush bookkeeping: with A = idle_base + pending = occurrences handed to the L1 bins since the last whole
  // flush, idle_mark[i] = A when region group i was last emptied and idle_base = min(idle_mark): `pending` is what
  // the group that has waited LONGEST may hold per its share -- the bound the segments' capacity is sized for
  uint64_t idle_base = 0;
  std::vector<uint64_t> idle_mark;
  // Settled launches (snap_*): `pending` counts a stream launch with the number of START POSITIONS it covers, an
  // upper bound of the tuples it yields (20 % above the truth for 150 bp reads at k = 31, 72 % at k = 63: a flush
  // -- a whole table pass -- came that much too early).  After every such launch the device's k-mer counter is
  // copied to a pinned ring slot behind the kernel; when the copy has landed (event query, never a wait) the
  // launches up to it are "settled": what they really yielded is known, and the difference is taken off `pending`.
  // Round 5: every entry is one launch -- its reservation `ub`, the bin set that took it and which counter it feeds
  // (0: `kmers`, stream launches; 1: `binned`, the owner side of exchange v3, whose reservation is an upper bound of what
  // the received records hold) -- so graphs with several colours (bin sets) and the shards of a multi-GPU table settle
  // too: yield = counter behind the launch - counter behind the previous snapped point (a launch that found the ring
  // full is not snapped: its yield is then charged to the next entry, which only makes the books more cautious).
  static constexpr uint32_t kSnap = 32;
  struct Snap { uint64_t ub; int set; int which; bool is_base; hipEvent_t ev; };
  Counters *h_snap = nullptr;  // pinned [kSnap]: the counters as of slot i
  Snap snap[kSnap] = {};
  uint32_t snap_head = 0, snap_tail = 0;  // ring: [tail, head) in flight
  uint64_t snap_last[2] = {0, 0};         // the two counters at the last settled point
  bool snap_base_known = true;            // (a new graph: counters and bins are zero)
  uint64_t n_flushes = 0;                 // flushes of the partition bins since create / reset (each = a table pass per colour)
  // ---- build --intersect (ctx_build.c:341-363,384-413) ----
  int hidden = -1;              // colour that holds the intersection graphs' edges, or -1
  int ncols_vis = 0;            // colours that are exported / scanned (ncols, or ncols - 1)
  uint32_t *d_readstrt = nullptr;  // --remove-pcr: T(node) per (slot, orientation), see k_pcr_starts
  bool must_exist = false;      // reads only update k-mers already in the graph
  int pending_colour = 0;
  // ---- optional per-kernel timing (mcx_graph_configure("profile", 1)) ----
  bool profile = false;
  int overlap = -1;  // flush overlap: -1 = MCX_FLUSH_OVERLAP / default (on), 0 / 1 = mcx_graph_configure("flush_overlap", v)
  struct Span { const char *name; hipEvent_t a, b; };
  std::vector<Span> spans;
  // ---- multi-GPU table (mcx_multi.h): a shard knows its group; the handle the caller holds is a
  // facade (as_group) whose calls are dealt out to the shards ----
  mcx_group *group = nullptr;
  int gidx = 0;
  mcx_group *as_group = nullptr;
  uint32_t own_lbo = 0;  // > 0: a shard of a group that deals the keys out by minimizer (exchange format v3): log2 shards
};

// how the kernels that walk records / reads on every shard tell their own keys (mcx_kernels.h: OwnerSpec)
static uint8_t *touch_base(const mcx_graph *g) { return reinterpret_cast<uint8_t *>(g->t.touch) - kTouchHdr; }

static OwnerSpec owner_spec(const mcx_graph *g)
{
  if (g->own_lbo) return OwnerSpec{2u, g->own_lbo, (uint32_t)g->gidx, g->k};
  return OwnerSpec{g->t.lbo ? 1u : 0u, 0u, 0u, g->k};
}

static int flush_deferred(mcx_graph *g);
static void free_defer(mcx_graph *g);
static void snap_push(mcx_graph *g, bool is_base);
static void snap_push(mcx_graph *g, bool is_base, uint64_t ub, int set, int which);
static void sets_release(mcx_graph *g)
{
  std::fill(g->set_colour.begin(), g->set_colour.end(), -1);
  std::fill(g->set_pending.begin(), g->set_pending.end(), 0);
}
struct StreamLaunch;
static int group_submit_stream(mcx_group *G, int idx, const StreamLaunch &L, int colour);
static void group_destroy(mcx_group *G);
static int grp_drain(mcx_group *G);
static int grp_sync(mcx_group *G);
static int grp_device_stats(mcx_group *G, mcx_load_stats *out);
static int grp_add_reads(mcx_group *G, int colour, const uint8_t *bases, const uint8_t *quals, const uint64_t *off,
                         uint64_t nreads, uint8_t fq, uint8_t hp, mcx_load_stats *stats_accum);
static int grp_add_reads_pcr(mcx_group *G, int colour, const uint8_t *bases, const uint8_t *quals, const uint64_t *off,
                             uint64_t nreads, uint8_t fq1, uint8_t fq2, uint8_t hp, int paired, int matedir,
                             mcx_load_stats *stats_accum);
static int grp_add_reads_must_exist(mcx_group *G, int colour, const uint8_t *bases, const uint8_t *quals, const uint64_t *off,
                                    uint64_t nreads, uint8_t fq, uint8_t hp, mcx_load_stats *stats_accum);
static int grp_intersect_finish(mcx_group *G, uint64_t *removed);
static int grp_add_records(mcx_group *G, const void *recs, uint64_t nrecs, int file_ncols, const int32_t *from_col,
                           const int32_t *into_col, int nmap, uint32_t flags, mcx_records_stats *stats_accum);
static int grp_export(mcx_group *G, mcx_graph *f, int sorted, mcx_sink_fn sink, void *ctx);
static int grp_part_of_pointer(mcx_group *G, const void *d_ptr, mcx_graph **part);
static mcx_graph *grp_part(mcx_group *G, int i);
static int grp_n(mcx_group *G);
static uint64_t grp_spilled(mcx_group *G);
#define NO_GROUP(g, what) do { if ((g) && (g)->as_group) return fail(MCX_ERR_ARG, what " takes a shard, not the multi-GPU handle"); } while (0)

extern "C" const char *mcx_last_error(void) { return g_err; }
extern "C" const char *mcx_version(void) { return "mccortex_amd 0.1 (gfx950)"; }

extern "C" int mcx_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int mcx_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(MCX_ERR_NODEVICE, "no HIP device %d", device);
  HIP_TRY(hipSetDevice(device));
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return MCX_OK;
}

static int check_k(int k)
{
  if (k < 3 || k > 127 || !(k & 1)) return fail(MCX_ERR_ARG, "kmer size must be odd and 3..127 (got %d)", k);
  return MCX_OK;
}

extern "C" int mcx_graph_create(mcx_graph **out, int kmer_size, int ncols, uint64_t capacity_kmers, int device)
{
  return mcx_graph_create_shard(out, kmer_size, ncols, capacity_kmers, device, 1, 0);
}

// log2 of the number of regions of a table of nsub sub-tables that is one of 2^lbo hash-prefix shards
static uint32_t region_bits(uint64_t nsub, uint32_t lbo)
{
  uint32_t lb1 = 0;
  const uint32_t lb1_max = lbo ? std::min<uint32_t>(9, 11 - lbo) : 9;  // shards x regions <= 2048 sender bins
  while (lb1 < lb1_max && (2ull << lb1) <= nsub) lb1++;             // up to 512 regions ...
  while (lb1 < 11 && ((nsub + (1ull << lb1) - 1) >> lb1) > (uint64_t)kMaxBins) lb1++;  // ... more for huge tables
  return lb1;
}

extern "C" int mcx_graph_create_shard(mcx_graph **out, int kmer_size, int ncols, uint64_t capacity_kmers, int device,
                                      int nparts, int part)
{
  if (!out) return fail(MCX_ERR_ARG, "null handle pointer");
  *out = nullptr;
  if (check_k(kmer_size) != MCX_OK) return MCX_ERR_ARG;
  if (nparts < 1 || nparts > 32 || (nparts & (nparts - 1)) || part < 0 || part >= nparts)
    return fail(MCX_ERR_ARG, "shards must be a power of two <= 32 and 0 <= part < shards (got %d of %d)", part, nparts);
  if (ncols < 1 || ncols > 4096) return fail(MCX_ERR_ARG, "ncols out of range: %d", ncols);
  if (kmer_size > 63 && nparts != 1)
    return fail(MCX_ERR_ARG, "k > 63 (three- and four-word keys) builds on one device: the exchange formats carry at most two key words");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(MCX_ERR_NODEVICE, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(MCX_ERR_NODEVICE, "device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));

  mcx_graph *g = new mcx_graph();
  g->k = kmer_size;
  g->W = words_for_k(kmer_size);
  g->ncols = ncols;
  g->ncols_vis = ncols;
  g->device = device;
  // geometry of the quotient-hashed table: 2^lb1 regions x spb sub-tables x 4096 (W=2: 2048) slots
  const uint64_t sub_slots = 1ull << sub_shift_for_words(g->W);
  uint64_t nsub = (std::max<uint64_t>(capacity_kmers, 1024) + sub_slots - 1) / sub_slots;
  uint32_t lbo = 0;
  while ((1 << lbo) < nparts) lbo++;
  uint32_t lb1 = region_bits(nsub, lbo);
  if (const char *e = getenv("MCX_LB1")) { const uint32_t v = (uint32_t)atoi(e); if (!lbo && v >= 6 && v <= 11 && ((nsub + (1ull << v) - 1) >> v) <= (uint64_t)kMaxBins && ((nsub + (1ull << v) - 1) >> v) >= 1) lb1 = v; }  // experiments
  if (lbo && lb1 + lbo > 11) {
    // (owner, region) bins of the sender kernel: at most 2048, and mix_bucket() takes its bits below
    // the lb1 + lbo <= 12 it assumes.  Per shard that is 2048 / shards regions x 2048 sub-tables.
    const uint64_t max_slots = ((uint64_t)kMaxBins << (11 - lbo)) * sub_slots;
    delete g;
    return fail(MCX_ERR_ARG, "capacity per device too large for a table split over %d devices: %llu slots requested, at most %llu "
                "(use more devices or a smaller -n / -m)", nparts, (unsigned long long)(nsub * sub_slots), (unsigned long long)max_slots);
  }
  const uint64_t spb = (nsub + (1ull << lb1) - 1) >> lb1;
  nsub = spb << lb1;
  if (nsub >= (1ull << 31)) { delete g; return fail(MCX_ERR_ARG, "capacity too large"); }
  // overflow area behind the hash-addressed slots (mcx_kernels.h, ovf_start): 1/32 of the table,
  // at least one sub-table's worth
  uint64_t novf = std::max<uint64_t>(sub_slots, (nsub * sub_slots / 32 + sub_slots - 1) / sub_slots * sub_slots);
  if (const char *e = getenv("MCX_OVERFLOW_SLOTS")) novf = strtoull(e, nullptr, 10) / kBucket * kBucket;  // tests
  if (novf / kBucket >= (1ull << 32)) novf = ((1ull << 32) - 1) * kBucket;
  const uint64_t slots = nsub * sub_slots + novf;
  g->t.nmain = nsub * sub_slots;
  g->t.nslots = slots;
  g->t.lb1 = lb1;
  g->t.lbo = lbo;
  g->t.part = (uint32_t)part;
  g->t.spb = (uint32_t)spb;
  g->t.S = (uint32_t)(g->W + ncols);
  // one colour: records [key words, value]; several: key array, then one value array per colour
  g->t.max_probe = (uint32_t)sub_slots;  // a probe sequence never leaves its sub-table
  { const char *e = getenv("MCX_DEFER"); if (e) g->defer = atoi(e) != 0; }
  // k = 65 .. 127 (mccortex95 / mccortex127: three- and four-word keys) takes the fused kernel with one HBM atomic per
  // occurrence: the partition path's tuple formats and LDS images are laid out for one and two key words
  if (g->W > 2) g->defer = false;
  { const char *e = getenv("MCX_GRID_STREAM"); if (e) g->grid_stream = atoi(e); }  // experiments
  { const char *e = getenv("MCX_REP1"); if (e && atoi(e) >= 1 && atoi(e) <= 4096) { g->rep1 = (uint32_t)atoi(e); g->rep1_forced = true; } }
  { const char *e = getenv("MCX_GRID_SPLIT"); if (e) g->grid_split = atoi(e); }
  { const char *e = getenv("MCX_GRID_INSERT"); if (e) g->grid_insert = atoi(e); }
  g->table_bytes = slots * g->t.S * 8;

  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  g->grid = prop.multiProcessorCount * 8;

#define CREATE_TRY(expr)                                                            \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      int rc = fail(_e == hipErrorOutOfMemory ? MCX_ERR_NOMEM : MCX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
      mcx_graph_destroy(g);                                                         \
      return rc;                                                                    \
    }                                                                               \
  } while (0)
  CREATE_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
  CREATE_TRY(hipMalloc((void **)&g->t.rec, g->table_bytes));
  if (ncols == 1) {
    g->t.KS = g->t.VS = (uint32_t)(g->W + 1);
    g->t.val = g->t.rec + g->W;
    g->t.VC = 0;
  } else {
    g->t.KS = (uint32_t)g->W;
    g->t.VS = 1;
    g->t.val = g->t.rec + g->t.nslots * (uint64_t)g->W;
    g->t.VC = g->t.nslots;
  }
  CREATE_TRY(hipMalloc((void **)&g->d_ctr, sizeof(Counters)));
  CREATE_TRY(hipHostMalloc((void **)&g->h_ctr, sizeof(Counters), hipHostMallocDefault));
  CREATE_TRY(hipHostMalloc((void **)&g->h_full, sizeof(uint32_t), hipHostMallocDefault));
  *g->h_full = 0;
  g->touch_bytes = (1 + ((g->t.nmain >> sub_shift_for_words(g->W)) + 31) / 32) * 4;
  {  // (kTouchHdr bytes of slow-path counters in front of the flags: TableView::touch)
    uint8_t *tb = nullptr;
    CREATE_TRY(hipMalloc((void **)&tb, kTouchHdr + g->touch_bytes));
    g->t.touch = reinterpret_cast<uint32_t *>(tb + kTouchHdr);
  }
  CREATE_TRY(hipMemsetAsync(g->t.rec, 0, g->table_bytes, g->stream));
  CREATE_TRY(hipMemsetAsync(touch_base(g), 0, kTouchHdr + g->touch_bytes, g->stream));
  CREATE_TRY(hipMemsetAsync(g->d_ctr, 0, sizeof(Counters), g->stream));
  CREATE_TRY(hipStreamSynchronize(g->stream));
#undef CREATE_TRY
  *out = g;
  return MCX_OK;
}

extern "C" void mcx_graph_destroy(mcx_graph *g)
{
  if (g && g->as_group) { group_destroy(g->as_group); delete g; return; }
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  if (g->cstream) { (void)hipStreamSynchronize(g->cstream); (void)hipStreamDestroy(g->cstream); }
  for (int i = 0; i < mcx_graph::kStageBufs; i++) {
    if (g->h_stage[i]) (void)hipHostFree(g->h_stage[i]);
    if (g->d_stage[i]) (void)hipFree(g->d_stage[i]);
    if (g->ev[i]) (void)hipEventDestroy(g->ev[i]);
    if (g->ev_copy[i]) (void)hipEventDestroy(g->ev_copy[i]);
    if (g->ev_wait0[i]) (void)hipEventDestroy(g->ev_wait0[i]);
    if (g->ev_wait1[i]) (void)hipEventDestroy(g->ev_wait1[i]);
  }
  free_defer(g);
  for (auto &sp : g->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  if (g->t.rec) (void)hipFree(g->t.rec);
  if (g->t.touch) (void)hipFree(touch_base(g));
  if (g->stream2) {
    (void)hipStreamDestroy(g->stream2);
    for (int i = 0; i < 2; i++) { (void)hipEventDestroy(g->ev_split[i]); (void)hipEventDestroy(g->ev_ins[i]); }
  }
  if (g->d_ctr) (void)hipFree(g->d_ctr);
  if (g->d_readstrt) (void)hipFree(g->d_readstrt);
  if (g->h_ctr) (void)hipHostFree(g->h_ctr);
  if (g->h_full) (void)hipHostFree(g->h_full);
  if (g->h_snap) {
    (void)hipHostFree(g->h_snap);
    for (uint32_t i = 0; i < mcx_graph::kSnap; i++) (void)hipEventDestroy(g->snap[i].ev);
  }
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
}

extern "C" int mcx_graph_reset(mcx_graph *g)
{
  if (g && g->as_group) {
    int rc = grp_drain(g->as_group);
    for (int i = 0; rc == MCX_OK && i < grp_n(g->as_group); i++) rc = mcx_graph_reset(grp_part(g->as_group, i));
    return rc;
  }
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  HIP_TRY(hipSetDevice(g->device));
  HIP_TRY(hipMemsetAsync(g->t.rec, 0, g->table_bytes, g->stream));
  HIP_TRY(hipMemsetAsync(touch_base(g), 0, kTouchHdr + g->touch_bytes, g->stream));
  HIP_TRY(hipMemsetAsync(g->d_ctr, 0, sizeof(Counters), g->stream));
  g->n_flushes = 0;
  if (g->h_full) { HIP_TRY(hipStreamSynchronize(g->stream)); *g->h_full = 0; }
  if (g->l1_cnt) HIP_TRY(hipMemsetAsync(g->l1_cnt, 0, (size_t)g->nsets * g->b1 * g->rep1 * 8, g->stream));
  if (g->l2_cnt) HIP_TRY(hipMemsetAsync(g->l2_cnt, 0, (size_t)g->l2_regions * g->subs_per_bin * 8, g->stream));
  if (g->d_readstrt) HIP_TRY(hipMemsetAsync(g->d_readstrt, 0xff, g->t.nslots * 8, g->stream));
  g->pending = g->pending_l2 = 0;  // buffered tuples are discarded with the table
  g->idle_next = 0; g->idle_base = 0; g->idle_mark.clear();
  sets_release(g);
  snap_push(g, true);
  return MCX_OK;
}

extern "C" int mcx_graph_capacity(const mcx_graph *g, uint64_t *slots, uint64_t *bytes)
{
  if (g && g->as_group) {
    uint64_t s_ = 0, b_ = 0;
    for (int i = 0; i < grp_n(g->as_group); i++) { uint64_t s1 = 0, b1 = 0; mcx_graph_capacity(grp_part(g->as_group, i), &s1, &b1); s_ += s1; b_ += b1; }
    if (slots) *slots = s_;
    if (bytes) *bytes = b_;
    return MCX_OK;
  }
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (slots) *slots = g->t.nslots;
  if (bytes) *bytes = g->table_bytes;
  return MCX_OK;
}

extern "C" void *mcx_graph_stream(mcx_graph *g) { return !g ? nullptr : g->as_group ? (void *)grp_part(g->as_group, 0)->stream : (void *)g->stream; }

// ---------------------------------------------------------------------------
// kernel dispatch
// ---------------------------------------------------------------------------
struct StreamLaunch {
  const uint8_t *stream;
  uint64_t nbytes, pos_lo, pos_hi;
  unsigned char *flag;
  const uint32_t *code = nullptr;  // packed form of the stream (StreamArgs), `stream` is null then
  const uint16_t *inv = nullptr;
};

#define DISPATCH_WC(g, F, ...)                                                        \
  do {                                                                                \
    if ((g)->W == 1) { if ((g)->ncols == 1) F<1, true>(__VA_ARGS__); else F<1, false>(__VA_ARGS__); } \
    else { if ((g)->ncols == 1) F<2, true>(__VA_ARGS__); else F<2, false>(__VA_ARGS__); }            \
  } while (0)

// ... for what exists for keys of three and four words as well (k = 65 .. 127: the fused insert, record load, export)
#define DISPATCH_WC4(g, F, ...)                                                       \
  do {                                                                                \
    const bool one_ = (g)->ncols == 1;                                                \
    switch ((g)->W) {                                                                 \
      case 1: if (one_) F<1, true>(__VA_ARGS__); else F<1, false>(__VA_ARGS__); break; \
      case 2: if (one_) F<2, true>(__VA_ARGS__); else F<2, false>(__VA_ARGS__); break; \
      case 3: if (one_) F<3, true>(__VA_ARGS__); else F<3, false>(__VA_ARGS__); break; \
      default: if (one_) F<4, true>(__VA_ARGS__); else F<4, false>(__VA_ARGS__); break; \
    }                                                                                 \
  } while (0)
#define LAUNCH_W4(W_, KERNEL, ...)                                       \
  do {                                                                   \
    switch (W_) {                                                        \
      case 1: hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__); break;       \
      case 2: hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__); break;       \
      case 3: hipLaunchKernelGGL((KERNEL<3>), __VA_ARGS__); break;       \
      default: hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__); break;      \
    }                                                                    \
  } while (0)
#define NO_WIDE(g, what) do { if ((g) && (g)->W > 2) return fail(MCX_ERR_ARG, what " is not available for k > 63"); } while (0)

// optional per-kernel timing with HIP events on the handle's stream
struct SpanGuard {
  mcx_graph *g; size_t idx; bool on;
  SpanGuard(mcx_graph *g_, const char *name) : g(g_), idx(0), on(g_->profile) {
    if (!on) return;
    mcx_graph::Span sp{name, nullptr, nullptr};
    (void)hipEventCreate(&sp.a); (void)hipEventCreate(&sp.b);
    (void)hipEventRecord(sp.a, g->stream);
    idx = g->spans.size();
    g->spans.push_back(sp);
  }
  ~SpanGuard() { if (on) (void)hipEventRecord(g->spans[idx].b, g->stream); }
};

template <class K> static void allow_lds(K kernel, size_t bytes)
{
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static StreamArgs make_args(mcx_graph *g, const StreamLaunch &L)
{
  StreamArgs a;
  a.stream = L.stream; a.code = L.code; a.inv = L.inv; a.nbytes = L.nbytes; a.pos_lo = L.pos_lo; a.pos_hi = L.pos_hi;
  a.tile0 = L.pos_lo / kTile;
  a.ntiles = (L.pos_hi + kTile - 1) / kTile;
  a.k = g->k; a.ctr = g->d_ctr; a.flag = L.flag;
  return a;
}

template <int W, bool ONECOL> static void launch_direct_t(mcx_graph *g, const StreamLaunch &L, int colour)
{
  const StreamArgs a = make_args(g, L);
  const uint64_t nt = a.ntiles > a.tile0 ? a.ntiles - a.tile0 : 0;
  if (!nt) return;
  InsertSink<W, ONECOL> is{g->t, (uint32_t)colour};
  SpanGuard sp(g, "k_stream");
  const dim3 grid((unsigned)std::min<uint64_t>(nt, (uint64_t)g->grid));
  if (a.code) hipLaunchKernelGGL((k_stream<W, ONECOL, true>), grid, dim3(kThreads), 0, g->stream, a, is);
  else hipLaunchKernelGGL((k_stream<W, ONECOL, false>), grid, dim3(kThreads), 0, g->stream, a, is);
}

template <int W, bool ONECOL, bool FULL, int SH, bool PK>
static void launch_bin_stream_pk(mcx_graph *g, const StreamArgs &a, uint64_t nt, int colour, BinSpec bs, BinOut out)
{
  InsertSink<W, ONECOL> is{g->t, (uint32_t)colour};
  static bool once_dev[64] = {false};  // per device: the attribute belongs to the function on one device
  bool &once = once_dev[g->device & 63];
  if (!once) {
    allow_lds(k_stream_bin<W, ONECOL, 512, FULL, SH, PK>, sizeof(BinLds<W, 512, FULL>));
    allow_lds(k_stream_bin<W, ONECOL, 1024, FULL, SH, PK>, sizeof(BinLds<W, 1024, FULL>));
    allow_lds(k_stream_bin<W, ONECOL, kMaxBins, FULL, SH, PK>, sizeof(BinLds<W, kMaxBins, FULL>));
    once = true;
  }
  SpanGuard sp(g, "k_stream_bin");
  // region bins of an unsharded one-word table: 512-thread blocks, two tiles sorted as one (MCX_STREAM_T=256: the old geometry)
  if constexpr (W == 1 && !FULL && SH == 0) {
    // MCX_STREAM_FC=1: fixed-capacity LDS segments instead of the LDS sort (mcx_streamfc.h; an experiment of round 6 that
    // lost to the sort -- profiles/r06_experiments.md -- and is kept compiled for the record: 30.3 against 22.0 ms at C2)
    static const int fc_cap = [] { const char *e = getenv("MCX_STREAM_FC"); return e ? atoi(e) : 0; }();
    if (fc_cap && bs.nlocal <= 512) {
      const dim3 gridf((unsigned)std::min<uint64_t>(nt, (uint64_t)(g->grid_stream ? g->grid_stream : g->grid)));
#define MCX_FC_LAUNCH(CAPV, PRIVV) { \
        static bool once_f[64] = {false}; \
        if (!once_f[g->device & 63]) { allow_lds(k_stream_fc<ONECOL, PK, CAPV, PRIVV>, sizeof(FcLds<CAPV>)); once_f[g->device & 63] = true; } \
        hipLaunchKernelGGL((k_stream_fc<ONECOL, PK, CAPV, PRIVV>), gridf, dim3(kThreads), sizeof(FcLds<CAPV>), g->stream, a, bs, out, is); }
      // (private replicas: one per block of the launch -- MCX_REP1 >= the grid)
      if (bs.rep >= gridf.x && bs.rep > 8) MCX_FC_LAUNCH(10, true)
      else MCX_FC_LAUNCH(10, false)
#undef MCX_FC_LAUNCH
      return;
    }
    static const bool wide = [] { const char *e = getenv("MCX_STREAM_T"); return !e || atoi(e) == 512; }();
    if (wide && bs.nlocal <= 512) {
      using GeoW = Geo<512, 512 * kPosPerLane>;
      static bool once_w[64] = {false};
      if (!once_w[g->device & 63]) { allow_lds(k_stream_bin<W, ONECOL, 512, FULL, SH, PK, 512>, sizeof(BinLds<W, 512, FULL, GeoW>)); once_w[g->device & 63] = true; }
      // (8 blocks per CU for 2 resident: 22.2-22.3 ms per 6 G occurrences at C2 against 22.8 with 4 and 23.5 with 2 --
      // a finer tail; 64 K blocks: 26.0)
      const dim3 gridw((unsigned)std::min<uint64_t>((nt + 1) / 2, (uint64_t)(g->grid_stream ? g->grid_stream : g->grid)));
      hipLaunchKernelGGL((k_stream_bin<W, ONECOL, 512, FULL, SH, PK, 512>), gridw, dim3(512), sizeof(BinLds<W, 512, FULL, GeoW>), g->stream, a, bs, out, is);
      return;
    }
  }
  const dim3 grid((unsigned)std::min<uint64_t>(nt, (uint64_t)(g->grid_stream ? g->grid_stream : g->grid)));
  // the histogram capacity sets the LDS footprint and with it the blocks per CU: 512 and 1024 bins
  // leave room for 4 blocks (W=1), 2048 for 2
  if (bs.nlocal <= 512)
    hipLaunchKernelGGL((k_stream_bin<W, ONECOL, 512, FULL, SH, PK>), grid, dim3(kThreads), sizeof(BinLds<W, 512, FULL>), g->stream, a, bs, out, is);
  else if (bs.nlocal <= 1024)
    hipLaunchKernelGGL((k_stream_bin<W, ONECOL, 1024, FULL, SH, PK>), grid, dim3(kThreads), sizeof(BinLds<W, 1024, FULL>), g->stream, a, bs, out, is);
  else
    hipLaunchKernelGGL((k_stream_bin<W, ONECOL, kMaxBins, FULL, SH, PK>), grid, dim3(kThreads), sizeof(BinLds<W, kMaxBins, FULL>), g->stream, a, bs, out, is);
}

template <int W, bool ONECOL, bool FULL, int SH>
static void launch_bin_stream_t(mcx_graph *g, const StreamLaunch &L, int colour, BinSpec bs, BinOut out)
{
  const StreamArgs a = make_args(g, L);
  const uint64_t nt = a.ntiles > a.tile0 ? a.ntiles - a.tile0 : 0;
  if (!nt) return;
  if (!FULL && a.code) launch_bin_stream_pk<W, ONECOL, FULL, SH, !FULL>(g, a, nt, colour, bs, out);  // (owner bins of full tuples only exist for ASCII streams)
  else launch_bin_stream_pk<W, ONECOL, FULL, SH, false>(g, a, nt, colour, bs, out);
}

template <int W, bool ONECOL, bool IN_FULL, bool SHARD>
static void launch_bin_tuples_t(mcx_graph *g, TupleIn in, int colour, BinSpec bs, BinOut out)
{
  // the split of one-word tuples into <= 512 sub-table bins runs with 512-thread blocks and tiles of
  // 8192 tuples (MCX_SPLIT_T=256 for the old geometry)
  constexpr bool kWide = W == 1 && !IN_FULL;
  static const bool wide = kWide && [] { const char *e = getenv("MCX_SPLIT_T"); return !e || atoi(e) == 512; }();
  const bool use_wide = wide && bs.nlocal <= 1024;
  // ... and into up to 2048 bins (tables beyond 2^32 slots) with ONE 1024-thread block per CU and tiles of 16384 tuples:
  // 8 tuples per bin and tile where the 256-thread kernel has 2 (C2-stress, 2^33 slots: 56 -> 36 ms per 6 G tuples, round 4)
  const bool use_huge = wide && !use_wide && bs.nlocal <= (uint32_t)kMaxBins;
  const uint64_t tile = use_huge ? 1024 * 16 : use_wide ? 512 * 16 : kTile;
  const uint64_t nchunks = (in.seg_cap + tile - 1) / tile * in.nseg;
  if (!nchunks) return;
  InsertSink<W, ONECOL> is{g->t, (uint32_t)colour};
  using GeoW = Geo<512, 512 * 16>;
  static bool once_dev[64] = {false};  // per device: the attribute belongs to the function on one device
  bool &once = once_dev[g->device & 63];
  if (!once) {
    allow_lds(k_tuples_bin<W, ONECOL, 512, IN_FULL, SHARD>, sizeof(BinLds<W, 512, false>));
    allow_lds(k_tuples_bin<W, ONECOL, 1024, IN_FULL, SHARD>, sizeof(BinLds<W, 1024, false>));
    allow_lds(k_tuples_bin<W, ONECOL, kMaxBins, IN_FULL, SHARD>, sizeof(BinLds<W, kMaxBins, false>));
    if constexpr (kWide) allow_lds(k_tuples_bin<W, ONECOL, 512, IN_FULL, SHARD, 512>, sizeof(BinLds<W, 512, false, GeoW>));
    if constexpr (kWide) allow_lds(k_tuples_bin<W, ONECOL, 1024, IN_FULL, SHARD, 512>, sizeof(BinLds<W, 1024, false, GeoW>));
    if constexpr (kWide) allow_lds(k_tuples_bin<W, ONECOL, kMaxBins, IN_FULL, SHARD, 1024>, sizeof(BinLds<W, kMaxBins, false, Geo<1024, 1024 * 16>>));
    once = true;
  }
  SpanGuard sp(g, "k_tuples_bin");
  const uint64_t gmax = (uint64_t)(g->grid_split ? g->grid_split : g->grid * 4);
  if constexpr (kWide) {
    if (use_huge) {
      using GeoH = Geo<1024, 1024 * 16>;
      const dim3 gridh((unsigned)std::min<uint64_t>(nchunks, g->grid_split ? gmax : gmax / 4));
      hipLaunchKernelGGL((k_tuples_bin<W, ONECOL, kMaxBins, IN_FULL, SHARD, 1024>), gridh, dim3(1024), sizeof(BinLds<W, kMaxBins, false, GeoH>), g->stream, in, bs, out, is, g->d_ctr);
      return;
    }
    if (use_wide) {
      const dim3 gridw((unsigned)std::min<uint64_t>(nchunks, g->grid_split ? gmax : gmax / 2));
      if (bs.nlocal <= 512)
        hipLaunchKernelGGL((k_tuples_bin<W, ONECOL, 512, IN_FULL, SHARD, 512>), gridw, dim3(512), sizeof(BinLds<W, 512, false, GeoW>), g->stream, in, bs, out, is, g->d_ctr);
      else
        hipLaunchKernelGGL((k_tuples_bin<W, ONECOL, 1024, IN_FULL, SHARD, 512>), gridw, dim3(512), sizeof(BinLds<W, 1024, false, GeoW>), g->stream, in, bs, out, is, g->d_ctr);
      return;
    }
  }
  const dim3 grid((unsigned)std::min<uint64_t>(nchunks, gmax));
  if (bs.nlocal <= 512)
    hipLaunchKernelGGL((k_tuples_bin<W, ONECOL, 512, IN_FULL, SHARD>), grid, dim3(kThreads), sizeof(BinLds<W, 512, false>), g->stream, in, bs, out, is, g->d_ctr);
  else if (bs.nlocal <= 1024)
    hipLaunchKernelGGL((k_tuples_bin<W, ONECOL, 1024, IN_FULL, SHARD>), grid, dim3(kThreads), sizeof(BinLds<W, 1024, false>), g->stream, in, bs, out, is, g->d_ctr);
  else
    hipLaunchKernelGGL((k_tuples_bin<W, ONECOL, kMaxBins, IN_FULL, SHARD>), grid, dim3(kThreads), sizeof(BinLds<W, kMaxBins, false>), g->stream, in, bs, out, is, g->d_ctr);
}
// keys of other shards can only show up in a sharded table (lbo > 0): the check is compiled apart
template <int W, bool ONECOL> static void launch_bin_region_stream(mcx_graph *g, const StreamLaunch &L, int colour, BinSpec bs, BinOut out)
{
  if (g->t.lbo) launch_bin_stream_t<W, ONECOL, false, 1>(g, L, colour, bs, out);
  else launch_bin_stream_t<W, ONECOL, false, 0>(g, L, colour, bs, out);
}
template <int W, bool ONECOL> static void launch_split_regions(mcx_graph *g, TupleIn in, int colour, BinSpec bs, BinOut out)
{ launch_bin_tuples_t<W, ONECOL, false, false>(g, in, colour, bs, out); }
template <int W, bool ONECOL> static void launch_bin_received(mcx_graph *g, TupleIn in, int colour, BinSpec bs, BinOut out)
{
  if (g->t.lbo) launch_bin_tuples_t<W, ONECOL, true, true>(g, in, colour, bs, out);
  else launch_bin_tuples_t<W, ONECOL, true, false>(g, in, colour, bs, out);
}

// sub-table bin `off` and the ones behind it in its half (the two halves of the flush overlap may be two allocations)
static uint64_t *l2_ptr(const mcx_graph *g, uint64_t off)
{
  if (g->l2_hi && off >= g->l2_hi_off) return g->l2_hi + (off - g->l2_hi_off) * g->cap2 * g->W;
  return g->l2_keys + off * g->cap2 * g->W;
}

template <int W, bool ONECOL> static void launch_lds_insert_t(mcx_graph *g, int colour, uint32_t sub0, uint32_t nsub)
{
  const size_t lds = Sub<W>::kSlots * (W + 1) * 8 + LdsQueue<W>::kTuples * 8 * W;
  static bool once_dev[64] = {false};  // per device: the attribute belongs to the function on one device
  bool &once = once_dev[g->device & 63];
  if (!once) { allow_lds(k_lds_insert<W, ONECOL>, lds); once = true; }
  BinOut bins{l2_ptr(g, g->l2_off), nullptr, g->l2_cnt + g->l2_off, g->cap2, nullptr, nullptr, nullptr, 0};
  SpanGuard sp(g, "k_lds_insert");
  hipLaunchKernelGGL((k_lds_insert<W, ONECOL>), dim3((unsigned)std::min<uint64_t>(nsub, (uint64_t)(g->grid_insert ? g->grid_insert : g->grid * 4))),
                     dim3(LdsCfg<W>::kThreads), lds, g->stream, g->t, (uint32_t)colour, bins, sub0, nsub, g->d_ctr);
}

template <int W, bool ONECOL>
static void launch_insert_tuples_t(mcx_graph *g, int colour, const uint64_t *keys, const uint8_t *edges, uint64_t n, uint32_t only_own = 0)
{
  const int grid = (int)std::min<uint64_t>((n + kThreads * kBatch - 1) / (kThreads * kBatch), (uint64_t)g->grid);
  InsertSink<W, ONECOL> s{g->t, (uint32_t)colour};
  SpanGuard sp(g, "k_insert_tuples");
  hipLaunchKernelGGL((k_insert_tuples<W, ONECOL>), dim3(grid), dim3(kThreads), 0, g->stream, s, keys, edges, n, g->d_ctr, only_own);
}

// ---- deferred path bookkeeping -------------------------------------------------------------
static void free_defer(mcx_graph *g)
{
  (void)hipFree(g->l1_keys); (void)hipFree(g->l1_cnt);
  (void)hipFree(g->l2_base); (void)hipFree(g->l2_hi); (void)hipFree(g->l2_cnt);
  g->l1_keys = g->l2_keys = g->l2_base = g->l2_hi = nullptr; g->l1_cnt = g->l2_cnt = nullptr;
  g->cap1 = g->cap2 = 0;
  g->l2_regions = 0;
  g->set_colour.clear(); g->set_pending.clear();
}

// The sub-table (L2) bins cover `regions` regions at a time.  A flush walks the L1 bins in groups
// of that many regions: split the group by sub-table, apply it in LDS, reuse the same bins for the
// next group -- so the bins take a fraction (group / regions) of what bins for the whole table
// would (80 GB for 8 G occurrences per flush on the bench shape).  Blocks received from other shards are split on arrival
// (mcx_graph_add_segments_dev), which needs bins for all regions at once.
// The write pattern of the split without the split: every block appends 128-byte runs to the sub-table bins of its
// region, the blocks of a region interleaved.  Where an allocation of sub-table bins happens to lie in HBM decides
// whether the split runs at 22.0 or at 19.5-20.5 ms per 6 G occurrences (round 6: constant for an allocation whatever
// the offset inside it, different from one allocation to the next, profiles/r06_experiments.md), and this probe's
// time follows it (5.3 ms where the split takes 20.5, 6.05 where it takes 22.2): ensure_l2 uses it to choose.
__global__ __launch_bounds__(512) void k_l2_probe(uint64_t *keys, uint64_t cap, uint32_t nregions, uint32_t spb, uint32_t iters,
                                                  uint32_t jit_mask = 0 /* experiment: every bin starts a pseudo-random (hash & mask) words late */)
{
  const uint32_t region = blockIdx.x % nregions, blk = blockIdx.x / nregions, nblk = gridDim.x / nregions;
  const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63u;
  for (uint32_t it = 0; it < iters; it++)
    for (uint32_t k = 0; k < spb / 32u; k++) {
      const uint32_t front = w * 4u + (l >> 4) + 32u * k;
      const uint32_t seg = region * spb + front;
      const uint64_t jit = ((seg * 0x9E3779B1u) >> 8) & jit_mask & ~15u;  // (whole 128-byte lines)
      const uint64_t at = (uint64_t)seg * cap + jit + ((uint64_t)it * nblk + blk) * 16u + (l & 15u);
      if (jit + ((uint64_t)it * nblk + blk) * 16u + 16u <= cap) keys[at] = at;
    }
}

// One allocation of `bytes` for `nreg` regions of sub-table bins, or the best of `tries` by the probe (losers are held
// until the choice is made -- a freed one would be handed out again -- as far as HBM has room; `taken`: allocations
// the caller holds meanwhile).  *ms_out: the winner's probe time (0: not probed).
static uint64_t *l2_place(mcx_graph *g, size_t bytes, uint32_t nreg, int tries, float *ms_out)
{
  std::vector<uint64_t *> cand;
  std::vector<float> ms;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (tries > 1 && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) { (void)hipGetLastError(); tries = 1; }
  const uint32_t iters = (uint32_t)std::max<uint64_t>(8, g->cap2 * g->W / 1024);  // (the whole depth of the bins)
  for (int i = 0; i < std::max(tries, 1); i++) {
    if (i > 0) {
      size_t fr = 0, tot = 0;
      if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < bytes + (size_t)(tot / 16)) { (void)hipGetLastError(); break; }
    }
    uint64_t *p = nullptr;
    if (hipMalloc((void **)&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
    cand.push_back(p);
    float t = 0.f;
    if (tries > 1)
      for (int rep = 0; rep < 2; rep++) {  // (the second run counts)
        (void)hipEventRecord(e0, g->stream);
        hipLaunchKernelGGL(k_l2_probe, dim3(nreg * 64), dim3(512), 0, g->stream, p, g->cap2 * g->W, nreg, g->subs_per_bin, iters);
        (void)hipEventRecord(e1, g->stream);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) { (void)hipGetLastError(); t = 0.f; }
      }
    ms.push_back(t);
    // (two kinds of places: ~1.5 ms and ~2.25 ms at C2's size.  One that is 30 % faster than the slowest seen is of
    // the fast kind: no need to look further)
    float worst = 0.f, best_t = 0.f;
    for (float x : ms) { worst = std::max(worst, x); if (x > 0.f && (best_t == 0.f || x < best_t)) best_t = x; }
    if (tries > 1 && best_t > 0.f && best_t < 0.70f * worst) break;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (cand.empty()) return nullptr;
  size_t best = 0;
  for (size_t i = 1; i < cand.size(); i++) if (ms[i] > 0.f && (ms[best] <= 0.f || ms[i] < ms[best])) best = i;
  for (size_t i = 0; i < cand.size(); i++) if (i != best) (void)hipFree(cand[i]);
  if (getenv("MCX_TIMING") && cand.size() > 1) {
    fprintf(stderr, "[timing] sub-table bins (%u regions): %zu placements probed:", nreg, cand.size());
    for (size_t i = 0; i < cand.size(); i++) fprintf(stderr, " %.3f%s", ms[i], i == best ? "*" : "");
    fprintf(stderr, " ms\n");
  }
  if (ms_out) *ms_out = ms[best];
  return cand[best];
}

static uint32_t flush_group(const mcx_graph *g);
// whole: the caller walks the bins as ONE array (split on arrival): no halves
static int ensure_l2(mcx_graph *g, uint32_t regions, bool whole = false)
{
  if (g->l2_keys && g->l2_regions >= regions && !(g->l2_hi && (whole || (regions > g->l2_regions / 2 && regions != g->l2_regions)))) return MCX_OK;
  if (g->pending_l2) return fail(MCX_ERR_ARG, "internal: sub-table bins resized while in use");
  HIP_TRY(hipStreamSynchronize(g->stream));
  (void)hipFree(g->l2_base); (void)hipFree(g->l2_hi); (void)hipFree(g->l2_cnt);
  g->l2_keys = g->l2_base = g->l2_hi = nullptr; g->l2_cnt = nullptr; g->l2_regions = 0; g->l2_hi_off = 0;
  const uint64_t nb = (uint64_t)regions * g->subs_per_bin;
  int tries = g->place_tries;
  { const char *e = getenv("MCX_PLACE_BINS"); if (e) tries = atoi(e); }
  if (g->subs_per_bin % 32 || regions < 8) tries = 1;  // (the probe's pattern needs whole rows of fronts)
  // Placed bins for the two halves of the flush overlap: each half an allocation of its own, chosen by itself (a place
  // in HBM that is good for 16 GB at a stretch is rarer than one that is good for 8).  Only when the caller asked for
  // exactly the two groups of a flush (never for the whole-table bins of the sharded receive path: those are walked as
  // one array).
  const uint32_t G = flush_group(g);
  const bool halves = !whole && tries > 1 && regions == 2 * G && !g->t.lbo && !g->group;
  if (halves) {
    const size_t hbytes = (nb / 2) * g->cap2 * 8 * g->W;
    g->l2_base = l2_place(g, hbytes, G, tries, nullptr);
    if (g->l2_base) g->l2_hi = l2_place(g, hbytes, G, tries, nullptr);
    if (g->l2_base && !g->l2_hi) { (void)hipFree(g->l2_base); g->l2_base = nullptr; }
    g->l2_hi_off = nb / 2;
  } else {
    g->l2_base = l2_place(g, nb * g->cap2 * 8 * g->W, std::min<uint32_t>(regions, 32), tries, nullptr);
  }
  if (!g->l2_base || hipMalloc((void **)&g->l2_cnt, nb * 8) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(g->l2_base); (void)hipFree(g->l2_hi); g->l2_base = g->l2_hi = nullptr;
    return fail(MCX_ERR_NOMEM, "out of device memory for the sub-table bins (%llu regions)", (unsigned long long)regions);
  }
  HIP_TRY(hipMemsetAsync(g->l2_cnt, 0, nb * 8, g->stream));
  g->l2_keys = g->l2_base;
  g->l2_regions = regions;
  return MCX_OK;
}

// Flush overlap (default on since round 3; MCX_FLUSH_OVERLAP=0 switches it off): the LDS insert of
// region group g runs on a second stream beside the split of group g + 1 (two halves of sub-table
// bins).  Worth 1.3-2.8 % of the C2 step (tools/sweep.sh, round 3); the per-kernel durations of the
// two kernels then include each other's interference (a co-running insert launch takes about twice
// as long as alone), so kernel-by-kernel profiles are taken with it off (tools/prof.sh).
static bool flush_overlap(const mcx_graph *g)
{
  static const bool on = [] { const char *e = getenv("MCX_FLUSH_OVERLAP"); return !e || atoi(e) != 0; }();
  return g->overlap < 0 ? on : g->overlap != 0;  // (mcx_graph_configure("flush_overlap", 0 | 1) overrides the environment)
}

// regions per flush group: enough sub-tables for one full wave of LDS-insert workgroups, few
// enough for the split output to stay cache resident
static uint32_t flush_group(const mcx_graph *g)
{
  uint32_t G = g->flush_regions;
  if (!G) {
    const char *e = getenv("MCX_FLUSH_REGIONS");
    G = e ? (uint32_t)strtoul(e, nullptr, 10) : 0;
  }
  // measured on the bench shape (512 regions x 512 sub-tables, profiles/r01g): groups of 32..512
  // regions run at the same speed (1, 2, 4 regions per step: launch tails dominate, the hoped-for
  // cache residency of one region's split does not pay); 16 K sub-tables per step = 32 waves of
  // insert workgroups, and the bins take 1/16 of what bins for the whole table would
  // ... and never fewer than 32 regions: with 8 the XCD-aware chunk order of the split leaves one
  // region per XCD and its blocks all reserve from the same counters (4x slower, C2-stress)
  if (!G) G = std::max<uint32_t>(32u, 16384u / std::max<uint32_t>(1u, g->subs_per_bin));
  return std::min<uint32_t>(G, g->b1);
}

static int ensure_defer(mcx_graph *g)
{
  if (g->l1_keys) return MCX_OK;
  g->nsub = (uint32_t)(g->t.nmain >> sub_shift_for_words(g->W));
  g->b1 = 1u << g->t.lb1;        // L1 bins = regions of the quotient hash
  g->subs_per_bin = g->t.spb;
  // packed tuples need 2k - lb1 <= 56 quotient bits in the top word, and the histograms must fit
  if (g->t.lb1 < 6 || g->b1 > (uint32_t)kMaxBins || g->subs_per_bin > (uint32_t)kMaxBins) { g->defer = false; return MCX_OK; }
  uint64_t tcap = g->defer_tuples;
  if (!tcap) {
    const char *e = getenv("MCX_DEFER_TUPLES");
    if (e) {
      tcap = strtoull(e, nullptr, 10);
    } else {
      // Default flush size.  Every flush streams the whole table through LDS once (2 x 16 B per
      // slot), so the more occurrences a flush applies the better: up to 64 per slot, within 40 %
      // of the HBM that is free once the table stands (the part has 288 GB: a 16 GiB table leaves
      // room for 8 G occurrences per flush), never below 1 M.
      size_t fr = 0, tot = 0;
      (void)hipMemGetInfo(&fr, &tot);
      // L1 segments (x 1.06) + the group's sub-table bins; a shard of a multi-GPU table splits on arrival
      // and needs sub-table bins for the whole table (x 1.25) as well
      const uint64_t per_tuple = 8ull * g->W * (g->t.lbo ? 240 : 118) / 100 + 1;
      const uint64_t by_mem = (uint64_t)((double)fr * 0.40) / per_tuple;
      tcap = std::max<uint64_t>(std::min<uint64_t>(std::min<uint64_t>(64 * g->t.nslots, by_mem), 1ull << 33), 1ull << 20);
    }
  }
  // Allocate the bins; if HBM is short halve the flush size (down to 1M occurrences), and if even
  // that does not fit build with the direct path (same graph, just slower).
  for (;; tcap /= 2) {
    if (tcap < (1ull << 20)) { g->defer = false; g->defer_tuples = 0; (void)hipGetLastError(); return MCX_OK; }
    g->defer_tuples = tcap;
    g->nsets = g->ncols > 1 ? kL1Sets : 1;
    if (const char *e = getenv("MCX_L1_SETS")) { const int v = atoi(e); if (v >= 1 && v <= 32) g->nsets = (uint32_t)v; }  // tests / experiments
    g->set_cap = tcap / g->nsets;
    // (32 replicas only where a segment still holds many tiles: the split reads segments tile by tile)
    if (!g->rep1_forced) g->rep1 = g->nsets != 1 ? 8 : g->set_cap / g->b1 / 64 >= (1u << 16) ? 64 : g->set_cap / g->b1 / 32 >= (1u << 16) ? 32 : 8;
    g->cap1 = (uint64_t)((double)g->set_cap / g->b1 / g->rep1 * (g->b1 == 1 ? 1.02 : 1.06)) + (g->nsets > 1 ? 2048 : 8192);
    static const double cap2_slack = [] { const char *e = getenv("MCX_CAP2_SLACK"); const double v = e ? atof(e) : 0; return v >= 1.0 && v <= 4.0 ? v : 1.25; }();  // experiments
    g->cap2 = g->nsub == 1 ? tcap + 8192 : (uint64_t)((double)tcap / g->nsub * cap2_slack) + 1024;
    g->cap1 = (g->cap1 + 1) & ~1ull;  // even: every segment starts 16-byte aligned (vector loads)
    g->cap2 = (g->cap2 + 1) & ~1ull;
    if (g->cap1 >= 0xFFFFFFFFull || g->cap2 >= 0xFFFFFFFFull) continue;
    const uint64_t n1 = (uint64_t)g->nsets * g->b1 * g->rep1 * g->cap1;
    const bool ok = hipMalloc((void **)&g->l1_keys, n1 * 8 * g->W) == hipSuccess &&
                    hipMalloc((void **)&g->l1_cnt, (size_t)g->nsets * g->b1 * g->rep1 * 8) == hipSuccess &&
                    ensure_l2(g, flush_overlap(g) ? std::min<uint32_t>(2 * flush_group(g), g->b1) : flush_group(g)) == MCX_OK;
    if (ok) break;
    free_defer(g);
    (void)hipGetLastError();  // clear the sticky out-of-memory error
  }
  HIP_TRY(hipMemsetAsync(g->l1_cnt, 0, (size_t)g->nsets * g->b1 * g->rep1 * 8, g->stream));
  g->set_colour.assign(g->nsets, -1);
  g->set_pending.assign(g->nsets, 0);
  return MCX_OK;
}

// the L1 bins of set `set` as the output of a binning kernel
static BinOut l1_out(const mcx_graph *g, int set)
{
  const uint64_t segs = (uint64_t)g->b1 * g->rep1;
  return BinOut{g->l1_keys + (uint64_t)set * segs * g->cap1 * g->W, nullptr, g->l1_cnt + (uint64_t)set * segs, g->cap1, nullptr, nullptr, nullptr, 0};
}

// The second stream of the flush overlap.  Creating a stream takes tens of milliseconds on this runtime (60 ms each in the
// rocprofv3 HIP trace of a CLI build, round 5): "prepare" does it while the first batch is being parsed, so that the
// closing flush of a host-fed build does not pay for it.
static int ensure_flush_stream(mcx_graph *g)
{
  if (g->stream2) return MCX_OK;
  HIP_TRY(hipStreamCreateWithFlags(&g->stream2, hipStreamNonBlocking));
  for (int i = 0; i < 2; i++) {
    if (!g->ev_split[i]) HIP_TRY(hipEventCreateWithFlags(&g->ev_split[i], hipEventDisableTiming));
    if (!g->ev_ins[i]) HIP_TRY(hipEventCreateWithFlags(&g->ev_ins[i], hipEventDisableTiming));
  }
  return MCX_OK;
}

// Split the L1 bins by sub-table and let one workgroup per sub-table apply its tuples in LDS,
// one group of regions at a time.
static int flush_deferred(mcx_graph *g)
{
  if (!g->pending && !g->pending_l2) {
    // Nothing buffered -- but a set may still be bound to a colour: the settled-launch bookkeeping (snap_poll) takes
    // what a launch did not yield off the books, down to zero for launches that yielded no k-mer at all (reads
    // shorter than k, all-N batches).  The bins are empty then, so the sets are free again; without this a later
    // request larger than a set (accepted only for a FREE set) found neither room nor a free set after "the flush"
    // ("internal: no L1 bin set after a flush": seed 210 of the round-4 soak).
    sets_release(g);
    if (!g->idle_mark.empty()) {  // idle flushes have emptied every region group: the books start afresh, as after a whole flush
      g->idle_next = 0; g->idle_base = 0; g->idle_mark.clear();
      snap_push(g, true);
    }
    return MCX_OK;
  }
  HIP_TRY(hipSetDevice(g->device));
  // tuples that were split on arrival occupy bins of all regions: everything goes through them
  const uint32_t G = g->pending_l2 ? g->b1 : std::min(flush_group(g), g->l2_regions);
  if (g->pending_l2 && g->l2_regions < g->b1) return fail(MCX_ERR_ARG, "internal: split tuples without bins");
  // Flush overlap (MCX_FLUSH_OVERLAP=1, experiment): the split is HBM-bound, the LDS insert bound by
  // instruction issue; with two halves of sub-table bins the insert of group g runs on a second
  // stream beside the split of group g + 1 (grids sized to share the CUs: grid_split / grid_insert).
  const bool overlap = flush_overlap(g) && g->pending && !g->pending_l2 && g->l2_regions >= 2 * G && G < g->b1;
  hipStream_t s1 = g->stream;
  if (overlap) {
    int rc2 = ensure_flush_stream(g);
    if (rc2 != MCX_OK) return rc2;
    HIP_TRY(hipEventRecord(g->ev_split[0], s1));  // the second stream starts behind everything queued so far
    HIP_TRY(hipStreamWaitEvent(g->stream2, g->ev_split[0], 0));
  }
  // One pass over the table per colour that has tuples pending: the colour whose tuples already sit
  // in the sub-table bins (split on arrival) first -- that empties them --, then the colours of the
  // L1 sets in the order in which they first took a set.
  std::vector<int> colours;
  if (g->pending_l2) colours.push_back(g->pending_colour);
  for (uint32_t s = 0; s < g->nsets && g->pending; s++) {
    const int c = g->set_colour[s];
    if (c >= 0 && g->set_pending[s] && std::find(colours.begin(), colours.end(), c) == colours.end()) colours.push_back(c);
  }
  uint32_t gi = 0;
  for (size_t ci = 0; ci < colours.size(); ci++) {
    const int colour = colours[ci];
    TupleIn in{};
    uint32_t nmine = 0;  // sets of this colour
    for (uint32_t s = 0; s < g->nsets && g->pending; s++)
      if (g->set_colour[s] == colour && g->set_pending[s]) in.set_map[nmine++] = (uint8_t)s;
    for (uint32_t r0 = 0; r0 < g->b1; r0 += G, gi++) {
      const uint32_t ng = std::min(G, g->b1 - r0);
      const int hb = overlap ? (int)(gi & 1u) : 0;
      g->l2_off = (uint64_t)hb * G * g->subs_per_bin;
      if (overlap && gi >= 2) HIP_TRY(hipStreamWaitEvent(s1, g->ev_ins[hb], 0));  // this half's previous insert has emptied it
      if (nmine) {
        // segment s = (replica q of the colour's sets, region r0 + s % ng): physical replica through set_map
        in.keys = g->l1_keys + (uint64_t)r0 * g->cap1 * g->W;
        in.edges = nullptr;
        in.counts = g->l1_cnt + r0;
        in.seg_cap = g->cap1;
        in.nseg = ng * g->rep1 * nmine;
        in.seg_group = ng;
        in.seg_stride = g->b1;
        in.set_rep = g->nsets > 1 ? g->rep1 : 0;
        BinSpec bs{BIN_SUBLOCAL, 0, g->subs_per_bin, 1, ng * g->subs_per_bin, ng, 0, r0};
        BinOut out{l2_ptr(g, g->l2_off), nullptr, g->l2_cnt + g->l2_off, g->cap2, nullptr, nullptr, nullptr, 0};
        DISPATCH_WC(g, launch_split_regions, g, in, colour, bs, out);
        HIP_TRY(hipGetLastError());
      }
      if (overlap) {
        HIP_TRY(hipEventRecord(g->ev_split[hb], s1));
        HIP_TRY(hipStreamWaitEvent(g->stream2, g->ev_split[hb], 0));
        g->stream = g->stream2;
      }
      DISPATCH_WC(g, launch_lds_insert_t, g, colour, r0 * g->subs_per_bin, ng * g->subs_per_bin);
      g->stream = s1;
      HIP_TRY(hipGetLastError());
      if (overlap) HIP_TRY(hipEventRecord(g->ev_ins[hb], g->stream2));
    }
  }
  g->l2_off = 0;
  if (overlap)
    for (int i = 0; i < 2; i++) HIP_TRY(hipStreamWaitEvent(s1, g->ev_ins[i], 0));
  if (g->pending) HIP_TRY(hipMemsetAsync(g->l1_cnt, 0, (size_t)g->nsets * g->b1 * g->rep1 * 8, g->stream));
  g->pending = 0;
  g->pending_l2 = 0;
  g->n_flushes++;
  g->idle_next = 0; g->idle_base = 0; g->idle_mark.clear();
  sets_release(g);
  snap_push(g, true);
  return MCX_OK;
}

// Make room for `ub` more tuples of `colour` in the L1 bins: -> the set that takes them.  A set
// that is bound to the colour and has room, else a free set, else a flush (which frees them all).
// `ub` beyond a set's capacity is accepted for an EMPTY set (callers that only know an upper bound
// of what a device-side fill holds): a segment that overflows falls back to the direct insert.
// settled launches (mcx_graph::snap_*): record the counters behind what has been enqueued so far
static void snap_push(mcx_graph *g, bool is_base, uint64_t ub, int set, int which)
{
  if (!g->h_snap) {
    if (hipHostMalloc((void **)&g->h_snap, sizeof(Counters) * mcx_graph::kSnap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); g->h_snap = nullptr; return; }
    for (uint32_t i = 0; i < mcx_graph::kSnap; i++)
      if (hipEventCreateWithFlags(&g->snap[i].ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        for (uint32_t j = 0; j < i; j++) { (void)hipEventDestroy(g->snap[j].ev); g->snap[j].ev = nullptr; }
        (void)hipHostFree(g->h_snap);
        g->h_snap = nullptr;
        return;
      }
  }
  if (is_base) {  // the bins are empty: what is in flight describes launches that no longer count
    g->snap_tail = g->snap_head;
    g->snap_base_known = false;
  }
  if (g->snap_head - g->snap_tail >= mcx_graph::kSnap) return;  // ring full: this launch's yield is charged to the next entry
  const uint32_t i = g->snap_head % mcx_graph::kSnap;
  if (hipMemcpyAsync(&g->h_snap[i], g->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
      hipEventRecord(g->snap[i].ev, g->stream) != hipSuccess) { (void)hipGetLastError(); return; }
  g->snap[i].ub = ub;
  g->snap[i].set = set;
  g->snap[i].which = which;
  g->snap[i].is_base = is_base;
  g->snap_head++;
}
static void snap_push(mcx_graph *g, bool is_base) { snap_push(g, is_base, 0, 0, 0); }

// ... and take what the settled launches did not yield off the books (never waits)
static void snap_poll(mcx_graph *g)
{
  if (!g->h_snap) return;
  while (g->snap_tail != g->snap_head) {
    const uint32_t i = g->snap_tail % mcx_graph::kSnap;
    if (hipEventQuery(g->snap[i].ev) != hipSuccess) { (void)hipGetLastError(); break; }
    const uint64_t v[2] = {g->h_snap[i].kmers, g->h_snap[i].binned};
    const mcx_graph::Snap &e = g->snap[i];
    if (e.is_base) {
      g->snap_base_known = true;
    } else if (g->snap_base_known && g->idle_mark.empty() && (size_t)e.set < g->set_pending.size()) {
      // (once an idle flush has marked a region group the books are kept in its units: flush_if_device_idle)
      // (other entries feed the counters too -- the direct path, launches the full ring did not snap: the yield is an
      // upper bound, the slack a lower one)
      const uint64_t yielded = std::min<uint64_t>(e.ub, v[e.which] - g->snap_last[e.which]);
      const uint64_t d = e.ub - yielded;
      g->pending -= std::min(g->pending, d);
      g->set_pending[e.set] -= std::min(g->set_pending[e.set], d);
    }
    g->snap_last[0] = v[0];
    g->snap_last[1] = v[1];
    g->snap_tail++;
  }
}

// Is there a bin set that takes `need` more occurrences of `colour`?
static bool defer_has_room(const mcx_graph *g, int colour, uint64_t need)
{
  for (uint32_t s = 0; s < g->nsets; s++)
    if (g->set_colour[s] < 0 || (g->set_colour[s] == colour && g->set_pending[s] + need <= g->set_cap)) return true;
  return false;
}
// A launch is booked with an upper bound (one occurrence per start position of its piece of stream; every position of a
// piece for an owner of exchange v3) and the excess comes off the books when the launch has settled (snap_poll).  The
// host runs ahead of the device, so a window can look full while most of it is such unsettled excess: before that costs
// a flush -- a pass over the whole table -- wait for the oldest launches, one at a time, until there is room or
// nothing is left to settle.  (C2: 151 booked positions per 120 k-mers; 20 steps fit a 12.4 G window instead of 15.3 G.)
static void settle_for_room(mcx_graph *g, int colour, uint64_t need)
{
  if (!g->h_snap) return;
  while (!defer_has_room(g, colour, need) && g->snap_tail != g->snap_head) {
    if (hipEventSynchronize(g->snap[g->snap_tail % mcx_graph::kSnap].ev) != hipSuccess) { (void)hipGetLastError(); return; }
    const uint32_t before = g->snap_tail;
    snap_poll(g);
    if (g->snap_tail == before) return;  // (cannot happen: the event has completed)
  }
}

static int defer_reserve(mcx_graph *g, int colour, uint64_t ub, int *set_out = nullptr)
{
  if (set_out) *set_out = 0;
  int rc = ensure_defer(g);
  if (rc != MCX_OK || !g->defer) return rc;
  snap_poll(g);
  settle_for_room(g, colour, ub);
  for (int pass = 0; pass < 2; pass++) {
    int free_set = -1;
    for (uint32_t s = 0; s < g->nsets; s++) {
      if (g->set_colour[s] == colour && g->set_pending[s] + ub <= g->set_cap) {
        g->set_pending[s] += ub; g->pending += ub;
        if (set_out) *set_out = (int)s;
        return MCX_OK;
      }
      if (g->set_colour[s] < 0 && free_set < 0) free_set = (int)s;
    }
    if (free_set >= 0) {
      g->set_colour[free_set] = colour;
      g->set_pending[free_set] = ub; g->pending += ub;
      if (set_out) *set_out = free_set;
      return MCX_OK;
    }
    rc = flush_deferred(g);
    if (rc != MCX_OK) return rc;
  }
  return fail(MCX_ERR_ARG, "internal: no L1 bin set after a flush");
}

// the sub-table bins take tuples of one colour at a time (sharded receive path: split on arrival)
static int l2_reserve(mcx_graph *g, int colour, uint64_t ub)
{
  int rc = ensure_defer(g);
  if (rc != MCX_OK || !g->defer) return rc;
  if ((g->pending_l2 && colour != g->pending_colour) ||
      ((g->pending_l2 || g->pending) && g->pending_l2 + g->pending + ub > g->defer_tuples)) {
    rc = flush_deferred(g);
    if (rc != MCX_OK) return rc;
  }
  g->pending_colour = colour;
  return MCX_OK;
}

// "Hash table is full" while input is still coming in: the device's flag (raised by the first insert that finds neither its
// sub-table nor the overflow area free, mcx_kernels.h table_flag_full) follows every submitted piece of stream to pinned
// memory; a later submission that finds it set ends the call -- the reference dies at that insert (hash_table.c:119-123),
// this build within a few pieces of it.  Every stream entry point comes through here (ASCII, packed, -Q/-H, PCR, device
// streams); the shards of a multi-GPU table report at their sync.
static int full_poll(mcx_graph *g)
{
  if (!g->h_full || !g->t.touch) return MCX_OK;
  if (*g->h_full) return fail(MCX_ERR_FULL, "Hash table is full");
  HIP_TRY(hipMemcpyAsync(g->h_full, g->t.touch - 2, sizeof(uint32_t), hipMemcpyDeviceToHost, g->stream));
  return MCX_OK;
}

static int submit_stream(mcx_graph *g, const StreamLaunch &L, int colour)
{
  if (g->group) return group_submit_stream(g->group, g->gidx, L, colour);  // shard of a multi-GPU table
  { int rc = full_poll(g); if (rc != MCX_OK) return rc; }
  if (g->defer) { int rc = ensure_defer(g); if (rc != MCX_OK) return rc; }
  if (!g->defer) {
    DISPATCH_WC4(g, launch_direct_t, g, L, colour);
    HIP_TRY(hipGetLastError());
    return MCX_OK;
  }
  // pieces of at most defer_tuples start positions (an upper bound of the tuples they yield)
  snap_poll(g);
  for (uint64_t lo = L.pos_lo; lo < L.pos_hi;) {
    settle_for_room(g, colour, std::min<uint64_t>(L.pos_hi - lo, g->set_cap));
    // a piece goes to ONE set: what is left of the colour's current set, or a fresh set
    uint64_t room = g->set_cap;
    for (uint32_t s = 0; s < g->nsets; s++)
      if (g->set_colour[s] == colour && g->set_pending[s] < g->set_cap) { room = g->set_cap - g->set_pending[s]; break; }
    if (room < (uint64_t)kTile * 16 && room < L.pos_hi - lo) room = g->set_cap;  // (not worth a launch: take a fresh set)
    const uint64_t hi = std::min(L.pos_hi, lo + room);
    int set = 0;
    int rc = defer_reserve(g, colour, hi - lo, &set);
    if (rc != MCX_OK) return rc;
    StreamLaunch P = L;
    P.pos_lo = lo; P.pos_hi = hi;
    BinSpec bs{BIN_GROUP, 0, g->b1, g->rep1, g->b1, 1, 0, 0};
    BinOut out = l1_out(g, set);
    DISPATCH_WC(g, launch_bin_region_stream, g, P, colour, bs, out);
    HIP_TRY(hipGetLastError());
    snap_push(g, false, hi - lo, set, 0);
    lo = hi;
  }
  return MCX_OK;
}

static int ensure_stage(mcx_graph *g);



extern "C" int mcx_graph_configure(mcx_graph *g, const char *key, uint64_t value)
{
  if (g && key && g->as_group) {
    const bool isec = !strcmp(key, "intersect");
    if (isec && g->ncols < 2) return fail(MCX_ERR_ARG, "intersect mode needs one colour more than the output has");
    if (!strcmp(key, "defer") && !value) return fail(MCX_ERR_ARG, "a multi-GPU table always uses the partitioned insert");
    int rc = grp_drain(g->as_group);
    for (int i = 0; rc == MCX_OK && i < grp_n(g->as_group); i++) rc = mcx_graph_configure(grp_part(g->as_group, i), key, value);
    // the facade's view of the colours follows its shards' -- only once every shard has accepted the key
    // (a shard that failed leaves the handle in its old view; intersect mode cannot be switched off again,
    // so a partial failure is reported and the caller must destroy the handle)
    if (isec && rc == MCX_OK) {
      g->hidden = g->ncols - 1;
      g->ncols_vis = g->ncols - 1;
    }
    return rc;
  }
  if (!g || !key) return fail(MCX_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(g->device));
  if (!strcmp(key, "defer")) {
    int rc = flush_deferred(g);
    if (rc != MCX_OK) return rc;
    g->defer = value != 0 && g->W <= 2;  // (k > 63: the fused kernel only)
    return MCX_OK;
  }
  if (!strcmp(key, "debug_realloc_l2")) {
    // experiment (tools/exp_split_var.py): move the sub-table bins to another place in HBM, leaving a hole of `value`
    // bytes behind -- the split's time follows the placement of its output (profiles/r06_experiments.md)
    int rc = flush_deferred(g);
    if (rc != MCX_OK) return rc;
    if (!g->l2_keys || g->l2_hi) return MCX_OK;
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (g->stream2) HIP_TRY(hipStreamSynchronize(g->stream2));
    static std::vector<void *> holes;
    void *hole = nullptr;
    if (value && hipMalloc(&hole, value) == hipSuccess) holes.push_back(hole);
    uint64_t *nk = nullptr;
    HIP_TRY(hipMalloc((void **)&nk, (uint64_t)g->l2_regions * g->subs_per_bin * g->cap2 * 8 * g->W));
    (void)hipFree(g->l2_base);
    g->l2_base = g->l2_keys = nk;
    return MCX_OK;
  }
  if (!strcmp(key, "place_bins")) {  // 1 = take the first allocation of sub-table bins; n = the best of n by the write-pattern probe
    if (value < 1 || value > 32) return fail(MCX_ERR_ARG, "place_bins: 1..32");
    g->place_tries = (int)value;
    return MCX_OK;
  }
  if (!strcmp(key, "defer_tuples")) {
    int rc = flush_deferred(g);
    if (rc != MCX_OK) return rc;
    HIP_TRY(hipStreamSynchronize(g->stream));
    free_defer(g);
    g->defer_tuples = value;
    return MCX_OK;
  }
  if (!strcmp(key, "flush_overlap")) {  // 1: the insert of region group g runs beside the split of g + 1 (default); 0: one kernel at a time
    int rc = flush_deferred(g);        // (isolated per-kernel durations: bench.py's roofline.kernels, tools/prof.sh)
    if (rc != MCX_OK) return rc;
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->overlap = value != 0;
    if (g->l1_keys && g->overlap) return ensure_l2(g, std::min<uint32_t>(2 * flush_group(g), g->b1));
    return MCX_OK;
  }
  if (!strcmp(key, "flush_regions")) {  // regions split + applied per step of a flush (0 = automatic)
    int rc = flush_deferred(g);
    if (rc != MCX_OK) return rc;
    g->flush_regions = (uint32_t)value;
    if (g->l1_keys) return ensure_l2(g, flush_group(g));
    return MCX_OK;
  }
  if (!strcmp(key, "intersect")) {  // the last colour becomes the hidden holder of the intersection edges
    if (!value) return fail(MCX_ERR_ARG, "intersect mode cannot be switched off");
    if (g->ncols < 2) return fail(MCX_ERR_ARG, "intersect mode needs one colour more than the output has");
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->hidden = g->ncols - 1;
    g->ncols_vis = g->ncols - 1;
    g->defer = false;  // every update goes straight to the table
    return MCX_OK;
  }
  if (!strcmp(key, "must_exist")) {
    int rc = flush_deferred(g);
    if (rc != MCX_OK) return rc;
    g->must_exist = value != 0;
    if (g->must_exist) g->defer = false;
    return MCX_OK;
  }
  if (!strcmp(key, "prepare")) {  // allocate now what the first mcx_graph_add_reads would: pinned staging, bins
    const double t0 = now_s();
    int rc = ensure_stage(g);
    const double t1 = now_s();
    if (rc == MCX_OK && g->defer && !g->must_exist) rc = ensure_defer(g);
    const double t2 = now_s();
    if (rc == MCX_OK && g->defer && !g->must_exist && flush_overlap(g)) rc = ensure_flush_stream(g);
    if (getenv("MCX_TIMING"))
      fprintf(stderr, "[timing] prepare: pinned staging + copy stream %.1f ms, partition workspace (%.1f GB) %.1f ms, flush stream %.1f ms\n", (t1 - t0) * 1e3,
              g->l1_keys ? (double)g->nsets * g->b1 * g->rep1 * g->cap1 * 8 * g->W / 1e9 : 0.0, (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
    return rc;
  }
  if (!strcmp(key, "grid_stream")) { g->grid_stream = (int)value; return MCX_OK; }
  if (!strcmp(key, "grid_split")) { g->grid_split = (int)value; return MCX_OK; }
  if (!strcmp(key, "grid_insert")) { g->grid_insert = (int)value; return MCX_OK; }
  if (!strcmp(key, "profile")) {
    HIP_TRY(hipStreamSynchronize(g->stream));
    for (auto &sp : g->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    g->spans.clear();
    g->profile = value != 0;
    return MCX_OK;
  }
  return fail(MCX_ERR_ARG, "unknown configuration key '%s'", key);
}

// Text report "kernel calls total_ms" per line of the spans recorded since profiling was enabled.
extern "C" int mcx_graph_profile(mcx_graph *g, char *buf, size_t buflen)
{
  if (g && buf && buflen && g->as_group) {  // one report per shard, each line prefixed with its index
    size_t o = 0;
    buf[0] = '\0';
    std::vector<char> tmp(buflen);
    for (int i = 0; i < grp_n(g->as_group); i++) {
      int rc = mcx_graph_profile(grp_part(g->as_group, i), tmp.data(), tmp.size());
      if (rc != MCX_OK) return rc;
      for (char *ln = strtok(tmp.data(), "\n"); ln; ln = strtok(nullptr, "\n")) {
        int n = snprintf(buf + o, buflen - o, "%s@%d%s\n", std::string(ln, strcspn(ln, " ")).c_str(), i, ln + strcspn(ln, " "));
        if (n < 0 || (size_t)n >= buflen - o) return MCX_OK;
        o += (size_t)n;
      }
    }
    return MCX_OK;
  }
  if (!g || !buf || !buflen) return fail(MCX_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(g->device));
  HIP_TRY(hipStreamSynchronize(g->stream));
  struct Acc { const char *name; int calls; double ms; };
  std::vector<Acc> acc;
  for (auto &sp : g->spans) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) continue;
    bool found = false;
    for (auto &a : acc) if (!strcmp(a.name, sp.name)) { a.calls++; a.ms += ms; found = true; }
    if (!found) acc.push_back({sp.name, 1, ms});
  }
  size_t o = 0;
  buf[0] = '\0';
  for (auto &a : acc) {
    int n = snprintf(buf + o, buflen - o, "%s %d %.4f\n", a.name, a.calls, a.ms);
    if (n < 0 || (size_t)n >= buflen - o) break;
    o += (size_t)n;
  }
  return MCX_OK;
}

extern "C" int mcx_graph_add_stream_dev(mcx_graph *g, int colour, const void *d_stream, uint64_t nbytes)
{
  if (g && g->as_group) {  // the shard on whose device the stream lives k-merises it
    mcx_graph *part = nullptr;
    int rc = grp_part_of_pointer(g->as_group, d_stream, &part);
    if (rc != MCX_OK) return rc;
    return mcx_graph_add_stream_dev(part, colour, d_stream, nbytes);
  }
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (colour < 0 || colour >= g->ncols) return fail(MCX_ERR_ARG, "colour %d out of range", colour);
  if (((uintptr_t)d_stream & 15) != 0) return fail(MCX_ERR_ARG, "stream must be 16-byte aligned");
  if (!nbytes) return MCX_OK;
  HIP_TRY(hipSetDevice(g->device));
  StreamLaunch L{(const uint8_t *)d_stream, nbytes, 0, nbytes, nullptr};
  return submit_stream(g, L, colour);
}

// Packed form of a device-resident stream: the conversion (what a GPU-side parser would emit
// directly) and the entry that consumes it.
extern "C" int mcx_pack_stream_dev(const void *d_stream, uint64_t nbytes, void *d_code, void *d_inv, void *hip_stream)
{
  if (!d_stream || !d_code || !d_inv) return fail(MCX_ERR_ARG, "null argument");
  if (((uintptr_t)d_stream & 15) != 0) return fail(MCX_ERR_ARG, "stream must be 16-byte aligned");
  if (!nbytes) return MCX_OK;
  const uint64_t nch = (nbytes + 15) / 16;
  hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                     (const uint8_t *)d_stream, nbytes, (uint32_t *)d_code, (uint16_t *)d_inv);
  HIP_TRY(hipGetLastError());
  return MCX_OK;
}

extern "C" int mcx_graph_add_packed_dev(mcx_graph *g, int colour, const void *d_code, const void *d_inv, uint64_t npos)
{
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (g->as_group) {
    mcx_graph *part = nullptr;
    int rc = grp_part_of_pointer(g->as_group, d_code, &part);
    if (rc != MCX_OK) return rc;
    return mcx_graph_add_packed_dev(part, colour, d_code, d_inv, npos);
  }
  if (colour < 0 || colour >= g->ncols) return fail(MCX_ERR_ARG, "colour %d out of range", colour);
  if (!d_code || !d_inv) return fail(MCX_ERR_ARG, "null argument");
  if (!npos) return MCX_OK;
  if (g->must_exist) return fail(MCX_ERR_ARG, "device streams cannot be loaded in must-exist mode");
  HIP_TRY(hipSetDevice(g->device));
  StreamLaunch L{nullptr, npos, 0, npos, nullptr, (const uint32_t *)d_code, (const uint16_t *)d_inv};
  return submit_stream(g, L, colour);
}

extern "C" int mcx_graph_partition_stream_dev(mcx_graph *g, const void *d_stream, uint64_t nbytes, int nparts,
                                              uint64_t bin_capacity, void *d_keys, void *d_edges, void *d_counts)
{
  NO_GROUP(g, "mcx_graph_partition_stream_dev");
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  NO_WIDE(g, "mcx_graph_partition_stream_dev");
  if (nparts < 1 || nparts > kMaxBins) return fail(MCX_ERR_ARG, "nparts must be 1..%d", kMaxBins);
  if (bin_capacity >= 0xFFFFFFFFull) return fail(MCX_ERR_ARG, "bin capacity must be below 2^32 tuples");
  if (((uintptr_t)d_stream & 15) != 0) return fail(MCX_ERR_ARG, "stream must be 16-byte aligned");
  if (!nbytes) return MCX_OK;
  HIP_TRY(hipSetDevice(g->device));
  StreamLaunch L{(const uint8_t *)d_stream, nbytes, 0, nbytes, nullptr};
  BinSpec bs{BIN_OWNER, (uint32_t)nparts, (uint32_t)nparts, 1, (uint32_t)nparts, 1, 0, 0};
  BinOut out{(uint64_t *)d_keys, (uint8_t *)d_edges, (unsigned long long *)d_counts, bin_capacity, nullptr, nullptr, nullptr, 0};
  if (g->W == 1) launch_bin_stream_t<1, true, true, 0>(g, L, 0, bs, out);
  else launch_bin_stream_t<2, true, true, 0>(g, L, 0, bs, out);
  HIP_TRY(hipGetLastError());
  return MCX_OK;
}

extern "C" int mcx_graph_insert_tuples_dev(mcx_graph *g, int colour, const void *d_keys, const void *d_edges, uint64_t n)
{
  NO_GROUP(g, "mcx_graph_insert_tuples_dev");
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (colour < 0 || colour >= g->ncols) return fail(MCX_ERR_ARG, "colour %d out of range", colour);
  if (!n) return MCX_OK;
  HIP_TRY(hipSetDevice(g->device));
  if (g->defer) { int rc = ensure_defer(g); if (rc != MCX_OK) return rc; }
  if (!g->defer) {
    DISPATCH_WC4(g, launch_insert_tuples_t, g, colour, (const uint64_t *)d_keys, (const uint8_t *)d_edges, n);
    HIP_TRY(hipGetLastError());
    return MCX_OK;
  }
  for (uint64_t lo = 0; lo < n;) {
    const uint64_t cnt = std::min(n - lo, g->set_cap);
    int set = 0;
    int rc = defer_reserve(g, colour, cnt, &set);
    if (rc != MCX_OK) return rc;
    TupleIn in{(const uint64_t *)d_keys + lo * g->W, (const uint8_t *)d_edges + lo, nullptr, cnt, 1, 1, 1};
    BinSpec bs{BIN_GROUP, 0, g->b1, g->rep1, g->b1, 1, 0, 0};
    BinOut out = l1_out(g, set);
    DISPATCH_WC(g, launch_bin_received, g, in, colour, bs, out);
    HIP_TRY(hipGetLastError());
    lo += cnt;
  }
  return MCX_OK;
}

// experiments (tools/exp_hbm_map.py): the split's write-pattern probe on any device buffer, timed
extern "C" int mcx_debug_probe(void *d_buf, uint64_t cap_words, uint32_t nregions, uint32_t spb, uint32_t iters, uint32_t jit_mask, float *ms_out)
{
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  float t = 0.f;
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_l2_probe, dim3(nregions * 64), dim3(512), 0, 0, (uint64_t *)d_buf, cap_words, nregions, spb, iters, jit_mask);
    (void)hipEventRecord(e1, 0);
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (ms_out) *ms_out = t;
  return MCX_OK;
}

// ---- `hashtest` (src/commands/ctx_exp_hashtest.c:40-69) ----
template <int W> __global__ void k_hashtest_keys(uint64_t *keys, uint64_t first, uint64_t n)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    keys[i * W] = first + i;  // bkmer.b[0] = i
    for (int w = 1; w < W; w++) keys[i * W + w] = 0;
  }
}
template <int W> __global__ void k_hashtest_func(uint64_t lo, uint64_t hi, unsigned int *out)
{
  uint32_t h = 0;
  for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (uint64_t)gridDim.x * blockDim.x) {
    Kmer<W> key;
    key.w[0] = i;
    for (int w = 1; w < W; w++) key.w[w] = 0;
    h ^= kmer_hash<W>(key, 0, nullptr);
  }
  for (int d = 32; d; d >>= 1) h ^= __shfl_xor(h, d, 64);
  if ((threadIdx.x & 63) == 0 && h) atomicXor(out, h);
}

extern "C" int mcx_graph_hashtest(mcx_graph *g, uint64_t first, uint64_t n)
{
  NO_GROUP(g, "mcx_graph_hashtest");
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (first + n < first) return fail(MCX_ERR_ARG, "key range wraps");
  HIP_TRY(hipSetDevice(g->device));
  const uint64_t chunk = std::min<uint64_t>(n, 1ull << 26);  // 64 M keys = 512 MiB (W = 1) of scratch
  if (!chunk) return mcx_graph_sync(g);
  uint64_t *d_keys = nullptr;
  uint8_t *d_edges = nullptr;
  if (hipMalloc((void **)&d_keys, chunk * 8 * g->W) != hipSuccess || hipMalloc((void **)&d_edges, chunk) != hipSuccess) {
    (void)hipGetLastError(); (void)hipFree(d_keys); (void)hipFree(d_edges);
    return fail(MCX_ERR_NOMEM, "out of device memory for the hashtest key buffer");
  }
  int rc = MCX_OK;
  if (hipMemsetAsync(d_edges, 0, chunk, g->stream) != hipSuccess) rc = fail(MCX_ERR_HIP, "hipMemsetAsync failed");
  for (uint64_t lo = 0; lo < n && rc == MCX_OK; lo += chunk) {
    const uint64_t cnt = std::min(chunk, n - lo);
    LAUNCH_W4(g->W, k_hashtest_keys, dim3(4096), dim3(256), 0, g->stream, d_keys, first + lo, cnt);
    rc = mcx_graph_insert_tuples_dev(g, 0, d_keys, d_edges, cnt);
    // (the binning kernel that reads the buffer runs on the same stream as the generator of the next chunk)
  }
  const int rc2 = mcx_graph_sync(g);
  (void)hipFree(d_keys); (void)hipFree(d_edges);
  return rc != MCX_OK ? rc : rc2;
}

extern "C" int mcx_hashtest_func(int device, int kmer_size, uint64_t n, uint32_t nparts, uint64_t *hash_out)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(MCX_ERR_NODEVICE, "no HIP device %d", device);
  if (kmer_size < 3 || kmer_size > 127 || !(kmer_size & 1)) return fail(MCX_ERR_ARG, "kmer size %d", kmer_size);
  if (!nparts || !hash_out) return fail(MCX_ERR_ARG, "nparts / hash_out");
  HIP_TRY(hipSetDevice(device));
  unsigned int *d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, (size_t)nparts * 4));
  std::vector<unsigned int> h(nparts, 0);
  bool ok = hipMemset(d, 0, (size_t)nparts * 4) == hipSuccess;
  const int W = words_for_k(kmer_size);
  for (uint32_t p = 0; p < nparts && ok; p++) {
    const uint64_t lo = (uint64_t)p * (n / nparts), hi = p + 1 == nparts ? n : lo + n / nparts;
    if (hi <= lo) continue;
    const unsigned grid = (unsigned)std::min<uint64_t>((hi - lo + 255) / 256, 8192);
    LAUNCH_W4(W, k_hashtest_func, dim3(grid), dim3(256), 0, 0, lo, hi, d + p);
  }
  ok = ok && hipMemcpy(h.data(), d, (size_t)nparts * 4, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  if (!ok) { (void)hipGetLastError(); return fail(MCX_ERR_HIP, "hashtest kernels failed"); }
  uint64_t sum = 0;
  for (uint32_t p = 0; p < nparts; p++) sum += h[p];
  *hash_out = sum;
  return MCX_OK;
}

extern "C" int mcx_graph_insert_tuple_segments_dev(mcx_graph *g, int colour, const void *d_keys, const void *d_edges,
                                                   const void *d_counts, uint32_t nseg, uint64_t seg_cap)
{
  NO_GROUP(g, "mcx_graph_insert_tuple_segments_dev");
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (colour < 0 || colour >= g->ncols) return fail(MCX_ERR_ARG, "colour %d out of range", colour);
  if (!nseg || !seg_cap) return MCX_OK;
  if (!d_keys || !d_edges || !d_counts) return fail(MCX_ERR_ARG, "null tuple buffers");
  HIP_TRY(hipSetDevice(g->device));
  int rc = ensure_defer(g);
  if (rc != MCX_OK) return rc;
  if (!g->defer) return fail(MCX_ERR_ARG, "tuple segments need the deferred insert path (table too small or defer=0)");
  const uint64_t ub = std::min<uint64_t>((uint64_t)nseg * seg_cap, g->set_cap);  // the fills are only known on the device
  int set = 0;
  rc = defer_reserve(g, colour, ub, &set);
  if (rc != MCX_OK) return rc;
  TupleIn in{(const uint64_t *)d_keys, (const uint8_t *)d_edges, (const unsigned long long *)d_counts, seg_cap, nseg, nseg, nseg};
  BinSpec bs{BIN_GROUP, 0, g->b1, g->rep1, g->b1, 1, 0, 0};
  BinOut out = l1_out(g, set);
  DISPATCH_WC(g, launch_bin_received, g, in, colour, bs, out);
  HIP_TRY(hipGetLastError());
  return MCX_OK;
}

// ---------------------------------------------------------------------------
// sharded build, exchange format v2: packed tuples binned by (owner, region) on the sender
// ---------------------------------------------------------------------------
static const uint32_t kShardRep = 8;

extern "C" int mcx_graph_shard_layout(mcx_graph *g, uint64_t tuples_per_call, uint32_t *segs_per_owner,
                                      uint64_t *seg_cap, uint64_t *ov_cap)
{
  NO_GROUP(g, "mcx_graph_shard_layout");
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  NO_WIDE(g, "mcx_graph_shard_layout");
  const uint32_t nparts = 1u << g->t.lbo, b1 = 1u << g->t.lb1;
  const uint64_t nseg = (uint64_t)nparts * kShardRep * b1;
  if (segs_per_owner) *segs_per_owner = kShardRep * b1;
  // Poisson spread of a segment is ~sqrt(mean): 6 % + 2048 covers > 10 sigma for the bench shapes;
  // anything beyond (hot k-mers) goes to the owner's overflow bin
  if (seg_cap) *seg_cap = (uint64_t)((double)tuples_per_call / (double)nseg * 1.06) + 2048;
  if (ov_cap) *ov_cap = std::max<uint64_t>(1u << 16, tuples_per_call / nparts / 64);
  return MCX_OK;
}

static int shard_bins_launch(mcx_graph *g, const StreamLaunch &L, void *d_keys, void *d_counts, uint64_t seg_cap,
                             void *d_ov_keys, void *d_ov_edges, void *d_ov_counts, uint64_t ov_cap, bool spill = false);

extern "C" int mcx_graph_shard_bins_dev(mcx_graph *g, const void *d_stream, uint64_t nbytes, void *d_keys,
                                        void *d_counts, uint64_t seg_cap, void *d_ov_keys, void *d_ov_edges,
                                        void *d_ov_counts, uint64_t ov_cap)
{
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  NO_WIDE(g, "mcx_graph_shard_bins_dev");
  if (g->as_group) return fail(MCX_ERR_ARG, "device-pointer exchange calls take a shard, not the multi-GPU handle");
  if (!nbytes) return MCX_OK;
  StreamLaunch L{(const uint8_t *)d_stream, nbytes, 0, nbytes, nullptr};
  return shard_bins_launch(g, L, d_keys, d_counts, seg_cap, d_ov_keys, d_ov_edges, d_ov_counts, ov_cap);
}

static int shard_bins_launch(mcx_graph *g, const StreamLaunch &L, void *d_keys, void *d_counts, uint64_t seg_cap,
                             void *d_ov_keys, void *d_ov_edges, void *d_ov_counts, uint64_t ov_cap, bool spill)
{
  if (!L.code && ((uintptr_t)L.stream & 15) != 0) return fail(MCX_ERR_ARG, "stream must be 16-byte aligned");
  if (g->t.lb1 + g->t.lbo > 11) return fail(MCX_ERR_ARG, "too many (owner, region) bins");
  if (seg_cap >= 0xFFFFFFFFull) return fail(MCX_ERR_ARG, "segment capacity must be below 2^32 tuples");
  HIP_TRY(hipSetDevice(g->device));
  const uint32_t nparts = 1u << g->t.lbo, b1 = 1u << g->t.lb1;
  BinSpec bs{BIN_GLOBAL, nparts, nparts * b1, kShardRep, b1, 1, g->t.lb1, spill ? 1u : 0u};
  BinOut out{(uint64_t *)d_keys, nullptr, (unsigned long long *)d_counts, seg_cap,
             (uint64_t *)d_ov_keys, (uint8_t *)d_ov_edges, (unsigned long long *)d_ov_counts, ov_cap};
  if (g->W == 1) launch_bin_stream_t<1, true, false, 2>(g, L, 0, bs, out);
  else launch_bin_stream_t<2, true, false, 2>(g, L, 0, bs, out);
  HIP_TRY(hipGetLastError());
  return MCX_OK;
}

extern "C" int mcx_graph_add_segments_dev(mcx_graph *g, int colour, const void *d_keys, const void *d_counts,
                                          uint32_t nseg, uint64_t seg_cap, uint64_t ntuples)
{
  NO_GROUP(g, "mcx_graph_add_segments_dev");
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  NO_WIDE(g, "mcx_graph_add_segments_dev");
  if (colour < 0 || colour >= g->ncols) return fail(MCX_ERR_ARG, "colour %d out of range", colour);
  if (!nseg) return MCX_OK;
  HIP_TRY(hipSetDevice(g->device));
  int rc = ensure_defer(g);
  if (rc != MCX_OK) return rc;
  if (!g->defer) return fail(MCX_ERR_ARG, "packed segments need the deferred insert path (table too small or defer=0)");
  if (nseg % g->b1) return fail(MCX_ERR_ARG, "segments must cover whole sets of %u regions", g->b1);
  rc = l2_reserve(g, colour, ntuples);
  if (rc != MCX_OK) return rc;
  rc = ensure_l2(g, g->b1, true);  // split on arrival: bins for every region
  if (rc != MCX_OK) return rc;
  TupleIn in{(const uint64_t *)d_keys, nullptr, (const unsigned long long *)d_counts, seg_cap, nseg, nseg, nseg};
  BinSpec bs{BIN_SUBLOCAL, 0, g->subs_per_bin, 1, g->nsub, g->b1, 0, 0};
  BinOut out{g->l2_keys, nullptr, g->l2_cnt, g->cap2, nullptr, nullptr, nullptr, 0};
  DISPATCH_WC(g, launch_split_regions, g, in, colour, bs, out);
  HIP_TRY(hipGetLastError());
  g->pending_l2 += ntuples;
  return MCX_OK;
}

// ---------------------------------------------------------------------------
// sharded build, exchange format v3: super-k-mer records (mcx_superk.h)
// ---------------------------------------------------------------------------
static const uint32_t kSuperkRep = 8;

extern "C" int mcx_superk_supported(int kmer_size) { return kmer_size >= kSuperkMinK && kmer_size <= 63 && (kmer_size & 1); }
extern "C" int mcx_superk_record_bytes(int kmer_size) { return mcx_superk_supported(kmer_size) ? 16 * words_for_k(kmer_size) : 0; }

extern "C" uint32_t mcx_superk_owner(const uint64_t *key_words, int kmer_size, int nparts)
{
  uint32_t lbo = 0;
  while ((1 << lbo) < nparts) lbo++;
  return words_for_k(kmer_size) == 1 ? superk_owner(0, key_words[0], kmer_size, lbo) : superk_owner(key_words[0], key_words[1], kmer_size, lbo);
}

extern "C" int mcx_graph_superk_layout(mcx_graph *g, int nparts, uint64_t positions_per_call, uint32_t *segs_per_owner,
                                       uint64_t *seg_cap)
{
  NO_GROUP(g, "mcx_graph_superk_layout");
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  NO_WIDE(g, "mcx_graph_superk_layout");
  if (nparts < 1 || nparts > 32 || (nparts & (nparts - 1))) return fail(MCX_ERR_ARG, "shards must be a power of two <= 32");
  if (segs_per_owner) *segs_per_owner = kSuperkRep;
  // ~2.3 records per 16 positions on random reads; room for 4 (a bin that overflows is reported as
  // MCX_ERR_FULL at the next sync of the sending handle: nothing is lost silently)
  if (seg_cap) *seg_cap = positions_per_call / 4 / ((uint64_t)nparts * kSuperkRep) + 4096;
  return MCX_OK;
}

extern "C" int mcx_graph_superk_bins_dev(mcx_graph *g, const void *d_stream, uint64_t nbytes, int nparts, void *d_recs,
                                         void *d_counts, uint64_t seg_cap)
{
  NO_GROUP(g, "mcx_graph_superk_bins_dev");
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  NO_WIDE(g, "mcx_graph_superk_bins_dev");
  if (!mcx_superk_supported(g->k)) return fail(MCX_ERR_ARG, "super-k-mer records need an odd k in %d..63 (got %d)", kSuperkMinK, g->k);
  if (nparts < 1 || nparts > 32 || (nparts & (nparts - 1))) return fail(MCX_ERR_ARG, "shards must be a power of two <= 32");
  if (((uintptr_t)d_stream & 15) != 0) return fail(MCX_ERR_ARG, "stream must be 16-byte aligned");
  if (!nbytes) return MCX_OK;
  HIP_TRY(hipSetDevice(g->device));
  uint32_t lbo = 0;
  while ((1 << lbo) < nparts) lbo++;
  StreamLaunch L{(const uint8_t *)d_stream, nbytes, 0, nbytes, nullptr};
  const StreamArgs a = make_args(g, L);
  const uint64_t nt = a.ntiles > a.tile0 ? a.ntiles - a.tile0 : 0;
  if (!nt) return MCX_OK;
  SuperkOut out{d_recs, (unsigned long long *)d_counts, seg_cap, lbo, kSuperkRep, nullptr, nullptr, nullptr, 0};
  SpanGuard sp(g, "k_stream_superk");
  const dim3 grid((unsigned)std::min<uint64_t>(nt, (uint64_t)g->grid));
  if (g->W == 1) hipLaunchKernelGGL((k_stream_superk<1, false>), grid, dim3(kThreads), 0, g->stream, a, out);
  else hipLaunchKernelGGL((k_stream_superk<2, false>), grid, dim3(kThreads), 0, g->stream, a, out);
  HIP_TRY(hipGetLastError());
  return MCX_OK;
}

// the same for a launch description (ASCII or packed stream, owned range) and with a spill area: the
// sender step of the in-process multi-GPU table (mcx_multi.h)
static int superk_bins_launch(mcx_graph *g, const StreamLaunch &L, const SuperkOut &out)
{
  if (!L.code && ((uintptr_t)L.stream & 15) != 0) return fail(MCX_ERR_ARG, "stream must be 16-byte aligned");
  HIP_TRY(hipSetDevice(g->device));
  const StreamArgs a = make_args(g, L);
  const uint64_t nt = a.ntiles > a.tile0 ? a.ntiles - a.tile0 : 0;
  if (!nt) return MCX_OK;
  SpanGuard sp(g, "k_stream_superk");
  const dim3 grid((unsigned)std::min<uint64_t>(nt, (uint64_t)g->grid));
  if (g->W == 1) {
    if (a.code) hipLaunchKernelGGL((k_stream_superk<1, true>), grid, dim3(kThreads), 0, g->stream, a, out);
    else hipLaunchKernelGGL((k_stream_superk<1, false>), grid, dim3(kThreads), 0, g->stream, a, out);
  } else {
    if (a.code) hipLaunchKernelGGL((k_stream_superk<2, true>), grid, dim3(kThreads), 0, g->stream, a, out);
    else hipLaunchKernelGGL((k_stream_superk<2, false>), grid, dim3(kThreads), 0, g->stream, a, out);
  }
  HIP_TRY(hipGetLastError());
  return MCX_OK;
}

template <int W, bool ONECOL> static void launch_superk_bin(mcx_graph *g, SuperkIn in, int colour, BinSpec bs, BinOut out)
{
  const uint64_t nunits = (in.seg_cap + kSkChunk - 1) / kSkChunk * in.nseg;  // chunks of records, one block walks one at a time
  InsertSink<W, ONECOL> is{g->t, (uint32_t)colour};
  const size_t lds512 = ((sizeof(BinLds<W, 512, false>) + 15) & ~(size_t)15) + kSkMapBytes;
  const size_t ldsmax = ((sizeof(BinLds<W, kMaxBins, false>) + 15) & ~(size_t)15) + kSkMapBytes;
  static bool once_dev[64] = {false};  // per device: the attribute belongs to the function on one device
  bool &once = once_dev[g->device & 63];
  if (!once) {
    allow_lds(k_superk_bin<W, ONECOL, 512>, lds512);
    allow_lds(k_superk_bin<W, ONECOL, kMaxBins>, ldsmax);
    once = true;
  }
  SpanGuard sp(g, "k_superk_bin");
  const dim3 grid((unsigned)std::min<uint64_t>(nunits, (uint64_t)g->grid * 4));
  if (bs.nlocal <= 512)
    hipLaunchKernelGGL((k_superk_bin<W, ONECOL, 512>), grid, dim3(kThreads), lds512, g->stream, in, g->k, bs, out, is, g->d_ctr);
  else
    hipLaunchKernelGGL((k_superk_bin<W, ONECOL, kMaxBins>), grid, dim3(kThreads), ldsmax, g->stream, in, g->k, bs, out, is, g->d_ctr);
}

extern "C" int mcx_graph_add_superk_dev(mcx_graph *g, int colour, const void *d_recs, const void *d_counts, uint32_t nseg,
                                        uint64_t seg_cap, uint64_t kmers_upper_bound)
{
  NO_GROUP(g, "mcx_graph_add_superk_dev");
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  NO_WIDE(g, "mcx_graph_add_superk_dev");
  if (colour < 0 || colour >= g->ncols_vis) return fail(MCX_ERR_ARG, "colour %d out of range", colour);
  if (!mcx_superk_supported(g->k)) return fail(MCX_ERR_ARG, "super-k-mer records need an odd k in %d..63 (got %d)", kSuperkMinK, g->k);
  if (g->t.lbo) return fail(MCX_ERR_ARG, "super-k-mer shards use ordinary (unsharded) tables");
  if (!nseg || !seg_cap) return MCX_OK;
  HIP_TRY(hipSetDevice(g->device));
  int rc = ensure_defer(g);
  if (rc != MCX_OK) return rc;
  if (!g->defer) return fail(MCX_ERR_ARG, "super-k-mer records need the deferred insert path (table too small or defer=0)");
  int set = 0;
  const uint64_t ub = std::min(kmers_upper_bound, g->set_cap);
  rc = defer_reserve(g, colour, ub, &set);
  if (rc != MCX_OK) return rc;
  SuperkIn in{d_recs, (const unsigned long long *)d_counts, seg_cap, nseg};
  BinSpec bs{BIN_GROUP, 0, g->b1, g->rep1, g->b1, 1, 0, 0};
  BinOut out = l1_out(g, set);
  DISPATCH_WC(g, launch_superk_bin, g, in, colour, bs, out);
  HIP_TRY(hipGetLastError());
  // the caller's bound is 16 k-mers per record slot; what the records really held (Counters::binned) comes off the
  // books when this launch has settled
  snap_push(g, false, ub, set, 1);
  return MCX_OK;
}

extern "C" uint32_t mcx_graph_key_owner(const mcx_graph *g, const uint64_t *key_words)
{
  if (!g) return 0;
  if (g->as_group) g = grp_part(g->as_group, 0);
  if (g->W > 2) return 0;  // (k > 63: one device)
  if (g->own_lbo) return g->W == 1 ? superk_owner(0, key_words[0], g->k, g->own_lbo) : superk_owner(key_words[0], key_words[1], g->k, g->own_lbo);
  const uint32_t lbq = g->t.lb1 + g->t.lbo;
  uint32_t r = 0, m;
  if (g->W == 1) {
    m = region_mix<1>(key_quot<1>(Kmer<1>{{key_words[0]}}, lbq, r));
  } else {
    m = region_mix<2>(key_quot<2>(Kmer<2>{{key_words[0], key_words[1]}}, lbq, r));
  }
  return (r ^ mix_g(m, lbq)) >> g->t.lb1;
}

extern "C" uint32_t mcx_key_owner(const uint64_t *key_words, int kmer_size, int nparts)
{
  uint32_t h2 = 0;
  if (words_for_k(kmer_size) == 1) {
    Kmer<1> k{{key_words[0]}};
    kmer_hash<1>(k, 0, &h2);
  } else {
    Kmer<2> k{{key_words[0], key_words[1]}};
    kmer_hash<2>(k, 0, &h2);
  }
  return (uint32_t)(((uint64_t)h2 * (uint32_t)nparts) >> 32);
}

// ---------------------------------------------------------------------------
// host-buffer entry: stage reads as a '\n'-separated stream
// ---------------------------------------------------------------------------
// ---- 16 ASCII bases -> one code word + 16 invalid flags, on the host -----------------------------
// The same mapping as encode_words() on the device (mcx_kernels.h): codes A=0 C=1 G=2 T=3 from bits
// 1-2 of the character, first base on top; a position is invalid unless its character is one of
// ACGTacgt (dna.c:8-25).  Packed on the host, a position costs 3 bits on PCIe instead of 8 and the
// k-merising kernel starts from what it would otherwise compute in its tile prologue.
static inline void pack16_swar(const uint8_t *src, uint32_t *code, uint16_t *inv)
{
  uint32_t c = 0, v = 0;
  for (int h = 0; h < 2; h++) {
    uint64_t x;
    memcpy(&x, src + 8 * h, 8);
    const uint64_t t = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ULL;
    const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    c = (c << 16) | ((lo * 0x40100401u) >> 24 << 8) | ((hi * 0x40100401u) >> 24);
    const uint64_t u = x & 0xDFDFDFDFDFDFDFDFULL;
#define MCX_NZ64(z) ((((z) & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | (z))
    const uint64_t bad = MCX_NZ64(u ^ 0x4141414141414141ULL) & MCX_NZ64(u ^ 0x4343434343434343ULL) &
                         MCX_NZ64(u ^ 0x4747474747474747ULL) & MCX_NZ64(u ^ 0x5454545454545454ULL) & 0x8080808080808080ULL;
#undef MCX_NZ64
    // bit of byte i -> bit 63 - i: multiplier = sum of 2^(63 - 9 i) (no two products meet: no carries)
    v = (v << 8) | (uint32_t)(((bad >> 7) * 0x8040201008040201ULL) >> 56);
  }
  *code = c;
  *inv = (uint16_t)v;
}

#if defined(__x86_64__)
#include <immintrin.h>
// 32 positions per iteration, no scalar bit gathering: the 2-bit codes are folded with two
// multiply-adds (byte pairs -> nibbles -> bytes of four bases, first base on top), one byte shuffle
// puts the eight result bytes in the order of the two code words; the invalid flags are the sign
// mask of the byte-reversed compare results (first base = bit 15 of its word).  ~22 micro-ops per
// 32 bytes (the pext version: ~40): 2.9 -> 4+ GB/s of bases per staging thread.
__attribute__((target("avx2"))) static void pack_block_avx2(const uint8_t *src, size_t n, uint32_t *code, uint16_t *inv)
{
  const __m256i m3 = _mm256_set1_epi8(3), mdf = _mm256_set1_epi8((char)0xDF);
  const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
  const __m256i w1 = _mm256_set1_epi16(0x0104);      // bytes (4, 1): first base of a pair on top
  const __m256i w2 = _mm256_set1_epi32(0x00010010);  // words (16, 1)
  const __m256i pick = _mm256_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                        12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0,
                                       15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
  const __m256i lanes = _mm256_setr_epi32(0, 4, 0, 0, 0, 0, 0, 0);
  for (size_t i = 0; i + 32 <= n; i += 32) {
    const __m256i v = _mm256_loadu_si256((const __m256i *)(src + i));
    const __m256i t = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), m3);
    const __m256i nib = _mm256_maddubs_epi16(t, w1);   // 16-bit lanes: base 2i * 4 + base 2i+1
    const __m256i byt = _mm256_madd_epi16(nib, w2);    // 32-bit lanes: four bases, the first on top
    const __m256i ord = _mm256_shuffle_epi8(byt, pick);  // per half: its four bytes, last first
    const __m128i two = _mm256_castsi256_si128(_mm256_permutevar8x32_epi32(ord, lanes));
    _mm_storel_epi64((__m128i *)(code + i / 16), two);
    const __m256i u = _mm256_and_si256(v, mdf);
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
    const uint32_t bad = ~(uint32_t)_mm256_movemask_epi8(_mm256_shuffle_epi8(ok, rev));  // bit 15 - j of a half = byte j is not a base
    memcpy(inv + i / 16, &bad, 4);
  }
}

// 64 positions per iteration with AVX-512 (BW + VL; the hosts of MI355X nodes are Zen 4 / Zen 5 or Sapphire Rapids):
// the same two multiply-adds fold the codes, vpmovdb gathers the sixteen result bytes and one 128-bit shuffle puts
// them in the order of the four code words; validity is ONE table lookup (the low nibble of an upper-cased base picks
// the letter that has it: 1 A, 3 C, 4 T, 7 G) and one compare into a mask register, on the byte-reversed lanes so
// that the first base of a word is its bit 15.  ~17 micro-ops per 64 bytes (AVX2: 22 per 32).
__attribute__((target("avx512f,avx512bw,avx512vl"))) static void pack_block_avx512(const uint8_t *src, size_t n, uint32_t *code, uint16_t *inv)
{
  const __m512i m3 = _mm512_set1_epi8(3), mdf = _mm512_set1_epi8((char)0xDF), m0f = _mm512_set1_epi8(0x0F);
  const __m512i w1 = _mm512_set1_epi16(0x0104), w2 = _mm512_set1_epi32(0x00010010);
  const __m128i bswap = _mm_setr_epi8(3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 15, 14, 13, 12);
  const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8((char)0xFF /* never equal: a NUL byte is no base */, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0));
  const __m512i rev = _mm512_broadcast_i32x4(_mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0));
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const __m512i v = _mm512_loadu_si512((const void *)(src + i));
    const __m512i t = _mm512_ternarylogic_epi64(_mm512_srli_epi16(v, 1), _mm512_srli_epi16(v, 2), m3, 0x28);  // (a ^ b) & c
    const __m512i byt = _mm512_madd_epi16(_mm512_maddubs_epi16(t, w1), w2);  // 32-bit lanes: four bases, the first on top
    _mm_storeu_si128((__m128i *)(code + i / 16), _mm_shuffle_epi8(_mm512_cvtepi32_epi8(byt), bswap));
    const __m512i u = _mm512_shuffle_epi8(_mm512_and_si512(v, mdf), rev);   // upper-cased, each 16-byte lane reversed
    const __mmask64 ok = _mm512_cmpeq_epi8_mask(_mm512_shuffle_epi8(lut, _mm512_and_si512(u, m0f)), u);
    const uint64_t bad = ~(uint64_t)ok;  // bit 16 g + 15 - j = byte j of lane g is not a base
    memcpy(inv + i / 16, &bad, 8);
  }
  if (i < n) pack_block_avx2(src + i, n - i, code + i / 16, inv + i / 16);  // (n is a multiple of 32)
}
#endif

// ---- reads -> packed stream in ONE pass (no intermediate ASCII block) ------------------------------
// What add_reads_packed() stages is the virtual stream "read, separator, read, separator, ..." in packed form.
// Until round 4 a staging thread first assembled 16 KiB of that stream as ASCII (a memcpy per read) and then packed
// the block: two passes over every base and a libc call per 150-base read -- 3.5 GB/s of bases per core, and the
// hosts of the GPU boxes give a job 16 cores' worth of time (cgroup cpu.max).  Here the reads are packed straight
// from where they lie (pack_reads_avx512).
struct PackReads {
  const uint8_t *bases;  // all reads, concatenated
  const uint64_t *off;   // off[r0 + i] .. off[r0 + i + 1]: read i of the chunk
  uint64_t r0, nreads;   // reads of the chunk (a piece of one long read: nreads = 1 with piece_from / piece_len)
  uint64_t pos0;         // position of the first base of read 0 (kCarry)
  bool piece;            // the chunk holds `piece_len` bases from bases + piece_from (and a separator if there is room)
  uint64_t piece_from, piece_len;
  uint64_t start_of(uint64_t i) const { return piece ? pos0 : pos0 + (off[r0 + i] - off[r0]) + i; }
  uint64_t len_of(uint64_t i) const { return piece ? piece_len : off[r0 + i + 1] - off[r0 + i]; }
  const uint8_t *ptr_of(uint64_t i) const { return piece ? bases + piece_from : bases + off[r0 + i]; }
};

#if defined(__x86_64__)
// positions [p_lo, p_hi) of the chunk's stream (both multiples of 64) -> code[p / 16], inv[p / 16].
// The reads lie back to back in `bases`, so the 64 positions of a block are 64 - m CONSECUTIVE source bytes with m
// separators dropped in where reads end: one load, one byte-expand under the mask of the non-separator lanes
// (vpexpandb, AVX-512 VBMI2) onto a vector of separators, then the same ~17 micro-ops as pack_block_avx512.  The
// only per-read work is one bit of the separator mask (150-base reads: 0.42 per block).
__attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi2,popcnt"))) static void pack_reads_avx512(const PackReads &R, uint64_t p_lo, uint64_t p_hi, uint32_t *code, uint16_t *inv)
{
  const __m512i m3 = _mm512_set1_epi8(3), mdf = _mm512_set1_epi8((char)0xDF), m0f = _mm512_set1_epi8(0x0F), nl = _mm512_set1_epi8('\n');
  const __m512i w1 = _mm512_set1_epi16(0x0104), w2 = _mm512_set1_epi32(0x00010010);
  const __m128i bswap = _mm_setr_epi8(3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 15, 14, 13, 12);
  const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8((char)0xFF, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0));
  const __m512i rev = _mm512_broadcast_i32x4(_mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0));
  constexpr uint64_t kNever = ~0ULL;
  const uint64_t n = R.nreads;
  // the read that holds position p_lo or whose separator it is (reads before it end before p_lo), and the source
  // byte that belongs at p_lo (reads are contiguous: position - pos0 - separators so far)
  uint64_t i = 0;
  if (n) {
    uint64_t lo = 0, hi = n;
    while (lo + 1 < hi) { const uint64_t mid = (lo + hi) / 2; if (R.start_of(mid) <= p_lo) lo = mid; else hi = mid; }
    i = lo;
  }
  const uint64_t end_pos = n ? R.start_of(n - 1) + R.len_of(n - 1) + 1 : R.pos0;  // first position behind the last separator
  uint64_t next_sep = n ? R.start_of(i) + R.len_of(i) : kNever;
  while (next_sep < p_lo) { i++; next_sep = i < n ? R.start_of(i) + R.len_of(i) : kNever; }  // (p_lo behind the last read)
  const uint8_t *src = n ? R.ptr_of(0) + (std::min(p_lo, end_pos) - R.pos0 - std::min(i, n)) : nullptr;
  const uint8_t *const src_begin = n ? R.ptr_of(0) : nullptr;                                  // the chunk's bases:
  const uint8_t *const src_end = n ? src_begin + (end_pos - R.pos0 - n) : nullptr;             // [src_begin, src_end)
  for (uint64_t P = p_lo; P < p_hi; P += 64) {
    uint64_t sepm = 0;  // lanes that hold a separator
    while (next_sep < P + 64) {
      sepm |= 1ULL << (next_sep - P);
      i++;
      next_sep = i < n ? R.start_of(i) + R.len_of(i) : kNever;
    }
    if (P + 64 > end_pos) sepm |= end_pos <= P ? ~0ULL : ~0ULL << (end_pos - P);  // padding behind the last read
    if (P < R.pos0) sepm |= R.pos0 >= P + 64 ? ~0ULL : (1ULL << (R.pos0 - P)) - 1;  // (the carry's positions: not ours)
    const unsigned nsrc = 64u - (unsigned)__builtin_popcountll(sepm);
    __m512i v = nl;
    // The masked load + byte expand costs 2.8x the rest of the block on the boxes' hosts (Zen 5: 9.0 against 25 GB/s
    // of bases per thread, tools/pk/pack_variants.cpp, round 6), and most blocks do not need it: no separator at all
    // (58 % of the blocks of 150-base reads) is one plain load, exactly one separator is two plain loads one byte apart
    // and two blends.  Plain loads read 64 bytes whatever is needed: only where the source has them.
    if (sepm == 0 && src + 64 <= src_end) {
      v = _mm512_loadu_si512((const void *)src);
      src += 64;
    } else if (sepm && !(sepm & (sepm - 1)) && src > src_begin && src + 64 <= src_end) {
      const __mmask64 from = (__mmask64)(~0ULL << __builtin_ctzll(sepm));  // the separator's lane and the ones behind it
      v = _mm512_mask_blend_epi8(from, _mm512_loadu_si512((const void *)src), _mm512_loadu_si512((const void *)(src - 1)));
      v = _mm512_mask_blend_epi8((__mmask64)sepm, v, nl);
      src += 63;
    } else if (nsrc) {
      const __m512i x = _mm512_maskz_loadu_epi8(nsrc == 64 ? ~0ULL : (1ULL << nsrc) - 1, (const void *)src);
      v = _mm512_mask_expand_epi8(nl, (__mmask64)~sepm, x);
      src += nsrc;
    }
    const __m512i t = _mm512_ternarylogic_epi64(_mm512_srli_epi16(v, 1), _mm512_srli_epi16(v, 2), m3, 0x28);  // (a ^ b) & c
    const __m512i byt = _mm512_madd_epi16(_mm512_maddubs_epi16(t, w1), w2);  // 32-bit lanes: four bases, the first on top
    _mm_storeu_si128((__m128i *)(code + P / 16), _mm_shuffle_epi8(_mm512_cvtepi32_epi8(byt), bswap));
    const __m512i u = _mm512_shuffle_epi8(_mm512_and_si512(v, mdf), rev);   // upper-cased, each 16-byte lane reversed
    const uint64_t bad = ~(uint64_t)_mm512_cmpeq_epi8_mask(_mm512_shuffle_epi8(lut, _mm512_and_si512(u, m0f)), u);
    memcpy(inv + P / 16, &bad, 8);
  }
}
#endif

static bool pack_reads_available()
{
#if defined(__x86_64__)
  static const bool ok = __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512vbmi2") &&
                         !getenv("MCX_NO_AVX512") && !getenv("MCX_NO_AVX2") && !(getenv("MCX_FUSED_PACK") && atoi(getenv("MCX_FUSED_PACK")) == 0);
  return ok;
#else
  return false;
#endif
}

// n is a multiple of 32
static void pack_block(const uint8_t *src, size_t n, uint32_t *code, uint16_t *inv)
{
#if defined(__x86_64__)
  static const int level = getenv("MCX_NO_AVX2") ? 0 : (__builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") && !getenv("MCX_NO_AVX512")) ? 2
                           : __builtin_cpu_supports("avx2") ? 1 : 0;
  if (level == 2) { pack_block_avx512(src, n, code, inv); return; }
  if (level == 1) { pack_block_avx2(src, n, code, inv); return; }
#endif
  for (size_t i = 0; i < n; i += 16) pack16_swar(src + i, code + i / 16, inv + i / 16);
}

// test hook: the reads as the stream "128 separators, read, separator, read, separator, ..., padding to a multiple of 64"
// -> code[p / 16], inv[p / 16] for p < the returned number of positions; fused != 0: the one-pass packer (returns 0 if
// this host cannot run it), else assemble the ASCII stream and pack it block by block
extern "C" uint64_t mcx_pack_reads_host(const uint8_t *bases, const uint64_t *off, uint64_t nreads, uint32_t *code, uint16_t *inv, uint64_t cap_pos, int fused)
{
  const uint64_t L = nreads ? off[nreads] - off[0] + nreads : 0, total = kCarry + ((L + 63) & ~63ull);
  if (total > cap_pos) return 0;
  if (fused) {
    if (!pack_reads_available()) return 0;
#if defined(__x86_64__)
    uint8_t sep[kCarry];
    memset(sep, '\n', sizeof(sep));
    pack_block(sep, (size_t)kCarry, code, inv);
    const PackReads PR{bases, off, 0, nreads, kCarry, false, 0, 0};
    // in uneven pieces, as the staging threads take them
    uint64_t p = kCarry;
    for (uint64_t step = 64; p < total; step = total > (1u << 24) ? (1u << 18) : step * 3 % 8191 / 64 * 64 + 64) {
      const uint64_t q = std::min(total, p + step);
      pack_reads_avx512(PR, p, q, code, inv);
      p = q;
    }
#endif
    return total;
  }
  std::vector<uint8_t> txt(total, (uint8_t)'\n');
  for (uint64_t i = 0, p = kCarry; i < nreads; i++) {
    const uint64_t len = off[i + 1] - off[i];
    memcpy(txt.data() + p, bases + off[i], len);
    p += len + 1;
  }
  pack_block(txt.data(), (size_t)total, code, inv);
  return total;
}

extern "C" void mcx_pack_bases(const uint8_t *src, uint64_t n, uint32_t *code, uint16_t *inv)
{ /* test hook: n multiple of 32 */
  pack_block(src, (size_t)n, code, inv);
}

// threads that pack / copy reads into the pinned staging buffer (MCX_STAGE_THREADS, default: half the cores, at most 24:
// 16 -> 24 threads was +3 % on one host and +40 % on another, 32 threads -30 % on the first; round 2 / round 3 logs)
static int stage_threads()
{
  static const int n = [] {
    const char *e = getenv("MCX_STAGE_THREADS");
    const int hw = (int)std::thread::hardware_concurrency();
    const int v = e ? atoi(e) : std::max(1, std::min(24, hw / 2));
    return v < 1 ? 1 : v > 64 ? 64 : v;
  }();
  return n;
}

// where the host entry's wall clock goes (MCX_STAGE_TIMING=1: printed when the process ends)
struct StageTiming {
  double wait = 0, prep = 0, pack = 0, submit = 0; uint64_t chunks = 0, flushes = 0;
  ~StageTiming() { if (chunks && getenv("MCX_STAGE_TIMING")) fprintf(stderr, "[stage] %llu chunks: wait for buffer %.1f ms, prepare %.1f ms, pack %.1f ms, copy + launches %.1f ms (%llu background flushes)\n",
                                          (unsigned long long)chunks, wait * 1e3, prep * 1e3, pack * 1e3, submit * 1e3, (unsigned long long)flushes); }
};
static StageTiming g_stage_timing;

// Host-fed builds: when the device has caught up with the host (nothing queued on the graph's stream
// at the moment a packed chunk is about to be copied) the host is the bottleneck -- parsing, packing,
// PCIe -- and the device would sit idle until the last batch, then flush everything while the host
// waits.  An idle device therefore flushes ONE group of regions (split + LDS insert of 1 / 16 of the
// table on the bench shape, about as long as the host needs for its next chunk) and moves on to the
// next group the next time it is idle: the flush becomes a background pass that fills the idle time,
// and what is left for the closing flush is what arrived during the last turn.  (A whole flush at
// once was tried first: 15 ms during which the two staging buffers run dry and the HOST waits --
// 35.6 -> 27 G k-mers/s.)  A device that is the bottleneck is never idle here and keeps its large
// flushes.  One-colour graphs on one device only; MCX_IDLE_FLUSH=0 switches it off.
static int flush_if_device_idle(mcx_graph *g, int starved = -1 /* -1: ask the stream; 0 / 1: the caller knows (add_reads_packed) */)
{
  // 2: take every chance (tests); 3: every chance once 3/4 of the flush size is buffered (tests: a device that was
  // busy first, so that the idle flushes start on a nearly full workspace)
  static const int mode_env = [] { const char *e = getenv("MCX_IDLE_FLUSH"); return e ? atoi(e) : 1; }();
  if (mode_env == 3 && !(g->idle_base || g->pending >= g->defer_tuples / 4 * 3)) return MCX_OK;
  const int mode = mode_env == 3 ? 2 : mode_env;
  const bool on = mode != 0;
  if (!on || !g->defer || g->group || g->nsets != 1 || !g->pending || g->pending_l2 || g->set_colour.empty() || g->set_colour[0] < 0) return MCX_OK;
  const uint32_t G = std::min(flush_group(g), g->l2_regions);
  if (G >= g->b1) return MCX_OK;  // one group = the whole table: nothing incremental about it
  const uint32_t ngroups = (g->b1 + G - 1) / G;
  // worth a group's table pass: its share of 1 / 8 of the flush size, at least 16 M occurrences
  if (mode != 2 && g->pending / ngroups < std::max<uint64_t>(g->defer_tuples / 8 / ngroups, 1ull << 24)) return MCX_OK;
  if (mode != 2 && starved == 0) return MCX_OK;
  if (mode != 2 && starved < 0 && hipStreamQuery(g->stream) != hipSuccess) { (void)hipGetLastError(); return MCX_OK; }  // busy: the device is not waiting for us
  const uint32_t r0 = g->idle_next * G < g->b1 ? g->idle_next * G : 0;
  g->idle_next = (r0 / G + 1) % ngroups;
  g_stage_timing.flushes++;
  const uint32_t ng = std::min(G, g->b1 - r0);
  const int colour = g->set_colour[0];
  g->l2_off = 0;
  TupleIn in{g->l1_keys + (uint64_t)r0 * g->cap1 * g->W, nullptr, g->l1_cnt + r0, g->cap1, ng * g->rep1, ng, g->b1};
  BinSpec bs{BIN_SUBLOCAL, 0, g->subs_per_bin, 1, ng * g->subs_per_bin, ng, 0, r0};
  BinOut out{g->l2_keys, nullptr, g->l2_cnt, g->cap2, nullptr, nullptr, nullptr, 0};
  DISPATCH_WC(g, launch_split_regions, g, in, colour, bs, out);
  HIP_TRY(hipGetLastError());
  DISPATCH_WC(g, launch_lds_insert_t, g, colour, r0 * g->subs_per_bin, ng * g->subs_per_bin);
  HIP_TRY(hipGetLastError());
  // the group's L1 bins are empty again: its counters in all replicas with one call (rows of ng counters, b1 apart)
  HIP_TRY(hipMemset2DAsync(g->l1_cnt + r0, (size_t)g->b1 * 8, 0, (size_t)ng * 8, g->rep1, g->stream));
  // The bound on what is buffered.  A region group that has not been emptied since A_i occurrences had been
  // handed over holds its share of A - A_i; the flush trigger (defer_reserve) compares `pending` with the
  // capacity the segments were sized for, so `pending` must stay A - min_i(A_i): it only drops once the group
  // that has waited longest is emptied.  (Until round 3 a flushed group's share was subtracted at once: after
  // the first idle flush of a nearly full workspace the other groups could then be filled beyond their
  // segments -- correct through the direct-insert fallback, but slow and silent.)
  if (g->idle_mark.size() != ngroups) g->idle_mark.assign(ngroups, g->idle_base);
  g->idle_mark[r0 / G] = g->idle_base + g->pending;
  const uint64_t base = *std::min_element(g->idle_mark.begin(), g->idle_mark.end());
  const uint64_t freed = base - g->idle_base;
  g->idle_base = base;
  g->pending -= std::min(g->pending, freed);
  g->set_pending[0] -= std::min(g->set_pending[0], freed);
  return MCX_OK;
}

// Layout of a staging buffer.  ASCII chunks (MCX_PACKED=0, the fallback): kCarry + kStageBytes stream bytes, then the
// staged offsets.  Packed chunks (the default): code words, invalid flags, offsets back to back -- 112 MB instead of
// 192 MB at the default chunk size, and three of them are page-locked on the first call (hipHostMalloc pins about
// 4 GB/s: 60 of `prepare`'s 150-200 ms on the CLI's critical path before the first batch can be submitted; round 5).
static bool stage_packed()
{
  static const bool packed = [] { const char *e = getenv("MCX_PACKED"); return !e || atoi(e) != 0; }();
  return packed;
}
static uint64_t stage_inv_at() { return (((kCarry + kStageBytes + 64) / 16 * 4) + 63) & ~63ull; }  // packed: byte offset of the invalid flags
static uint64_t stage_off_region()
{
  if (!stage_packed()) return kCarry + kStageBytes + 256;
  return (stage_inv_at() + (kCarry + kStageBytes + 64) / 16 * 2 + 255) & ~255ull;
}

static int ensure_stage(mcx_graph *g)
{
  if (g->stage_alloc) return MCX_OK;
  // stream bytes + worst-case one offset per 2 bytes would be silly; offsets are
  // staged in a second region sized for reads of >= 15 bytes on average and the
  // filler stops a chunk when either region is full.
  const uint64_t bytes = stage_off_region() + (kStageBytes / 16 + 2) * sizeof(uint64_t);
  for (int i = 0; i < mcx_graph::kStageBufs; i++) {  // (per resource: a call that failed part-way is continued, not repeated)
    if (!g->h_stage[i]) HIP_TRY(hipHostMalloc((void **)&g->h_stage[i], bytes, hipHostMallocDefault));
    if (!g->d_stage[i]) HIP_TRY(hipMalloc((void **)&g->d_stage[i], bytes));
    if (!g->ev[i]) HIP_TRY(hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming));
    if (!g->ev_copy[i]) HIP_TRY(hipEventCreateWithFlags(&g->ev_copy[i], hipEventDisableTiming));
    if (!g->ev_wait0[i]) HIP_TRY(hipEventCreate(&g->ev_wait0[i]));
    if (!g->ev_wait1[i]) HIP_TRY(hipEventCreate(&g->ev_wait1[i]));
  }
  if (!g->cstream) HIP_TRY(hipStreamCreateWithFlags(&g->cstream, hipStreamNonBlocking));
  g->stage_alloc = bytes;
  return MCX_OK;
}

static int add_reads_qh(mcx_graph *g, int colour, const uint8_t *bases, const uint8_t *quals,
                        const uint64_t *off, uint64_t nreads, uint8_t fq, uint8_t hp);
static int add_reads_must_exist(mcx_graph *g, int colour, const uint8_t *bases, const uint8_t *quals,
                                const uint64_t *off, uint64_t nreads, uint8_t fq, uint8_t hp);

// Host reads -> packed stream chunks in pinned memory -> HBM -> the front end.  A chunk is the same
// virtual stream the ASCII path stages (128 positions of carry from the previous chunk, then whole
// reads each followed by one separator, or a piece of a read longer than a chunk), padded with
// separators to a multiple of 64 positions, but it travels as 3 bits per position (pack_block).
// The staging threads each assemble and pack blocks of 64-aligned positions; nothing else touches
// the bases on the host.
static int add_reads_packed(mcx_graph *g, int colour, const uint8_t *bases, const uint64_t *off, uint64_t nreads,
                            unsigned char *d_flags)
{
  const uint64_t off_region = stage_off_region();           // byte offset of the staged offsets (8-aligned)
  const uint64_t max_offs = kStageBytes / 16;
  const uint64_t cap_pos = kCarry + kStageBytes + 64;       // positions a chunk may hold
  const uint64_t inv_at = stage_inv_at();                   // byte offset of the invalid flags inside a staging buffer
  uint32_t carry_code[kCarry / 16];
  uint16_t carry_inv[kCarry / 16];
  for (uint64_t i = 0; i < kCarry / 16; i++) { carry_code[i] = 0; carry_inv[i] = 0xFFFF; }
  uint64_t r = 0, r_pos = 0;  // next read, bytes of it already staged
  const int T = stage_threads();
  const bool fused = pack_reads_available();

  // One chunk on its way through plan -> pack -> submit.  Two are alive at a time (round 4): while the caller
  // enqueues chunk n (copies, background flush, kernels: ~0.15 ms of HIP calls) and waits for the staging pair of
  // chunk n + 1 to come free, the pool already packs chunk n + 1 -- the packing threads used to idle through both.
  struct Job {
    int b = 0;
    uint8_t *hs = nullptr;
    uint32_t *hcode = nullptr;
    uint16_t *hinv = nullptr;
    uint64_t *hoff = nullptr;
    uint64_t r0 = 0, nwhole = 0, L = 0, Lp = 0, total = 0;
    long long piece_of = -1;
    uint64_t piece_from = 0, piece_data = 0;  // a piece: where its bases start in `bases`, how many there are
    std::atomic<uint64_t> next_run{0};
    std::function<void(int)> fn;
    bool async = false;
  } jobs[2];

  // plan: the staging pair, the reads (or the piece of a long read) the chunk holds
  auto plan = [&](Job &J) -> int {
    J.b = g->cur;
    g->cur = (g->cur + 1) % mcx_graph::kStageBufs;
    const double tq0 = now_s();
    HIP_TRY(hipEventSynchronize(g->ev[J.b]));  // previous use of this pair finished
    g_stage_timing.wait += now_s() - tq0;
    J.hs = g->h_stage[J.b];
    J.hcode = reinterpret_cast<uint32_t *>(J.hs);
    J.hinv = reinterpret_cast<uint16_t *>(J.hs + inv_at);
    J.hoff = reinterpret_cast<uint64_t *>(J.hs + off_region);
    J.r0 = r;
    J.nwhole = 0; J.L = 0; J.piece_of = -1; J.piece_from = 0; J.piece_data = 0;
    // whole reads that fit: read i lands at position kCarry + (off[r0 + i] - off[r0]) + i (each is
    // followed by one separator), which grows with i: the count is found by bisection, not by a walk
    if (r_pos == 0) {
      const uint64_t r0 = J.r0;
      uint64_t lo = 0, hi = std::min<uint64_t>(nreads - r0, max_offs);  // largest n with n reads taking <= kStageBytes positions
      auto need = [&](uint64_t n) { return off[r0 + n] - off[r0] + n; };
      if (need(hi) <= kStageBytes) lo = hi;
      else while (lo + 1 < hi) { const uint64_t mid = (lo + hi) / 2; if (need(mid) <= kStageBytes) lo = mid; else hi = mid; }
      J.nwhole = lo;
      J.L = need(lo);
    }
    r += J.nwhole;
    if (J.nwhole == 0) {  // a read longer than a chunk (or its tail): a piece of it on its own
      const uint64_t len = off[r + 1] - off[r], remain = len - r_pos;
      uint64_t take = std::min(remain, kStageBytes - 64);
      if (take < remain) take &= ~63ull;  // pieces end on a 64-position boundary: no padding inside a read
      J.piece_of = (long long)r;
      J.piece_from = off[r] + r_pos;
      J.piece_data = take;
      J.L = take;
      r_pos += take;
      if (r_pos == len) { J.L += 1; r++; r_pos = 0; }  // its separator
    }
    J.hoff[J.nwhole] = kCarry + J.L;
    J.Lp = (J.L + 63) & ~63ull;
    J.total = kCarry + J.Lp;
    return MCX_OK;
  };

  // pack: the chunk's stream, position p >= kCarry = base of a read or a separator, as code words + invalid flags.
  // The threads take runs of kRun blocks from a shared counter (not a fixed share each): on a host that is shared
  // with other jobs a thread that loses its core for a moment would otherwise hold up the whole chunk.
  auto pack_begin = [&](Job &J) {
    memcpy(J.hcode, carry_code, sizeof(carry_code));  // (the previous chunk's last positions: its packing has finished)
    memcpy(J.hinv, carry_inv, sizeof(carry_inv));
    J.next_run.store(0, std::memory_order_relaxed);
    Job *jp = &J;
    J.fn = [jp, bases, off, T, fused](int ti) {
      Job &J = *jp;
      constexpr uint64_t BLK = 16384, kRun = 16;
      const uint64_t r0 = J.r0, nwhole = J.nwhole, Lp = J.Lp;
      const uint64_t nblk = (Lp + BLK - 1) / BLK;
      auto start_of = [&](uint64_t i) { return kCarry + (off[r0 + i] - off[r0]) + i; };  // first position of whole read i
      // the staged offsets of this thread's share of the reads (k_read_flags_packed reads them)
      for (uint64_t q = nwhole * (uint64_t)ti / (uint64_t)T, qe = nwhole * (uint64_t)(ti + 1) / (uint64_t)T; q < qe; q++) J.hoff[q] = start_of(q);
#if defined(__x86_64__)
      if (fused) {  // one pass from the reads to the packed chunk (pack_reads_avx512)
        const PackReads PR{bases, off, r0, J.piece_of >= 0 ? 1 : nwhole, kCarry, J.piece_of >= 0, J.piece_from, J.piece_data};
        for (;;) {
          const uint64_t b_lo = J.next_run.fetch_add(kRun, std::memory_order_relaxed), b_hi = std::min(nblk, b_lo + kRun);
          if (b_lo >= nblk) break;
          pack_reads_avx512(PR, kCarry + b_lo * BLK, std::min(kCarry + b_hi * BLK, kCarry + Lp), J.hcode, J.hinv);
        }
        return;
      }
#endif
      uint8_t buf[BLK];  // hosts without AVX-512 VBMI2: assemble 16 KiB of the stream as ASCII, pack the block
      for (;;) {
        const uint64_t b_lo = J.next_run.fetch_add(kRun, std::memory_order_relaxed), b_hi = std::min(nblk, b_lo + kRun);
        if (b_lo >= nblk) break;
        uint64_t i = 0;  // first read that reaches into the run's range of positions
        if (J.piece_of < 0 && nwhole) {
          const uint64_t p0 = kCarry + b_lo * BLK;
          uint64_t lo = 0, hi = nwhole;  // last read that starts at or before p0
          while (lo + 1 < hi) { const uint64_t mid = (lo + hi) / 2; if (start_of(mid) <= p0) lo = mid; else hi = mid; }
          i = lo;
        }
        for (uint64_t bk = b_lo; bk < b_hi; bk++) {
          const uint64_t p0 = kCarry + bk * BLK, n = std::min(BLK, kCarry + Lp - p0);
          if (J.piece_of >= 0) {
            const uint64_t at = p0 - kCarry, data = at < J.piece_data ? std::min(n, J.piece_data - at) : 0;
            memcpy(buf, bases + J.piece_from + at, data);
            memset(buf + data, '\n', n - data);  // the read's separator (if it ends here) and the padding
          } else {
            uint64_t p = p0;
            while (p < p0 + n) {
              if (i >= nwhole) { memset(buf + (p - p0), '\n', p0 + n - p); break; }
              const uint64_t s_ = start_of(i), len = off[r0 + i + 1] - off[r0 + i];
              if (p < s_ + len) {
                const uint64_t cnt = std::min(s_ + len - p, p0 + n - p);
                memcpy(buf + (p - p0), bases + off[r0 + i] + (p - s_), cnt);
                p += cnt;
              } else {  // the read's separator, then the next read
                buf[p - p0] = '\n';
                p++;
                i++;
              }
            }
          }
          pack_block(buf, (size_t)n, J.hcode + p0 / 16, J.hinv + p0 / 16);
        }
      }
    };
    J.async = T > 1 && J.Lp >= (1u << 20);
    if (J.async) StagePool::get().begin(T, J.fn);
    else for (int ti = 0; ti < T; ti++) J.fn(ti);  // small chunk: every share on this thread
  };
  auto pack_end = [&](Job &J) {
    if (J.async) { StagePool::get().finish(); J.async = false; }
    memcpy(carry_code, J.hcode + J.total / 16 - kCarry / 16, sizeof(carry_code));
    memcpy(carry_inv, J.hinv + J.total / 16 - kCarry / 16, sizeof(carry_inv));
  };
  // (an error return must not leave pool threads working on this frame)
  struct Drain { Job *jobs; ~Drain() { for (int i = 0; i < 2; i++) if (jobs[i].async) { StagePool::get().finish(); jobs[i].async = false; } } } drain{jobs};

  // submit: copies on the copy stream, background flush if the device was starved, kernels
  auto submit = [&](Job &J) -> int {
    const int b = J.b;
    uint8_t *ds = g->d_stage[b];
    // The chunk travels on the copy stream; the graph's stream waits for it between two timing events.  How long the
    // previous use of this buffer pair made the compute stream wait (the events' distance) says whether the device
    // is starved by the host side -- packing or PCIe: then one region group is flushed in that idle time, ahead of
    // the wait.  (Until round 4 the copies ran on the graph's stream and "idle" was hipStreamQuery of it: with PCIe
    // as the bottleneck that stream is never empty, no background flush ever ran, and the whole flush -- 39 ms for
    // the bench's 6 G occurrences -- came after the last chunk.)
    HIP_TRY(hipMemcpyAsync(ds, J.hcode, J.total / 16 * 4, hipMemcpyHostToDevice, g->cstream));
    HIP_TRY(hipMemcpyAsync(ds + inv_at, J.hinv, J.total / 16 * 2, hipMemcpyHostToDevice, g->cstream));
    if (J.nwhole)
      HIP_TRY(hipMemcpyAsync(ds + off_region, J.hoff, (J.nwhole + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, g->cstream));
    HIP_TRY(hipEventRecord(g->ev_copy[b], g->cstream));
    {
      int starved = hipStreamQuery(g->stream) == hipSuccess ? 1 : 0;  // nothing queued at all: certainly waiting for us
      (void)hipGetLastError();
      float wait_ms = 0.f;
      if (!starved && g->ev_wait_used[b] && hipEventElapsedTime(&wait_ms, g->ev_wait0[b], g->ev_wait1[b]) == hipSuccess) starved = wait_ms > 0.15f;
      (void)hipGetLastError();
      int rc_ = flush_if_device_idle(g, starved);
      if (rc_ != MCX_OK) return rc_;
    }
    HIP_TRY(hipEventRecord(g->ev_wait0[b], g->stream));
    HIP_TRY(hipStreamWaitEvent(g->stream, g->ev_copy[b], 0));
    HIP_TRY(hipEventRecord(g->ev_wait1[b], g->stream));
    g->ev_wait_used[b] = true;
    StreamLaunch SL{nullptr, J.total, kCarry - (uint64_t)g->k, J.total - (uint64_t)g->k, J.piece_of >= 0 ? d_flags + J.piece_of : nullptr,
                    reinterpret_cast<const uint32_t *>(ds), reinterpret_cast<const uint16_t *>(ds + inv_at)};
    int rc = submit_stream(g, SL, colour);  // (polls the table's "full" flag: full_poll)
    // whatever happens, the staging buffer's event follows the copies already queued on it: a caller that goes on after
    // an error must not reuse the buffer unordered
    if (rc != MCX_OK) { (void)hipEventRecord(g->ev[b], g->stream); return rc; }
    HIP_TRY(hipSetDevice(g->device));
    if (J.nwhole) {
      hipLaunchKernelGGL(k_read_flags_packed, dim3((unsigned)((J.nwhole + 255) / 256)), dim3(256), 0, g->stream,
                         reinterpret_cast<const uint16_t *>(ds + inv_at), (const uint64_t *)(ds + off_region), J.nwhole, g->k, d_flags + J.r0);
      if (hipGetLastError() != hipSuccess) { (void)hipEventRecord(g->ev[b], g->stream); return fail(MCX_ERR_HIP, "k_read_flags_packed launch failed"); }
    }
    HIP_TRY(hipEventRecord(g->ev[b], g->stream));
    g_stage_timing.chunks++;
    return MCX_OK;
  };

  if (!nreads) return MCX_OK;
  int cur = 0, rc = plan(jobs[0]);
  if (rc != MCX_OK) return rc;
  pack_begin(jobs[0]);
  for (;;) {
    Job &J = jobs[cur];
    const double t0 = now_s();
    pack_end(J);  // (its last positions are the next chunk's carry)
    const double t1 = now_s();
    g_stage_timing.pack += t1 - t0;
    const bool more = r < nreads;
    if (more) {
      rc = plan(jobs[cur ^ 1]);
      if (rc != MCX_OK) return rc;
      pack_begin(jobs[cur ^ 1]);
    }
    const double t2 = now_s();
    rc = submit(J);
    if (rc != MCX_OK) return rc;
    g_stage_timing.submit += now_s() - t2;
    if (!more) break;
    cur ^= 1;
  }
  return MCX_OK;
}

extern "C" int mcx_graph_add_reads(mcx_graph *g, int colour, const uint8_t *bases, const uint8_t *quals,
                                   const uint64_t *off, uint64_t nreads, uint8_t fq_cutoff_abs,
                                   uint8_t hp_cutoff, mcx_load_stats *stats_accum)
{
  if (g && g->as_group) {
    if (colour < 0 || colour >= g->ncols) return fail(MCX_ERR_ARG, "colour %d out of range", colour);
    if (nreads && (!bases || !off)) return fail(MCX_ERR_ARG, "null read buffers");
    if (nreads && colour >= g->ncols_vis) return fail(MCX_ERR_ARG, "colour %d is the intersection colour", colour);
    if (grp_part(g->as_group, 0)->must_exist)
      return grp_add_reads_must_exist(g->as_group, colour, bases, quals, off, nreads, fq_cutoff_abs, hp_cutoff, stats_accum);
    return grp_add_reads(g->as_group, colour, bases, quals, off, nreads, fq_cutoff_abs, hp_cutoff, stats_accum);
  }
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (colour < 0 || colour >= g->ncols) return fail(MCX_ERR_ARG, "colour %d out of range", colour);
  if (nreads && (!bases || !off)) return fail(MCX_ERR_ARG, "null read buffers");
  HIP_TRY(hipSetDevice(g->device));
  if (stats_accum) {
    stats_accum->num_se_reads += nreads;
    stats_accum->total_bases_read += nreads ? off[nreads] - off[0] : 0;
  }
  if (!nreads) return MCX_OK;
  if (colour >= g->ncols_vis) return fail(MCX_ERR_ARG, "colour %d is the intersection colour", colour);
  if (g->must_exist) return add_reads_must_exist(g, colour, bases, quals, off, nreads, fq_cutoff_abs, hp_cutoff);
  if ((fq_cutoff_abs > 0 && quals) || hp_cutoff > 0)
    return add_reads_qh(g, colour, bases, quals, off, nreads, fq_cutoff_abs, hp_cutoff);

  int rc = ensure_stage(g);
  if (rc != MCX_OK) return rc;

  unsigned char *d_flags = nullptr;
  HIP_TRY(hipMallocAsync((void **)&d_flags, nreads, g->stream));
  HIP_TRY(hipMemsetAsync(d_flags, 0, nreads, g->stream));

  const bool packed = stage_packed();
  if (packed) {
    rc = add_reads_packed(g, colour, bases, off, nreads, d_flags);
    if (rc != MCX_OK) { (void)hipFreeAsync(d_flags, g->stream); return rc; }  // (stream-ordered: behind the kernels that write the flags)
    hipLaunchKernelGGL(k_count_flags, dim3(256), dim3(256), 0, g->stream, (const unsigned char *)d_flags, nreads, g->d_ctr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipFreeAsync(d_flags, g->stream));
    return MCX_OK;
  }

  const uint64_t off_region = stage_off_region();  // byte offset of the staged offsets (8-aligned)
  const uint64_t max_offs = kStageBytes / 16;
  uint64_t r = 0, r_pos = 0;  // next read, bytes of it already staged
  uint8_t carry[kCarry];
  memset(carry, '\n', kCarry);
  while (r < nreads) {
    const int b = g->cur;
    g->cur = (g->cur + 1) % mcx_graph::kStageBufs;
    HIP_TRY(hipEventSynchronize(g->ev[b]));  // previous use of this buffer finished
    uint8_t *hs = g->h_stage[b];
    uint64_t *hoff = reinterpret_cast<uint64_t *>(hs + off_region);
    memcpy(hs, carry, kCarry);
    uint64_t L = 0, nwhole = 0;
    const uint64_t r0 = r;
    long long piece_of = -1;  // >= 0: chunk holds a piece of this (long) read only
    if (r_pos == 0 && stage_threads() > 1) {
      // Whole reads that fit this chunk, found without copying; then a few threads copy their
      // share (read r lands at (off[r] - off[r0]) + (r - r0): every read is followed by one
      // separator).  One thread moves ~10 GB/s into pinned memory, PCIe takes ~25 GB/s.
      uint64_t nf = 0, bytes = 0;
      while (r0 + nf < nreads && nf < max_offs) {
        const uint64_t len = off[r0 + nf + 1] - off[r0 + nf];
        if (bytes + len + 1 > kStageBytes) break;
        bytes += len + 1;
        nf++;
      }
      if (nf >= 4096) {
        const int T = stage_threads();
        auto work = [&](int ti) {
          const uint64_t lo = nf * (uint64_t)ti / (uint64_t)T, hi = nf * (uint64_t)(ti + 1) / (uint64_t)T;
          for (uint64_t i = lo; i < hi; i++) {
            const uint64_t at = (off[r0 + i] - off[r0]) + i, len = off[r0 + i + 1] - off[r0 + i];
            hoff[i] = kCarry + at;
            memcpy(hs + kCarry + at, bases + off[r0 + i], len);
            hs[kCarry + at + len] = '\n';
          }
        };
        StagePool::get().run(T, work);
        nwhole = nf; L = bytes; r += nf;
      }
    }
    while (r < nreads) {
      const uint64_t len = off[r + 1] - off[r];
      const uint64_t remain = len - r_pos;
      if (r_pos == 0 && remain + 1 <= kStageBytes - L && nwhole < max_offs) {
        hoff[nwhole++] = kCarry + L;
        memcpy(hs + kCarry + L, bases + off[r], len);
        L += len;
        hs[kCarry + L++] = '\n';
        r++;
      } else if (L == 0) {  // read longer than a chunk (or its tail): stage a piece on its own
        const uint64_t take = std::min(remain, kStageBytes - 1);
        memcpy(hs + kCarry, bases + off[r] + r_pos, take);
        L = take;
        r_pos += take;
        piece_of = (long long)r;
        if (r_pos == len) { hs[kCarry + L++] = '\n'; r++; r_pos = 0; }
        break;
      } else {
        break;
      }
    }
    hoff[nwhole] = kCarry + L;
    const uint64_t total = kCarry + L;
    memcpy(carry, hs + total - kCarry, kCarry);
    // pad the tail so the 16-byte chunk loads of the kernel stay inside the copy
    memset(hs + total, '\n', 64);
    { int rc_ = flush_if_device_idle(g); if (rc_ != MCX_OK) return rc_; }
    HIP_TRY(hipMemcpyAsync(g->d_stage[b], hs, total + 64, hipMemcpyHostToDevice, g->stream));
    if (nwhole)
      HIP_TRY(hipMemcpyAsync(g->d_stage[b] + off_region, hoff, (nwhole + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, g->stream));
    StreamLaunch SL{g->d_stage[b], total, kCarry - (uint64_t)g->k, total - (uint64_t)g->k,
                    piece_of >= 0 ? d_flags + piece_of : nullptr};
    rc = submit_stream(g, SL, colour);
    if (rc != MCX_OK) return rc;
    if (nwhole) {
      hipLaunchKernelGGL(k_read_flags, dim3((unsigned)((nwhole + 255) / 256)), dim3(256), 0, g->stream,
                         (const uint8_t *)g->d_stage[b], (const uint64_t *)(g->d_stage[b] + off_region), nwhole,
                         g->k, d_flags + r0);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(g->ev[b], g->stream));
  }
  hipLaunchKernelGGL(k_count_flags, dim3(256), dim3(256), 0, g->stream, (const unsigned char *)d_flags, nreads, g->d_ctr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipFreeAsync(d_flags, g->stream));
  return MCX_OK;
}

// build --intersect: reads are uploaded whole (bases, qualities, offsets) and one lane per read
// restates the reference's must-exist loading loop (k_reads_must_exist)
static int add_reads_must_exist(mcx_graph *g, int colour, const uint8_t *bases, const uint8_t *quals,
                                const uint64_t *off, uint64_t nreads, uint8_t fq, uint8_t hp)
{
  const uint64_t base0 = off[0], nb = off[nreads] - off[0];
  uint8_t *d_bases = nullptr, *d_quals = nullptr;
  uint64_t *d_off = nullptr;
  std::vector<uint64_t> rel(nreads + 1);
  for (uint64_t i = 0; i <= nreads; i++) rel[i] = off[i] - base0;
  auto cleanup = [&]() { (void)hipFree(d_bases); (void)hipFree(d_quals); (void)hipFree(d_off); };
  if (hipMalloc((void **)&d_bases, nb + 16) != hipSuccess || hipMalloc((void **)&d_off, (nreads + 1) * 8) != hipSuccess ||
      (quals && fq > 0 && hipMalloc((void **)&d_quals, nb + 16) != hipSuccess)) {
    cleanup();
    (void)hipGetLastError();
    return fail(MCX_ERR_NOMEM, "out of device memory for a read batch");
  }
  HIP_TRY(hipMemcpyAsync(d_bases, bases + base0, nb, hipMemcpyHostToDevice, g->stream));
  if (d_quals) HIP_TRY(hipMemcpyAsync(d_quals, quals + base0, nb, hipMemcpyHostToDevice, g->stream));
  HIP_TRY(hipMemcpyAsync(d_off, rel.data(), (nreads + 1) * 8, hipMemcpyHostToDevice, g->stream));
  const unsigned blocks = (unsigned)((nreads + 127) / 128);
  {
    SpanGuard sp(g, "k_reads_must_exist");
    LAUNCH_W4(g->W, k_reads_must_exist, dim3(blocks), dim3(128), 0, g->stream, g->t, (const uint8_t *)d_bases,
              (const uint8_t *)d_quals, (const uint64_t *)d_off, nreads, g->k, (uint32_t)fq, (uint32_t)hp, (uint32_t)colour, g->d_ctr, (uint8_t *)nullptr, 0u, owner_spec(g));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(g->stream));
  cleanup();
  return MCX_OK;
}

extern "C" int mcx_graph_intersect_finish(mcx_graph *g, uint64_t *removed)
{
  if (g && g->as_group) return grp_intersect_finish(g->as_group, removed);
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (g->hidden < 0) return fail(MCX_ERR_ARG, "the graph is not in intersect mode");
  HIP_TRY(hipSetDevice(g->device));
  int rc = flush_deferred(g);
  if (rc != MCX_OK) return rc;
  DevBuf<unsigned long long> d_removed;
  unsigned long long h_removed = 0;
  HIP_TRY(d_removed.alloc(1));
  HIP_TRY(hipMemsetAsync(d_removed, 0, 8, g->stream));
  hipLaunchKernelGGL(k_intersect_finish, dim3(g->grid), dim3(256), 0, g->stream, g->t, (uint32_t)g->ncols_vis, (uint32_t)g->hidden,
                     g->d_ctr, d_removed.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(&h_removed, d_removed, 8, hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(hipStreamSynchronize(g->stream));
  if (removed) *removed = h_removed;
  return MCX_OK;
}

// ---------------------------------------------------------------------------
// build --remove-pcr (build_graph_from_reads_mt with prefs.remove_pcr_dups, build_graph.c:192-231)
// ---------------------------------------------------------------------------
extern "C" int mcx_graph_pcr_reset(mcx_graph *g)
{
  if (g && g->as_group) {
    int rc = MCX_OK;
    for (int i = 0; rc == MCX_OK && i < grp_n(g->as_group); i++) rc = mcx_graph_pcr_reset(grp_part(g->as_group, i));
    return rc;
  }
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  HIP_TRY(hipSetDevice(g->device));
  if (g->d_readstrt) HIP_TRY(hipMemsetAsync(g->d_readstrt, 0xff, g->t.nslots * 8, g->stream));
  return MCX_OK;
}

// Reads that need the reference's own contig rules (-Q / -H) and / or the duplicate filter: uploaded
// whole (bases, qualities, offsets); the kept reads are cut into contigs on the device
// (k_qh_contigs = seq_contig_start/end) and written out as a fresh separator stream for the
// ordinary front end.  filter: run seq_reads_are_novel first (ndup = reads / pairs dropped).
// The steps are separate functions because a multi-GPU table interleaves them across its shards
// (mcx_multi.h).
struct CutJob {  // one batch of whole reads on one device
  mcx_graph *g = nullptr;
  uint64_t nreads = 0, nunits = 0, nb = 0;
  uint32_t pmask = 0, fq1 = 0, fq2 = 0, hp = 0;
  uint8_t *d_bases = nullptr, *d_quals = nullptr, *d_keep = nullptr;
  uint64_t *d_off = nullptr, *d_node = nullptr;
  unsigned long long *d_ndup = nullptr;
  CutJob() = default;
  CutJob(const CutJob &) = delete;
  CutJob &operator=(const CutJob &) = delete;
  ~CutJob()
  {
    if (!g) return;
    (void)hipSetDevice(g->device);
    (void)hipFree(d_bases); (void)hipFree(d_quals); (void)hipFree(d_keep); (void)hipFree(d_off); (void)hipFree(d_node); (void)hipFree(d_ndup);
  }
};
#define CUT_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return fail(_e == hipErrorOutOfMemory ? MCX_ERR_NOMEM : MCX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// reads -> device (asynchronous on the handle's stream); mates turned to FF when the filter runs
static int cut_upload(CutJob &J, mcx_graph *g, const uint8_t *bases, const uint8_t *quals, const uint64_t *off, uint64_t nreads,
                      uint8_t fq1, uint8_t fq2, uint8_t hp, bool filter, int paired, int matedir)
{
  J.g = g;
  J.nreads = nreads;
  J.nunits = paired ? nreads / 2 : nreads;
  J.pmask = paired ? 1u : 0u; J.fq1 = fq1; J.fq2 = fq2; J.hp = hp;
  const uint64_t base0 = off[0];
  J.nb = off[nreads] - off[0];
  CUT_TRY(hipSetDevice(g->device));
  std::vector<uint64_t> rel(nreads + 1);
  for (uint64_t i = 0; i <= nreads; i++) rel[i] = off[i] - base0;
  const bool use_q = quals && (fq1 > 0 || fq2 > 0);
  CUT_TRY(hipMalloc((void **)&J.d_bases, J.nb + 16));
  if (use_q) CUT_TRY(hipMalloc((void **)&J.d_quals, J.nb + 16));
  CUT_TRY(hipMalloc((void **)&J.d_off, (nreads + 1) * 8));
  if (filter) {
    CUT_TRY(hipMalloc((void **)&J.d_node, nreads * 8));
    CUT_TRY(hipMalloc((void **)&J.d_keep, nreads));
    CUT_TRY(hipMalloc((void **)&J.d_ndup, 8));
    CUT_TRY(hipMemsetAsync(J.d_ndup, 0, 8, g->stream));
  }
  CUT_TRY(hipMemcpyAsync(J.d_bases, bases + base0, J.nb, hipMemcpyHostToDevice, g->stream));
  if (J.d_quals) CUT_TRY(hipMemcpyAsync(J.d_quals, quals + base0, J.nb, hipMemcpyHostToDevice, g->stream));
  CUT_TRY(hipMemcpyAsync(J.d_off, rel.data(), (nreads + 1) * 8, hipMemcpyHostToDevice, g->stream));
  CUT_TRY(hipStreamSynchronize(g->stream));  // `rel` is a local
  const unsigned blocks = (unsigned)((nreads + 127) / 128);
  if (filter && matedir)
    hipLaunchKernelGGL(k_pcr_orient, dim3(blocks), dim3(128), 0, g->stream, J.d_bases, J.d_quals, (const uint64_t *)J.d_off, nreads, J.pmask, (uint32_t)matedir);
  CUT_TRY(hipGetLastError());
  return MCX_OK;
}

// start node of every read (created if new) and T(node) = min over the reads that start there
static int cut_starts(CutJob &J)
{
  mcx_graph *g = J.g;
  CUT_TRY(hipSetDevice(g->device));
  if (!g->d_readstrt) {  // 2 x u32 per slot, only ever allocated for --remove-pcr
    if (hipMalloc((void **)&g->d_readstrt, g->t.nslots * 8) != hipSuccess) {
      (void)hipGetLastError();
      return fail(MCX_ERR_NOMEM, "out of device memory for the read-start table");
    }
    CUT_TRY(hipMemsetAsync(g->d_readstrt, 0xff, g->t.nslots * 8, g->stream));
  }
  const unsigned blocks = (unsigned)((J.nreads + 127) / 128);
  LAUNCH_W4(g->W, k_pcr_starts, dim3(blocks), dim3(128), 0, g->stream, g->t, (const uint8_t *)J.d_bases, (const uint8_t *)J.d_quals,
            (const uint64_t *)J.d_off, J.nreads, g->k, J.fq1, J.fq2, J.pmask, J.hp, g->d_readstrt, J.d_node, g->d_ctr, owner_spec(g));
  CUT_TRY(hipGetLastError());
  return MCX_OK;
}

// the kept reads (d_keep, or all), cut into contigs by the reference's rules, become a separator
// stream for the ordinary front end.  Synchronous.
static int cut_finish(CutJob &J, int colour)
{
  mcx_graph *g = J.g;
  CUT_TRY(hipSetDevice(g->device));
  const uint64_t nreads = J.nreads;
  const unsigned blocks = (unsigned)((nreads + 127) / 128);
  DevBuf<uint64_t> d_sizes, d_ooff;
  DevBuf<uint8_t> d_out, d_tmp;
  size_t tmp_bytes = 0;
  CUT_TRY(d_sizes.alloc(nreads + 1));
  CUT_TRY(d_ooff.alloc(nreads + 1));
  CUT_TRY(hipMemsetAsync(d_sizes, 0, (nreads + 1) * 8, g->stream));
  hipLaunchKernelGGL(k_qh_contigs, dim3(blocks), dim3(128), 0, g->stream, (const uint8_t *)J.d_bases, (const uint8_t *)J.d_quals,
                     (const uint64_t *)J.d_off, nreads, g->k, J.fq1, J.fq2, J.pmask, J.hp, (const uint8_t *)J.d_keep, 0, d_sizes.p,
                     (const uint64_t *)nullptr, (uint8_t *)nullptr);
  CUT_TRY(hipGetLastError());
  CUT_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, d_sizes.p, d_ooff.p, (uint64_t)0, nreads + 1, rocprim::plus<uint64_t>(), g->stream));
  CUT_TRY(d_tmp.alloc(tmp_bytes ? tmp_bytes : 16));
  CUT_TRY(rocprim::exclusive_scan((void *)d_tmp.p, tmp_bytes, d_sizes.p, d_ooff.p, (uint64_t)0, nreads + 1, rocprim::plus<uint64_t>(), g->stream));
  uint64_t out_bytes = 0;
  CUT_TRY(hipMemcpyAsync(&out_bytes, d_ooff.p + nreads, 8, hipMemcpyDeviceToHost, g->stream));
  CUT_TRY(hipStreamSynchronize(g->stream));
  int rc = MCX_OK;
  if (out_bytes) {
    CUT_TRY(d_out.alloc(out_bytes + 64));
    CUT_TRY(hipMemsetAsync(d_out.p + out_bytes, '\n', 64, g->stream));
    hipLaunchKernelGGL(k_qh_contigs, dim3(blocks), dim3(128), 0, g->stream, (const uint8_t *)J.d_bases, (const uint8_t *)J.d_quals,
                       (const uint64_t *)J.d_off, nreads, g->k, J.fq1, J.fq2, J.pmask, J.hp, (const uint8_t *)J.d_keep, 1, d_sizes.p,
                       (const uint64_t *)d_ooff.p, d_out.p);
    CUT_TRY(hipGetLastError());
    StreamLaunch SL{d_out.p, out_bytes, 0, out_bytes, nullptr};
    rc = submit_stream(g, SL, colour);
    CUT_TRY(hipSetDevice(g->device));
  }
  hipLaunchKernelGGL(k_count_sizes, dim3(256), dim3(256), 0, g->stream, (const uint64_t *)d_sizes.p, (const uint8_t *)J.d_keep, nreads, g->d_ctr);
  CUT_TRY(hipGetLastError());
  CUT_TRY(hipStreamSynchronize(g->stream));  // d_out is read by the kernels submit_stream launched
  return rc;
}

static int add_reads_cut(mcx_graph *g, int colour, const uint8_t *bases, const uint8_t *quals, const uint64_t *off,
                         uint64_t nreads, uint8_t fq_cutoff_abs1, uint8_t fq_cutoff_abs2, uint8_t hp_cutoff,
                         bool filter, int paired, int matedir, unsigned long long *ndup)
{
  CutJob J;
  int rc = cut_upload(J, g, bases, quals, off, nreads, fq_cutoff_abs1, fq_cutoff_abs2, hp_cutoff, filter, paired, matedir);
  if (rc != MCX_OK) return rc;
  unsigned long long h_ndup = 0;
  if (filter) {
    SpanGuard sp(g, "k_pcr_filter");
    rc = cut_starts(J);
    if (rc != MCX_OK) return rc;
    const unsigned blocks = (unsigned)((nreads + 127) / 128), ublocks = (unsigned)((J.nunits + 255) / 256);
    hipLaunchKernelGGL(k_pcr_decide, dim3(ublocks), dim3(256), 0, g->stream, (const uint64_t *)J.d_node, (const uint32_t *)g->d_readstrt,
                       J.nunits, J.pmask, J.d_keep, J.d_ndup);
    hipLaunchKernelGGL(k_pcr_commit, dim3(blocks), dim3(128), 0, g->stream, (const uint64_t *)J.d_node, nreads, g->d_readstrt);
    CUT_TRY(hipGetLastError());
    CUT_TRY(hipMemcpyAsync(&h_ndup, J.d_ndup, 8, hipMemcpyDeviceToHost, g->stream));
  }
  rc = cut_finish(J, colour);  // (synchronises the stream: h_ndup has arrived)
  if (ndup) *ndup = h_ndup;
  return rc;
}

// -Q / -H without the filter
static int add_reads_qh(mcx_graph *g, int colour, const uint8_t *bases, const uint8_t *quals,
                        const uint64_t *off, uint64_t nreads, uint8_t fq, uint8_t hp)
{
  return add_reads_cut(g, colour, bases, quals, off, nreads, fq, fq, hp, false, 0, 0, nullptr);
}

extern "C" int mcx_graph_add_reads_pcr(mcx_graph *g, int colour, const uint8_t *bases, const uint8_t *quals,
                                       const uint64_t *off, uint64_t nreads, uint8_t fq_cutoff_abs1,
                                       uint8_t fq_cutoff_abs2, uint8_t hp_cutoff, int paired, int matedir,
                                       mcx_load_stats *stats_accum)
{
  if (g && g->as_group) {
    if (colour < 0 || colour >= g->ncols) return fail(MCX_ERR_ARG, "colour %d out of range", colour);
    if (nreads && (!bases || !off)) return fail(MCX_ERR_ARG, "null read buffers");
    return grp_add_reads_pcr(g->as_group, colour, bases, quals, off, nreads, fq_cutoff_abs1, fq_cutoff_abs2, hp_cutoff, paired, matedir, stats_accum);
  }
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (colour < 0 || colour >= g->ncols_vis) return fail(MCX_ERR_ARG, "colour %d out of range", colour);
  if (nreads && (!bases || !off)) return fail(MCX_ERR_ARG, "null read buffers");
  if (paired && (nreads & 1)) return fail(MCX_ERR_ARG, "paired reads come in twos (%llu reads)", (unsigned long long)nreads);
  if (matedir < 0 || matedir > 3) return fail(MCX_ERR_ARG, "mate pair orientation %d: 0 FF, 1 FR, 2 RF, 3 RR", matedir);
  if (nreads >= (1ull << 32)) return fail(MCX_ERR_ARG, "too many reads in one batch");
  if (g->must_exist || g->hidden >= 0) return fail(MCX_ERR_ARG, "Cannot use --remove-pcr and --intersect");  // build_graph.c:198
  if (g->t.lbo) return fail(MCX_ERR_ARG, "duplicate removal on a sharded table goes through the multi-GPU handle");
  HIP_TRY(hipSetDevice(g->device));
  if (stats_accum) {
    if (paired) stats_accum->num_pe_reads += nreads; else stats_accum->num_se_reads += nreads;  // build_graph.c:211-212
    stats_accum->total_bases_read += nreads ? off[nreads] - off[0] : 0;
  }
  if (!nreads) return MCX_OK;
  unsigned long long ndup = 0;
  const int rc = add_reads_cut(g, colour, bases, quals, off, nreads, fq_cutoff_abs1, fq_cutoff_abs2, hp_cutoff, true,
                               paired, matedir, &ndup);
  if (stats_accum) {
    if (paired) stats_accum->num_dup_pe_pairs += ndup; else stats_accum->num_dup_se_reads += ndup;
  }
  return rc;
}

// ---------------------------------------------------------------------------
// sync / stats
// ---------------------------------------------------------------------------
static int fetch_counters(mcx_graph *g)
{
  HIP_TRY(hipSetDevice(g->device));
  static const bool timing = getenv("MCX_TIMING") != nullptr;
  const double t0 = timing ? now_s() : 0;
  const uint64_t had = g->pending + g->pending_l2;
  { int rc = flush_deferred(g); if (rc != MCX_OK) return rc; }
  const double t1 = timing ? now_s() : 0;
  HIP_TRY(hipMemcpyAsync(g->h_ctr, g->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(hipStreamSynchronize(g->stream));
  if (timing && had) fprintf(stderr, "[timing] closing flush: %.1f ms to enqueue (%llu occurrences booked), %.1f ms until the device was idle\n",
                             (t1 - t0) * 1e3, (unsigned long long)had, (now_s() - t1) * 1e3);
  if (g->h_ctr->full) return fail(MCX_ERR_FULL, "Hash table is full");
  if (g->h_ctr->bin_over) return fail(MCX_ERR_FULL, "partition bin overflow");
  return MCX_OK;
}

extern "C" int mcx_graph_sync(mcx_graph *g)
{
  if (g && g->as_group) return grp_sync(g->as_group);
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  return fetch_counters(g);
}

extern "C" int mcx_graph_nkmers(mcx_graph *g, uint64_t *n)
{
  if (g && n && g->as_group) {
    mcx_load_stats st;
    int rc = grp_device_stats(g->as_group, &st);
    *n = st.num_kmers_novel;
    return rc;
  }
  if (!g || !n) return fail(MCX_ERR_ARG, "null argument");
  int rc = fetch_counters(g);
  *n = g->h_ctr->novel;
  return rc;
}

extern "C" int mcx_graph_device_stats(mcx_graph *g, mcx_load_stats *out)
{
  if (g && out && g->as_group) return grp_device_stats(g->as_group, out);
  if (!g || !out) return fail(MCX_ERR_ARG, "null argument");
  int rc = fetch_counters(g);
  memset(out, 0, sizeof(*out));
  const Counters &c = *g->h_ctr;
  out->num_good_reads = c.good_reads;
  out->num_bad_reads = c.bad_reads;
  out->contigs_parsed = c.contigs;
  out->num_kmers_loaded = c.kmers - c.absent;  // must-exist mode: found k-mers only (build_graph.c:175-177)
  out->num_kmers_novel = c.novel;
  // sum of contig lengths = k-mers + (k-1) per contig (build_graph.c:173-176)
  out->total_bases_loaded = c.kmers + (uint64_t)(g->k - 1) * c.contigs;
  return rc;
}

extern "C" int mcx_graph_insert_stats(mcx_graph *g, mcx_insert_stats *out)
{
  if (!g || !out) return fail(MCX_ERR_ARG, "null argument");
  memset(out, 0, sizeof(*out));
  if (g->as_group) {
    int rc = grp_drain(g->as_group), first = MCX_OK;
    if (rc != MCX_OK) return rc;
    for (int i = 0; i < grp_n(g->as_group); i++) {
      mcx_insert_stats s;
      rc = mcx_graph_insert_stats(grp_part(g->as_group, i), &s);
      if (rc != MCX_OK && first == MCX_OK) first = rc;
      out->fallback_inserts += s.fallback_inserts; out->foreign_inserts += s.foreign_inserts; out->flushes += s.flushes;
    }
    out->spilled = grp_spilled(g->as_group);
    return first;
  }
  int rc = fetch_counters(g);  // (flushes what is buffered: the counters are final)
  unsigned long long h[2] = {0, 0};
  HIP_TRY(hipMemcpy(h, touch_base(g), sizeof(h), hipMemcpyDeviceToHost));
  out->fallback_inserts = h[0];
  out->foreign_inserts = h[1];
  out->flushes = g->n_flushes;
  return rc;
}

// ---------------------------------------------------------------------------
// bulk load of .ctx records (build --graph; graph_load, src/graph/graphs_load.c:86-214)
// ---------------------------------------------------------------------------
extern "C" int mcx_graph_add_records(mcx_graph *g, const void *recs, uint64_t nrecs, int file_ncols,
                                     const int32_t *from_col, const int32_t *into_col, int nmap, uint32_t flags,
                                     mcx_records_stats *stats_accum)
{
  if (g && g->as_group) return grp_add_records(g->as_group, recs, nrecs, file_ncols, from_col, into_col, nmap, flags, stats_accum);
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (file_ncols < 1 || file_ncols > 10000) return fail(MCX_ERR_ARG, "bad number of file colours: %d", file_ncols);
  if (nmap < 1 || !from_col || !into_col) return fail(MCX_ERR_ARG, "empty colour filter");
  for (int m = 0; m < nmap; m++) {
    if (from_col[m] < 0 || from_col[m] >= file_ncols)
      return fail(MCX_ERR_ARG, "filter entry %d reads colour %d of a %d colour file", m, from_col[m], file_ncols);
    if (into_col[m] < 0 || into_col[m] >= g->ncols)
      return fail(MCX_ERR_ARG, "filter entry %d loads into colour %d of a %d colour graph", m, into_col[m], g->ncols);
  }
  if (nrecs && !recs) return fail(MCX_ERR_ARG, "null records");
  if ((flags & MCX_RECORDS_MASK_EDGES) && g->hidden < 0) return fail(MCX_ERR_ARG, "edge masking needs intersect mode");
  HIP_TRY(hipSetDevice(g->device));
  int rc = ensure_stage(g);
  if (rc != MCX_OK) return rc;
  HIP_TRY(hipStreamSynchronize(g->stream));  // the staging buffers are ours now
  const uint64_t rec_bytes = 8ull * g->W + 5ull * (uint64_t)file_ncols;
  DevBuf<int32_t> d_into;  // from[nmap] then into[nmap]
  DevBuf<RecordStats> d_st;
  RecordStats h_st;
  HIP_TRY(d_into.alloc(2 * (size_t)nmap));
  HIP_TRY(d_st.alloc(1));
  memset(&h_st, 0, sizeof(h_st));
  h_st.first_oversized = h_st.first_zero_covg = h_st.first_edges_no_covg = ~0ULL;
  HIP_TRY(hipMemcpyAsync(d_into, from_col, sizeof(int32_t) * (size_t)nmap, hipMemcpyHostToDevice, g->stream));
  HIP_TRY(hipMemcpyAsync(d_into.p + nmap, into_col, sizeof(int32_t) * (size_t)nmap, hipMemcpyHostToDevice, g->stream));
  HIP_TRY(hipMemcpyAsync(d_st, &h_st, sizeof(h_st), hipMemcpyHostToDevice, g->stream));
  const uint64_t per_chunk = std::max<uint64_t>(1, std::min<uint64_t>(kStageBytes, g->stage_alloc) / rec_bytes);  // (a staging buffer holds stage_alloc bytes)
  int cur = 0;
  for (uint64_t r0 = 0; r0 < nrecs; r0 += per_chunk, cur ^= 1) {
    const uint64_t n = std::min(per_chunk, nrecs - r0);
    HIP_TRY(hipEventSynchronize(g->ev[cur]));  // the kernel that last read this staging pair is done
    memcpy(g->h_stage[cur], (const uint8_t *)recs + r0 * rec_bytes, n * rec_bytes);
    HIP_TRY(hipMemcpyAsync(g->d_stage[cur], g->h_stage[cur], n * rec_bytes, hipMemcpyHostToDevice, g->stream));
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)g->grid);
    SpanGuard sp(g, "k_load_records");
    LAUNCH_W4(g->W, k_load_records, dim3(grid), dim3(256), 0, g->stream, g->t, g->d_stage[cur], n, r0, (uint32_t)file_ncols,
              d_into.p, d_into.p + nmap, (uint32_t)nmap, flags & MCX_RECORDS_MUST_EXIST, (flags & MCX_RECORDS_MASK_EDGES) ? g->hidden : -1, g->k, g->d_ctr, d_st.p, owner_spec(g));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(g->ev[cur], g->stream));
  }
  HIP_TRY(hipMemcpyAsync(&h_st, d_st, sizeof(h_st), hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(hipStreamSynchronize(g->stream));
  if (stats_accum) {
    stats_accum->nkmers_read += nrecs;
    stats_accum->nkmers_loaded += h_st.loaded;
    stats_accum->nkmers_novel += h_st.novel;
    auto first = [](int64_t &dst, unsigned long long v, uint64_t base) {
      if (v != ~0ULL && dst < 0) dst = (int64_t)(base + v);
    };
    const uint64_t base = stats_accum->nkmers_read - nrecs;
    first(stats_accum->first_oversized, h_st.first_oversized, base);
    first(stats_accum->first_zero_covg, h_st.first_zero_covg, base);
    first(stats_accum->first_edges_no_covg, h_st.first_edges_no_covg, base);
  }
  if (h_st.first_oversized != ~0ULL)
    return fail(MCX_ERR_ARG, "oversized kmer in record %llu [kmer: %d]", (unsigned long long)h_st.first_oversized, g->k);
  return MCX_OK;
}

// ---------------------------------------------------------------------------
// export
// ---------------------------------------------------------------------------
// MCX_TIMING=1: stage clock of the export (stderr)
static void export_clock(const char *what)
{
  static const bool on = getenv("MCX_TIMING") != nullptr;
  if (!on) return;
  static auto t0 = std::chrono::steady_clock::now();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  fprintf(stderr, "[timing]   export %8.1f ms  %s\n", ms, what);
}

// Scratch of the export: per k-mer 8 (key word) + 8 (slot) + 4 x 8 (sort keys / permutation, in
// and out) bytes, two more words for two-word keys, plus the radix sort's own temporary.  When that
// does not fit beside the table (a table at 75 % load that takes more than a third of HBM), the
// deferred-insert workspace is released first, and if it still does not fit the key space is
// walked in 2^p ranges of the k-mer's top bits (k_compact's prefix filter): ranges in ascending
// order keep a sorted export sorted, and a range that turns out fuller than its arrays (real
// genomes are not uniform over prefixes) is split in two and redone.
template <int W>
static int export_t(mcx_graph *g, int sorted, mcx_sink_fn sink, void *ctx)
{
  export_clock("start");
  int rc = fetch_counters(g);
  if (rc != MCX_OK) return rc;
  const uint64_t n = g->h_ctr->novel;
  if (n == 0) return MCX_OK;
  hipStream_t st = g->stream;
  const uint32_t recsz = 8u * W + 5u * (uint32_t)g->ncols_vis;
  const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / recsz);
  const uint64_t per_kmer = 8 * (uint64_t)W + 8 + 32 + (W >= 2 ? 8 : 0) + 8 /* radix temporary, roughly */;
  const uint64_t fixed = 2 * chunk * recsz + (64ull << 20);
  auto free_bytes = []() { size_t f = 0, t = 0; return hipMemGetInfo(&f, &t) == hipSuccess ? (uint64_t)f : 0; };
  uint32_t pbits = 0;
  {
    uint64_t avail = free_bytes();
    if (const char *e = getenv("MCX_EXPORT_SCRATCH")) avail = std::min<uint64_t>(avail, strtoull(e, nullptr, 10));  // tests
    if (n * per_kmer + fixed > avail && g->l1_keys && !g->pending && !g->pending_l2) {
      HIP_TRY(hipStreamSynchronize(st));
      free_defer(g);  // the bins come back with the next batch of reads
      avail = free_bytes();
      if (const char *e = getenv("MCX_EXPORT_SCRATCH")) avail = std::min<uint64_t>(avail, strtoull(e, nullptr, 10));
    }
    const int key_bits = 2 * g->k;
    while (pbits < 16 && (int)pbits + 2 <= key_bits && (n >> pbits) * per_kmer * 5 / 4 + fixed > avail) pbits++;
    if ((n >> pbits) * per_kmer * 5 / 4 + fixed > avail)
      return fail(MCX_ERR_NOMEM, "not enough free HBM to dump the graph (%llu MB free)", (unsigned long long)(avail >> 20));
  }

  hipEvent_t done[2] = {nullptr, nullptr};
  uint8_t *d_rec2[2] = {nullptr, nullptr}, *h_rec2[2] = {nullptr, nullptr};
  unsigned long long *d_cursor = nullptr;
  uint64_t *d_k0 = nullptr, *d_k1 = nullptr, *d_slot = nullptr, *d_idx = nullptr, *d_idx2 = nullptr, *d_ks = nullptr, *d_ks2 = nullptr;
  void *d_tmp = nullptr;
  // Scratch arrays come out of the partition workspace when the graph has one and it is empty (after the closing flush
  // it is: 18-65 GB of HBM with nothing in it) -- a bump allocator over it instead of seven hipMalloc / hipFree of
  // gigabytes each: 40 ms of an export on a good day, 200+ ms when the driver has to hand over memory another process
  // freed moments before (`export ... compacted / sorted`: 38 / 52 ms against 149 / 282 ms, round 5).  What does not
  // fit the pool is allocated as before.
  // (two pools: the region bins and the sub-table bins.  The command's 2.1 G-occurrence window gives 18.5 + 2.9 GB; C2's
  // 337 M k-mers need 5 x 2.7 GB of arrays + 5.4 GB for the radix sort, and the one array that found no room made a
  // hipMalloc of 150 ms inside the bench process)
  const bool bins_idle = g->l1_keys && !g->pending && !g->pending_l2;
  uint8_t *pool[2] = {bins_idle ? reinterpret_cast<uint8_t *>(g->l1_keys) : nullptr, bins_idle ? reinterpret_cast<uint8_t *>(g->l2_keys) : nullptr};
  const uint64_t pool_bytes[2] = {pool[0] ? (uint64_t)g->nsets * g->b1 * g->rep1 * g->cap1 * 8 * g->W : 0,
                                  pool[1] ? (uint64_t)(g->l2_hi ? g->l2_regions / 2 : g->l2_regions) * g->subs_per_bin * g->cap2 * 8 * g->W : 0};  // (placed bins: the first half's allocation)
  uint64_t pool_off[2] = {0, 0};
  std::vector<void *> owned;
  auto salloc = [&](void **p, size_t bytes) -> hipError_t {
    const uint64_t need = ((uint64_t)bytes + 255) & ~255ull;
    for (int i = 0; i < 2; i++)
      if (pool[i] && pool_off[i] + need <= pool_bytes[i]) { *p = pool[i] + pool_off[i]; pool_off[i] += need; return hipSuccess; }
    const double t0 = now_s();
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) owned.push_back(*p);
    if (getenv("MCX_TIMING")) fprintf(stderr, "[timing]   export: %.2f GB of scratch found no room in the idle bins (%.1f + %.1f GB): hipMalloc %.1f ms\n", bytes / 1e9,
                                      pool_bytes[0] / 1e9, pool_bytes[1] / 1e9, (now_s() - t0) * 1e3);
    return e;
  };
  auto free_range = [&]() {
    for (void *q : owned) (void)hipFree(q);
    owned.clear();
    pool_off[0] = pool_off[1] = 0;
    d_k0 = d_k1 = d_slot = d_idx = d_idx2 = d_ks = d_ks2 = nullptr; d_tmp = nullptr;
  };
  // The chunk buffers: the host entry's staging pairs when the graph has them (pinned host + device memory that sits
  // idle once the reads are in: the device is idle here, fetch_counters has just returned) -- page-locking two fresh
  // 64 MiB buffers took 23 ms on a good day and 263 ms on a bad one (`export ... buffers allocated`, round 5).
  const bool borrow = g->stage_alloc >= chunk * recsz && g->h_stage[0] && g->h_stage[1] && g->d_stage[0] && g->d_stage[1];
  auto cleanup = [&]() {
    free_range();
    (void)hipFree(d_cursor);
    for (int i = 0; i < 2; i++) {
      if (done[i]) (void)hipEventDestroy(done[i]);
      if (borrow) continue;
      (void)hipFree(d_rec2[i]);
      if (h_rec2[i]) (void)hipHostFree(h_rec2[i]);
    }
  };
#define EXP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      cleanup();                                                                          \
      return fail(_e == hipErrorOutOfMemory ? MCX_ERR_NOMEM : MCX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    }                                                                                     \
  } while (0)
  export_clock("counters fetched");
  EXP_TRY(hipMalloc((void **)&d_cursor, 8));
  // records are produced and copied to pinned memory in 64 MiB chunks, double buffered: while the
  // sink consumes chunk i on the host, the device emits and copies chunk i + 1
  for (int i = 0; i < 2; i++) {
    if (borrow) {
      d_rec2[i] = g->d_stage[i];
      h_rec2[i] = g->h_stage[i];
    } else {
      EXP_TRY(hipMalloc((void **)&d_rec2[i], chunk * recsz));
      EXP_TRY(hipHostMalloc((void **)&h_rec2[i], chunk * recsz, hipHostMallocDefault));
    }
    EXP_TRY(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
  }
  export_clock("buffers allocated");

  struct Range { uint32_t prefix, bits; };
  std::vector<Range> todo;  // a stack: the lowest prefix on top
  for (uint32_t p = 1u << pbits; p-- > 0;) todo.push_back({p, pbits});
  uint64_t emitted = 0;
  while (!todo.empty()) {
    const Range r = todo.back();
    todo.pop_back();
    const uint64_t cap = r.bits == 0 ? n : std::min<uint64_t>(n, (n >> r.bits) * 5 / 4 + 65536);
    EXP_TRY(salloc((void **)&d_k0, cap * 8));
    if (W == 2) EXP_TRY(salloc((void **)&d_k1, cap * 8));
    EXP_TRY(salloc((void **)&d_slot, cap * 8));
    EXP_TRY(hipMemsetAsync(d_cursor, 0, 8, st));
    hipLaunchKernelGGL((k_compact<W>), dim3(g->grid), dim3(kThreads), 0, st, g->t, d_k0, d_k1, d_slot, d_cursor, cap, g->k, r.prefix, r.bits);
    EXP_TRY(hipGetLastError());
    unsigned long long found = 0;
    EXP_TRY(hipMemcpyAsync(&found, d_cursor, 8, hipMemcpyDeviceToHost, st));
    EXP_TRY(hipStreamSynchronize(st));
    if (r.bits == 0 && found != n) { cleanup(); return fail(MCX_ERR_HIP, "table scan found %llu nodes, counter says %llu", found, (unsigned long long)n); }
    if (found > cap) {  // fuller than expected: two halves, the lower one first
      free_range();
      if ((int)r.bits + 1 > 2 * g->k || r.bits >= 30) { cleanup(); return fail(MCX_ERR_NOMEM, "not enough free HBM to dump the graph"); }
      todo.push_back({r.prefix * 2 + 1, r.bits + 1});
      todo.push_back({r.prefix * 2, r.bits + 1});
      continue;
    }
    const uint64_t m = found;
    if (m == 0) { free_range(); continue; }
    export_clock("compacted");
    // permutation of the compacted entries: by key (sorted) or by slot (table order)
    size_t tmp_bytes = 0;
    const uint64_t *first_key = sorted ? (W == 2 ? d_k1 : d_k0) : d_slot;  // (W > 2: a throw-away pass, see below)
    // (the sort's own temporary first: it is the largest block -- two more arrays of m words -- and the bump allocator
    // fills the first pool in order; the smaller arrays find room in what is left or in the second pool)
    EXP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, first_key, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, m, 0, 64, st));
    EXP_TRY(salloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
    EXP_TRY(salloc((void **)&d_idx, m * 8));
    EXP_TRY(salloc((void **)&d_idx2, m * 8));
    EXP_TRY(salloc((void **)&d_ks, m * 8));
    if (sorted && W >= 2) EXP_TRY(salloc((void **)&d_ks2, m * 8));
    const unsigned gb = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(k_iota, dim3(gb), dim3(256), 0, st, d_idx, m);
    EXP_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, first_key, d_ks, d_idx, d_idx2, m, 0, 64, st));
    uint64_t *perm = d_idx2;
    if (sorted && W == 2) {  // LSD: stable second pass on the most significant word
      hipLaunchKernelGGL(k_gather_u64, dim3(gb), dim3(256), 0, st, (const uint64_t *)d_k0, (const uint64_t *)d_idx2, d_ks2, m);
      EXP_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_ks2, d_ks, d_idx2, d_idx, m, 0, 64, st));
      perm = d_idx;
    }
    if (sorted && W > 2) {
      // k > 63: the pass above ordered by the top word only (d_k0).  Redo as an LSD sort over all W words, least
      // significant first, every pass stable; a pass fetches its word from the table through the permutation so far.
      hipLaunchKernelGGL(k_iota, dim3(gb), dim3(256), 0, st, d_idx, m);
      uint64_t *cur = d_idx, *nxt = d_idx2;
      for (int w = W - 1; w >= 0; w--) {
        hipLaunchKernelGGL(k_gather_keyword, dim3(gb), dim3(256), 0, st, g->t, (const uint64_t *)d_slot, (const uint64_t *)cur, (uint32_t)w, d_ks2, m);
        EXP_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_ks2, d_ks, cur, nxt, m, 0, 64, st));
        std::swap(cur, nxt);
      }
      perm = cur;
    }
    EXP_TRY(hipStreamSynchronize(st));
    export_clock("sorted");
    auto produce = [&](uint64_t first, int b) -> hipError_t {
      const uint64_t cnt = std::min(chunk, m - first);
      hipLaunchKernelGGL((k_emit_records<W>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, g->t,
                         (const uint64_t *)d_slot, (const uint64_t *)perm, first, cnt, (uint32_t)g->ncols_vis, d_rec2[b]);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      e = hipMemcpyAsync(h_rec2[b], d_rec2[b], cnt * recsz, hipMemcpyDeviceToHost, st);
      if (e != hipSuccess) return e;
      return hipEventRecord(done[b], st);
    };
    EXP_TRY(produce(0, 0));
    int b = 0;
    for (uint64_t first = 0; first < m; first += chunk, b ^= 1) {
      const uint64_t cnt = std::min(chunk, m - first);
      if (first + chunk < m) EXP_TRY(produce(first + chunk, b ^ 1));
      EXP_TRY(hipEventSynchronize(done[b]));
      if (sink(ctx, h_rec2[b], cnt * recsz) != 0) { cleanup(); return fail(MCX_ERR_SINK, "export sink failed"); }
    }
    emitted += m;
    free_range();
  }
  export_clock("records delivered");
#undef EXP_TRY
  cleanup();
  if (emitted != n) return fail(MCX_ERR_HIP, "export delivered %llu of %llu nodes", (unsigned long long)emitted, (unsigned long long)n);
  return MCX_OK;
}

extern "C" int mcx_graph_export(mcx_graph *g, int sorted, mcx_sink_fn sink, void *ctx)
{
  if (g && sink && g->as_group) return grp_export(g->as_group, g, sorted, sink, ctx);
  if (!g || !sink) return fail(MCX_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(g->device));
  switch (g->W) {
    case 1: return export_t<1>(g, sorted, sink, ctx);
    case 2: return export_t<2>(g, sorted, sink, ctx);
    case 3: return export_t<3>(g, sorted, sink, ctx);
    default: return export_t<4>(g, sorted, sink, ctx);
  }
}

// ---------------------------------------------------------------------------
// table scans: per-colour k-mer / coverage totals and the k-mer coverage histogram
// ---------------------------------------------------------------------------
static int covg_scan(mcx_graph *g, uint64_t *nkmers, uint64_t *sumcov, uint64_t *hist, uint32_t nbins)
{
  HIP_TRY(hipSetDevice(g->device));
  int rc = flush_deferred(g);
  if (rc != MCX_OK) return rc;
  const uint32_t nc = (uint32_t)g->ncols_vis;
  const uint32_t lbins = hist ? std::min<uint32_t>(nbins, 4096u) : 0;
  DevBuf<unsigned long long> d_out, d_hist;
  HIP_TRY(d_out.alloc(2 * nc));
  HIP_TRY(hipMemsetAsync(d_out, 0, 2 * nc * 8, g->stream));
  if (hist) {
    HIP_TRY(d_hist.alloc(nbins));
    HIP_TRY(hipMemsetAsync(d_hist, 0, (size_t)nbins * 8, g->stream));
  }
  {
    SpanGuard sp(g, "k_covg_scan");
    hipLaunchKernelGGL(k_covg_scan, dim3(g->grid), dim3(256), (2 * nc + lbins) * 8, g->stream, g->t, nc, d_out.p, d_hist.p, nbins, lbins);
  }
  HIP_TRY(hipGetLastError());
  std::vector<unsigned long long> h(2 * nc);
  HIP_TRY(hipMemcpyAsync(h.data(), d_out, 2 * nc * 8, hipMemcpyDeviceToHost, g->stream));
  if (hist) HIP_TRY(hipMemcpyAsync(hist, d_hist, (size_t)nbins * 8, hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(hipStreamSynchronize(g->stream));
  for (uint32_t c = 0; c < nc; c++) {
    if (nkmers) nkmers[c] = h[c];
    if (sumcov) sumcov[c] = h[nc + c];
  }
  return MCX_OK;
}

extern "C" int mcx_graph_checksum(mcx_graph *g, uint64_t *checksum, uint64_t *nkmers)
{
  if (g && checksum && g->as_group) {  // an order-independent sum: the shards' sums add up
    int rc = grp_sync(g->as_group);
    uint64_t cs = 0, nk = 0;
    for (int i = 0; rc == MCX_OK && i < grp_n(g->as_group); i++) {
      uint64_t c1 = 0, n1 = 0;
      rc = mcx_graph_checksum(grp_part(g->as_group, i), &c1, &n1);
      cs += c1; nk += n1;
    }
    *checksum = cs;
    if (nkmers) *nkmers = nk;
    return rc;
  }
  if (!g || !checksum) return fail(MCX_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(g->device));
  int rc = flush_deferred(g);
  if (rc != MCX_OK) return rc;
  DevBuf<unsigned long long> d_out;
  unsigned long long h_out[2] = {0, 0};
  HIP_TRY(d_out.alloc(2));
  HIP_TRY(hipMemsetAsync(d_out, 0, 16, g->stream));
  {
    SpanGuard sp(g, "k_checksum");
    hipLaunchKernelGGL(k_checksum, dim3(g->grid), dim3(256), 0, g->stream, g->t, (uint32_t)g->W, (uint32_t)g->ncols_vis, d_out.p);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h_out, d_out, 16, hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(hipStreamSynchronize(g->stream));
  *checksum = h_out[0];
  if (nkmers) *nkmers = h_out[1];
  return MCX_OK;
}

extern "C" uint64_t mcx_records_checksum(const void *recs, uint64_t nrecs, int kmer_size, int ncols)
{
  const int W = words_for_k(kmer_size);
  const size_t rb = 8 * (size_t)W + 5 * (size_t)ncols;
  const uint8_t *p = (const uint8_t *)recs;
  uint64_t sum = 0;
  std::vector<uint32_t> cv((size_t)ncols);
  for (uint64_t i = 0; i < nrecs; i++, p += rb) {
    uint64_t kw[4] = {0, 0, 0, 0};
    memcpy(kw, p, 8 * (size_t)W);
    memcpy(cv.data(), p + 8 * W, 4 * (size_t)ncols);
    sum += record_hash(kw, W, cv.data(), p + 8 * W + 4 * ncols, (uint32_t)ncols);
  }
  return sum;
}

// ---------------------------------------------------------------------------
// device ceilings (mcx_ubench.h)
// ---------------------------------------------------------------------------
#ifdef MCX_PHASES  // tools/variants.sh build only (mcx_defer.h: per-phase time of the build kernels)
extern "C" int mcx_debug_phases(uint64_t *out24, int reset)
{
  if (out24) HIP_TRY(hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 24));
  if (reset) { unsigned long long z[24] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z))); }
  return MCX_OK;
}
#endif

extern "C" int mcx_ubench_stream(int device, uint64_t bytes, double *copy_gbs, double *read_gbs, double *write_gbs)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(MCX_ERR_NODEVICE, "no HIP device %d", device);
  HIP_TRY(hipSetDevice(device));
  const uint64_t n = bytes / 16;
  if (n < (1u << 20)) return fail(MCX_ERR_ARG, "buffers of at least 16 MiB");
  ulonglong2 *a = nullptr, *b = nullptr;
  unsigned long long *sink = nullptr;
  if (hipMalloc((void **)&a, n * 16) != hipSuccess || hipMalloc((void **)&b, n * 16) != hipSuccess || hipMalloc((void **)&sink, 8) != hipSuccess) {
    (void)hipGetLastError(); (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
    return fail(MCX_ERR_NOMEM, "out of device memory for the streaming microbenchmark");
  }
  (void)hipMemset(a, 1, n * 16); (void)hipMemset(b, 2, n * 16);
  const int grid = 8192;
  double ms[3] = {0, 0, 0};
  int rc = ubench_time([&] { k_ubench_stream<0, 8><<<grid, 256>>>(a, b, n, sink); }, 5, &ms[0]);
  if (!rc) rc = ubench_time([&] { k_ubench_stream<1, 8><<<grid, 256>>>(a, b, n, sink); }, 5, &ms[1]);
  if (!rc) rc = ubench_time([&] { k_ubench_stream<2, 8><<<grid, 256>>>(a, b, n, sink); }, 5, &ms[2]);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
  if (rc) return fail(MCX_ERR_HIP, "streaming microbenchmark failed");
  if (copy_gbs) *copy_gbs = 2.0 * n * 16 / ms[0] / 1e6;
  if (read_gbs) *read_gbs = 1.0 * n * 16 / ms[1] / 1e6;
  if (write_gbs) *write_gbs = 1.0 * n * 16 / ms[2] / 1e6;
  return MCX_OK;
}

extern "C" int mcx_ubench_random_rmw(int device, uint64_t table_bytes, uint64_t nupdates, double *rmw_per_s, double *load16_per_s,
                                     double *load_rmw_per_s)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(MCX_ERR_NODEVICE, "no HIP device %d", device);
  HIP_TRY(hipSetDevice(device));
  const uint64_t nrec = table_bytes / 16;
  if (nrec < 1024 || nupdates < (1u << 20)) return fail(MCX_ERR_ARG, "table of at least 16 KiB, at least 1 M updates");
  uint64_t *tab = nullptr;
  unsigned long long *sink = nullptr;
  if (hipMalloc((void **)&tab, nrec * 16) != hipSuccess || hipMalloc((void **)&sink, 8) != hipSuccess) {
    (void)hipGetLastError(); (void)hipFree(tab); (void)hipFree(sink);
    return fail(MCX_ERR_NOMEM, "out of device memory for the random-access microbenchmark");
  }
  (void)hipMemset(tab, 0, nrec * 16);
  double ms[3] = {0, 0, 0};
  uint64_t seed = 7;
  int rc = ubench_time([&] { k_ubench_rmw<1, 4><<<2048, 256>>>(tab, nrec, nupdates, seed++, sink); }, 2, &ms[0]);
  if (!rc) rc = ubench_time([&] { k_ubench_rmw<0, 4><<<2048, 256>>>(tab, nrec, nupdates, seed++, sink); }, 2, &ms[1]);
  if (!rc) rc = ubench_time([&] { k_ubench_rmw<2, 4><<<2048, 256>>>(tab, nrec, nupdates, seed++, sink); }, 2, &ms[2]);
  (void)hipFree(tab); (void)hipFree(sink);
  if (rc) return fail(MCX_ERR_HIP, "random-access microbenchmark failed");
  if (rmw_per_s) *rmw_per_s = (double)nupdates / ms[0] * 1e3;
  if (load16_per_s) *load16_per_s = (double)nupdates / ms[1] * 1e3;
  if (load_rmw_per_s) *load_rmw_per_s = (double)nupdates / ms[2] * 1e3;
  return MCX_OK;
}

extern "C" int mcx_graph_kmer_covg(mcx_graph *g, uint64_t *nkmers, uint64_t *sumcov)
{
  if (g && g->as_group) {
    int rc = grp_sync(g->as_group);
    std::vector<uint64_t> a((size_t)g->ncols), b((size_t)g->ncols);
    for (int c = 0; c < g->ncols; c++) { if (nkmers) nkmers[c] = 0; if (sumcov) sumcov[c] = 0; }
    for (int i = 0; rc == MCX_OK && i < grp_n(g->as_group); i++) {
      rc = mcx_graph_kmer_covg(grp_part(g->as_group, i), a.data(), b.data());
      for (int c = 0; c < g->ncols; c++) { if (nkmers) nkmers[c] += a[c]; if (sumcov) sumcov[c] += b[c]; }
    }
    return rc;
  }
  if (!g) return fail(MCX_ERR_ARG, "null graph");
  if (g->ncols > 2048) return fail(MCX_ERR_ARG, "too many colours for the scan");
  return covg_scan(g, nkmers, sumcov, nullptr, 0);
}

extern "C" int mcx_graph_covg_histogram(mcx_graph *g, uint64_t *hist, uint32_t nbins)
{
  if (g && hist && g->as_group) {
    int rc = grp_sync(g->as_group);
    std::vector<uint64_t> h(nbins);
    for (uint32_t b = 0; b < nbins; b++) hist[b] = 0;
    for (int i = 0; rc == MCX_OK && i < grp_n(g->as_group); i++) {
      rc = mcx_graph_covg_histogram(grp_part(g->as_group, i), h.data(), nbins);
      for (uint32_t b = 0; b < nbins; b++) hist[b] += h[b];
    }
    return rc;
  }
  if (!g || !hist) return fail(MCX_ERR_ARG, "null argument");
  if (nbins < 2) return fail(MCX_ERR_ARG, "the histogram needs at least two bins");
  if (g->ncols > 2048) return fail(MCX_ERR_ARG, "too many colours for the scan");
  return covg_scan(g, nullptr, nullptr, hist, nbins);
}

// ---------------------------------------------------------------------------
// sort / sortedness check of .ctx records on the device (src/commands/ctx_sort.c:133-152,
// src/commands/ctx_index.c:136-137)
// ---------------------------------------------------------------------------
template <int W>
static int sort_records_t(uint8_t *recs, uint64_t n, uint32_t rec_bytes, int device, bool check_only, int64_t *first_unsorted)
{
  hipStream_t st = nullptr;
  uint8_t *d_in = nullptr, *d_out = nullptr;
  uint64_t *d_k0 = nullptr, *d_k1 = nullptr, *d_ks = nullptr, *d_ks2 = nullptr, *d_idx = nullptr, *d_idx2 = nullptr;
  unsigned long long *d_bad = nullptr;
  void *d_tmp = nullptr;
  size_t tmp_bytes = 0;
  auto cleanup = [&]() {
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_k0); (void)hipFree(d_k1); (void)hipFree(d_ks); (void)hipFree(d_ks2);
    (void)hipFree(d_idx); (void)hipFree(d_idx2); (void)hipFree(d_bad); (void)hipFree(d_tmp);
    if (st) (void)hipStreamDestroy(st);
  };
#define SRT_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      cleanup();                                                                          \
      return fail(_e == hipErrorOutOfMemory ? MCX_ERR_NOMEM : MCX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    }                                                                                     \
  } while (0)
  SRT_TRY(hipSetDevice(device));
  SRT_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const uint64_t bytes = n * rec_bytes;
  SRT_TRY(hipMalloc((void **)&d_in, bytes));
  SRT_TRY(hipMalloc((void **)&d_k0, n * 8));
  if (W == 2) SRT_TRY(hipMalloc((void **)&d_k1, n * 8));
  SRT_TRY(hipMemcpyAsync(d_in, recs, bytes, hipMemcpyHostToDevice, st));
  const unsigned gb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL((k_record_keys<W>), dim3(gb), dim3(256), 0, st, d_in, n, rec_bytes, d_k0, d_k1);
  SRT_TRY(hipGetLastError());
  if (check_only) {
    unsigned long long bad = ~0ULL;
    SRT_TRY(hipMalloc((void **)&d_bad, 8));
    SRT_TRY(hipMemcpyAsync(d_bad, &bad, 8, hipMemcpyHostToDevice, st));
    if (W > 2) hipLaunchKernelGGL(k_check_sorted_records, dim3(gb), dim3(256), 0, st, (const uint8_t *)d_in, rec_bytes, (uint32_t)W, n, d_bad);
    else hipLaunchKernelGGL((k_check_sorted<W>), dim3(gb), dim3(256), 0, st, d_k0, d_k1, n, d_bad);
    SRT_TRY(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st));
    SRT_TRY(hipStreamSynchronize(st));
    *first_unsorted = bad == ~0ULL ? -1 : (int64_t)bad;
    cleanup();
    return MCX_OK;
  }
  SRT_TRY(hipMalloc((void **)&d_idx, n * 8));
  SRT_TRY(hipMalloc((void **)&d_idx2, n * 8));
  SRT_TRY(hipMalloc((void **)&d_ks, n * 8));
  hipLaunchKernelGGL(k_iota, dim3(gb), dim3(256), 0, st, d_idx, n);
  const uint64_t *first_key = W == 2 ? d_k1 : d_k0;
  SRT_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, first_key, d_ks, d_idx, d_idx2, n, 0, 64, st));
  SRT_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
  SRT_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, first_key, d_ks, d_idx, d_idx2, n, 0, 64, st));
  uint64_t *perm = d_idx2;
  if (W == 2) {  // LSD: stable second pass on the most significant word
    SRT_TRY(hipMalloc((void **)&d_ks2, n * 8));
    hipLaunchKernelGGL(k_gather_u64, dim3(gb), dim3(256), 0, st, (const uint64_t *)d_k0, (const uint64_t *)d_idx2, d_ks2, n);
    SRT_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_ks2, d_ks, d_idx2, d_idx, n, 0, 64, st));
    perm = d_idx;
  }
  if (W > 2) {  // k > 63: LSD over all W words, least significant first, each word fetched through the permutation so far
    SRT_TRY(hipMalloc((void **)&d_ks2, n * 8));
    hipLaunchKernelGGL(k_iota, dim3(gb), dim3(256), 0, st, d_idx, n);
    uint64_t *cur = d_idx, *nxt = d_idx2;
    for (int w = W - 1; w >= 0; w--) {
      hipLaunchKernelGGL(k_gather_record_word, dim3(gb), dim3(256), 0, st, (const uint8_t *)d_in, rec_bytes, (const uint64_t *)cur, (uint32_t)w, d_ks2, n);
      SRT_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_ks2, d_ks, cur, nxt, n, 0, 64, st));
      std::swap(cur, nxt);
    }
    perm = cur;
  }
  (void)hipFree(d_ks); d_ks = nullptr;
  SRT_TRY(hipMalloc((void **)&d_out, bytes));
  hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, st, d_in, (const uint64_t *)perm, n, rec_bytes, d_out);
  SRT_TRY(hipGetLastError());
  SRT_TRY(hipMemcpyAsync(recs, d_out, bytes, hipMemcpyDeviceToHost, st));
  SRT_TRY(hipStreamSynchronize(st));
#undef SRT_TRY
  cleanup();
  return MCX_OK;
}

static int sort_args(const void *recs, uint64_t nrecs, int kmer_size, int ncols, uint32_t *rec_bytes)
{
  if (kmer_size < 3 || kmer_size > 127 || !(kmer_size & 1)) return fail(MCX_ERR_ARG, "kmer size must be odd and within 3..127 (got %d)", kmer_size);
  if (ncols < 1 || ncols > 10000) return fail(MCX_ERR_ARG, "bad number of colours: %d", ncols);
  if (nrecs && !recs) return fail(MCX_ERR_ARG, "null records");
  if (nrecs >= (1ull << 32) * 16) return fail(MCX_ERR_ARG, "too many records for one call");
  *rec_bytes = 8u * (uint32_t)words_for_k(kmer_size) + 5u * (uint32_t)ncols;
  return MCX_OK;
}

extern "C" int mcx_sort_records(void *recs, uint64_t nrecs, int kmer_size, int ncols, int device)
{
  uint32_t rb;
  int rc = sort_args(recs, nrecs, kmer_size, ncols, &rb);
  if (rc != MCX_OK || nrecs < 2) return rc;
  int64_t dummy;
  switch (words_for_k(kmer_size)) {
    case 1: return sort_records_t<1>((uint8_t *)recs, nrecs, rb, device, false, &dummy);
    case 2: return sort_records_t<2>((uint8_t *)recs, nrecs, rb, device, false, &dummy);
    case 3: return sort_records_t<3>((uint8_t *)recs, nrecs, rb, device, false, &dummy);
    default: return sort_records_t<4>((uint8_t *)recs, nrecs, rb, device, false, &dummy);
  }
}

extern "C" int mcx_records_sorted(const void *recs, uint64_t nrecs, int kmer_size, int ncols, int device, int64_t *first_unsorted)
{
  uint32_t rb;
  if (!first_unsorted) return fail(MCX_ERR_ARG, "null argument");
  *first_unsorted = -1;
  int rc = sort_args(recs, nrecs, kmer_size, ncols, &rb);
  if (rc != MCX_OK || nrecs < 2) return rc;
  switch (words_for_k(kmer_size)) {
    case 1: return sort_records_t<1>((uint8_t *)recs, nrecs, rb, device, true, first_unsorted);
    case 2: return sort_records_t<2>((uint8_t *)recs, nrecs, rb, device, true, first_unsorted);
    case 3: return sort_records_t<3>((uint8_t *)recs, nrecs, rb, device, true, first_unsorted);
    default: return sort_records_t<4>((uint8_t *)recs, nrecs, rb, device, true, first_unsorted);
  }
}

// ---------------------------------------------------------------------------
// host-side primitives (rows A-C), same templates the kernels use
// ---------------------------------------------------------------------------
extern "C" void mcx_kmer_from_str(const char *seq, int k, uint64_t *out)
{
  const int W = words_for_k(k);
  uint64_t w[4] = {0, 0, 0, 0};  // 256-bit shift register, w[3] = least significant
  for (int i = 0; i < k; i++) {
    for (int j = 0; j < 3; j++) w[j] = (w[j] << 2) | (w[j + 1] >> 62);
    w[3] = (w[3] << 2) | base_code((unsigned char)seq[i]);
  }
  for (int j = 0; j < W; j++) out[j] = w[4 - W + j];
}

template <int W> static void kmer_canonical_t(const uint64_t *in, int k, uint64_t *key_out, uint32_t &o)
{
  Kmer<W> fw;
  for (int i = 0; i < W; i++) fw.w[i] = in[i];
  const Kmer<W> key = canonical<W>(fw, revcomp<W>(fw, k), o);
  for (int i = 0; i < W; i++) key_out[i] = key.w[i];
}
extern "C" void mcx_kmer_canonical(const uint64_t *in, int k, uint64_t *key_out, int *orient_out)
{
  uint32_t o = 0;
  switch (words_for_k(k)) {
    case 1: kmer_canonical_t<1>(in, k, key_out, o); break;
    case 2: kmer_canonical_t<2>(in, k, key_out, o); break;
    case 3: kmer_canonical_t<3>(in, k, key_out, o); break;
    default: kmer_canonical_t<4>(in, k, key_out, o); break;
  }
  if (orient_out) *orient_out = (int)o;
}

template <int W> static uint32_t kmer_hash_t(const uint64_t *key, uint32_t initval)
{
  Kmer<W> x;
  for (int i = 0; i < W; i++) x.w[i] = key[i];
  return kmer_hash<W>(x, initval, nullptr);
}
extern "C" uint32_t mcx_kmer_hash(const uint64_t *key, int k, uint32_t initval)
{
  switch (words_for_k(k)) {
    case 1: return kmer_hash_t<1>(key, initval);
    case 2: return kmer_hash_t<2>(key, initval);
    case 3: return kmer_hash_t<3>(key, initval);
    default: return kmer_hash_t<4>(key, initval);
  }
}

#include "mcx_multi.h"

static mcx_graph *grp_part(mcx_group *G, int i) { return G->part[i]; }
static int grp_n(mcx_group *G) { return G->n; }
static uint64_t grp_spilled(mcx_group *G) { return G->spilled; }
// the shard on whose device `d_ptr` lives (shards that share a device take turns)
static int grp_part_of_pointer(mcx_group *G, const void *d_ptr, mcx_graph **part)
{
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, d_ptr) != hipSuccess) { (void)hipGetLastError(); return fail(MCX_ERR_ARG, "not a device pointer"); }
  static unsigned turn = 0;
  std::vector<int> on;
  for (int i = 0; i < G->n; i++) if (G->part[i]->device == at.device) on.push_back(i);
  if (on.empty()) return fail(MCX_ERR_ARG, "the stream lives on device %d, which holds no shard of this table", at.device);
  *part = G->part[on[turn++ % on.size()]];
  return MCX_OK;
}
