// mcx_multi.h -- one table over several GPUs, driven by ONE host process (included by mcx_api.hip).
//
// What it replaces: the batch loop of `build` (src/commands/ctx_build.c:384-407) feeding ONE hash
// table from -t worker threads.  Here the table is split by quotient-hash prefix into one shard per
// GPU (mcx_graph_create_shard: owner = top bits of the address word G) and
// mcx_graph_create_multi() returns an ordinary mcx_graph handle -- a facade -- whose calls are dealt
// out to the shards, so that the host program (host/cmd_build.c) is the same for one GPU and for
// eight: reads, -Q / -H, --remove-pcr, --graph loads, statistics and the export all go through it.
//
// Two exchange formats, as in the process-per-GPU driver (mccortex_amd/shard.py):
//   v3 (odd k in 29..63, the default there): the owner of a k-mer is a hash of its canonical minimizer;
//      READS travel -- per-owner super-k-mer records, about a third of v2's bytes on the links --, the
//      owner k-merises what it receives, and every shard holds an ordinary table (mcx_superk.h);
//   v2 (any k; MCX_MULTI_EXCHANGE=v2 forces it): the table is sharded by quotient-hash prefix and
//      packed OCCURRENCES travel.
// Either way a piece's blocks have a fixed size, go out with one peer copy per (owner, buffer) and carry
// their fills with them: the host never reads a count.
//
// Data path of a batch of reads in format v2 (the same kernels bench.py drives through
// torch.distributed / RCCL with one process per GPU):
//   1. the batch is cut into one contiguous piece per shard; shard i stages its piece and k-merises
//      it with the sender kernel (k_stream_bin, BIN_GLOBAL): packed occurrences binned by
//      (owner, region) into fixed-size blocks, one block per owner                     [stream i]
//   2. block j of shard i goes to shard j's receive slot for sender i: hipMemcpyPeerAsync over
//      xGMI, a direct copy per pair (every pair of GPUs of a node is one hop)      [copy stream i]
//   3. shard j splits what it received by sub-table (k_tuples_bin); its LDS insert applies it at
//      the next flush                                                                   [stream j]
// Send and receive sets exist kSets (3) times and the three steps are ordered with HIP events only,
// so the sender kernel of piece n + 1 overlaps with the copies of piece n and the owners' split of
// piece n - 1.  Nothing here reads a table, so nothing flushes one.  The same device may be named
// more than once (-D 0,0): the peer copy then is a device-local copy, which is how the path is
// tested on a one-GPU box.
#pragma once

namespace {

struct XBuf {  // one block set: packed tuples by (owner, region) + per-owner overflow bins (full tuples)
  uint64_t *keys = nullptr;
  unsigned long long *counts = nullptr;
  uint64_t *ov_keys = nullptr;    // send sets: [N][ov_cap], then the spill area [sp_cap] (BinOut::ov_keys)
  uint8_t *ov_edges = nullptr;
  unsigned long long *ov_counts = nullptr;  // send sets: [N] fills, [N] = spill fill, [N + 1] = spill capacity
  // format v3: super-k-mer records by (owner, replica segment)
  void *recs = nullptr;                      // send: [N][segs][sk_cap] records; receive: [segs][sk_cap]
  unsigned long long *fills_rm = nullptr;    // send: fills as the sender kernel writes them, [segs][N]
  unsigned long long *fills = nullptr;       // send: the same, [N][segs] (one row per owner: what travels); receive: [segs]
  void *sp_recs = nullptr;                   // send: spill area (SuperkOut::sp_recs / sp_own / sp_count)
  uint8_t *sp_own = nullptr;
  unsigned long long *sp_count = nullptr;
};

}  // namespace

// Send / receive sets per (sender, owner): the sender kernel of piece n + 1 overlaps the copies of piece n and the owners'
// kernels of piece n - 1, and the host may run one piece further ahead -- it reads a set's spill count (pinned memory,
// written behind the sender kernel) when it REUSES the set, i.e. it waits for the sender kernel of kSets pieces ago.  With
// two sets that capped the run-ahead at two pieces (round 4's advisor); three since round 5.
constexpr int kSets = 3;

struct mcx_group {
  int n = 0;
  std::vector<mcx_graph *> part;
  // geometry of one exchange piece (at most max_pos k-mer start positions)
  uint32_t segs = 0;
  uint64_t seg_cap = 0, ov_cap = 0, max_pos = 0;
  bool v3 = false;      // exchange format v3 (minimizer ownership, ordinary per-shard tables)
  uint32_t sk_segs = 0; // v3: replica segments per owner
  uint64_t sk_cap = 0;  // v3: records per segment
  uint64_t sp_cap = 0;  // spill area of a send set: every occurrence of a piece fits, so no input can overflow
  std::vector<std::array<unsigned long long *, kSets>> h_spill;  // [i][b] pinned: spill fill of the set's last piece
  std::vector<std::array<int, kSets>> spill_colour;               // colour of that piece
  // send[i][b]: blocks for all owners on shard i's device; recv[j][i][b]: shard i's block on shard j's device
  std::vector<std::array<XBuf, kSets>> send;
  std::vector<std::vector<std::array<XBuf, kSets>>> recv;
  std::vector<hipStream_t> cs;                                     // copy stream of sender i (its device)
  std::vector<std::array<hipEvent_t, kSets>> filled, sent;             // [i][b]
  std::vector<std::vector<std::array<hipEvent_t, kSets>>> arrived;     // [j][i][b], recorded on cs[i]
  std::vector<std::vector<std::array<hipEvent_t, kSets>>> consumed;    // [j][i][b], recorded on shard j's stream
  std::vector<int> cur;
  std::vector<std::array<bool, kSets>> used;
  bool buffers = false;
  uint64_t spilled = 0;  // occurrences (v2) / records (v3) that went through a sender's spill area (mcx_graph_insert_stats)
  bool peer_ok = true;  // every pair of distinct devices can map each other's memory (kernels may write to a peer)
};

#define GRP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(_e == hipErrorOutOfMemory ? MCX_ERR_NOMEM : MCX_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                  hipGetErrorString(_e), __FILE__, __LINE__);                                      \
  } while (0)

static void group_free_buffers(mcx_group *G)
{
  for (int i = 0; i < G->n; i++) {
    (void)hipSetDevice(G->part[i]->device);
    for (int b = 0; b < kSets; b++) {
      if ((size_t)i < G->send.size()) {
        XBuf &x = G->send[i][b];
        (void)hipFree(x.keys); (void)hipFree(x.counts); (void)hipFree(x.ov_keys); (void)hipFree(x.ov_edges); (void)hipFree(x.ov_counts);
        (void)hipFree(x.recs); (void)hipFree(x.fills_rm); (void)hipFree(x.fills); (void)hipFree(x.sp_recs); (void)hipFree(x.sp_own); (void)hipFree(x.sp_count);
        x = XBuf();
        if ((size_t)i < G->h_spill.size() && G->h_spill[i][b]) { (void)hipHostFree(G->h_spill[i][b]); G->h_spill[i][b] = nullptr; }
      }
      if ((size_t)i < G->recv.size())
        for (auto &r : G->recv[i]) {
          XBuf &x = r[b];
          (void)hipFree(x.keys); (void)hipFree(x.counts); (void)hipFree(x.ov_keys); (void)hipFree(x.ov_edges); (void)hipFree(x.ov_counts);
          (void)hipFree(x.recs); (void)hipFree(x.fills);
          x = XBuf();
        }
    }
  }
  G->buffers = false;
}

// Geometry of one exchange piece of at most max_pos k-mer start positions (everything group_alloc_buffers sizes).
// v3: ~2.3 records per 16 positions on random reads; room for 3: what does not fit a segment goes to the sender's
// spill area and is routed by the host.  A record is a run of >= 1 k-mers, so a piece yields at most one record per
// start position: a spill area of max_pos records cannot overflow WHATEVER the input is -- one owner taking
// everything, or an owner change at every k-mer.  v2: a segment holds mean + 8 sigma of the Poisson fill + room for
// a short run of one k-mer; what does not fit goes to the owner's overflow bin (hot k-mers), beyond that to the
// spill area, which holds a whole piece as well.
static int group_layout(mcx_group *G, uint64_t max_pos)
{
  mcx_graph *g0 = G->part[0];
  const int N = G->n, W = g0->W;
  G->max_pos = max_pos;
  if (G->v3) {
    G->sk_segs = kSuperkRep;
    G->sk_cap = max_pos * 3 / 16 / ((uint64_t)N * G->sk_segs) + 4096;
    G->sp_cap = max_pos + 16;
    if (const char *e = getenv("MCX_MULTI_SKCAP")) G->sk_cap = std::max<uint64_t>(16, strtoull(e, nullptr, 10));  // tests: force the spill path
    return MCX_OK;
  }
  const uint32_t b1 = 1u << g0->t.lb1;
  G->segs = kShardRep * b1;
  const double mean = (double)max_pos / ((double)N * G->segs);
  G->seg_cap = ((uint64_t)(mean + 8.0 * sqrt(mean + 1.0)) + 64 + 1) & ~1ull;
  G->ov_cap = (std::max<uint64_t>(1u << 16, max_pos / (uint64_t)N / 16) + 15) & ~15ull;  // (16-byte aligned edge-byte rows: k_copy_filled)
  G->sp_cap = max_pos;
  // k_copy_filled addresses segment js at js * ceil(cap * item_bytes / 16) 16-byte units: that is the real stride
  // only when a segment is a whole number of units
  if ((G->seg_cap * 8 * W) % 16 || (G->ov_cap * 8 * W) % 16 || G->ov_cap % 16)
    return fail(MCX_ERR_ARG, "internal: exchange segment stride is not a multiple of 16 bytes (seg_cap %llu, ov_cap %llu)",
                (unsigned long long)G->seg_cap, (unsigned long long)G->ov_cap);
  return MCX_OK;
}

// HBM the exchange buffers of the present layout take on ONE device: kSets send sets + its receive slots for all N senders.
// With the default piece (128 Mi positions): 9.3 GB (v3, k <= 31), 18.2 GB (v3, k = 63), 11.1 GB (v2, k <= 31) whatever
// the number of devices, of which the spill areas are 6.8 / 13.3 / 3.6 GB -- the price of "no input can overflow".
static uint64_t group_buffer_bytes(const mcx_group *G)
{
  const uint64_t N = (uint64_t)G->n, W = (uint64_t)G->part[0]->W;
  if (G->v3) {
    const uint64_t recb = 16 * W, blk_recs = (uint64_t)G->sk_segs * G->sk_cap;
    const uint64_t send = N * blk_recs * recb + 2 * N * G->sk_segs * 8 + G->sp_cap * (recb + 1) + 8;
    const uint64_t recv = N * (blk_recs * recb + (uint64_t)G->sk_segs * 8);
    return kSets * (send + recv);
  }
  const uint64_t blk = (uint64_t)G->segs * G->seg_cap;
  const uint64_t send = N * blk * 8 * W + N * G->segs * 8 + (N * G->ov_cap + G->sp_cap) * (8 * W + 1) + (N + 2) * 8;
  const uint64_t recv = N * (blk * 8 * W + (uint64_t)G->segs * 8 + G->ov_cap * (8 * W + 1) + 8);
  return kSets * (send + recv);
}

static int group_alloc_buffers(mcx_group *G)
{
  const int N = G->n, W = G->part[0]->W;
  G->send.clear(); G->recv.clear();  // (a failed attempt has been released by group_free_buffers)
  G->send.resize(N);
  G->recv.assign(N, std::vector<std::array<XBuf, kSets>>(N));
  G->h_spill.assign(N, {});
  G->spill_colour.assign(N, {});
  if (G->v3) {
    const uint64_t recb = 16ull * W;
    const uint64_t blk_recs = (uint64_t)G->sk_segs * G->sk_cap;
    for (int i = 0; i < N; i++) {
      GRP_TRY(hipSetDevice(G->part[i]->device));
      for (int b = 0; b < kSets; b++) {
        XBuf &s = G->send[i][b];
        GRP_TRY(hipMalloc((void **)&s.recs, (uint64_t)N * blk_recs * recb));
        GRP_TRY(hipMalloc((void **)&s.fills_rm, (uint64_t)N * G->sk_segs * 8));
        GRP_TRY(hipMalloc((void **)&s.fills, (uint64_t)N * G->sk_segs * 8));
        GRP_TRY(hipMalloc((void **)&s.sp_recs, G->sp_cap * recb));
        GRP_TRY(hipMalloc((void **)&s.sp_own, G->sp_cap));
        GRP_TRY(hipMalloc((void **)&s.sp_count, 8));
        GRP_TRY(hipHostMalloc((void **)&G->h_spill[i][b], 8, hipHostMallocDefault));
        *G->h_spill[i][b] = 0;
        for (int src = 0; src < N; src++) {
          XBuf &r = G->recv[i][src][b];
          GRP_TRY(hipMalloc((void **)&r.recs, blk_recs * recb));
          GRP_TRY(hipMalloc((void **)&r.fills, (uint64_t)G->sk_segs * 8));
        }
      }
    }
    return MCX_OK;
  }
  const uint64_t blk = (uint64_t)G->segs * G->seg_cap;  // tuples of one owner's block
  for (int i = 0; i < N; i++) {
    GRP_TRY(hipSetDevice(G->part[i]->device));
    for (int b = 0; b < kSets; b++) {
      XBuf &s = G->send[i][b];
      GRP_TRY(hipMalloc((void **)&s.keys, (uint64_t)N * blk * 8 * W));
      GRP_TRY(hipMalloc((void **)&s.counts, (uint64_t)N * G->segs * 8));
      GRP_TRY(hipMalloc((void **)&s.ov_keys, ((uint64_t)N * G->ov_cap + G->sp_cap) * 8 * W));
      GRP_TRY(hipMalloc((void **)&s.ov_edges, (uint64_t)N * G->ov_cap + G->sp_cap));
      GRP_TRY(hipMalloc((void **)&s.ov_counts, (uint64_t)(N + 2) * 8));
      GRP_TRY(hipMemcpy(s.ov_counts + N + 1, &G->sp_cap, 8, hipMemcpyHostToDevice));
      GRP_TRY(hipHostMalloc((void **)&G->h_spill[i][b], 8, hipHostMallocDefault));
      *G->h_spill[i][b] = 0;
      for (int src = 0; src < N; src++) {
        XBuf &r = G->recv[i][src][b];
        GRP_TRY(hipMalloc((void **)&r.keys, blk * 8 * W));
        GRP_TRY(hipMalloc((void **)&r.counts, (uint64_t)G->segs * 8));
        GRP_TRY(hipMalloc((void **)&r.ov_keys, G->ov_cap * 8 * W));
        GRP_TRY(hipMalloc((void **)&r.ov_edges, G->ov_cap));
        GRP_TRY(hipMalloc((void **)&r.ov_counts, 8));
      }
    }
  }
  return MCX_OK;
}

static uint64_t group_default_piece()
{
  uint64_t max_pos = std::max<uint64_t>(kStageBytes + kCarry, 32ull << 20);  // a staged chunk of mcx_graph_add_reads is one piece
  if (const char *e = getenv("MCX_MULTI_PIECE")) max_pos = std::max<uint64_t>(4096, strtoull(e, nullptr, 10));  // tests
  return max_pos;
}

// Exchange buffers, allocated on first use.  A piece covers at most max_pos start positions: a staged chunk of
// mcx_graph_add_reads (128 Mi positions by default) is one piece, longer device streams are cut to this size.  When the
// buffers do not fit beside the tables -- they are allocated after the tables and the partition workspace -- the piece
// is halved (down to 8 Mi positions: more launches per chunk, same result) instead of failing the build.
static int group_ensure_buffers(mcx_group *G)
{
  if (G->buffers) return MCX_OK;
  for (uint64_t max_pos = group_default_piece();; max_pos /= 2) {
    int rc = group_layout(G, max_pos);
    if (rc != MCX_OK) return rc;
    rc = group_alloc_buffers(G);
    if (rc == MCX_OK) break;
    group_free_buffers(G);
    if (rc != MCX_ERR_NOMEM || max_pos / 2 < (8ull << 20)) return rc;
    (void)hipGetLastError();
    if (getenv("MCX_TIMING")) fprintf(stderr, "[timing] exchange buffers for pieces of %llu positions do not fit: halving the piece\n", (unsigned long long)max_pos);
  }
  G->buffers = true;
  return MCX_OK;
}

// copies between the shards: k_copy_filled over peer-mapped pointers (default) or whole blocks with hipMemcpyPeerAsync
// (MCX_MULTI_COPY=memcpy; also what is used when a pair of devices has no peer access)
static bool group_copy_kernel(const mcx_group *G)
{
  static const bool want = [] { const char *e = getenv("MCX_MULTI_COPY"); return !(e && !strcmp(e, "memcpy")); }();
  return want && G->peer_ok;
}

// Hot k-mers.  A segment holds mean + 8 sigma, an owner's overflow bin max(64 K, piece / N / 16) more;
// one k-mer tens of thousands of times in a piece (poly-G reads, satellite repeats, a homopolymer
// contig) goes beyond both.  The sender then appends the occurrence to the spill area of its send set
// (bin_writeout), which holds a whole piece, and copies the fill to pinned memory behind the kernel.
// The host looks at that number when the send set is next used and when the group is drained --
// moments at which the kernel has long finished, so the common case costs no wait -- and, if it is
// not zero, hands the spilled tuples to every shard, which inserts the keys it owns with the
// lock-free direct insert (k_insert_tuples, only_own).  The one-GPU path does the same on the spot
// (bin_writeout -> probe_insert); here the table is on other devices, hence the detour.
static int group_route_spill(mcx_group *G, int idx, int b)
{
  if (!G->used[idx][b]) return MCX_OK;
  mcx_graph *me = G->part[idx];
  GRP_TRY(hipSetDevice(me->device));
  GRP_TRY(hipEventSynchronize(G->filled[idx][b]));
  const uint64_t n = std::min<uint64_t>(*G->h_spill[idx][b], G->sp_cap);
  if (!n) return MCX_OK;
  *G->h_spill[idx][b] = 0;
  G->spilled += n;
  const int N = G->n, W = me->W, colour = G->spill_colour[idx][b];
  const XBuf &s = G->send[idx][b];
  if (G->v3) {  // records: every shard picks its own out of the spill and k-merises them
    const uint64_t recb = 16ull * W;
    for (int j = 0; j < N; j++) {
      mcx_graph *own = G->part[j];
      GRP_TRY(hipSetDevice(own->device));
      DevBuf<uint8_t> r, o, d;  // (freed on every way out; the owner's stream is synchronised before the normal one)
      DevBuf<unsigned long long> cnt;
      GRP_TRY(r.alloc(n * recb));
      GRP_TRY(o.alloc(n));
      GRP_TRY(d.alloc(n * recb));
      GRP_TRY(cnt.alloc(1));
      GRP_TRY(hipMemcpyPeerAsync(r, own->device, s.sp_recs, me->device, n * recb, own->stream));
      GRP_TRY(hipMemcpyPeerAsync(o, own->device, s.sp_own, me->device, n, own->stream));
      GRP_TRY(hipMemsetAsync(cnt, 0, 8, own->stream));
      const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
      if (W == 1) hipLaunchKernelGGL(k_superk_pick<1>, dim3(blocks), dim3(256), 0, own->stream, (const void *)r.p, (const uint8_t *)o.p, n, (uint32_t)j, (void *)d.p, cnt.p);
      else hipLaunchKernelGGL(k_superk_pick<2>, dim3(blocks), dim3(256), 0, own->stream, (const void *)r.p, (const uint8_t *)o.p, n, (uint32_t)j, (void *)d.p, cnt.p);
      GRP_TRY(hipGetLastError());
      int rc = mcx_graph_add_superk_dev(own, colour, d.p, cnt.p, 1, n, n * 16);
      const hipError_t se = hipStreamSynchronize(own->stream);  // the buffers are released when this scope ends
      if (rc != MCX_OK) return rc;
      GRP_TRY(se);  // (an asynchronous fault of the pick / the owner kernel is reported here, at the piece that caused it)
    }
    GRP_TRY(hipSetDevice(me->device));
    return MCX_OK;
  }
  const uint64_t at = (uint64_t)N * G->ov_cap;
  for (int j = 0; j < N; j++) {
    mcx_graph *own = G->part[j];
    GRP_TRY(hipSetDevice(own->device));
    DevBuf<uint64_t> k;
    DevBuf<uint8_t> e;
    GRP_TRY(k.alloc(n * W));
    GRP_TRY(e.alloc(n));
    GRP_TRY(hipMemcpyPeerAsync(k.p, own->device, s.ov_keys + at * W, me->device, n * 8 * W, own->stream));
    GRP_TRY(hipMemcpyPeerAsync(e.p, own->device, s.ov_edges + at, me->device, n, own->stream));
    DISPATCH_WC(own, launch_insert_tuples_t, own, colour, (const uint64_t *)k.p, (const uint8_t *)e.p, n, 1u);
    const hipError_t le = hipGetLastError();
    const hipError_t se = hipStreamSynchronize(own->stream);  // the buffers are released when this scope ends
    GRP_TRY(le);
    GRP_TRY(se);
  }
  GRP_TRY(hipSetDevice(me->device));
  return MCX_OK;
}

// Steps 1-3 for the start positions [L.pos_lo, L.pos_hi) of a stream that is resident on shard
// idx's device (the sharded counterpart of submit_stream).
static int group_submit_stream(mcx_group *G, int idx, const StreamLaunch &L, int colour)
{
  int rc = group_ensure_buffers(G);
  if (rc != MCX_OK) return rc;
  const int N = G->n;
  mcx_graph *me = G->part[idx];
  const int W = me->W;
  if (G->v3) {
    const uint64_t recb = 16ull * W, blk_recs = (uint64_t)G->sk_segs * G->sk_cap;
    uint32_t lbo = 0;
    while ((1 << lbo) < N) lbo++;
    for (uint64_t lo = L.pos_lo; lo < L.pos_hi;) {
      const uint64_t hi = std::min(L.pos_hi, lo + G->max_pos);
      const int b = G->cur[idx];
      G->cur[idx] = (G->cur[idx] + 1) % kSets;
      XBuf &s = G->send[idx][b];
      rc = group_route_spill(G, idx, b);  // what this set's previous piece spilled (nearly always nothing)
      if (rc != MCX_OK) return rc;
      // 1. sender: minimizers, per-owner records
      GRP_TRY(hipSetDevice(me->device));
      if (G->used[idx][b]) GRP_TRY(hipStreamWaitEvent(me->stream, G->sent[idx][b], 0));  // its last copies have left
      GRP_TRY(hipMemsetAsync(s.fills_rm, 0, (uint64_t)N * G->sk_segs * 8, me->stream));
      GRP_TRY(hipMemsetAsync(s.sp_count, 0, 8, me->stream));
      StreamLaunch P = L;
      P.pos_lo = lo; P.pos_hi = hi;
      SuperkOut so{s.recs, s.fills_rm, G->sk_cap, lbo, G->sk_segs, s.sp_recs, s.sp_own, s.sp_count, G->sp_cap};
      rc = superk_bins_launch(me, P, so);
      if (rc != MCX_OK) return rc;
      hipLaunchKernelGGL(k_transpose_fills, dim3(((unsigned)N * G->sk_segs + 63) / 64), dim3(64), 0, me->stream,
                         (const unsigned long long *)s.fills_rm, s.fills, G->sk_segs, (uint32_t)N);
      GRP_TRY(hipGetLastError());
      GRP_TRY(hipMemcpyAsync(G->h_spill[idx][b], s.sp_count, 8, hipMemcpyDeviceToHost, me->stream));
      G->spill_colour[idx][b] = colour;
      GRP_TRY(hipEventRecord(G->filled[idx][b], me->stream));
      // 2. one copy of the fills and one of the records per owner: whole segments, so no count is read on the host
      GRP_TRY(hipStreamWaitEvent(G->cs[idx], G->filled[idx][b], 0));
      if (group_copy_kernel(G)) {  // the filled parts only, one launch (k_copy_filled)
        PeerDst pd{};
        for (int j = 0; j < N; j++) {
          if (G->used[idx][b]) GRP_TRY(hipStreamWaitEvent(G->cs[idx], G->consumed[j][idx][b], 0));  // the slot is free again
          pd.data[j] = G->recv[j][idx][b].recs;
          pd.fills[j] = G->recv[j][idx][b].fills;
        }
        hipLaunchKernelGGL(k_copy_filled, dim3(2048), dim3(256), 0, G->cs[idx], (const ulonglong2 *)s.recs, (const unsigned long long *)s.fills, pd,
                           (uint32_t)N, G->sk_segs, G->sk_cap, (uint32_t)recb);
        GRP_TRY(hipGetLastError());
        for (int j = 0; j < N; j++) GRP_TRY(hipEventRecord(G->arrived[j][idx][b], G->cs[idx]));
      } else
      for (int j = 0; j < N; j++) {
        mcx_graph *own = G->part[j];
        XBuf &r = G->recv[j][idx][b];
        if (G->used[idx][b]) GRP_TRY(hipStreamWaitEvent(G->cs[idx], G->consumed[j][idx][b], 0));  // the slot is free again
        GRP_TRY(hipMemcpyPeerAsync(r.fills, own->device, s.fills + (uint64_t)j * G->sk_segs, me->device, (uint64_t)G->sk_segs * 8, G->cs[idx]));
        GRP_TRY(hipMemcpyPeerAsync(r.recs, own->device, (const uint8_t *)s.recs + (uint64_t)j * blk_recs * recb, me->device, blk_recs * recb, G->cs[idx]));
        GRP_TRY(hipEventRecord(G->arrived[j][idx][b], G->cs[idx]));
      }
      GRP_TRY(hipEventRecord(G->sent[idx][b], G->cs[idx]));
      // 3. owners: k-merise what arrived into their region bins.  The reservation is a true upper bound -- every start
      // position of the piece may belong to one owner -- and what the records really held comes off the owner's books
      // when the launch has settled (Counters::binned, snap_*).  (Until round 5 it was an estimate, 1.25 x piece / N:
      // an owner far above its share overflowed its bins into the per-occurrence insert without anyone noticing.)
      const uint64_t share = hi - lo;
      for (int j = 0; j < N; j++) {
        mcx_graph *own = G->part[j];
        XBuf &r = G->recv[j][idx][b];
        GRP_TRY(hipSetDevice(own->device));
        GRP_TRY(hipStreamWaitEvent(own->stream, G->arrived[j][idx][b], 0));
        rc = mcx_graph_add_superk_dev(own, colour, r.recs, r.fills, G->sk_segs, G->sk_cap, std::min(share, blk_recs * 16));
        if (rc != MCX_OK) return rc;
        GRP_TRY(hipEventRecord(G->consumed[j][idx][b], own->stream));
      }
      G->used[idx][b] = true;
      lo = hi;
    }
    GRP_TRY(hipSetDevice(me->device));
    return MCX_OK;
  }
  const uint64_t blk = (uint64_t)G->segs * G->seg_cap;
  for (uint64_t lo = L.pos_lo; lo < L.pos_hi;) {
    const uint64_t hi = std::min(L.pos_hi, lo + G->max_pos);
    const int b = G->cur[idx];
    G->cur[idx] = (G->cur[idx] + 1) % kSets;
    XBuf &s = G->send[idx][b];
    rc = group_route_spill(G, idx, b);  // what this set's previous piece spilled (nearly always nothing)
    if (rc != MCX_OK) return rc;
    // 1. sender
    GRP_TRY(hipSetDevice(me->device));
    if (G->used[idx][b]) GRP_TRY(hipStreamWaitEvent(me->stream, G->sent[idx][b], 0));  // its last copies have left
    GRP_TRY(hipMemsetAsync(s.counts, 0, (uint64_t)N * G->segs * 8, me->stream));
    GRP_TRY(hipMemsetAsync(s.ov_counts, 0, (uint64_t)(N + 1) * 8, me->stream));
    StreamLaunch P = L;
    P.pos_lo = lo; P.pos_hi = hi;
    rc = shard_bins_launch(me, P, s.keys, s.counts, G->seg_cap, s.ov_keys, s.ov_edges, s.ov_counts, G->ov_cap, true);
    if (rc != MCX_OK) return rc;
    GRP_TRY(hipMemcpyAsync(G->h_spill[idx][b], s.ov_counts + N, 8, hipMemcpyDeviceToHost, me->stream));
    G->spill_colour[idx][b] = colour;
    GRP_TRY(hipEventRecord(G->filled[idx][b], me->stream));
    // 2. one copy per (owner, buffer): fixed sizes, so the fills travel with the blocks and the host
    // never has to read them
    GRP_TRY(hipStreamWaitEvent(G->cs[idx], G->filled[idx][b], 0));
    const bool ck = group_copy_kernel(G);
    if (ck) {  // the packed tuples: filled parts of all (owner, segment) bins in one launch; the overflow bins below as before
      PeerDst pd{};
      for (int j = 0; j < N; j++) {
        if (G->used[idx][b]) GRP_TRY(hipStreamWaitEvent(G->cs[idx], G->consumed[j][idx][b], 0));  // the slot is free again
        pd.data[j] = G->recv[j][idx][b].keys;
        pd.fills[j] = G->recv[j][idx][b].counts;
      }
      hipLaunchKernelGGL(k_copy_filled, dim3(2048), dim3(256), 0, G->cs[idx], (const ulonglong2 *)s.keys, (const unsigned long long *)s.counts, pd,
                         (uint32_t)N, G->segs, G->seg_cap, (uint32_t)(8 * W));
      // the owners' overflow bins (full tuples: key words, then edge bytes): one segment per owner, filled part only
      for (int j = 0; j < N; j++) { pd.data[j] = G->recv[j][idx][b].ov_keys; pd.fills[j] = G->recv[j][idx][b].ov_counts; }
      hipLaunchKernelGGL(k_copy_filled, dim3(256), dim3(256), 0, G->cs[idx], (const ulonglong2 *)s.ov_keys, (const unsigned long long *)s.ov_counts, pd,
                         (uint32_t)N, 1u, G->ov_cap, (uint32_t)(8 * W));
      for (int j = 0; j < N; j++) pd.data[j] = G->recv[j][idx][b].ov_edges;
      hipLaunchKernelGGL(k_copy_filled, dim3(256), dim3(256), 0, G->cs[idx], (const ulonglong2 *)s.ov_edges, (const unsigned long long *)s.ov_counts, pd,
                         (uint32_t)N, 1u, G->ov_cap, 1u);
      GRP_TRY(hipGetLastError());
    }
    for (int j = 0; j < N; j++) {
      mcx_graph *own = G->part[j];
      XBuf &r = G->recv[j][idx][b];
      if (!ck) {
        if (G->used[idx][b]) GRP_TRY(hipStreamWaitEvent(G->cs[idx], G->consumed[j][idx][b], 0));  // the slot is free again
        GRP_TRY(hipMemcpyPeerAsync(r.counts, own->device, s.counts + (uint64_t)j * G->segs, me->device, (uint64_t)G->segs * 8, G->cs[idx]));
        GRP_TRY(hipMemcpyPeerAsync(r.keys, own->device, s.keys + (uint64_t)j * blk * W, me->device, blk * 8 * W, G->cs[idx]));
        GRP_TRY(hipMemcpyPeerAsync(r.ov_counts, own->device, s.ov_counts + j, me->device, 8, G->cs[idx]));
        GRP_TRY(hipMemcpyPeerAsync(r.ov_keys, own->device, s.ov_keys + (uint64_t)j * G->ov_cap * W, me->device, G->ov_cap * 8 * W, G->cs[idx]));
        GRP_TRY(hipMemcpyPeerAsync(r.ov_edges, own->device, s.ov_edges + (uint64_t)j * G->ov_cap, me->device, G->ov_cap, G->cs[idx]));
      }
      GRP_TRY(hipEventRecord(G->arrived[j][idx][b], G->cs[idx]));
    }
    GRP_TRY(hipEventRecord(G->sent[idx][b], G->cs[idx]));
    // 3. owners: split by sub-table; the overflow bins take the full-tuple path.  Booked with a true upper bound:
    // an owner's block holds at most `blk` tuples, and a piece at most hi - lo.
    const uint64_t share = hi - lo;
    for (int j = 0; j < N; j++) {
      mcx_graph *own = G->part[j];
      XBuf &r = G->recv[j][idx][b];
      GRP_TRY(hipSetDevice(own->device));
      GRP_TRY(hipStreamWaitEvent(own->stream, G->arrived[j][idx][b], 0));
      rc = mcx_graph_add_segments_dev(own, colour, r.keys, r.counts, G->segs, G->seg_cap, std::min(share, blk));
      if (rc != MCX_OK) return rc;
      rc = mcx_graph_insert_tuple_segments_dev(own, colour, r.ov_keys, r.ov_edges, r.ov_counts, 1, G->ov_cap);
      if (rc != MCX_OK) return rc;
      GRP_TRY(hipEventRecord(G->consumed[j][idx][b], own->stream));
    }
    G->used[idx][b] = true;
    lo = hi;
  }
  GRP_TRY(hipSetDevice(me->device));
  return MCX_OK;
}

static void group_destroy(mcx_group *G)
{
  if (!G) return;
  for (int i = 0; i < G->n; i++) {
    if (!G->part[i]) continue;
    (void)hipSetDevice(G->part[i]->device);
    if (G->part[i]->stream) (void)hipStreamSynchronize(G->part[i]->stream);
    if ((size_t)i < G->cs.size() && G->cs[i]) (void)hipStreamSynchronize(G->cs[i]);
  }
  group_free_buffers(G);
  for (int i = 0; i < G->n; i++) {
    if (!G->part[i]) continue;
    (void)hipSetDevice(G->part[i]->device);
    if ((size_t)i < G->cs.size() && G->cs[i]) (void)hipStreamDestroy(G->cs[i]);
    for (int b = 0; b < kSets; b++) {
      if ((size_t)i < G->filled.size() && G->filled[i][b]) (void)hipEventDestroy(G->filled[i][b]);
      if ((size_t)i < G->sent.size() && G->sent[i][b]) (void)hipEventDestroy(G->sent[i][b]);
    }
  }
  for (auto &row : G->arrived) for (auto &e : row) for (int b = 0; b < kSets; b++) if (e[b]) (void)hipEventDestroy(e[b]);
  for (auto &row : G->consumed) for (auto &e : row) for (int b = 0; b < kSets; b++) if (e[b]) (void)hipEventDestroy(e[b]);
  for (int j = 0; j < G->n; j++) {
    if (!G->part[j]) continue;
    G->part[j]->group = nullptr;
    mcx_graph_destroy(G->part[j]);
  }
  delete G;
}

// Peer self-test at start-up: the can-access-peer matrix as the runtime reports it, and for every ordered pair of
// distinct devices ONE k_copy_filled launch on the sender's copy stream that writes a pattern and its fill into a
// buffer on the owner's device through the peer-mapped pointer, read back from the owner.  A pair that fails turns
// the kernel copy off for the whole group (hipMemcpyPeerAsync of whole blocks instead: what MCX_MULTI_COPY=memcpy
// selects by hand) and says so once on stderr -- a build must not find out at its first piece, or not at all, that
// stores to a peer do not land.  (A fault inside the kernel cannot be caught here: on a node where peer mappings are
// broken outright, MCX_MULTI_COPY=memcpy skips the test.)  MCX_TIMING=1 prints the matrix.
static void group_peer_selftest(mcx_group *G, const int *devices)
{
  const int N = G->n;
  const bool verbose = getenv("MCX_TIMING") != nullptr;
  bool distinct = false;
  for (int i = 0; i < N; i++) for (int j = 0; j < i; j++) if (devices[i] != devices[j]) distinct = true;
  if (!distinct) return;
  if (verbose) {
    fprintf(stderr, "[timing] peer access (row = from device, column = to device; 1 = mapped):\n");
    for (int i = 0; i < N; i++) {
      fprintf(stderr, "[timing]   %2d:", devices[i]);
      for (int j = 0; j < N; j++) {
        int can = devices[i] == devices[j];
        if (!can) (void)hipDeviceCanAccessPeer(&can, devices[i], devices[j]);
        fprintf(stderr, " %d", can);
      }
      fprintf(stderr, "\n");
    }
  }
  if (!group_copy_kernel(G)) {
    if (!G->peer_ok) fprintf(stderr, "mcx: not every pair of the %d devices can map each other's memory: copies between the shards use hipMemcpyPeerAsync\n", N);
    return;
  }
  constexpr uint32_t kItems = 1024;  // 16 KiB of 16-byte items
  std::vector<ulonglong2> pat(kItems), back(kItems);
  const char *why = nullptr;
  int bad_i = -1, bad_j = -1;
  for (int i = 0; i < N && !why; i++)
    for (int j = 0; j < N && !why; j++) {
      if (devices[i] == devices[j]) continue;
      bool seen = false;  // (a device named twice: test each ordered pair of DEVICES once)
      for (int a = 0; a < i; a++) for (int b = 0; b < N; b++) if (devices[a] == devices[i] && devices[b] == devices[j]) seen = true;
      for (int b = 0; b < j; b++) if (devices[b] == devices[j]) seen = true;
      if (seen) continue;
      ulonglong2 *src = nullptr, *dst = nullptr;
      unsigned long long *fsrc = nullptr, *fdst = nullptr;
      const unsigned long long fill = kItems - 3, zero = 0;
      for (uint32_t t = 0; t < kItems; t++) pat[t] = make_ulonglong2(0x9E3779B97F4A7C15ull * (t + 1) + (uint64_t)i, ((uint64_t)i << 32 | (uint64_t)j) ^ t);
      auto ok = [&](hipError_t e, const char *what) { if (e != hipSuccess && !why) { why = what; (void)hipGetLastError(); } return e == hipSuccess; };
      bool good = ok(hipSetDevice(devices[j]), "hipSetDevice") && ok(hipMalloc((void **)&dst, kItems * 16), "hipMalloc") && ok(hipMalloc((void **)&fdst, 8), "hipMalloc") &&
                  ok(hipMemset(dst, 0, kItems * 16), "hipMemset") && ok(hipMemcpy(fdst, &zero, 8, hipMemcpyHostToDevice), "hipMemcpy") &&
                  ok(hipSetDevice(devices[i]), "hipSetDevice") && ok(hipMalloc((void **)&src, kItems * 16), "hipMalloc") && ok(hipMalloc((void **)&fsrc, 8), "hipMalloc") &&
                  ok(hipMemcpy(src, pat.data(), kItems * 16, hipMemcpyHostToDevice), "hipMemcpy") && ok(hipMemcpy(fsrc, &fill, 8, hipMemcpyHostToDevice), "hipMemcpy");
      if (good) {
        PeerDst pd{};
        pd.data[0] = dst; pd.fills[0] = fdst;
        hipLaunchKernelGGL(k_copy_filled, dim3(4), dim3(256), 0, G->cs[i], (const ulonglong2 *)src, (const unsigned long long *)fsrc, pd, 1u, 1u, (uint64_t)kItems, 16u);
        good = ok(hipGetLastError(), "k_copy_filled launch") && ok(hipStreamSynchronize(G->cs[i]), "k_copy_filled over a peer mapping");
      }
      unsigned long long got_fill = 0;
      if (good) good = ok(hipSetDevice(devices[j]), "hipSetDevice") && ok(hipMemcpy(back.data(), dst, kItems * 16, hipMemcpyDeviceToHost), "read back") &&
                       ok(hipMemcpy(&got_fill, fdst, 8, hipMemcpyDeviceToHost), "read back");
      if (good && (got_fill != fill || memcmp(back.data(), pat.data(), (size_t)fill * 16) != 0)) { why = "the stores did not land (pattern mismatch)"; good = false; }
      if (!good) { bad_i = devices[i]; bad_j = devices[j]; }
      (void)hipSetDevice(devices[i]); (void)hipFree(src); (void)hipFree(fsrc);
      (void)hipSetDevice(devices[j]); (void)hipFree(dst); (void)hipFree(fdst);
    }
  if (why) {
    G->peer_ok = false;
    fprintf(stderr, "mcx: peer self-test failed from device %d to device %d (%s): copies between the shards fall back to hipMemcpyPeerAsync "
                    "(MCX_MULTI_COPY=memcpy selects that without the test)\n", bad_i, bad_j, why);
  } else if (verbose) {
    fprintf(stderr, "[timing] peer self-test: k_copy_filled round trip passed for every ordered pair of devices\n");
  }
  (void)hipSetDevice(devices[0]);
}

// HBM per device that the exchange buffers of an N-device table will take (sizing: host/cmd_build.c adds it to the
// table when it checks -m / -n against the free HBM).  Same arithmetic as group_ensure_buffers' first attempt.
extern "C" int mcx_multi_exchange_bytes(int kmer_size, int ndevices, uint64_t capacity_kmers, uint64_t *bytes_per_device)
{
  if (!bytes_per_device) return fail(MCX_ERR_ARG, "null result pointer");
  *bytes_per_device = 0;
  if (ndevices <= 1) return MCX_OK;
  if (ndevices > 32 || (ndevices & (ndevices - 1))) return fail(MCX_ERR_ARG, "the number of devices must be a power of two <= 32 (got %d)", ndevices);
  if (check_k(kmer_size) != MCX_OK) return MCX_ERR_ARG;
  mcx_group G;
  mcx_graph proto;
  proto.k = kmer_size; proto.W = words_for_k(kmer_size);
  const char *e = getenv("MCX_MULTI_EXCHANGE");
  G.n = ndevices;
  G.v3 = mcx_superk_supported(kmer_size) && !(e && !strcmp(e, "v2"));
  G.part.assign(1, &proto);
  if (!G.v3) {  // the region count of a shard's table decides the segment count
    const uint64_t min_shard = 64ull << sub_shift_for_words(proto.W);
    const uint64_t per = std::max<uint64_t>((capacity_kmers + (uint64_t)ndevices - 1) / (uint64_t)ndevices, min_shard);
    const uint64_t sub_slots = 1ull << sub_shift_for_words(proto.W);
    uint32_t lbo = 0;
    while ((1 << lbo) < ndevices) lbo++;
    proto.t.lb1 = region_bits((std::max<uint64_t>(per, 1024) + sub_slots - 1) / sub_slots, lbo);
  }
  int rc = group_layout(&G, group_default_piece());
  if (rc == MCX_OK) *bytes_per_device = group_buffer_bytes(&G);
  G.part.clear();
  return rc;
}

extern "C" int mcx_graph_create_multi(mcx_graph **out, int kmer_size, int ncols, uint64_t capacity_kmers,
                                      const int *devices, int ndevices)
{
  if (!out) return fail(MCX_ERR_ARG, "null handle pointer");
  *out = nullptr;
  if (!devices || ndevices < 1) return fail(MCX_ERR_ARG, "no devices given");
  if (ndevices == 1) return mcx_graph_create(out, kmer_size, ncols, capacity_kmers, devices[0]);
  if (ndevices > 32 || (ndevices & (ndevices - 1)))
    return fail(MCX_ERR_ARG, "the table is split by hash prefix: the number of devices must be a power of two <= 32 (got %d)", ndevices);
  if (check_k(kmer_size) != MCX_OK) return MCX_ERR_ARG;
  if (kmer_size > 63)
    return fail(MCX_ERR_ARG, "k > 63 (three- and four-word keys) builds on one device: the exchange formats carry at most two key words");
  const int W = words_for_k(kmer_size);
  // every shard needs the partitioned insert path (>= 64 regions of one sub-table each)
  const uint64_t min_shard = 64ull << sub_shift_for_words(W);
  const uint64_t per = std::max<uint64_t>((capacity_kmers + (uint64_t)ndevices - 1) / (uint64_t)ndevices, min_shard);
  mcx_group *G = new mcx_group();
  G->n = ndevices;
  G->part.assign(ndevices, nullptr);
  {
    const char *e = getenv("MCX_MULTI_EXCHANGE");
    G->v3 = mcx_superk_supported(kmer_size) && !(e && !strcmp(e, "v2"));
  }
  uint32_t lbo = 0;
  while ((1 << lbo) < ndevices) lbo++;
  for (int i = 0; i < ndevices; i++) {
    // v3: an ordinary table per device (the keys are dealt out by minimizer); v2: a hash-prefix shard
    int rc = G->v3 ? mcx_graph_create(&G->part[i], kmer_size, ncols, per, devices[i])
                   : mcx_graph_create_shard(&G->part[i], kmer_size, ncols, per, devices[i], ndevices, i);
    if (rc != MCX_OK) { group_destroy(G); return rc; }
    G->part[i]->group = G;
    G->part[i]->gidx = i;
    if (G->v3) G->part[i]->own_lbo = lbo;
  }
  G->cs.assign(ndevices, nullptr);
  G->filled.assign(ndevices, {});
  G->sent.assign(ndevices, {});
  G->arrived.assign(ndevices, std::vector<std::array<hipEvent_t, kSets>>(ndevices, std::array<hipEvent_t, kSets>{}));
  G->consumed.assign(ndevices, std::vector<std::array<hipEvent_t, kSets>>(ndevices, std::array<hipEvent_t, kSets>{}));
  G->cur.assign(ndevices, 0);
  G->used.assign(ndevices, {});
#define MK_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { int rc_ = fail(MCX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); group_destroy(G); return rc_; } } while (0)
  for (int i = 0; i < ndevices; i++) {
    MK_TRY(hipSetDevice(devices[i]));
    for (int j = 0; j < ndevices; j++)  // direct xGMI copies instead of staging through the host
      if (devices[j] != devices[i]) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
          const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
          if (e != hipSuccess) { (void)hipGetLastError(); if (e != hipErrorPeerAccessAlreadyEnabled) G->peer_ok = false; }
        } else {
          G->peer_ok = false;
        }
      }
    MK_TRY(hipStreamCreateWithFlags(&G->cs[i], hipStreamNonBlocking));
    for (int b = 0; b < kSets; b++) {
      MK_TRY(hipEventCreateWithFlags(&G->filled[i][b], hipEventDisableTiming));
      MK_TRY(hipEventCreateWithFlags(&G->sent[i][b], hipEventDisableTiming));
      for (int j = 0; j < ndevices; j++) MK_TRY(hipEventCreateWithFlags(&G->arrived[j][i][b], hipEventDisableTiming));
    }
  }
  for (int j = 0; j < ndevices; j++) {
    MK_TRY(hipSetDevice(devices[j]));
    for (int i = 0; i < ndevices; i++)
      for (int b = 0; b < kSets; b++) MK_TRY(hipEventCreateWithFlags(&G->consumed[j][i][b], hipEventDisableTiming));
  }
#undef MK_TRY
  group_peer_selftest(G, devices);
  mcx_graph *f = new mcx_graph();  // the facade: no device state of its own
  f->k = kmer_size; f->W = W; f->ncols = ncols; f->ncols_vis = ncols;
  f->device = devices[0];
  f->as_group = G;
  *out = f;
  return MCX_OK;
}

extern "C" int mcx_graph_ndevices(const mcx_graph *g) { return !g ? 0 : g->as_group ? g->as_group->n : 1; }

// ---- the facade's calls -----------------------------------------------------------------------
// everything in flight between the shards has landed and been handed to its owner
static int grp_drain(mcx_group *G)
{
  if (G->buffers)
    for (int i = 0; i < G->n; i++)
      for (int b = 0; b < kSets; b++) {
        int rc = group_route_spill(G, i, b);
        if (rc != MCX_OK) return rc;
      }
  for (int i = 0; i < G->n; i++) {
    GRP_TRY(hipSetDevice(G->part[i]->device));
    GRP_TRY(hipStreamSynchronize(G->part[i]->stream));
    GRP_TRY(hipStreamSynchronize(G->cs[i]));
  }
  return MCX_OK;
}

static int grp_sync(mcx_group *G)
{
  int rc = grp_drain(G), first = MCX_OK;
  if (rc != MCX_OK) return rc;
  for (int i = 0; i < G->n; i++) {
    rc = mcx_graph_sync(G->part[i]);
    if (rc != MCX_OK && first == MCX_OK) first = rc;
  }
  return first;
}

static int grp_device_stats(mcx_group *G, mcx_load_stats *out)
{
  int rc = grp_drain(G), first = MCX_OK;
  if (rc != MCX_OK) return rc;
  memset(out, 0, sizeof(*out));
  for (int i = 0; i < G->n; i++) {
    mcx_load_stats s;
    rc = mcx_graph_device_stats(G->part[i], &s);
    if (rc != MCX_OK && first == MCX_OK) first = rc;
    out->num_good_reads += s.num_good_reads; out->num_bad_reads += s.num_bad_reads;
    out->contigs_parsed += s.contigs_parsed; out->num_kmers_loaded += s.num_kmers_loaded;
    out->num_kmers_novel += s.num_kmers_novel; out->total_bases_loaded += s.total_bases_loaded;
  }
  return first;
}

static int grp_add_reads(mcx_group *G, int colour, const uint8_t *bases, const uint8_t *quals, const uint64_t *off,
                         uint64_t nreads, uint8_t fq, uint8_t hp, mcx_load_stats *stats_accum)
{
  if (stats_accum) {
    stats_accum->num_se_reads += nreads;
    stats_accum->total_bases_read += nreads ? off[nreads] - off[0] : 0;
  }
  for (int i = 0; i < G->n; i++) {  // one contiguous piece of the batch per shard
    const uint64_t lo = nreads * (uint64_t)i / (uint64_t)G->n, hi = nreads * (uint64_t)(i + 1) / (uint64_t)G->n;
    if (hi == lo) continue;
    int rc = mcx_graph_add_reads(G->part[i], colour, bases, quals, off + lo, hi - lo, fq, hp, nullptr);
    if (rc != MCX_OK) return rc;
  }
  return MCX_OK;
}

// build --intersect on a sharded table: reads only update k-mers that are already in the graph, and an
// edge needs both of its k-mers found (ctx_build.c:341-363,409-413; build_graph.c:99-150) -- but
// consecutive k-mers of a read live on different shards.  Every shard gets the whole batch and walks
// it twice (k_reads_must_exist phases 1 and 2): first it records, per k-mer occurrence, whether a
// k-mer it owns was found; the arrays are OR-ed on device 0 and handed back to every shard; then
// each shard applies the reference's rule to the nodes it owns.  Not a fast path (as on one device).
static int grp_add_reads_must_exist(mcx_group *G, int colour, const uint8_t *bases, const uint8_t *quals, const uint64_t *off,
                                    uint64_t nreads, uint8_t fq, uint8_t hp, mcx_load_stats *stats_accum)
{
  if (stats_accum) {
    stats_accum->num_se_reads += nreads;
    stats_accum->total_bases_read += nreads ? off[nreads] - off[0] : 0;
  }
  if (!nreads) return MCX_OK;
  int rc = grp_drain(G);
  if (rc != MCX_OK) return rc;
  const int N = G->n;
  const uint64_t base0 = off[0], nb = off[nreads] - off[0];
  std::vector<uint64_t> rel(nreads + 1);
  for (uint64_t i = 0; i <= nreads; i++) rel[i] = off[i] - base0;
  std::vector<uint8_t *> d_bases(N, nullptr), d_quals(N, nullptr), d_present(N, nullptr);
  std::vector<uint64_t *> d_off(N, nullptr);
  uint8_t *d_tmp = nullptr;  // on device 0
  auto cleanup = [&]() {
    for (int i = 0; i < N; i++) {
      (void)hipSetDevice(G->part[i]->device);
      (void)hipStreamSynchronize(G->part[i]->stream);
      (void)hipFree(d_bases[i]); (void)hipFree(d_quals[i]); (void)hipFree(d_present[i]); (void)hipFree(d_off[i]);
    }
    (void)hipSetDevice(G->part[0]->device);
    (void)hipFree(d_tmp);
  };
#define ISEC_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { int rc_ = fail(_e == hipErrorOutOfMemory ? MCX_ERR_NOMEM : MCX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); cleanup(); return rc_; } } while (0)
  const unsigned blocks = (unsigned)((nreads + 127) / 128);
  auto launch = [&](int i, uint32_t phase) {
    mcx_graph *g = G->part[i];
    if (g->W == 1)
      hipLaunchKernelGGL((k_reads_must_exist<1>), dim3(blocks), dim3(128), 0, g->stream, g->t, (const uint8_t *)d_bases[i], (const uint8_t *)d_quals[i],
                         (const uint64_t *)d_off[i], nreads, g->k, (uint32_t)fq, (uint32_t)hp, (uint32_t)colour, g->d_ctr, d_present[i], phase, owner_spec(g));
    else
      hipLaunchKernelGGL((k_reads_must_exist<2>), dim3(blocks), dim3(128), 0, g->stream, g->t, (const uint8_t *)d_bases[i], (const uint8_t *)d_quals[i],
                         (const uint64_t *)d_off[i], nreads, g->k, (uint32_t)fq, (uint32_t)hp, (uint32_t)colour, g->d_ctr, d_present[i], phase, owner_spec(g));
  };
  for (int i = 0; i < N; i++) {  // upload, phase 1
    mcx_graph *g = G->part[i];
    ISEC_TRY(hipSetDevice(g->device));
    ISEC_TRY(hipMalloc((void **)&d_bases[i], nb + 16));
    ISEC_TRY(hipMalloc((void **)&d_off[i], (nreads + 1) * 8));
    ISEC_TRY(hipMalloc((void **)&d_present[i], nb + 16));
    if (quals && fq > 0) ISEC_TRY(hipMalloc((void **)&d_quals[i], nb + 16));
    ISEC_TRY(hipMemcpyAsync(d_bases[i], bases + base0, nb, hipMemcpyHostToDevice, g->stream));
    if (d_quals[i]) ISEC_TRY(hipMemcpyAsync(d_quals[i], quals + base0, nb, hipMemcpyHostToDevice, g->stream));
    ISEC_TRY(hipMemcpyAsync(d_off[i], rel.data(), (nreads + 1) * 8, hipMemcpyHostToDevice, g->stream));
    ISEC_TRY(hipMemsetAsync(d_present[i], 0, nb + 16, g->stream));
    launch(i, 1u);
    ISEC_TRY(hipGetLastError());
  }
  for (int i = 0; i < N; i++) { ISEC_TRY(hipSetDevice(G->part[i]->device)); ISEC_TRY(hipStreamSynchronize(G->part[i]->stream)); }
  {  // OR the shards' answers on device 0, hand the result back
    mcx_graph *g0 = G->part[0];
    ISEC_TRY(hipSetDevice(g0->device));
    ISEC_TRY(hipMalloc((void **)&d_tmp, nb + 16));
    for (int i = 1; i < N; i++) {
      ISEC_TRY(hipMemcpyPeerAsync(d_tmp, g0->device, d_present[i], G->part[i]->device, nb, g0->stream));
      hipLaunchKernelGGL(k_or_bytes, dim3((unsigned)std::min<uint64_t>((nb + 255) / 256, 4096)), dim3(256), 0, g0->stream, d_present[0], (const uint8_t *)d_tmp, nb);
      ISEC_TRY(hipGetLastError());
    }
    ISEC_TRY(hipStreamSynchronize(g0->stream));
    for (int i = 1; i < N; i++) ISEC_TRY(hipMemcpyPeer(d_present[i], G->part[i]->device, d_present[0], g0->device, nb));
  }
  for (int i = 0; i < N; i++) {  // phase 2
    ISEC_TRY(hipSetDevice(G->part[i]->device));
    launch(i, 2u);
    ISEC_TRY(hipGetLastError());
  }
  cleanup();
#undef ISEC_TRY
  return MCX_OK;
}

static int grp_intersect_finish(mcx_group *G, uint64_t *removed)
{
  int rc = grp_drain(G);
  uint64_t total = 0;
  for (int i = 0; rc == MCX_OK && i < G->n; i++) {
    uint64_t r = 0;
    rc = mcx_graph_intersect_finish(G->part[i], &r);
    total += r;
  }
  if (removed) *removed = total;
  return rc;
}

// --remove-pcr on a sharded table: every shard sees the whole batch and answers for the start
// nodes it owns (k_pcr_claim); the answers are copied to every shard, each reaches the same
// verdict (k_pcr_decide_claims) and loads its share of the kept reads.
static int grp_add_reads_pcr(mcx_group *G, int colour, const uint8_t *bases, const uint8_t *quals, const uint64_t *off,
                             uint64_t nreads, uint8_t fq1, uint8_t fq2, uint8_t hp, int paired, int matedir,
                             mcx_load_stats *stats_accum)
{
  if (paired && (nreads & 1)) return fail(MCX_ERR_ARG, "paired reads come in twos (%llu reads)", (unsigned long long)nreads);
  if (matedir < 0 || matedir > 3) return fail(MCX_ERR_ARG, "mate pair orientation %d: 0 FF, 1 FR, 2 RF, 3 RR", matedir);
  if (nreads >= (1ull << 32)) return fail(MCX_ERR_ARG, "too many reads in one batch");
  if (stats_accum) {
    if (paired) stats_accum->num_pe_reads += nreads; else stats_accum->num_se_reads += nreads;
    stats_accum->total_bases_read += nreads ? off[nreads] - off[0] : 0;
  }
  if (!nreads) return MCX_OK;
  const int N = G->n;
  std::vector<std::unique_ptr<CutJob>> J;
  std::vector<uint8_t *> claims(N, nullptr);     // on shard j: claims of all shards, [N][nreads]
  std::vector<hipEvent_t> claimed(N, nullptr);   // shard i's own claims are ready
  auto cleanup = [&]() {
    for (int i = 0; i < N; i++) {
      (void)hipSetDevice(G->part[i]->device);
      (void)hipStreamSynchronize(G->part[i]->stream);
      (void)hipFree(claims[i]);
      if (claimed[i]) (void)hipEventDestroy(claimed[i]);
    }
  };
#define PCR_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { int rc_ = fail(_e == hipErrorOutOfMemory ? MCX_ERR_NOMEM : MCX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); cleanup(); return rc_; } } while (0)
  const unsigned blocks = (unsigned)((nreads + 127) / 128);
  int rc = MCX_OK;
  for (int i = 0; i < N && rc == MCX_OK; i++) {
    mcx_graph *g = G->part[i];
    J.emplace_back(new CutJob());
    rc = cut_upload(*J[i], g, bases, quals, off, nreads, fq1, fq2, hp, true, paired, matedir);
    if (rc == MCX_OK) rc = cut_starts(*J[i]);
    if (rc != MCX_OK) break;
    PCR_TRY(hipSetDevice(g->device));
    PCR_TRY(hipMalloc((void **)&claims[i], (uint64_t)N * nreads));
    PCR_TRY(hipEventCreateWithFlags(&claimed[i], hipEventDisableTiming));
    hipLaunchKernelGGL(k_pcr_claim, dim3(blocks), dim3(128), 0, g->stream, (const uint64_t *)J[i]->d_node, (const uint32_t *)g->d_readstrt,
                       nreads, J[i]->pmask, claims[i] + (uint64_t)i * nreads);
    PCR_TRY(hipGetLastError());
    PCR_TRY(hipEventRecord(claimed[i], g->stream));
  }
  if (rc != MCX_OK) { cleanup(); return rc; }
  unsigned long long h_ndup = 0;
  const uint64_t nunits = J[0]->nunits;
  for (int j = 0; j < N; j++) {  // gather the claims, decide, commit
    mcx_graph *g = G->part[j];
    PCR_TRY(hipSetDevice(g->device));
    for (int i = 0; i < N; i++)
      if (i != j) {
        PCR_TRY(hipStreamWaitEvent(g->stream, claimed[i], 0));
        PCR_TRY(hipMemcpyPeerAsync(claims[j] + (uint64_t)i * nreads, g->device, claims[i] + (uint64_t)i * nreads, G->part[i]->device, nreads, g->stream));
      }
    const uint64_t u_lo = nunits * (uint64_t)j / (uint64_t)N, u_hi = nunits * (uint64_t)(j + 1) / (uint64_t)N;
    hipLaunchKernelGGL(k_pcr_decide_claims, dim3((unsigned)((nunits + 255) / 256)), dim3(256), 0, g->stream, (const uint8_t *)claims[j],
                       (uint32_t)N, nreads, nunits, J[j]->pmask, u_lo, u_hi, J[j]->d_keep, J[j]->d_ndup);
    hipLaunchKernelGGL(k_pcr_commit, dim3(blocks), dim3(128), 0, g->stream, (const uint64_t *)J[j]->d_node, nreads, g->d_readstrt);
    PCR_TRY(hipGetLastError());
    if (j == 0) PCR_TRY(hipMemcpyAsync(&h_ndup, J[0]->d_ndup, 8, hipMemcpyDeviceToHost, g->stream));
  }
  for (int j = 0; j < N; j++) {  // (a shard's own claims must not be freed while another still copies them)
    PCR_TRY(hipSetDevice(G->part[j]->device));
    PCR_TRY(hipStreamSynchronize(G->part[j]->stream));
  }
  for (int j = 0; j < N && rc == MCX_OK; j++) rc = cut_finish(*J[j], colour);
  cleanup();
#undef PCR_TRY
  if (stats_accum) {
    if (paired) stats_accum->num_dup_pe_pairs += h_ndup; else stats_accum->num_dup_se_reads += h_ndup;
  }
  return rc;
}

static int grp_add_records(mcx_group *G, const void *recs, uint64_t nrecs, int file_ncols, const int32_t *from_col,
                           const int32_t *into_col, int nmap, uint32_t flags, mcx_records_stats *stats_accum)
{
  // (must-exist / masked-edge loads of build --intersect: every record is dealt with by the shard that
  // owns its key, where the intersection colour of that key lives as well -- k_load_records)
  int rc = grp_drain(G);
  if (rc != MCX_OK) return rc;
  mcx_records_stats tot;
  memset(&tot, 0, sizeof(tot));
  tot.first_oversized = tot.first_zero_covg = tot.first_edges_no_covg = -1;
  const uint64_t base = stats_accum ? stats_accum->nkmers_read : 0;
  for (int i = 0; i < G->n; i++) {  // every shard scans all records and keeps the keys it owns (k_load_records)
    mcx_records_stats s;
    memset(&s, 0, sizeof(s));
    s.first_oversized = s.first_zero_covg = s.first_edges_no_covg = -1;
    s.nkmers_read = base;
    rc = mcx_graph_add_records(G->part[i], recs, nrecs, file_ncols, from_col, into_col, nmap, flags, &s);
    if (rc != MCX_OK) return rc;
    tot.nkmers_loaded += s.nkmers_loaded;
    tot.nkmers_novel += s.nkmers_novel;
    auto first = [](int64_t &d, int64_t v) { if (v >= 0 && (d < 0 || v < d)) d = v; };
    first(tot.first_oversized, s.first_oversized);
    first(tot.first_zero_covg, s.first_zero_covg);
    first(tot.first_edges_no_covg, s.first_edges_no_covg);
  }
  if (stats_accum) {
    stats_accum->nkmers_read += nrecs;
    stats_accum->nkmers_loaded += tot.nkmers_loaded;
    stats_accum->nkmers_novel += tot.nkmers_novel;
    if (stats_accum->first_oversized < 0) stats_accum->first_oversized = tot.first_oversized;
    if (stats_accum->first_zero_covg < 0) stats_accum->first_zero_covg = tot.first_zero_covg;
    if (stats_accum->first_edges_no_covg < 0) stats_accum->first_edges_no_covg = tot.first_edges_no_covg;
  }
  return MCX_OK;
}

// The shards hold disjoint key sets.  Unsorted: one shard after the other.  Sorted: every shard
// exports its own records sorted (its own device sort, which walks the key space in ranges when its
// scratch would not fit beside the table) into host memory, and the host merges the N sorted runs
// straight into the sink's 64 MiB pieces -- nothing is gathered on one device, so a graph that needs
// several GPUs to build does not need one GPU to hold it for `--sort`.  Small graphs (MCX_MULTI_SORT_DEV
// records at most, default 32 M) keep the faster way: one device sort of the gathered records.
static int grp_export(mcx_group *G, mcx_graph *f, int sorted, mcx_sink_fn sink, void *ctx)
{
  int rc = grp_sync(G);
  if (rc != MCX_OK) return rc;
  if (!sorted) {
    for (int i = 0; i < G->n; i++) {
      rc = mcx_graph_export(G->part[i], 0, sink, ctx);
      if (rc != MCX_OK) return rc;
    }
    return MCX_OK;
  }
  uint64_t total = 0;
  for (int i = 0; i < G->n; i++) {
    uint64_t n = 0;
    rc = mcx_graph_nkmers(G->part[i], &n);
    if (rc != MCX_OK) return rc;
    total += n;
  }
  const int W = f->W;
  const uint64_t recsz = 8ull * W + 5ull * (uint64_t)f->ncols_vis;  // (an intersect build does not export its hidden colour)
  const char *dsm = getenv("MCX_MULTI_SORT_DEV");
  const uint64_t dev_sort_max = dsm ? strtoull(dsm, nullptr, 10) : (32ull << 20);
  const bool on_device = total <= dev_sort_max;
  std::vector<std::vector<uint8_t>> run(on_device ? 1 : G->n);
  struct Collect { std::vector<uint8_t> *v; };
  auto collect = [](void *p, const void *r, size_t nb) -> int {
    Collect *cc = (Collect *)p;
    const uint8_t *b = (const uint8_t *)r;
    try { cc->v->insert(cc->v->end(), b, b + nb); } catch (...) { return 1; }
    return 0;
  };
  for (int i = 0; i < G->n; i++) {
    Collect c{&run[on_device ? 0 : i]};
    rc = mcx_graph_export(G->part[i], on_device ? 0 : 1, collect, &c);
    if (rc != MCX_OK) return rc == MCX_ERR_SINK ? fail(MCX_ERR_NOMEM, "out of host memory for %llu records", (unsigned long long)total) : rc;
  }
  const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / recsz) * recsz;
  if (on_device) {
    std::vector<uint8_t> &all = run[0];
    rc = mcx_sort_records(all.data(), all.size() / recsz, f->k, f->ncols_vis, G->part[0]->device);
    if (rc != MCX_OK) return rc;
    for (uint64_t o = 0; o < all.size(); o += chunk)
      if (sink(ctx, all.data() + o, (size_t)std::min<uint64_t>(chunk, all.size() - o)) != 0) return fail(MCX_ERR_SINK, "export sink failed");
    return MCX_OK;
  }
  // N-way merge: records compare by their key words, word 0 (the top word) first (hash_table.c:371)
  auto key_word = [&](const uint8_t *r, int w) { uint64_t x; memcpy(&x, r + 8 * w, 8); return x; };
  auto less = [&](const uint8_t *a, const uint8_t *b) {
    for (int w = 0; w < W; w++) { const uint64_t x = key_word(a, w), y = key_word(b, w); if (x != y) return x < y; }
    return false;
  };
  struct Head { const uint8_t *p, *end; };
  std::vector<Head> heap;
  for (auto &v : run) if (!v.empty()) heap.push_back(Head{v.data(), v.data() + v.size()});
  auto cmp = [&](const Head &a, const Head &b) { return less(b.p, a.p); };  // min-heap on the head records
  std::make_heap(heap.begin(), heap.end(), cmp);
  std::vector<uint8_t> out;
  out.reserve(chunk);
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), cmp);
    Head &h = heap.back();
    // everything of this run below the next run's head goes out in one piece
    const uint8_t *limit = heap.size() > 1 ? heap.front().p : nullptr;
    do {
      out.insert(out.end(), h.p, h.p + recsz);
      h.p += recsz;
      if (out.size() >= chunk) {
        if (sink(ctx, out.data(), out.size()) != 0) return fail(MCX_ERR_SINK, "export sink failed");
        out.clear();
      }
    } while (h.p < h.end && (!limit || less(h.p, limit)));
    if (h.p < h.end) std::push_heap(heap.begin(), heap.end(), cmp); else heap.pop_back();
  }
  if (!out.empty() && sink(ctx, out.data(), out.size()) != 0) return fail(MCX_ERR_SINK, "export sink failed");
  return MCX_OK;
}
