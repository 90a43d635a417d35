// mcx_superk.h -- exchange format v3 of the sharded build: reads travel, not occurrences.
//
// The owner of a k-mer is a function of its canonical MINIMIZER (the smallest hashed canonical
// 13-mer inside it), so consecutive k-mers of a read mostly share their owner and go to it as one
// piece of 2-bit sequence ("super-k-mer") instead of one 8-byte tuple each: a 16-byte record
// carries the lane's 48-base window plus (start, length, two flags) of one run of up to 16
// consecutive k-mers with the same owner -- ~2.3 records per 16 positions = ~2.3 B per occurrence
// on the links instead of 8.5.  The owner k-merises what it receives and feeds the same region
// bins as the single-GPU path.  Two-word keys (k = 33..63) use a 32-byte record with an 80-base
// window: their minimizer windows are long (k - 12 m-mers), so most runs are the lane's whole 16
// positions and a record carries ~3 B per occurrence against the 17 of format v2.  Restricted to
// k >= 29 (M = 13, so that all 16 minimizer windows of a lane share a common middle part); smaller
// k use format v2.
//
// Reference semantics are unaffected: which GPU holds a k-mer is free, the tuples
// (canonical key, colour, edge byte) an owner derives are exactly those of mcx_defer.h.
#pragma once
#include "mcx_defer.h"

namespace mcx {

// (kMmer, mmer_hash, owner_of_minimizer, superk_owner: mcx_kernels.h, "Who owns a key")

#if defined(__HIPCC__)

// record, one-word keys (16 bytes): a.x = bases -1 .. 30 of the lane's window (2 bits each, first on
// top); a.y = bases 31 .. 46 in the high half, header in the low half.
// two-word keys (32 bytes): a.x = bases -1 .. 30, a.y = bases 31 .. 62, b.x = bases 63 .. 78 in the
// high half, header in the low half; b.y unused.
constexpr uint32_t kSkStartMask = 0xFu, kSkLenShift = 4, kSkPrevOk = 1u << 8, kSkNextOk = 1u << 9;
template <int W> struct SkRec;
template <> struct SkRec<1> { ulonglong2 a; };
template <> struct SkRec<2> { ulonglong2 a, b; };
template <int W> __device__ __forceinline__ uint32_t sk_hdr(const SkRec<W> &r);
template <> __device__ __forceinline__ uint32_t sk_hdr<1>(const SkRec<1> &r) { return (uint32_t)r.a.y; }
template <> __device__ __forceinline__ uint32_t sk_hdr<2>(const SkRec<2> &r) { return (uint32_t)r.b.x; }
template <int W> __device__ __forceinline__ SkRec<W> sk_make(uint64_t w0, uint64_t w1, uint64_t w2, uint32_t hdr);
template <> __device__ __forceinline__ SkRec<1> sk_make<1>(uint64_t w0, uint64_t w1, uint64_t, uint32_t hdr)
{
  SkRec<1> r;
  r.a = make_ulonglong2(w0, (w1 & 0xFFFFFFFF00000000ULL) | hdr);
  return r;
}
template <> __device__ __forceinline__ SkRec<2> sk_make<2>(uint64_t w0, uint64_t w1, uint64_t w2, uint32_t hdr)
{
  SkRec<2> r;
  r.a = make_ulonglong2(w0, w1);
  r.b = make_ulonglong2((w2 & 0xFFFFFFFF00000000ULL) | hdr, 0);
  return r;
}

struct SuperkOut {
  void *recs;                   // [nparts][rep][cap] records of SkRec<W>
  // fills, REPLICA-major [rep][nparts] (zeroed by the caller; > cap: records were dropped): the
  // blocks of one replica reserve from one 64-byte line, other replicas from other lines
  unsigned long long *counts;
  uint64_t cap;
  uint32_t lbo, rep;
  // Records that do not fit their segment (one owner far above its share: poly-G reads, a satellite
  // repeat) go to the sender's spill area, any owner, with the owner beside them; the host routes
  // them (mcx_multi.h).  sp_cap == 0: no spill area, such records are dropped and reported (bin_over).
  void *sp_recs;                 // [sp_cap] records
  uint8_t *sp_own;               // [sp_cap] owner of each
  unsigned long long *sp_count;  // fill
  uint64_t sp_cap;
};

template <int W> __device__ __forceinline__ bool sk_spill(const SuperkOut &out, const SkRec<W> &rec, uint32_t owner)
{
  if (!out.sp_cap) return false;
  const unsigned long long p = atomicAdd(out.sp_count, 1ULL);
  if (p >= out.sp_cap) return false;
  reinterpret_cast<SkRec<W> *>(out.sp_recs)[p] = rec;
  out.sp_own[p] = (uint8_t)owner;
  return true;
}

template <int W> struct SkCfg { static constexpr int kStage = 2048 / W; };  // records staged per tile in LDS (a tile of random reads makes ~600, ~350 at k = 63)

// ---------------------------------------------------------------------------
// sender: reads -> per-owner bins of super-k-mer records
// ---------------------------------------------------------------------------
template <int W, bool PK /* the stream comes packed (code words + invalid flags), as the host entry stages it */>
__global__ __launch_bounds__(kThreads, 4) void k_stream_superk(StreamArgs a, SuperkOut out)
{
  constexpr int kSkStage = SkCfg<W>::kStage;
  __shared__ uint32_t s_code[kChunks + 4];
  __shared__ uint32_t s_inv[kChunks / 2 + 4];
  __shared__ SkRec<W> s_rec[kSkStage];
  __shared__ uint8_t s_own[kSkStage];
  __shared__ uint32_t s_cnt[32], s_rank[32], s_total;
  __shared__ unsigned long long s_base[32];
  SkRec<W> *recs_out = reinterpret_cast<SkRec<W> *>(out.recs);

  const int tid = threadIdx.x;
  const int k = a.k;
  uint32_t n_kmers = 0, n_contigs = 0, dropped = 0;
  const uint32_t nparts = 1u << out.lbo;
  const uint32_t rep = blockIdx.x % out.rep;

  // the chunks of the NEXT tile are fetched into registers while the current one is processed
  TileSrc pre;
  pre.a = make_uint4(0, 0, 0, 0); pre.b = make_uint4(0, 0, 0, 0);
  {
    const uint64_t t0 = a.tile0 + blockIdx.x;
    if (t0 < a.ntiles) tile_fetch<PK>(a, t0, tid, pre);
  }
  for (uint64_t tile = a.tile0 + blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();
    tile_stage<PK>(a, pre, tid, s_code, s_inv);
    if (tid < 4) { s_code[kChunks + tid] = 0; s_inv[kChunks / 2 + tid] = 0xFFFFFFFFu; }
    if (tid < 32) { s_cnt[tid] = 0; s_rank[tid] = 0; }
    if (tid == 0) s_total = 0;
    {
      const uint64_t tn = tile + gridDim.x;
      if (tn < a.ntiles) tile_fetch<PK>(a, tn, tid, pre);
    }
    __syncthreads();

    const uint32_t pl = 16u * (uint32_t)(tid + 1);
    const uint64_t Vh = inv_win64(s_inv, pl);
    const uint64_t Vl = (W == 2) ? inv_win64(s_inv, pl + 64) : 0;
    const uint32_t prev_chunk_inv = (s_inv[(pl - 1) >> 5] >> (31 - ((pl - 1) & 31))) & 1u;
    const uint64_t P0 = tile * kTile + 16ull * (uint64_t)tid;
    const int j_lo = a.pos_lo > P0 ? (int)min((uint64_t)kPosPerLane, a.pos_lo - P0) : 0;
    const int j_hi = a.pos_hi > P0 ? (int)min((uint64_t)kPosPerLane, a.pos_hi - P0) : 0;
    uint32_t ok16, nok16, pok16;  // as in k_stream_bin
    {
      uint64_t Mh = Vh, Ml = Vl;
      for (int c = 1; c < k;) {
        const int s = min(c, k - c);
        if (W == 2) Mh |= (Mh << s) | (Ml >> (64 - s));
        else Mh |= Mh << s;
        if (W == 2) Ml |= Ml << s;
        c += s;
      }
      const uint32_t range = ((0x10000u >> j_lo) - 1u) & ~((0x10000u >> j_hi) - 1u);
      ok16 = ~(uint32_t)(Mh >> 48) & range;
      const int sh = 112 - k;  // base j + k: field of 16 flags starting at bit 112 - k of Vh : Vl
      const uint64_t nx = sh >= 64 ? (Vh >> (sh - 64)) : ((Vh << (64 - sh)) | (W == 2 ? (Vl >> sh) : 0));
      nok16 = ~(uint32_t)nx & 0xFFFFu;
      pok16 = ~((prev_chunk_inv << 15) | (uint32_t)(Vh >> 49)) & 0xFFFFu;
    }
    if (ok16) {
      n_kmers += __popc(ok16);
      n_contigs += __popc(ok16 & ~pok16);
      // window: base i (relative to the lane's first position) is base t = i + 1 of w0 : w1 : w2
      const uint64_t w0 = code_win64(s_code, pl - 1), w1 = code_win64(s_code, pl + 31);
      const uint64_t w2 = (W == 2) ? code_win64(s_code, pl + 63) : 0;
      auto nuc_at = [&](int t) -> uint32_t {
        if (W == 2 && t >= 64) return (uint32_t)(w2 >> (190 - 2 * t)) & 3u;
        return t < 32 ? (uint32_t)(w0 >> (62 - 2 * t)) & 3u : (uint32_t)(w1 >> (126 - 2 * t)) & 3u;
      };
      // canonical minimizer hash of every k-mer start j: min over m-mers p = j .. j + k - M.
      // All 16 windows contain p = 15 .. k - M; left of it a suffix minimum, right a prefix one.
      uint32_t f = 0, r = 0;
#pragma unroll
      for (int i = 0; i < kMmer - 1; i++) {
        const uint32_t n = nuc_at(i + 1);
        f = (f << 2) | n;
        r = (r >> 2) | ((3u - n) << (2 * kMmer - 2));
      }
      auto push = [&](uint32_t n) -> uint32_t {
        f = ((f << 2) | n) & kMmerMask;
        r = (r >> 2) | ((3u - n) << (2 * kMmer - 2));
        return mmer_hash(f < r ? f : r);
      };
      uint32_t h[15];
#pragma unroll
      for (int p = 0; p < 15; p++) h[p] = push(nuc_at(p + kMmer));  // m-mer p ends at base p + M - 1
      uint32_t common = 0xFFFFFFFFu;
      for (int p = 15; p <= k - kMmer; p++) common = min(common, push(nuc_at(p + kMmer)));
#pragma unroll
      for (int p = 13; p >= 0; p--) h[p] = min(h[p], h[p + 1]);       // suffix minima over p .. 14
      uint32_t own[16];
      uint32_t pre_min = 0xFFFFFFFFu;
      own[0] = owner_of_minimizer(min(common, h[0]), out.lbo);
#pragma unroll
      for (int j = 1; j < 16; j++) {
        pre_min = min(pre_min, push(nuc_at(k + j)));                  // m-mer k - M + j ends at base k + j - 1
        const uint32_t left = j < 15 ? h[j] : 0xFFFFFFFFu;
        own[j] = owner_of_minimizer(min(min(common, left), pre_min), out.lbo);
      }
      // runs of consecutive valid k-mers with one owner -> records staged in LDS
      int start = -1;
      uint32_t cur = 0;
#pragma unroll
      for (int j = 0; j <= 16; j++) {
        const bool v = j < 16 && (ok16 >> (15 - j) & 1u);
        const uint32_t o = j < 16 ? own[j < 16 ? j : 15] : 0;
        if (start >= 0 && (!v || o != cur)) {  // close the run [start, j)
          const uint32_t last = (uint32_t)j - 1u;
          uint32_t hdr = (uint32_t)start | ((last - (uint32_t)start) << kSkLenShift);
          if (pok16 >> (15 - start) & 1u) hdr |= kSkPrevOk;
          if (nok16 >> (15 - last) & 1u) hdr |= kSkNextOk;
          const uint32_t slot = atomicAdd(&s_total, 1u);
          const SkRec<W> rec = sk_make<W>(w0, w1, w2, hdr);
          if (slot < (uint32_t)kSkStage) { s_rec[slot] = rec; s_own[slot] = (uint8_t)cur; }
          else {  // staging full (pathological input): straight to the bin, one global atomic
            const unsigned long long pos = atomicAdd(&out.counts[rep * nparts + cur], 1ULL);
            if (pos < out.cap) recs_out[((uint64_t)cur * out.rep + rep) * out.cap + pos] = rec;
            else if (!sk_spill<W>(out, rec, cur)) dropped = 1;
          }
          start = -1;
        }
        if (v && start < 0) { start = j; cur = o; }
      }
    }
    __syncthreads();
    const uint32_t nst = min(s_total, (uint32_t)kSkStage);
    for (uint32_t q = tid; q < nst; q += kThreads) atomicAdd(&s_cnt[s_own[q]], 1u);
    __syncthreads();
    if ((uint32_t)tid < nparts && s_cnt[tid])  // one wave instruction reserves for every owner
      s_base[tid] = atomicAdd(&out.counts[rep * nparts + (uint32_t)tid], (unsigned long long)s_cnt[tid]);
    __syncthreads();
    for (uint32_t q = tid; q < nst; q += kThreads) {
      const uint32_t o = s_own[q];
      const unsigned long long pos = s_base[o] + atomicAdd(&s_rank[o], 1u);
      if (pos < out.cap) recs_out[((uint64_t)o * out.rep + rep) * out.cap + pos] = s_rec[q];
      else if (!sk_spill<W>(out, s_rec[q], o)) dropped = 1;
    }
  }
  block_add(&a.ctr->kmers, n_kmers);
  block_add(&a.ctr->contigs, n_contigs);
  if (dropped) a.ctr->bin_over = 1;
  if (a.flag && n_contigs) *a.flag = 1;  // (a piece of a read longer than a staging chunk holds a contig: the read is a good one)
}

// owner side of a spill: the records of `owner` out of a spill area (recs, own)[n] -> dst[0 .. *count)
template <int W>
__global__ void k_superk_pick(const void *recs, const uint8_t *own, uint64_t n, uint32_t owner, void *dst, unsigned long long *count)
{
  const SkRec<W> *src = reinterpret_cast<const SkRec<W> *>(recs);
  SkRec<W> *out = reinterpret_cast<SkRec<W> *>(dst);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (own[i] == owner) out[atomicAdd(count, 1ULL)] = src[i];
}
// fills [rep][nparts] (what the sender writes) -> [nparts][rep] (one contiguous row per owner, what travels)
__global__ void k_transpose_fills(const unsigned long long *in, unsigned long long *out, uint32_t rep, uint32_t nparts)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rep * nparts) out[(i % nparts) * rep + i / nparts] = in[i];
}

// The in-process exchange (mcx_multi.h): ONE launch on the sender's copy stream moves the FILLED part of every
// (owner, segment) of a send set into the owners' receive slots through peer-mapped pointers, and the fills with
// them -- the fills are read on the device, so the host still never sees a count.  (Until round 4 every owner's whole
// block went out with hipMemcpyPeerAsync: the segments' capacity, 3.6 B per occurrence at k = 31 against 2.4 B of
// filled records, and 2 N + N copy calls per piece on the one host thread.)  Segments are 16-byte aligned arrays of
// `unit16` 16-byte units per item (records: 1 or 2; packed one-word tuples: half a unit -- copied in pairs).
struct PeerDst {
  void *data[32];                  // owner j's receive slot for this sender: [segs][cap] items
  unsigned long long *fills[32];   // ... and its fills [segs]
};
constexpr uint32_t kCopyChunk = 8192;  // 16-byte units per unit of work (128 KiB)
__global__ __launch_bounds__(256) void k_copy_filled(const ulonglong2 *src, const unsigned long long *fills, PeerDst dst, uint32_t nparts,
                                                     uint32_t segs, uint64_t cap, uint32_t item_bytes)
{
  const uint64_t seg_units = (cap * item_bytes + 15) / 16;
  const uint64_t chunks_per_seg = (seg_units + kCopyChunk - 1) / kCopyChunk;
  const uint64_t nsegs = (uint64_t)nparts * segs, nunits = nsegs * chunks_per_seg;
  for (uint64_t v = blockIdx.x; v < nunits; v += gridDim.x) {
    const uint64_t js = v % nsegs, c = v / nsegs;  // segment-interleaved
    const uint32_t j = (uint32_t)(js / segs), sg = (uint32_t)(js % segs);
    const unsigned long long f = fills[js];
    if (c == 0 && threadIdx.x == 0) dst.fills[j][sg] = f;
    const uint64_t n16 = (min((uint64_t)f, cap) * item_bytes + 15) / 16;
    const uint64_t lo = c * kCopyChunk;
    if (lo >= n16) continue;
    const uint64_t hi = min(n16, lo + kCopyChunk);
    const ulonglong2 *sp = src + js * seg_units;
    ulonglong2 *dp = reinterpret_cast<ulonglong2 *>(dst.data[j]) + (uint64_t)sg * seg_units;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) dp[i] = sp[i];
  }
}

// ---------------------------------------------------------------------------
// owner: super-k-mer records -> region bins of packed tuples (same output as k_stream_bin)
// ---------------------------------------------------------------------------
struct SuperkIn {
  const void *recs;                  // [nseg][seg_cap] records of SkRec<W>
  const unsigned long long *counts;  // [nseg]; a fill above seg_cap is read as seg_cap
  uint64_t seg_cap;
  uint32_t nseg;
};

// Records hold 1..16 k-mers (16 when one GPU owns everything, 6-7 on average at 8 owners), and
// a lane that walks its record's 16 positions pays for all 16 whatever the run's length.  So the
// K-MERS -- not the records -- are dealt out to the lanes.  A block walks a chunk of a segment and
// takes, per tile, as many records (in fours: one 64-byte load per lane) as hold <= kTile k-mers;
// k-mer q of the tile (records in order, their k-mers in order) goes to lane q % kThreads as its
// (q / kThreads)-th, and every lane extracts its k-mers straight from the record's
// window.  All lanes carry the same number of k-mers (+-1) and tiles are full (>= kTile - 63
// k-mers) except at a chunk's end, so the cost per k-mer does not depend on the run lengths.
constexpr int kSkPerLane = 4;                      // candidate records per lane and tile
constexpr int kSkCand = kSkPerLane * kThreads;     // 1024
constexpr uint32_t kSkChunk = 1u << 13;            // records per unit of work (a block's walk)
constexpr size_t kSkMapBytes = (size_t)kTile * 2;  // s_map behind the BinLds block in dynamic LDS

template <int W, bool ONECOL, int NB>
__global__ __launch_bounds__(kThreads, (W == 1 ? 4 : 3)) void k_superk_bin(SuperkIn in, int k, BinSpec bs, BinOut out,
                                                                         InsertSink<W, ONECOL> isink, Counters *ctr)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  using LDS = BinLds<W, NB, false>;
  LDS &L = *reinterpret_cast<LDS *>(dyn_lds);
  // the taken records sit in the staging area (not written before bin_place); the k-mer -> (record,
  // index in run) map has its own 8 KB
  static_assert(sizeof(L.skey) >= (size_t)kSkCand * sizeof(SkRec<W>), "records fit the staging area");
  SkRec<W> *s_rec = reinterpret_cast<SkRec<W> *>(L.skey);
  uint16_t *s_map = reinterpret_cast<uint16_t *>(dyn_lds + ((sizeof(LDS) + 15) & ~(size_t)15));  // record | index << 10
  __shared__ uint32_t s_incl[kThreads + 1];
  __shared__ uint32_t s_wsum[kThreads / 64];
  __shared__ uint32_t s_T, s_nl;
  const int tid = threadIdx.x;
  uint32_t n_novel = 0, full = 0;
  unsigned long long n_binned = 0;  // (uniform: every tile's T)
  const uint32_t ob0 = (blockIdx.x % bs.rep) * bs.nout;
  const uint64_t chunks_per_seg = (in.seg_cap + kSkChunk - 1) / kSkChunk;
  const uint64_t nunits = chunks_per_seg * in.nseg;
  for (uint64_t v = blockIdx.x; v < nunits; v += gridDim.x) {
    const uint32_t seg = (uint32_t)(v % in.nseg);  // segment-interleaved
    const uint64_t c0 = (v / in.nseg) * kSkChunk;
    uint64_t cnt = in.counts[seg];
    if (cnt > in.seg_cap) cnt = in.seg_cap;
    if (c0 >= cnt) continue;  // uniform
    const uint64_t c1 = min(cnt, c0 + (uint64_t)kSkChunk);
    const SkRec<W> *recs = reinterpret_cast<const SkRec<W> *>(in.recs) + (uint64_t)seg * in.seg_cap;
    for (uint64_t pos = c0; pos < c1;) {  // uniform
      __syncthreads();
      for (uint32_t b = tid; b < bs.nlocal; b += kThreads) L.cnt[b] = 0;
      // four candidate records per lane, their k-mer counts, inclusive scan over the block
      SkRec<W> rr[kSkPerLane];
      uint32_t ll[kSkPerLane];
      uint32_t lsum = 0;
      {
        const uint64_t i = pos + (uint64_t)kSkPerLane * (uint64_t)tid;
#pragma unroll
        for (int q = 0; q < kSkPerLane; q++) {
          ll[q] = 0;
          if (i + q < c1) { rr[q] = recs[i + q]; ll[q] = ((sk_hdr<W>(rr[q]) >> kSkLenShift) & 0xFu) + 1u; }
          lsum += ll[q];
        }
      }
      uint32_t x = lsum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if ((tid & 63) >= d) x += y;
      }
      if ((tid & 63) == 63) s_wsum[tid >> 6] = x;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < kThreads / 64; w++)
        if (w < (tid >> 6)) x += s_wsum[w];
      s_incl[tid] = x;  // inclusive
      if (tid == 0) s_incl[kThreads] = 0xFFFFFFFFu;
      __syncthreads();
      // lanes are taken while the running total fits a tile: a prefix of the lanes (lane 0 always)
      const bool take = x <= (uint32_t)kTile;
      if (take && s_incl[tid + 1] > (uint32_t)kTile) { s_T = x; s_nl = (uint32_t)tid + 1u; }
      if (take) {
        uint32_t o = x - lsum;
        const uint32_t r0 = (uint32_t)kSkPerLane * (uint32_t)tid;
#pragma unroll
        for (int q = 0; q < kSkPerLane; q++) {
          if (ll[q]) s_rec[r0 + q] = rr[q];
          for (uint32_t i = 0; i < ll[q]; i++) s_map[o++] = (uint16_t)((r0 + (uint32_t)q) | (i << 10));
        }
      }
      __syncthreads();
      const uint32_t T = s_T;
      pos += (uint64_t)kSkPerLane * s_nl;
      n_binned += T;

      Kmer<W> tk[kPosPerLane];
      uint32_t tle[kPosPerLane];
      uint32_t vmask = 0;
      const int nj = (int)((T + kThreads - 1) / kThreads);  // uniform
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++) {
        const uint32_t q = (uint32_t)j * kThreads + (uint32_t)tid;
        if (j < nj && q < T) {  // (j < nj is uniform: whole iterations are skipped)
          const uint32_t m = s_map[q];
          const SkRec<W> rec = s_rec[m & 0x3ffu];
          const uint32_t hdr = sk_hdr<W>(rec);
          const uint32_t i = m >> 10, rlen = ((hdr >> kSkLenShift) & 0xFu) + 1u;
          const uint32_t p = (hdr & kSkStartMask) + i;       // the k-mer is bases p+1 .. p+k of the window
          const uint32_t sft = 2u * (p + 1u);                // 2 .. 32
          Kmer<W> fw, rc;
          uint32_t nuc_next;
          const uint64_t w0 = rec.a.x;
          if constexpr (W == 1) {
            const uint64_t w1 = rec.a.y & 0xFFFFFFFF00000000ULL;
            const uint64_t X = (w0 << sft) | (w1 >> (64u - sft));  // bases p+1 .. p+32 on top
            fw.w[0] = X >> (64 - 2 * k);
            // base after the k-mer = the one below the top k bases of X (k <= 31)
            nuc_next = (uint32_t)(X >> (62 - 2 * k)) & 3u;
          } else {
            const uint64_t w1 = rec.a.y, w2 = rec.b.x & 0xFFFFFFFF00000000ULL;
            const uint64_t A = (w0 << sft) | (w1 >> (64u - sft));  // bases p+1 .. p+32
            const uint64_t B = (w1 << sft) | (w2 >> (64u - sft));  // bases p+33 .. p+64
            const int s = 128 - 2 * k;                               // 2 .. 62: the k-mer is the top 2k bits of A : B
            fw.w[0] = A >> s;
            fw.w[W - 1] = (B >> s) | (A << (64 - s));
            nuc_next = (uint32_t)(B >> (s - 2)) & 3u;                // the base below them
          }
          rc = revcomp<W>(fw, k);
          const uint32_t prev_nuc = (uint32_t)(w0 >> (62u - 2u * p)) & 3u;  // base before it = window base p
          const bool next_ok = (i + 1u < rlen) || (hdr & kSkNextOk);
          const bool prev_ok = (i > 0u) || (hdr & kSkPrevOk);
          uint32_t o;
          const Kmer<W> key = canonical<W>(fw, rc, o);
          uint32_t e = 0;
          if (next_ok) e |= 1u << (nuc_next + 4u * o);
          if (prev_ok) e |= 1u << ((3u - prev_nuc) + 4u * (1u - o));
          uint32_t r;
          const uint32_t lbq = lbq_of(isink.t);
          const Kmer<W> qq = key_quot<W>(key, lbq, r);
          const uint32_t G = r ^ mix_g(region_mix<W>(qq), lbq);
          const uint32_t local = G & ((1u << isink.t.lb1) - 1u);
          tk[j] = tuple_pack<W>(qq, e);
          vmask |= 1u << j;
          // (the counting atomic returns the tuple's arrival index in its bin: with the bin's offset that is its
          // sorted position -- no second atomic per tuple, as in k_stream_bin)
          tle[j] = (local << 8) | (atomicAdd(&L.cnt[local], 1u) << 19);
        }
      }
      BinRes<NB> res;
      bin_reserve<LDS, NB>(L, bs, out, ob0, res);   // (its first barrier also ends the reads of s_rec / s_map)
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++)
        if (j < nj && (vmask & (1u << j))) tle[j] += L.off[(tle[j] >> 8) & 0x7ffu] << 19;
      bin_commit<LDS, NB>(L, bs, out, ob0, res);
      for (int round = 0; round < kRounds; round++) {
        if ((uint32_t)round * kStage < T) {  // uniform
#pragma unroll
          for (int j = 0; j < kPosPerLane; j++)
            if (j < nj && (vmask & (1u << j)))
              bin_place<W, false, LDS>(L, round, tle[j] >> 19, (tle[j] >> 8) & 0x7ffu, tk[j], 0);
        }
        bin_writeout<W, ONECOL, false, 0, LDS>(L, round, bs, out, ob0, 0, isink, n_novel, full);
      }
    }
  }
  block_add(&ctr->novel, n_novel);
  if (tid == 0 && n_binned) atomicAdd(&ctr->binned, n_binned);
  if (full == 1) ctr->full = 1;
  if (full == 2) ctr->bin_over = 1;
}

#endif  // __HIPCC__

}  // namespace mcx
