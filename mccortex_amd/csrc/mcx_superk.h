// mcx_superk.h -- exchange format v3 of the sharded build: reads travel, not occurrences.
//
// The owner of a k-mer is a function of its canonical MINIMIZER (the smallest hashed canonical
// 13-mer inside it), so consecutive k-mers of a read mostly share their owner and go to it as one
// piece of 2-bit sequence ("super-k-mer") instead of one 8-byte tuple each: a 16-byte record
// carries the lane's 48-base window plus (start, length, two flags) of one run of up to 16
// consecutive k-mers with the same owner -- ~2.3 records per 16 positions = ~2.3 B per occurrence
// on the links instead of 8.5.  The owner k-merises what it receives and feeds the same region
// bins as the single-GPU path.  Restricted to one-word keys with k >= 29 (M = 13, so that all 16
// minimizer windows of a lane share a common middle part); other k use format v2.
//
// Reference semantics are unaffected: which GPU holds a k-mer is free, the tuples
// (canonical key, colour, edge byte) an owner derives are exactly those of mcx_defer.h.
#pragma once
#include "mcx_defer.h"

namespace mcx {

constexpr int kMmer = 13;                                  // minimizer length
constexpr uint32_t kMmerMask = (1u << (2 * kMmer)) - 1u;
constexpr int kSuperkMinK = kMmer + 16;                    // 29

// hash of a canonical m-mer (its 2-bit value): multiply-xorshift, 32 bits
MCX_HD uint32_t mmer_hash(uint32_t c)
{
  const uint32_t x = c * 0x9E3779B1u;
  return x ^ (x >> 15);
}
// owner from the minimum hash: the minimum of many uniform values is biased towards 0, so it is
// mixed again before its top bits are taken
MCX_HD uint32_t owner_of_minimizer(uint32_t min_hash, uint32_t lbo)
{
  return lbo ? ((min_hash * 0x85EBCA6Bu) ^ (min_hash >> 13)) * 0xC2B2AE35u >> (32u - lbo) : 0u;
}

// Host/device reference: owner of a k-mer given as its 2-bit value (one word, k <= 31).  The
// kernels compute the same thing incrementally; tests compare shard contents against this.
MCX_HD uint32_t superk_owner(uint64_t kmer, int k, uint32_t lbo)
{
  uint32_t best = 0xFFFFFFFFu;
  for (int p = 0; p + kMmer <= k; p++) {
    const uint32_t f = (uint32_t)(kmer >> (2 * (k - kMmer - p))) & kMmerMask;
    uint32_t r = 0;
    for (int i = 0; i < kMmer; i++) r |= (3u - ((f >> (2 * i)) & 3u)) << (2 * (kMmer - 1 - i));
    const uint32_t h = mmer_hash(f < r ? f : r);
    best = h < best ? h : best;
  }
  return owner_of_minimizer(best, lbo);
}

#if defined(__HIPCC__)

// record: word 0 = bases -1 .. 30 of the lane's window (2 bits each, first on top); word 1 =
// bases 31 .. 46 in the high half, header in the low half
constexpr uint32_t kSkStartMask = 0xFu, kSkLenShift = 4, kSkPrevOk = 1u << 8, kSkNextOk = 1u << 9;

struct SuperkOut {
  ulonglong2 *recs;             // [nparts][rep][cap]
  // fills, REPLICA-major [rep][nparts] (zeroed by the caller; > cap: records were dropped): the
  // blocks of one replica reserve from one 64-byte line, other replicas from other lines
  unsigned long long *counts;
  uint64_t cap;
  uint32_t lbo, rep;
};

constexpr int kSkStage = 2048;  // records staged per tile in LDS (a tile of random reads makes ~600)

// ---------------------------------------------------------------------------
// sender: reads -> per-owner bins of super-k-mer records
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 4) void k_stream_superk(StreamArgs a, SuperkOut out)
{
  __shared__ uint32_t s_code[kChunks + 4];
  __shared__ uint32_t s_inv[kChunks / 2 + 4];
  __shared__ ulonglong2 s_rec[kSkStage];
  __shared__ uint8_t s_own[kSkStage];
  __shared__ uint32_t s_cnt[32], s_rank[32], s_total;
  __shared__ unsigned long long s_base[32];

  const int tid = threadIdx.x;
  const int k = a.k;
  uint32_t n_kmers = 0, n_contigs = 0, dropped = 0;
  const uint32_t nparts = 1u << out.lbo;
  const uint32_t rep = blockIdx.x % out.rep;

  // the chunks of the NEXT tile are fetched into registers while the current one is processed
  uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = make_uint4(0, 0, 0, 0);
  {
    const uint64_t t0 = a.tile0 + blockIdx.x;
    if (t0 < a.ntiles) {
      const int64_t r0 = (int64_t)(t0 * kTile) - 16;
      pre0 = load_chunk(a.stream, a.nbytes, r0 + 16 * (int64_t)tid);
      if (tid < kChunks - kThreads) pre1 = load_chunk(a.stream, a.nbytes, r0 + 16 * (int64_t)(tid + kThreads));
    }
  }
  for (uint64_t tile = a.tile0 + blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();
    {
      uint32_t code, inv;
      encode_words(pre0, code, inv);
      s_code[tid] = code;
      reinterpret_cast<uint16_t *>(s_inv)[tid ^ 1] = (uint16_t)inv;
      if (tid < kChunks - kThreads) {
        encode_words(pre1, code, inv);
        s_code[tid + kThreads] = code;
        reinterpret_cast<uint16_t *>(s_inv)[(tid + kThreads) ^ 1] = (uint16_t)inv;
      }
    }
    if (tid < 4) { s_code[kChunks + tid] = 0; s_inv[kChunks / 2 + tid] = 0xFFFFFFFFu; }
    if (tid < 32) { s_cnt[tid] = 0; s_rank[tid] = 0; }
    if (tid == 0) s_total = 0;
    {
      const uint64_t tn = tile + gridDim.x;
      if (tn < a.ntiles) {
        const int64_t r0 = (int64_t)(tn * kTile) - 16;
        pre0 = load_chunk(a.stream, a.nbytes, r0 + 16 * (int64_t)tid);
        if (tid < kChunks - kThreads) pre1 = load_chunk(a.stream, a.nbytes, r0 + 16 * (int64_t)(tid + kThreads));
      }
    }
    __syncthreads();

    const uint32_t pl = 16u * (uint32_t)(tid + 1);
    const uint64_t Vh = inv_win64(s_inv, pl);
    const uint32_t prev_chunk_inv = (s_inv[(pl - 1) >> 5] >> (31 - ((pl - 1) & 31))) & 1u;
    const uint64_t P0 = tile * kTile + 16ull * (uint64_t)tid;
    const int j_lo = a.pos_lo > P0 ? (int)min((uint64_t)kPosPerLane, a.pos_lo - P0) : 0;
    const int j_hi = a.pos_hi > P0 ? (int)min((uint64_t)kPosPerLane, a.pos_hi - P0) : 0;
    uint32_t ok16, nok16, pok16;  // as in k_stream_bin
    {
      uint64_t Mh = Vh;
      for (int c = 1; c < k;) { const int s = min(c, k - c); Mh |= Mh << s; c += s; }
      const uint32_t range = ((0x10000u >> j_lo) - 1u) & ~((0x10000u >> j_hi) - 1u);
      ok16 = ~(uint32_t)(Mh >> 48) & range;
      nok16 = ~(uint32_t)(Vh >> (48 - k)) & 0xFFFFu;
      pok16 = ~((prev_chunk_inv << 15) | (uint32_t)(Vh >> 49)) & 0xFFFFu;
    }
    if (ok16) {
      n_kmers += __popc(ok16);
      n_contigs += __popc(ok16 & ~pok16);
      // window: base i (relative to the lane's first position) is base i + 1 of (hiW : loW)
      const uint64_t hiW = code_win64(s_code, pl - 1), loW = code_win64(s_code, pl + 31);
      auto nuc_at = [&](int t) -> uint32_t {  // t = i + 1
        return t < 32 ? (uint32_t)(hiW >> (62 - 2 * t)) & 3u : (uint32_t)(loW >> (126 - 2 * t)) & 3u;
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
      uint32_t pre = 0xFFFFFFFFu;
      own[0] = owner_of_minimizer(min(common, h[0]), out.lbo);
#pragma unroll
      for (int j = 1; j < 16; j++) {
        pre = min(pre, push(nuc_at(k + j)));                          // m-mer k - M + j ends at base k + j - 1
        const uint32_t left = j < 15 ? h[j] : 0xFFFFFFFFu;
        own[j] = owner_of_minimizer(min(min(common, left), pre), out.lbo);
      }
      // runs of consecutive valid k-mers with one owner -> records staged in LDS
      const uint64_t w1hi = loW & 0xFFFFFFFF00000000ULL;
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
          const ulonglong2 rec = make_ulonglong2(hiW, w1hi | hdr);
          if (slot < (uint32_t)kSkStage) { s_rec[slot] = rec; s_own[slot] = (uint8_t)cur; }
          else {  // staging full (pathological input): straight to the bin, one global atomic
            const unsigned long long pos = atomicAdd(&out.counts[rep * nparts + cur], 1ULL);
            if (pos < out.cap) out.recs[((uint64_t)cur * out.rep + rep) * out.cap + pos] = rec;
            else dropped = 1;
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
      if (pos < out.cap) out.recs[((uint64_t)o * out.rep + rep) * out.cap + pos] = s_rec[q];
      else dropped = 1;
    }
  }
  block_add(&a.ctr->kmers, n_kmers);
  block_add(&a.ctr->contigs, n_contigs);
  if (dropped) a.ctr->bin_over = 1;
}

// ---------------------------------------------------------------------------
// owner: super-k-mer records -> region bins of packed tuples (same output as k_stream_bin)
// ---------------------------------------------------------------------------
struct SuperkIn {
  const ulonglong2 *recs;            // [nseg][seg_cap]
  const unsigned long long *counts;  // [nseg]; a fill above seg_cap is read as seg_cap
  uint64_t seg_cap;
  uint32_t nseg;
};

// Records hold 1..16 k-mers (16 when one GPU owns everything, 6-7 on average at 8 owners), and
// a lane that walks its record's 16 positions pays for all 16 whatever the run's length.  So the
// K-MERS -- not the records -- are dealt out to the lanes.  A block walks a chunk of a segment and
// takes, per tile, as many records (in fours: one 64-byte load per lane) as hold <= kTile k-mers;
// k-mer q of the tile (records in order, their k-mers in order) goes to lane q % kThreads as its
// (q / kThreads)-th, and every lane extracts its k-mers straight from the record's 48-base
// window.  All lanes carry the same number of k-mers (+-1) and tiles are full (>= kTile - 63
// k-mers) except at a chunk's end, so the cost per k-mer does not depend on the run lengths.
constexpr int kSkPerLane = 4;                      // candidate records per lane and tile
constexpr int kSkCand = kSkPerLane * kThreads;     // 1024
constexpr uint32_t kSkChunk = 1u << 13;            // records per unit of work (a block's walk)
constexpr size_t kSkMapBytes = (size_t)kTile * 2;  // s_map behind the BinLds block in dynamic LDS

template <bool ONECOL, int NB>
__global__ __launch_bounds__(kThreads, 4) void k_superk_bin(SuperkIn in, int k, BinSpec bs, BinOut out,
                                                            InsertSink<1, ONECOL> isink, Counters *ctr)
{
  constexpr int W = 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  using LDS = BinLds<W, NB, false>;
  LDS &L = *reinterpret_cast<LDS *>(dyn_lds);
  // the taken records sit in the staging area (not written before bin_place); the k-mer -> (record,
  // index in run) map has its own 8 KB
  static_assert(sizeof(L.skey) >= (size_t)kSkCand * 16, "records fit the staging area");
  ulonglong2 *s_rec = reinterpret_cast<ulonglong2 *>(L.skey);
  uint16_t *s_map = reinterpret_cast<uint16_t *>(dyn_lds + ((sizeof(LDS) + 15) & ~(size_t)15));  // record | index << 10
  __shared__ uint32_t s_incl[kThreads + 1];
  __shared__ uint32_t s_wsum[kThreads / 64];
  __shared__ uint32_t s_T, s_nl;
  const int tid = threadIdx.x;
  uint32_t n_novel = 0, full = 0;
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
    const ulonglong2 *recs = in.recs + (uint64_t)seg * in.seg_cap;
    for (uint64_t pos = c0; pos < c1;) {  // uniform
      __syncthreads();
      for (uint32_t b = tid; b < bs.nlocal; b += kThreads) L.cnt[b] = 0;
      // four candidate records per lane, their k-mer counts, inclusive scan over the block
      ulonglong2 ra = make_ulonglong2(0, 0), rb = ra, rc4 = ra, rd = ra;
      uint32_t la = 0, lb = 0, lc = 0, ld = 0;
      {
        const uint64_t i = pos + (uint64_t)kSkPerLane * (uint64_t)tid;
        auto rlen = [](const ulonglong2 &r) { return (((uint32_t)r.y >> kSkLenShift) & 0xFu) + 1u; };
        if (i + 0 < c1) { ra = recs[i + 0]; la = rlen(ra); }
        if (i + 1 < c1) { rb = recs[i + 1]; lb = rlen(rb); }
        if (i + 2 < c1) { rc4 = recs[i + 2]; lc = rlen(rc4); }
        if (i + 3 < c1) { rd = recs[i + 3]; ld = rlen(rd); }
      }
      const uint32_t lsum = la + lb + lc + ld;
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
        s_rec[r0 + 0] = ra; s_rec[r0 + 1] = rb; s_rec[r0 + 2] = rc4; s_rec[r0 + 3] = rd;
        for (uint32_t i = 0; i < la; i++) s_map[o++] = (uint16_t)((r0 + 0u) | (i << 10));
        for (uint32_t i = 0; i < lb; i++) s_map[o++] = (uint16_t)((r0 + 1u) | (i << 10));
        for (uint32_t i = 0; i < lc; i++) s_map[o++] = (uint16_t)((r0 + 2u) | (i << 10));
        for (uint32_t i = 0; i < ld; i++) s_map[o++] = (uint16_t)((r0 + 3u) | (i << 10));
      }
      __syncthreads();
      const uint32_t T = s_T;
      pos += (uint64_t)kSkPerLane * s_nl;

      Kmer<W> tk[kPosPerLane];
      uint32_t tle[kPosPerLane];
      uint32_t vmask = 0;
      const int nj = (int)((T + kThreads - 1) / kThreads);  // uniform
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++) {
        const uint32_t q = (uint32_t)j * kThreads + (uint32_t)tid;
        if (j < nj && q < T) {  // (j < nj is uniform: whole iterations are skipped)
          const uint32_t m = s_map[q];
          const ulonglong2 rec = s_rec[m & 0x3ffu];
          const uint64_t hiW = rec.x, loW = rec.y & 0xFFFFFFFF00000000ULL;
          const uint32_t hdr = (uint32_t)rec.y;
          const uint32_t i = m >> 10, rlen = ((hdr >> kSkLenShift) & 0xFu) + 1u;
          const uint32_t p = (hdr & kSkStartMask) + i;       // the k-mer is bases p+1 .. p+k of the window
          const uint32_t sft = 2u * (p + 1u);                // 2 .. 32
          const uint64_t X = (hiW << sft) | (loW >> (64u - sft));  // bases p+1 .. p+32 on top
          Kmer<W> fw, rc;
          fw.w[0] = X >> (64 - 2 * k);
          rc = revcomp<W>(fw, k);
          // base after the k-mer = the one below the top k bases of X (k <= 31); base before it = window base p
          const uint32_t nuc_next = (uint32_t)(X >> (62 - 2 * k)) & 3u;
          const uint32_t prev_nuc = (uint32_t)(hiW >> (62u - 2u * p)) & 3u;
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
          const uint32_t G = r ^ (region_mix<W>(qq) & ((1u << lbq) - 1u));
          const uint32_t local = G & ((1u << isink.t.lb1) - 1u);
          tk[j] = tuple_pack<W>(qq, e);
          tle[j] = local << 8;
          vmask |= 1u << j;
          atomicAdd(&L.cnt[local], 1u);
        }
      }
      BinRes<NB> res;
      bin_reserve<LDS, NB>(L, bs, out, ob0, res);   // (its first barrier also ends the reads of s_rec / s_map)
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++)
        if (j < nj && (vmask & (1u << j))) tle[j] |= bin_rank<LDS>(L, (tle[j] >> 8) & 0x7ffu) << 19;
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
  if (full == 1) ctr->full = 1;
  if (full == 2) ctr->bin_over = 1;
}

#endif  // __HIPCC__

}  // namespace mcx
