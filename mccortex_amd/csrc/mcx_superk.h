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

  for (uint64_t tile = a.tile0 + blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();
    const int64_t region0 = (int64_t)(tile * kTile) - 16;
    for (int c = tid; c < kChunks; c += kThreads) {
      uint32_t code, inv;
      encode_chunk(a.stream, a.nbytes, region0 + 16 * (int64_t)c, code, inv);
      s_code[c] = code;
      reinterpret_cast<uint16_t *>(s_inv)[c ^ 1] = (uint16_t)inv;
    }
    if (tid < 4) { s_code[kChunks + tid] = 0; s_inv[kChunks / 2 + tid] = 0xFFFFFFFFu; }
    if (tid < 32) { s_cnt[tid] = 0; s_rank[tid] = 0; }
    if (tid == 0) s_total = 0;
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

template <bool ONECOL, int NB>
__global__ __launch_bounds__(kThreads, 4) void k_superk_bin(SuperkIn in, int k, BinSpec bs, BinOut out,
                                                            InsertSink<1, ONECOL> isink, Counters *ctr)
{
  constexpr int W = 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  using LDS = BinLds<W, NB, false>;
  LDS &L = *reinterpret_cast<LDS *>(dyn_lds);
  const int tid = threadIdx.x;
  uint32_t n_novel = 0, full = 0;
  const uint64_t top_mask = ~0ULL >> (64 - 2 * k);
  const int first_shift = 2 * k - 2;
  const uint32_t ob0 = (blockIdx.x % bs.rep) * bs.nout;
  // a tile = kThreads records = up to 4096 k-mers: the partition machinery of mcx_defer.h as is
  const uint64_t tiles_per_seg = (in.seg_cap + kThreads - 1) / kThreads;
  const uint64_t ntiles = tiles_per_seg * in.nseg;
  for (uint64_t v = blockIdx.x; v < ntiles; v += gridDim.x) {
    const uint32_t seg = (uint32_t)(v % in.nseg);  // segment-interleaved
    const uint64_t i0 = (v / in.nseg) * kThreads;
    uint64_t cnt = in.counts[seg];
    if (cnt > in.seg_cap) cnt = in.seg_cap;
    if (i0 >= cnt) continue;  // uniform
    __syncthreads();
    for (uint32_t b = tid; b < bs.nlocal; b += kThreads) L.cnt[b] = 0;
    __syncthreads();
    Kmer<W> tk[kPosPerLane];
    uint32_t tle[kPosPerLane];
    uint32_t vmask = 0;
    if (i0 + (uint64_t)tid < cnt) {
      const ulonglong2 rec = in.recs[(uint64_t)seg * in.seg_cap + i0 + tid];
      const uint64_t hiW = rec.x, loW = rec.y & 0xFFFFFFFF00000000ULL;
      const uint32_t hdr = (uint32_t)rec.y;
      const uint32_t start = hdr & kSkStartMask, len = ((hdr >> kSkLenShift) & 0xFu) + 1u;
      const uint32_t run = ((0x10000u >> start) - 1u) & ~((0x10000u >> (start + len)) - 1u);  // bit 15 - j
      // k-mer at position 0 = bases 0 .. k-1 = window bases 1 .. k
      Kmer<W> fw, rc;
      fw.w[0] = (hiW << 2) >> (64 - 2 * k);
      rc = revcomp<W>(fw, k);
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++) {
        const uint32_t bit = 0x8000u >> j;
        const int tn = k + j + 1;  // window index of the base after k-mer j
        const uint32_t nuc_next = tn < 32 ? (uint32_t)(hiW >> (62 - 2 * tn)) & 3u : (uint32_t)(loW >> (126 - 2 * tn)) & 3u;
        const uint32_t prev_nuc = (uint32_t)(hiW >> (62 - 2 * j)) & 3u;  // window index j = base j - 1
        if (run & bit) {
          const bool next_ok = ((uint32_t)j + 1u < start + len) || (hdr & kSkNextOk);
          const bool prev_ok = ((uint32_t)j > start) || (hdr & kSkPrevOk);
          uint32_t o;
          const Kmer<W> key = canonical<W>(fw, rc, o);
          uint32_t e = 0;
          if (next_ok) e |= 1u << (nuc_next + 4u * o);
          if (prev_ok) e |= 1u << ((3u - prev_nuc) + 4u * (1u - o));
          uint32_t r;
          const uint32_t lbq = lbq_of(isink.t);
          const Kmer<W> q = key_quot<W>(key, lbq, r);
          const uint32_t G = r ^ (region_mix<W>(q) & ((1u << lbq) - 1u));
          const uint32_t local = G & ((1u << isink.t.lb1) - 1u);
          tk[j] = tuple_pack<W>(q, e);
          tle[j] = local << 8;
          vmask |= 1u << j;
          atomicAdd(&L.cnt[local], 1u);
        }
        fw.w[0] = ((fw.w[0] << 2) | nuc_next) & top_mask;
        rc.w[0] = (rc.w[0] >> 2) | ((uint64_t)(3u - nuc_next) << first_shift);
      }
    }
    BinRes<NB> res;
    bin_reserve<LDS, NB>(L, bs, out, ob0, res);
#pragma unroll
    for (int j = 0; j < kPosPerLane; j++)
      if (vmask & (1u << j)) tle[j] |= bin_rank<LDS>(L, (tle[j] >> 8) & 0x7ffu) << 19;
    bin_commit<LDS, NB>(L, bs, out, ob0, res);
    for (int round = 0; round < kRounds; round++) {
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++)
        if (vmask & (1u << j))
          bin_place<W, false, LDS>(L, round, tle[j] >> 19, (tle[j] >> 8) & 0x7ffu, tk[j], 0);
      bin_writeout<W, ONECOL, false, 0, LDS>(L, round, bs, out, ob0, 0, isink, n_novel, full);
    }
  }
  block_add(&ctr->novel, n_novel);
  if (full == 1) ctr->full = 1;
  if (full == 2) ctr->bin_over = 1;
}

#endif  // __HIPCC__

}  // namespace mcx
