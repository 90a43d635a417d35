// mcx_defer.h -- partition-then-insert path of the build ("deferred" mode) and the
// owner partition of the sharded build.  Included by mcx_api.hip after mcx_kernels.h.
//
// Why: on MI355X every device atomic costs one 64-byte line request and the chip retires
// ~17 G of them per second on a 16 GiB table (23.7 G/s even when the table sits in the
// Infinity Cache) -- profiles/r01_ubench_atomics*.log.  The fused kernel already runs at that
// ceiling, so the only way past it is to stop issuing one HBM atomic per k-mer occurrence:
//
//   1. k_stream_bin      k-merise reads, radix-partition the per-occurrence tuples by table
//                        region (the 2^lb1 "L1 bins" of the quotient hash)         (streaming)
//   2. k_tuples_bin      split every L1 bin by sub-table (4096 slots) into L2 bins  (streaming)
//   3. k_lds_insert      one workgroup per sub-table: slice -> LDS, apply its tuples with
//                        LDS atomics (find-or-insert, coverage +1, edge OR), slice -> HBM
//
// Inside the bins an occurrence is ONE 64-bit word per key word: the quotient q of the key
// (mcx_kernels.h, "Table addressing") with the edge byte in bits 56..63 of the top word; the
// remainder r is implied by the region the tuple sits in.  All HBM traffic is coalesced
// streaming; tuples that do not fit a bin (hot k-mers, skew) fall back to the lock-free direct
// insert of mcx_kernels.h, so no input can overflow.  The same binning kernel with BIN_OWNER
// produces the per-GPU bins of the sharded build (full keys + edge bytes, the exchange format).
#pragma once
#include "mcx_kernels.h"

namespace mcx {

// BIN_OWNER    reads -> per-owner bins of full tuples, owner from lookup3's 2nd word (exchange format v1)
// BIN_GROUP    tuples of THIS shard -> its region bins (packed)
// BIN_SUBLOCAL packed tuples of one region -> its sub-table bins
// BIN_GLOBAL   reads -> bins of every (owner, region) of the sharded table (packed), laid out
//              [owner][replica][region] so that each owner's block is one contiguous message
enum : int { BIN_OWNER = 0, BIN_GROUP = 1, BIN_SUBLOCAL = 2, BIN_GLOBAL = 3 };

struct BinSpec {
  int mode;
  uint32_t nparts;  // BIN_OWNER: number of owners
  uint32_t nlocal;  // bins a block can meet (size of the LDS histogram), <= kMaxBins
  // Output replication (replica-major): every output bin exists `rep` times (own counter, own
  // storage) and a block appends to replica blockIdx % rep.  All blocks reserving from the same
  // 64 counter lines once per tile serialise on those lines (~12 ns per same-line atomic:
  // 1.7 of 5 ms per 600 M tuples, profiles/r01b); 8 replicas in separate lines remove that, and
  // since block b is observed to run on XCD b % 8 each replica's write fronts stay in one L2,
  // which merges the 8-tuple runs into full-line write-backs (speed only, never correctness).
  uint32_t rep;
  uint32_t nout;     // output bins per replica
  uint32_t seg_mod;  // BIN_SUBLOCAL: input segment s holds region region0 + s % seg_mod
  uint32_t lb1;      // BIN_GLOBAL: log2 regions per owner
  uint32_t region0;  // BIN_SUBLOCAL: first region of the group being split (output bins are group-local);
                     // BIN_GLOBAL: 1 = a spill area follows the owners' overflow bins (BinOut::ov_counts)
};

// index of the output segment of local bin b
__device__ __forceinline__ uint32_t out_seg(const BinSpec &bs, uint32_t ob0, uint32_t b)
{
  if (bs.mode == BIN_GLOBAL)
    return ((((b >> bs.lb1) * bs.rep + blockIdx.x % bs.rep) << bs.lb1) | (b & ((1u << bs.lb1) - 1u)));
  return ob0 + b;
}

// Output bins.  Packed format (deferred path): `keys` holds W words per tuple, `edges` unused.
// Full format (owner bins): W key words + one edge byte per tuple.
struct BinOut {
  uint64_t *keys;              // [rep][nbins][cap][W]
  uint8_t *edges;              // [rep][nbins][cap] (full format only)
  unsigned long long *counts;  // [rep][nbins] fill (may exceed cap: the excess went to the fallback)
  uint64_t cap;                // tuples per (replica, bin) segment
  // BIN_GLOBAL only: per-owner overflow bins in full format for tuples beyond a segment's capacity,
  // then the sender's spill area for tuples beyond an overflow bin's capacity (bin_writeout)
  uint64_t *ov_keys;           // [nparts][ov_cap][W], then [spill capacity][W]
  uint8_t *ov_edges;           // [nparts][ov_cap], then [spill capacity]
  unsigned long long *ov_counts;  // [nparts] fills, [nparts] = spill fill, [nparts + 1] = spill capacity
  uint64_t ov_cap;
};

#define MCX_LDS_AS __attribute__((address_space(3)))

// Per-phase time of the three build kernels (tools/exp_phases.py, profiles/r04_phases.md): compiled in ONLY with
// -DMCX_PHASES (a tools/variants.sh build); in the product build the hooks expand to nothing.  Thread 0 of every
// block reads the 100 MHz clock at the phase boundaries (most of them barriers) and the per-block sums are added
// to g_phase[kernel][phase]; phase 7 counts the tiles / sub-table visits.
#ifdef MCX_PHASES
__device__ unsigned long long g_phase[3][8];
#define MCX_PH_DECL unsigned long long ph_t = wall_clock64(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MCX_PH(i) { const unsigned long long n_ = wall_clock64(); ph_acc[i] += n_ - ph_t; ph_t = n_; }
#define MCX_PH_COUNT ph_acc[7]++;
#define MCX_PH_DUMP(kid) { if (threadIdx.x == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_phase[kid][i_], ph_acc[i_]); }
#else
#define MCX_PH_DECL
#define MCX_PH(i)
#define MCX_PH_COUNT
#define MCX_PH_DUMP(kid)
#endif
constexpr int kMaxBins = 2048;
constexpr uint64_t kQMask = (1ull << 56) - 1;  // quotient bits of the top tuple word

// pack / unpack a tuple of the deferred path
template <int W> __device__ __forceinline__ Kmer<W> tuple_pack(const Kmer<W> &q, uint32_t e)
{
  Kmer<W> t = q;
  t.w[0] |= (uint64_t)e << 56;
  return t;
}
template <int W> __device__ __forceinline__ Kmer<W> tuple_q(const Kmer<W> &t)
{
  Kmer<W> q = t;
  q.w[0] &= kQMask;
  return q;
}

// The sorted tile is staged and written out in kRounds rounds of kStage tuples: the staging
// area is what limits blocks per CU (W=2: 84 KB for a whole tile = 1 block, 46 KB = 3 blocks).
constexpr int kRounds = 2;  // (1 round: k_stream_bin 22.8 -> 31.2 ms at C2, round 3)
constexpr int kStage = kTile / kRounds;

// Geometry of a binning block: T threads, TILE tuples per tile (staged in kRounds rounds).  The
// k-merising kernel and the super-k-mer kernels use 256 x 16; the split uses 512 x 16: its runs per
// sub-table bin are twice as long (128 B), so fewer of its lines are written in two halves.
template <int T_, int TILE_, int R_ = kRounds> struct Geo {
  static constexpr int kT = T_, kTileG = TILE_, kRoundsG = R_, kStageG = TILE_ / R_;
};
using Geo256 = Geo<kThreads, kTile>;

// LDS working set of one binning block (NB = histogram capacity)
template <int W, int NB, bool FULL, class G = Geo256> struct BinLds {
  using geo = G;
  uint64_t skey[G::kStageG * W];
  // per bin, for the write-out: bits 0..47 = (output tuple index of the bin's first tuple of this
  // tile) - (its sorted position in the tile), mod 2^48; bits 48..63 = sorted positions below this
  // value still fit the bin's segment (the rest overflows)
  unsigned long long gbase[NB];
  uint32_t cnt[NB + 64];  // [nlocal + lane] = the trash bins (positions without a tuple), one per lane
  uint32_t rnk[NB + 64];  // rank counters of the placement (zeroed with cnt at the top of a tile)
  uint32_t off[NB + 64];
  uint32_t wsum[G::kT / 64];
  uint16_t sbin[G::kStageG];
  uint8_t se[FULL ? G::kStageG : 16];
};

// The thread index, opaque to the optimiser: what is derived from it inside a tile loop (LDS
// addresses, masks) is then recomputed per tile in an instruction or two -- hoisted out of the
// loop these values were spilled to scratch, and a scratch reload waits for the next tile's
// prefetch (vmcnt counts in order).
// Used by k_stream_bin's tile loop only (the write-out loop of bin_writeout keeps threadIdx.x: see the
// note there and profiles/r03_writeout_fault.md).
__device__ __forceinline__ uint32_t tid_now()
{
  uint32_t t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// After the counting sweep: exclusive scan of the histogram (wave 0), then every thread issues
// the global reservations of its bins (one returning atomic per non-empty bin).  The results
// stay in registers: placement into the LDS staging area only needs off[], so the atomics'
// round trip overlaps with it; bin_commit() publishes the bases before the write-out.
template <int NB, int T = kThreads> struct BinRes { unsigned long long g0[(NB + T - 1) / T]; };

template <class LDS, int NB>
__device__ __forceinline__ void bin_reserve(LDS &L, const BinSpec &bs, const BinOut &out, uint32_t ob0, BinRes<NB, LDS::geo::kT> &res,
                                            bool trash_beyond = false)
{
  constexpr int kT = LDS::geo::kT;
  const int tid = threadIdx.x;
  __syncthreads();
  {  // every thread scans NB / kT consecutive bins; the waves' totals meet in LDS.  (One wave
     // scanning all bins while the other three wait at the barrier cost 8 % of the k-merising kernel.)
    constexpr int PERB = NB / kT;
    static_assert(NB % kT == 0, "bins per thread");
    uint32_t c[PERB], x = 0;
#pragma unroll
    for (int i = 0; i < PERB; i++) {
      const uint32_t b = (uint32_t)tid * PERB + i;
      c[i] = b < bs.nlocal ? L.cnt[b] : 0;
      x += c[i];
    }
    const uint32_t mine = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d, 64);
      if ((tid & 63) >= d) x += y;
    }
    if ((tid & 63) == 63) L.wsum[tid >> 6] = x;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kT / 64; w++) {
      const uint32_t ws = L.wsum[w];
      if (w < (tid >> 6)) before += ws;
      total += ws;
    }
    uint32_t o = before + x - mine;
#pragma unroll
    for (int i = 0; i < PERB; i++) {
      const uint32_t b = (uint32_t)tid * PERB + i;
      if (b < bs.nlocal) L.off[b] = o;
      o += c[i];
    }
    // [nlocal] = the tile's total; then where the trash bins start: behind the real tuples, or
    // (trash_beyond) past the tile, where the placement never looks
    if (tid < 64) L.off[bs.nlocal + tid] = tid && trash_beyond ? (uint32_t)LDS::geo::kTileG : total;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < (NB + kT - 1) / kT; q++) {
    const uint32_t b = (uint32_t)q * kT + tid;
    res.g0[q] = 0;
    if (b < bs.nlocal) {
      const uint32_t c = L.cnt[b];
      if (c) res.g0[q] = atomicAdd(&out.counts[out_seg(bs, ob0, b)], (unsigned long long)c);
    }
  }
  // (no barrier: the ranking that follows counts in rnk[], which was zeroed at the top of the tile)
}

constexpr unsigned long long kDstMask = (1ULL << 48) - 1;

template <class LDS, int NB>
__device__ __forceinline__ void bin_commit(LDS &L, const BinSpec &bs, const BinOut &out, uint32_t ob0, const BinRes<NB, LDS::geo::kT> &res)
{
  constexpr int kT = LDS::geo::kT;
#pragma unroll
  for (int q = 0; q < (NB + kT - 1) / kT; q++) {
    const uint32_t b = (uint32_t)q * kT + threadIdx.x;
    if (b < bs.nlocal) {
      const uint32_t o0 = L.off[b], o1 = L.off[b + 1];
      const unsigned long long g0 = res.g0[q];                       // start of the run in its segment
      const unsigned long long room = out.cap > g0 ? out.cap - g0 : 0;  // tuples that still fit
      const uint32_t lim = o0 + (uint32_t)min((unsigned long long)(o1 - o0), room);
      const unsigned long long dst = ((unsigned long long)out_seg(bs, ob0, b) * out.cap + g0 - o0) & kDstMask;
      L.gbase[b] = dst | ((unsigned long long)lim << 48);
    }
  }
}

// Sorted position of a tuple inside the tile (taken once, kept in a register)
template <class LDS> __device__ __forceinline__ uint32_t bin_rank(LDS &L, uint32_t local)
{
  return L.off[local] + atomicAdd(&L.rnk[local], 1u);
}

// Placement: drop a tuple at its sorted position if that position belongs to this round
template <int W, bool FULL, class LDS>
__device__ __forceinline__ void bin_place(LDS &L, int round, uint32_t p, uint32_t local, const Kmer<W> &t, uint32_t e)
{
  constexpr uint32_t kSt = LDS::geo::kStageG;
  if ((int)(p / kSt) != round) return;
  const uint32_t q = p % kSt;
  L.skey[q * W] = t.w[0];
  if (W == 2) L.skey[q * W + 1] = t.w[W - 1];
  L.sbin[q] = (uint16_t)local;
  if (FULL) L.se[q] = (uint8_t)e;
}

// Linear write-out of one round: consecutive lanes write consecutive tuples of a bin.  Tuples
// beyond a bin's capacity take the direct insert (deferred modes; the region of a packed tuple
// is its bin in BIN_GROUP and `region_of_seg` in BIN_SUBLOCAL) or raise bin_over (owner mode).
template <int W, bool ONECOL, bool FULL, int SH, class LDS>
__device__ __forceinline__ void bin_writeout(LDS &L, int round, const BinSpec &bs, const BinOut &out, uint32_t ob0,
                                             uint32_t region_of_seg, const InsertSink<W, ONECOL> &isink,
                                             uint32_t &novel, uint32_t &full)
{
  __syncthreads();
  const uint32_t n = L.off[bs.nlocal];
  constexpr uint32_t kSt = LDS::geo::kStageG;
  // Sorted positions [lo, hi) of the tile are staged in this round.  The loop runs over the POSITION p,
  // not over an index into the staging area counted from "n - lo": written as
  //     cnt = n > lo ? min(n - lo, kSt) : 0;  for (q = tid; q < cnt; ...)
  // the compiler (ROCm 7.2 clang, gfx950) turned the guarded difference into a saturating
  // subtraction and, in the variant whose loop index started from the opaque tid_now(), emitted it as
  // a plain v_add_u32 -kSt followed by v_min_u32: with fewer than kSt tuples in the tile (n < lo in
  // round 1) the difference wrapped, cnt became kSt, stale staging entries were "written out", failed
  // the capacity test and took the direct-insert branch with a stale bin as their region -> a wild
  // table address.  That was the unexplained fault of round 2 (profiles/r03_writeout_fault.md).  No
  // subtraction of n is left to get wrong.
  const uint32_t lo = (uint32_t)round * kSt;
  const uint32_t hi = min(n, lo + kSt);
  // one staged tuple (sorted position p, bin b, that bin's base word gb, tuple words t0 [t1]) -> its place
  auto emit = [&](uint32_t p, uint32_t b, unsigned long long gb, uint64_t t0, uint64_t t1) {
    if (p < (uint32_t)(gb >> 48)) {
      const uint64_t at = (gb + p) & kDstMask;
      uint64_t *kd = out.keys + at * W;
      kd[0] = t0;
      if (W == 2) kd[1] = t1;
      if (FULL) out.edges[at] = L.se[p - lo];
    } else if (FULL || bs.mode == BIN_OWNER) {
      full = 2;
    } else {  // packed tuple -> full key
      Kmer<W> tq;
      tq.w[0] = t0;
      if (W == 2) tq.w[W - 1] = t1;
      const uint32_t e = (uint32_t)(tq.w[0] >> 56);
      const Kmer<W> qq = tuple_q<W>(tq);
      const uint32_t lbq = lbq_of(isink.t);
      if (SH == 2) {  // BIN_GLOBAL ... appended to the owner's overflow bin (full format)
        const Kmer<W> key = key_unquot<W>(qq, lbq, b ^ mix_g(region_mix<W>(qq), lbq));
        const uint32_t owner = b >> bs.lb1;
        const unsigned long long pos = atomicAdd(&out.ov_counts[owner], 1ULL);
        if (pos < out.ov_cap) {
          uint64_t *kd = out.ov_keys + ((uint64_t)owner * out.ov_cap + pos) * W;
          kd[0] = key.w[0];
          if (W == 2) kd[1] = key.w[W - 1];
          out.ov_edges[(uint64_t)owner * out.ov_cap + pos] = (uint8_t)e;
        } else {
          // The owner's overflow bin is full as well (one k-mer tens of thousands of times in a
          // piece: poly-G reads, satellite repeats, a homopolymer contig).  Behind the nparts owner
          // bins sits the sender's spill area (any owner, full format; fill in ov_counts[nparts],
          // capacity in ov_counts[nparts + 1], 0 = none): it stays on the sender, and the host routes
          // it to the owners when it next looks (mcx_multi.h, group_route_spill).  With a spill area
          // as large as the piece nothing can be lost.
          const unsigned long long sp_cap = bs.region0 ? out.ov_counts[bs.nparts + 1] : 0;  // (BIN_GLOBAL: region0 = "a spill area follows")
          const unsigned long long sp = sp_cap ? atomicAdd(&out.ov_counts[bs.nparts], 1ULL) : 0;
          if (sp < sp_cap) {
            const uint64_t at = (uint64_t)bs.nparts * out.ov_cap + sp;
            uint64_t *kd = out.ov_keys + at * W;
            kd[0] = key.w[0];
            if (W == 2) kd[1] = key.w[W - 1];
            out.ov_edges[at] = (uint8_t)e;
          } else {
            full = 2;
          }
        }
      } else {                      // ... lock-free insert into the HBM table
        table_mark_written(isink.t);
        table_count_fallback(isink.t);
        const uint32_t region = bs.mode == BIN_GROUP ? b : region_of_seg;
        const Kmer<W> key = key_unquot<W>(qq, lbq, r_of<W>(isink.t, region, qq));
        const uint64_t slot = key_slot<W>(isink.t, key);
        const uint64_t cur = *key_ptr_t<W, ONECOL>(isink.t, slot);
        probe_insert<W, ONECOL>(isink.t, key, slot, cur, 0, e, isink.col, novel, full);
      }
    }
  };
  // (A lane taking two neighbouring sorted positions -- one 16-byte store where both go to the same bin at an
  // even index, else two 8-byte stores -- was tried in round 3: k_stream_bin 22.5 -> 30 ms, the unmerged pairs
  // write half lines per instruction.  One position per lane it is.)
  for (uint32_t p = lo + threadIdx.x; p < hi; p += LDS::geo::kT) {
    const uint32_t q = p - lo;
    const uint32_t b = L.sbin[q];
    emit(p, b, L.gbase[b], L.skey[q * W], W == 2 ? L.skey[q * W + W - 1] : 0);
  }
  if (round + 1 < LDS::geo::kRoundsG) __syncthreads();  // staging is reused by the next round
}

// The lane's 16 k-mers and their reverse complements as windows of two 96-bit registers (one-word
// keys, k <= 31), LEFT-aligned: with A = the lane's bases 0..47 (base 0 on top) and R = the reverse
// complement of bases 0..k+14, also top-aligned,
//     fw_j = top 64 bits of A << 2 j,        rc_j = top 64 bits of R << 2 (15 - j)
// hold the k-mer in their upper 2k bits and other bases below.  Comparing them picks the
// canonical strand all the same (k is odd: the two k-mers differ, so the highest differing bit is
// one of theirs), and ONE shift of the winner by 64 - 2k + lbq yields the quotient: no masks.
// After unrolling the window shifts are constants: two v_alignbit per strand and no dependency
// chain from one position to the next (rolling both strands cost 12 instructions per position).
struct LaneWin { uint32_t a2, a1, a0, r2, r1, r0; };
__device__ __forceinline__ uint32_t pairrev32(uint32_t x)  // order of the 16 two-bit groups reversed
{
  const uint32_t r = __builtin_bitreverse32(x);
  return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}
__device__ __forceinline__ LaneWin lane_win(const uint32_t *s_code, uint32_t pl, int k)
{
  LaneWin w;
  w.a2 = s_code[pl >> 4];  // pl is a multiple of 16: whole code words
  w.a1 = s_code[(pl >> 4) + 1];
  w.a0 = s_code[(pl >> 4) + 2];
  // reverse complement of the 48 bases: its first 33 - k bases belong to bases beyond k + 14
  const uint32_t c2 = pairrev32(~w.a0), c1 = pairrev32(~w.a1), c0 = pairrev32(~w.a2);
  const int t = 66 - 2 * k;  // 4..60 (uniform)
  if (t < 32) {
    w.r2 = (c2 << t) | (c1 >> (32 - t));
    w.r1 = (c1 << t) | (c0 >> (32 - t));
    w.r0 = c0 << t;
  } else {
    const uint64_t x = (((uint64_t)c1 << 32) | c0) << (t - 32);
    w.r2 = (uint32_t)(x >> 32);
    w.r1 = (uint32_t)x;
    w.r0 = 0;
  }
  return w;
}

// A key owned by another shard cannot be packed for this one (its remainder would be rebuilt
// with the wrong owner bits): it takes the direct insert.  Never happens when nparts == 1.
template <int W, bool ONECOL>
__device__ __noinline__ void foreign_insert(const InsertSink<W, ONECOL> &isink, Kmer<W> key, uint32_t e,
                                            uint32_t &novel, uint32_t &full)
{
  table_mark_written(isink.t);
  table_count_foreign(isink.t);
  const uint64_t slot = key_slot<W>(isink.t, key);
  const uint64_t cur = *key_ptr_t<W, ONECOL>(isink.t, slot);
  probe_insert<W, ONECOL>(isink.t, key, slot, cur, 0, e, isink.col, novel, full);
}

// ---------------------------------------------------------------------------
// 1. reads -> bins.  FULL = owner bins (key words + edge byte, nparts bins);
//    !FULL = region bins (packed tuples): SH 0 of an unsharded table, SH 1 of this shard (keys of
//    other shards take the direct insert), SH 2 of every shard (BIN_GLOBAL, exchange blocks).
//    The variants are compiled apart: code of the rare paths costs the common one registers.
// ---------------------------------------------------------------------------
// T = 256: one tile of 4096 positions per block iteration (4 blocks per CU).  T = 512 (region bins of an unsharded
// one-word table): the two halves of the block k-merise two neighbouring tiles, each with its own staged codes, and
// the block sorts the 8192 tuples as ONE tile -- half as many histogram scans, reservations and commits per tuple,
// and runs of 16 tuples (128 bytes) per bin instead of 8; 2 blocks per CU, the same 16 waves.
template <int W, bool ONECOL, int NB, bool FULL, int SH, bool PK, int T = kThreads>
__global__ __launch_bounds__(T, (W == 1 ? 4 : 3)) void k_stream_bin(StreamArgs a_arg, BinSpec bs, BinOut out_arg,
                                                                  InsertSink<W, ONECOL> isink_arg)
{
  constexpr int H = T / kThreads;  // tiles k-merised side by side
  static_assert(T == kThreads || (T == 2 * kThreads && !FULL), "256 threads, or 512 for packed bins");
  __shared__ uint32_t s_code_h[H][kChunks + 4];
  __shared__ uint32_t s_inv_h[H][kChunks / 2 + 4];
  // Arguments that are read once per tile (owned range), after the loop (counters), or
  // only by the rare paths (the table's description: bin overflow, foreign keys), are read from
  // LDS: held in scalar registers through the tile loop they exhausted the register file --
  // every tile paid ~80 v_readlane, and uniform values that had been moved to vector registers
  // were spilled to scratch, whose reloads wait for the next tile's prefetch (vmcnt is in order).
  // (Not the stream and bin pointers: through LDS they would be generic pointers and their loads
  // and stores FLAT operations, which every LDS wait would then wait for as well.)
  struct Cold { StreamArgs a; InsertSink<W, ONECOL> isink; };
  __shared__ Cold cold;
  if (threadIdx.x == 0) { cold.a = a_arg; cold.isink = isink_arg; }
  __syncthreads();  // (also for the blocks that have no tile: they read the counter pointers below)
  const StreamArgs &a = cold.a;   // pos_lo, pos_hi, ctr, flag
  const BinOut &out = out_arg;
  const InsertSink<W, ONECOL> &isink = cold.isink;
  const uint64_t a_tile0 = a_arg.tile0, a_ntiles = a_arg.ntiles;
  const uint32_t t_lb1 = isink_arg.t.lb1, t_lbq = isink_arg.t.lb1 + isink_arg.t.lbo, t_part = isink_arg.t.part;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  using LDS = BinLds<W, NB, FULL, Geo<T, T * kPosPerLane>>;
  LDS &L = *reinterpret_cast<LDS *>(dyn_lds);

  const int tid0 = threadIdx.x & (kThreads - 1);
  const uint64_t half0 = (uint64_t)(threadIdx.x / kThreads);
  const int k = a_arg.k;
  uint32_t n_kmers = 0, n_contigs = 0, n_novel = 0, full = 0;
  const uint64_t top_mask = (W == 1) ? (~0ULL >> (64 - 2 * k)) : (~0ULL >> (128 - 2 * k));
  const int first_shift = (W == 1) ? (2 * k - 2) : (2 * k - 66);
  const uint32_t ob0 = (blockIdx.x % bs.rep) * bs.nout;

  // the chunks of the NEXT tile are fetched into registers while the current one is processed
  TileSrc pre;
  pre.a = make_uint4(0, 0, 0, 0); pre.b = make_uint4(0, 0, 0, 0);
  {
    const uint64_t t0 = a_tile0 + (uint64_t)blockIdx.x * H + half0;
    if (t0 < a_ntiles) tile_fetch<PK>(a_arg, t0, tid0, pre);
  }
  MCX_PH_DECL
  for (uint64_t tile_b = a_tile0 + (uint64_t)blockIdx.x * H; tile_b < a_ntiles; tile_b += (uint64_t)gridDim.x * H) {
    // (opaque: what is derived from the thread index -- LDS addresses, masks -- is recomputed per
    // tile in an instruction or two; hoisted out of the loop it was spilled to scratch)
    const int tid_blk = (int)tid_now();               // index in the block: bins, counters
    const int tid = tid_blk & (kThreads - 1);         // index in the half: the tile's positions
    const uint64_t tile = tile_b + (uint64_t)(tid_blk / kThreads);
    const bool have_tile = H == 1 || tile < a_ntiles;  // (the last block iteration of an odd number of tiles: an empty half)
    uint32_t *s_code = s_code_h[H == 1 ? 0 : tid_blk / kThreads];
    uint32_t *s_inv = s_inv_h[H == 1 ? 0 : tid_blk / kThreads];
    MCX_PH(6) MCX_PH_COUNT
    // (No barrier here: what is written before the next one -- the tile's codes and flags, the
    // zeroed counters -- was last read before the write-out's entry barrier of the previous tile;
    // what a slower wave may still be reading, the staging area and the bin bases, is next
    // written after the three barriers of bin_reserve.)
    tile_stage<PK>(a_arg, pre, tid, s_code, s_inv);
    for (uint32_t b = tid_blk; b < bs.nlocal + 64; b += T) { L.cnt[b] = 0; L.rnk[b] = 0; }
    if (tid < 4) { s_code[kChunks + tid] = 0; s_inv[kChunks / 2 + tid] = 0xFFFFFFFFu; }
    {
      const uint64_t tn = tile + (uint64_t)gridDim.x * H;
      if (tn < a_ntiles) tile_fetch<PK>(a_arg, tn, tid, pre);
    }
    __syncthreads();
    MCX_PH(0)

    const uint32_t pl = 16u * (uint32_t)(tid + 1);
    const uint64_t Vh = inv_win64(s_inv, pl);
    const uint64_t Vl = (W == 2) ? inv_win64(s_inv, pl + 64) : 0;
    const uint32_t prev_chunk_inv = (s_inv[(pl - 1) >> 5] >> (31 - ((pl - 1) & 31))) & 1u;
    // positions owned by this launch: all 16 of every lane unless the tile straddles an end of the
    // launch's range (uniform test; the per-lane 64-bit arithmetic cost 25 instructions a tile)
    uint32_t range = have_tile ? 0xFFFFu : 0u;
    if (have_tile) {
      const uint64_t plo = a.pos_lo, phi = a.pos_hi, T0 = tile * kTile;
      if (T0 < plo || T0 + kTile > phi) {
        const uint64_t P0 = T0 + 16ull * (uint64_t)tid;
        const int j_lo = plo > P0 ? (int)min((uint64_t)kPosPerLane, plo - P0) : 0;
        const int j_hi = phi > P0 ? (int)min((uint64_t)kPosPerLane, phi - P0) : 0;
        range = ((0x10000u >> j_lo) - 1u) & ~((0x10000u >> j_hi) - 1u);
      }
    }
    // The lane's 16 positions as 16-bit masks, position j at bit 15 - j (base i of the lane's
    // window = bit 63 - i of Vh, then Vl):
    //   ok16   a whole k-mer of valid bases starts at j, and j is owned by this launch
    //   nok16  the base after that k-mer is valid (there is a successor)
    //   pok16  the base before j is valid (there is a predecessor)
    // "any invalid base in [j, j+k)" for all j at once = OR of k left-shifted copies of the
    // invalid flags, built by doubling.
    uint32_t ok16, nok16, pok16;
    {
      uint64_t Mh = Vh, Ml = Vl;
      for (int c = 1; c < k;) {  // uniform
        const int s = min(c, k - c);
        if (W == 2) Mh |= (Mh << s) | (Ml >> (64 - s));
        else Mh |= Mh << s;
        if (W == 2) Ml |= Ml << s;
        c += s;
      }
      ok16 = ~(uint32_t)(Mh >> 48) & range;
      // base j + k: bit 127 - k - j of Vh:Vl -> field of 16 starting at bit 112 - k
      const int sh = 112 - k;  // 49..109
      const uint64_t nx = sh >= 64 ? (Vh >> (sh - 64)) : ((Vh << (64 - sh)) | (W == 2 ? (Vl >> sh) : 0));
      nok16 = ~(uint32_t)nx & 0xFFFFu;
      pok16 = ~((prev_chunk_inv << 15) | (uint32_t)(Vh >> 49)) & 0xFFFFu;
    }

    // One pass over the lane's 16 positions: the tuples stay in registers (static indices
    // after unrolling) while the block histograms, reserves and then places them.  No branch
    // depends on whether a position holds a k-mer: a position without one (or whose key belongs
    // to another shard) is "binned" into a trash bin after the last real one -- it is counted,
    // ranked and placed like the others, at sorted positions beyond the tile's real tuples, which
    // the write-out never reaches.  (With `if (valid)` around every step a position cost 8 more
    // VALU instructions and five scalar ones, and the LDS atomics of the ranking were waited for
    // one by one.)
    Kmer<W> tk[kPosPerLane];    // FULL: canonical key; packed: quotient | edges << 56
    uint32_t tle[kPosPerLane];  // local bin | sorted position << 12 | edge byte << 24 (FULL)
    // one trash bin per lane (no same-address LDS atomics), never [nlocal] itself: off[nlocal] is the tile's total
    const uint32_t trash = bs.nlocal + 1u + ((uint32_t)tid_blk & 31u);
    n_kmers += __popc(ok16);
    n_contigs += __popc(ok16 & ~pok16);
    {
      // base before position j: the last base of the previous chunk, then the lane's own codes
      // (s_code[pl >> 4] holds the 16 bases of the lane, first base on top): constant shifts
      const uint32_t own = s_code[pl >> 4];
      const uint32_t before = s_code[(pl - 1) >> 4] & 3u;
      const uint32_t lbq = t_lbq;
      const uint32_t qmask = (1u << lbq) - 1u, lmask = (1u << t_lb1) - 1u;
      Kmer<W> fw, rc;
      uint64_t feed;
      LaneWin lw;
      if (W == 1) {
        lw = lane_win(s_code, pl, k);
        feed = code_win64(s_code, pl + (uint32_t)k);
      } else {
        const uint64_t hi = code_win64(s_code, pl), lo = code_win64(s_code, pl + 32);
        const int s = 128 - 2 * k;
        fw.w[0] = hi >> s;
        fw.w[W - 1] = (lo >> s) | (hi << (64 - s));
        rc = revcomp<W>(fw, k);
        feed = code_win64(s_code, pl + (uint32_t)k);
      }
      const uint32_t feed32 = (uint32_t)(feed >> 32);  // the bases after the lane's 16 k-mers
      uint32_t arr_prev = 0;
      const uint32_t key_sh = 64u - 2u * (uint32_t)k;  // one-word keys: left-aligned -> key
      const uint32_t mix_sh = (32u - lbq) & 31u;
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++) {
        const uint32_t prev_nuc = j == 0 ? before : ((own >> (32 - 2 * j)) & 3u);
        const uint32_t nuc_next = (feed32 >> (30 - 2 * j)) & 3u;
        // valid = all ones where a k-mer starts at j
        const uint32_t valid = 0u - ((ok16 >> (15 - j)) & 1u);
        const uint32_t nb = (nok16 >> (15 - j)) & 1u, pb = (pok16 >> (15 - j)) & 1u;
        uint32_t o, local;
        Kmer<W> key;
        uint64_t sel = 0;  // one-word keys: the canonical strand, left-aligned
        if (W == 1) {      // windows of the lane's 96-bit registers: constant shifts, no chain
          const uint32_t fh = j ? __builtin_amdgcn_alignbit(lw.a2, lw.a1, 32 - 2 * j) : lw.a2;
          const uint32_t fl = j ? __builtin_amdgcn_alignbit(lw.a1, lw.a0, 32 - 2 * j) : lw.a1;
          const uint32_t rh = j < 15 ? __builtin_amdgcn_alignbit(lw.r2, lw.r1, 2 + 2 * j) : lw.r2;
          const uint32_t rl = j < 15 ? __builtin_amdgcn_alignbit(lw.r1, lw.r0, 2 + 2 * j) : lw.r1;
          const uint64_t f = ((uint64_t)fh << 32) | fl, r = ((uint64_t)rh << 32) | rl;
          o = f < r ? 0u : 1u;
          sel = f < r ? f : r;
          key.w[0] = sel >> key_sh;
        } else {
          key = canonical<W>(fw, rc, o);
        }
        // edge byte: successor bit nuc_next + 4 o, predecessor bit (3 - prev_nuc) + 4 (1 - o)
        const uint32_t o4 = o << 2;
        const uint32_t e = (nb << (nuc_next | o4)) | (pb << ((prev_nuc ^ 7u) ^ o4));
        if (FULL) {
          uint32_t h2;
          kmer_hash<W>(key, 0, &h2);
          local = owner_of(h2, bs.nparts);
          tk[j] = key;
          local = (local & valid) | (trash & ~valid);
          tle[j] = local | (e << 24);
        } else {
          uint32_t G;
          Kmer<W> q;
          if (W == 1) {  // key, then quotient and remainder, from the left-aligned strand
            const uint64_t kk = sel >> key_sh;
            q.w[0] = kk >> lbq;
            G = ((uint32_t)kk ^ (region_mix<W>(q) >> mix_sh)) & qmask;  // = r ^ mix_g(): (owner, region)
          } else {
            uint32_t r;
            q = key_quot<W>(key, lbq, r);
            G = r ^ mix_g(region_mix<W>(q), lbq);
          }
          local = SH == 1 ? (G & lmask) : G;  // (SH 0: no owner bits; SH 2: bins of every shard)
          tk[j] = tuple_pack<W>(q, e);
          if (SH == 1 && valid && (G >> t_lb1) != t_part) {
            foreign_insert<W, ONECOL>(isink, key, e, n_novel, full);
            local = trash;  // not binned
          }
          local = (local & valid) | (trash & ~valid);
          tle[j] = local;
        }
        // The counting atomic returns the tuple's arrival index in its bin: with the bin's offset that
        // IS its sorted position, so the ranking needs no second atomic per tuple.  The result is
        // folded into tle one position later: its LDS round trip overlaps the next position's work.
        const uint32_t arr_now = atomicAdd(&L.cnt[local], 1u);
        if (j > 0) { tle[j - 1] |= arr_prev << 12; asm volatile("" : "+v"(tle[j - 1])); }
        arr_prev = arr_now;
        if (j == kPosPerLane - 1) tle[j] |= arr_now << 12;
        // one position at a time: VALU work gains nothing from interleaving positions, and their
        // temporaries together pushed tuples out to scratch
        if (W == 1) asm volatile("" : "+v"(tle[j]), "+v"(tk[j].w[0]));
        else asm volatile("" : "+v"(tle[j]), "+v"(tk[j].w[0]), "+v"(tk[j].w[W - 1]));
        if (W == 2) {
          fw.w[0] = ((fw.w[0] << 2) | (fw.w[W - 1] >> 62)) & top_mask;
          fw.w[W - 1] = (fw.w[W - 1] << 2) | nuc_next;
          rc.w[W - 1] = (rc.w[W - 1] >> 2) | (rc.w[0] << 62);
          rc.w[0] = (rc.w[0] >> 2) | ((uint64_t)(3u - nuc_next) << first_shift);
        }
      }
    }
    MCX_PH(1)
    BinRes<NB, T> res;
    bin_reserve<LDS, NB>(L, bs, out, ob0, res, !FULL);
    MCX_PH(2)
#pragma unroll
    for (int j = 0; j < kPosPerLane; j++) {  // sorted position goes into bits 12..24 of tle (FULL: 12..23)
      tle[j] += L.off[tle[j] & 0xfffu] << 12;  // arrival index -> sorted position
      // four returning atomics in flight, then their results are folded into tle (the opaque
      // statement also keeps the compiler from holding on to the bin index for the placement:
      // it spilled sixteen of them)
      if ((j & 3) == 3) asm volatile("" : "+v"(tle[j - 3]), "+v"(tle[j - 2]), "+v"(tle[j - 1]), "+v"(tle[j]));
    }
    bin_commit<LDS, NB>(L, bs, out, ob0, res);
    MCX_PH(3)
    for (int round = 0; round < kRounds; round++) {
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++)
        bin_place<W, FULL, LDS>(L, round, FULL ? (tle[j] >> 12) & 0xfffu : tle[j] >> 12, tle[j] & 0xfffu, tk[j], tle[j] >> 24);
      bin_writeout<W, ONECOL, FULL, SH, LDS>(L, round, bs, out, ob0, 0, isink, n_novel, full);
      if (round == 0) { MCX_PH(4) } else { MCX_PH(5) }
    }
  }
  MCX_PH_DUMP(0)

  block_add(&a.ctr->kmers, n_kmers);
  block_add(&a.ctr->contigs, n_contigs);
  block_add(&a.ctr->novel, n_novel);
  if (full == 1) a.ctr->full = 1;
  if (full == 2) a.ctr->bin_over = 1;
  if (a.flag && n_contigs) *a.flag = 1;
}

// ---------------------------------------------------------------------------
// 2. tuples -> bins.
//    IN_FULL  (tuples received from other GPUs: keys + edge bytes) -> region bins, BIN_GROUP
//    !IN_FULL (packed tuples of the region bins)                    -> sub-table bins, BIN_SUBLOCAL
//    Output is always packed.
// ---------------------------------------------------------------------------
struct TupleIn {
  const uint64_t *keys;              // [nseg][seg_cap][W]
  const uint8_t *edges;              // [nseg][seg_cap] (IN_FULL only)
  const unsigned long long *counts;  // [nseg] or nullptr (every segment holds seg_cap tuples)
  uint64_t seg_cap;
  uint32_t nseg;
  // segment s sits at physical index (s / seg_group) * seg_stride + s % seg_group: a group of
  // regions inside the replica-major L1 bins (seg_group == seg_stride: plain [nseg] array)
  uint32_t seg_group, seg_stride;
  // L1 bin sets (several colours pending at once, mcx_api.hip: "L1 bin sets"): the replica index
  // q = s / seg_group runs over the replicas of the sets a colour holds, set_rep replicas each;
  // its physical replica is set_map[q / set_rep] * set_rep + q % set_rep.  set_rep == 0: no sets.
  uint32_t set_rep;
  uint8_t set_map[32];
};
// physical index of segment s
__device__ __forceinline__ uint64_t tuple_seg_phys(const TupleIn &in, uint32_t seg)
{
  uint32_t q = seg / in.seg_group;
  if (in.set_rep) q = (uint32_t)in.set_map[q / in.set_rep] * in.set_rep + q % in.set_rep;
  return (uint64_t)q * in.seg_stride + seg % in.seg_group;
}

// T threads x 16 tuples per tile: 256 (4 blocks per CU), or 512 (2 blocks per CU, one-word keys; the launch
// bound is waves per SIMD, 4 either way): runs of
// 16 tuples = 128 bytes per sub-table bin and tile instead of 8.
// (R staging rounds, P tuples per lane.  Tried for two-word tuples in round 4, C4, isolated split 32.6 ms: 512 x 16 in
// four rounds: 109 ms, 432 bytes of scratch per lane; 512 x 8 -- 16 waves per CU instead of 8 -- 31.9 ms: that split
// moves its 140 GB at 4.4 TB/s either way.  Neither is launched.)
template <int W, bool ONECOL, int NB, bool IN_FULL, bool SHARD, int T = kThreads, int R = kRounds, int P = 16>
__global__ __launch_bounds__(T, (W == 1 || T > kThreads ? 4 : 3)) void k_tuples_bin(TupleIn in, BinSpec bs, BinOut out,
                                                                         InsertSink<W, ONECOL> isink_arg, Counters *ctr)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  using LDS = BinLds<W, NB, false, Geo<T, T * P, R>>;
  constexpr int kThreads = T, kTile = T * P;  // (shadow the 256 x 16 geometry of the other kernels)
  LDS &L = *reinterpret_cast<LDS *>(dyn_lds);
  // (as in k_stream_bin: the table's description is read from LDS by the rare paths, so that its
  // twenty arguments do not occupy scalar registers through the tile loop)
  __shared__ InsertSink<W, ONECOL> isink;
  if (threadIdx.x == 0) isink = isink_arg;
  __syncthreads();
  const uint32_t t_spb = isink_arg.t.spb, t_lb1 = isink_arg.t.lb1, t_lbq = isink_arg.t.lb1 + isink_arg.t.lbo, t_part = isink_arg.t.part;
  const int tid = threadIdx.x;
  uint32_t n_novel = 0, full = 0;
  const uint64_t chunks_per_seg = (in.seg_cap + kTile - 1) / kTile;
  // Work order.  Blocks that run at the same time must not all split the same region: they
  // would reserve from the same 64 counter lines (~12 ns per same-line atomic).  So chunks are
  // taken segment-interleaved, and -- XCD-aware -- block x only takes regions b with
  // b % 8 == x % 8: with the observed block -> XCD round-robin every sub-table bin is then
  // written from one XCD, whose L2 merges the 8-tuple runs into full lines (speed only).
  const bool xcd = bs.mode == BIN_SUBLOCAL && gridDim.x % 8 == 0 && bs.seg_mod % 8 == 0 && in.nseg % bs.seg_mod == 0;
  const uint32_t group = xcd ? blockIdx.x % 8 : 0;
  const uint32_t nseg_g = xcd ? in.nseg / 8 : in.nseg;  // segments this block may take
  // ... and within an XCD only kWin regions are split at a time, so that their 512 write fronts
  // each (64 B) stay resident in that XCD's L2 until the runs have filled whole lines.  Measured
  // (C2, ms per 12 G occurrences, two runs each): 16 regions 51.7; 8: 44.7 / 43.8; 4: 45.6 / 45.4;
  // 2: 42.5 / 41.3; 1: 43.7 / 41.7.
  constexpr uint32_t kWin = 2;
  const uint32_t bins_g = xcd ? bs.seg_mod / 8 : 1;      // regions of this group
  const uint32_t reps = xcd ? in.nseg / bs.seg_mod : 1;  // replicas per region
  const uint32_t win_segs = kWin * reps;
  const uint32_t nwin = (bins_g + kWin - 1) / kWin;
  const uint64_t nchunks = xcd ? (uint64_t)nwin * chunks_per_seg * win_segs : chunks_per_seg * nseg_g;
  const uint64_t v0 = xcd ? blockIdx.x / 8 : blockIdx.x, vstep = xcd ? gridDim.x / 8 : gridDim.x;
  const uint32_t lmask = (1u << t_lb1) - 1u;
  MCX_PH_DECL
  for (uint64_t v = v0; v < nchunks; v += vstep) {
    uint32_t seg;
    uint64_t start;
    if (xcd) {
      const uint64_t per_win = chunks_per_seg * win_segs;
      const uint32_t win = (uint32_t)(v / per_win);
      const uint64_t rem = v % per_win;
      const uint32_t s = (uint32_t)(rem % win_segs);  // segment-interleaved inside the window
      const uint32_t rgi = win * kWin + s % kWin;      // region index inside this XCD's group
      if (rgi >= bins_g) continue;                     // uniform
      seg = (s / kWin) * bs.seg_mod + group + 8 * rgi; // replica-major segment index
      start = (rem / win_segs) * kTile;
    } else {
      seg = (uint32_t)(v % nseg_g);  // segment-interleaved
      start = (v / nseg_g) * kTile;
    }
    const uint64_t pseg = tuple_seg_phys(in, seg);
    uint64_t cnt = in.counts ? (uint64_t)in.counts[pseg] : in.seg_cap;
    if (cnt > in.seg_cap) cnt = in.seg_cap;
    if (start >= cnt) continue;  // uniform across the block
    const uint32_t n = (uint32_t)min((uint64_t)kTile, cnt - start);
    const uint32_t lregion = bs.mode == BIN_SUBLOCAL ? seg % bs.seg_mod : 0;  // block-uniform
    const uint32_t region = bs.region0 + lregion;
    const uint32_t ob0 = (bs.mode == BIN_SUBLOCAL ? lregion * t_spb : 0) + (blockIdx.x % bs.rep) * bs.nout;
    MCX_PH(6) MCX_PH_COUNT
    __syncthreads();
    for (uint32_t b = tid; b < bs.nlocal + 64; b += kThreads) { L.cnt[b] = 0; L.rnk[b] = 0; }
    __syncthreads();
    MCX_PH(0)
    const uint32_t trash = bs.nlocal + 1u + ((uint32_t)tid & 31u);
    const uint64_t *kin = in.keys + (pseg * in.seg_cap + start) * W;
    const uint8_t *ein = IN_FULL ? in.edges + pseg * in.seg_cap + start : nullptr;
    // each lane keeps its kTile/kThreads tuples in registers: all loads are issued up front
    // (memory-level parallelism) and the placement sweep does not re-read HBM
    constexpr int PER = kTile / kThreads;
    Kmer<W> tk[PER];
    uint32_t ev[IN_FULL ? PER : 1], loc[PER];
    uint32_t okm = 0;  // tuples of this lane that are binned
    // 16-byte loads when the segments are 16-byte aligned (one-word tuples: a lane takes two
    // neighbours; which lane holds which tuple of the tile is irrelevant to the partition)
    const bool vec16 = !IN_FULL && (((uintptr_t)in.keys & 15u) == 0) && (W == 2 || (in.seg_cap & 1u) == 0);
    if (vec16) {
      const ulonglong2 *kin2 = reinterpret_cast<const ulonglong2 *>(kin);
#pragma unroll
      for (int q = 0; q < PER; q += (W == 1 ? 2 : 1)) {
        const uint32_t p = (uint32_t)(q / (W == 1 ? 2 : 1)) * kThreads + tid;
        const uint32_t i = W == 1 ? 2 * p : p;
        tk[q].w[0] = 0; tk[q + (W == 1 ? 1 : 0)].w[W - 1] = 0;
        if (i < n) {
          const ulonglong2 x = kin2[p];
          okm |= 1u << q;
          if (W == 1) {
            tk[q].w[0] = x.x;
            tk[q + (W == 1 ? 1 : 0)].w[0] = x.y;
            if (i + 1 < n) okm |= 2u << q;
          } else {
            tk[q].w[0] = x.x;
            tk[q].w[W - 1] = x.y;
          }
        }
      }
    } else {
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const uint32_t i = (uint32_t)q * kThreads + tid;
      if (i < n) okm |= 1u << q;
      tk[q].w[0] = 0; if (W == 2) tk[q].w[W - 1] = 0;
      if (IN_FULL) ev[q] = 0;
      if (i < n) {
        tk[q].w[0] = kin[(uint64_t)i * W];
        if (W == 2) tk[q].w[W - 1] = kin[(uint64_t)i * W + 1];
        if (IN_FULL) ev[q] = ein[i];
      }
    }
    }
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const uint32_t i = (uint32_t)q * kThreads + tid;
      uint32_t hb;
      if (IN_FULL) {  // full key -> region, packed tuple
        uint32_t r;
        const uint32_t lbq = t_lbq;
        const Kmer<W> key = tk[q];
        const Kmer<W> qq = key_quot<W>(key, lbq, r);
        const uint32_t G = r ^ mix_g(region_mix<W>(qq), lbq);
        hb = 0;
        loc[q] = G & lmask;
        tk[q] = tuple_pack<W>(qq, ev[q]);
        if (SHARD && (okm >> q & 1u) && (G >> t_lb1) != t_part) {
          foreign_insert<W, ONECOL>(isink, key, ev[q], n_novel, full);
          okm &= ~(1u << q);
        }
      } else {        // packed tuple of a known region -> sub-table inside the region
        hb = sub_hash<W>(tuple_q<W>(tk[q]));
        loc[q] = __umulhi(hb, t_spb);
      }
      if (loc[q] >= bs.nlocal) loc[q] = bs.nlocal - 1;  // cannot happen for well-formed bins
      // no branches on "this lane holds a tuple" (the tail of a segment, foreign keys): such a
      // slot goes to the lane's trash bin, whose sorted positions lie beyond the tile
      loc[q] = (okm >> q & 1u) ? loc[q] : trash;
      atomicAdd(&L.cnt[loc[q]], 1u);
      (void)i;
    }
    MCX_PH(1)
    BinRes<NB, T> res;
    bin_reserve<LDS, NB>(L, bs, out, ob0, res, true);
    MCX_PH(2)
#pragma unroll
    for (int q = 0; q < PER; q++)  // sorted position goes into the high half of loc
      loc[q] |= bin_rank<LDS>(L, loc[q]) << 16;
    bin_commit<LDS, NB>(L, bs, out, ob0, res);
    MCX_PH(3)
    for (int round = 0; round < R; round++) {
#pragma unroll
      for (int q = 0; q < PER; q++)
        bin_place<W, false, LDS>(L, round, loc[q] >> 16, loc[q] & 0xffffu, tk[q], 0);
      bin_writeout<W, ONECOL, false, 0, LDS>(L, round, bs, out, ob0, region, isink, n_novel, full);
      if (round == 0) { MCX_PH(4) } else { MCX_PH(5) }
    }
  }
  MCX_PH_DUMP(1)
  block_add(&ctr->novel, n_novel);
  if (full == 1) ctr->full = 1;
  if (full == 2) ctr->bin_over = 1;
}

// ---------------------------------------------------------------------------
// 3. LDS insert: one workgroup owns one sub-table
// ---------------------------------------------------------------------------
// The slice is held in LDS as Sub<W>::kSlots x (W key words + this colour's value word); other
// colours' value words stay untouched in HBM.  find-or-insert / coverage / edges are the same
// protocol as probe_insert, with LDS atomics.
// Geometry of the LDS-insert workgroup: 512 threads, two workgroups per CU (4 waves per SIMD); the next
// sub-table's slice is fetched into registers while the current one is applied; every thread keeps two
// batches of kBatch tuple loads in flight.  (Rejected with measurements, profiles/r03_experiments.md:
// 1 or 3 workgroups per CU, 3-6 batches in flight, 2 or 8 tuples per batch, 16-byte tuple loads, 4096-slot
// two-word sub-tables with 1024-thread workgroups, requesting the next sub-table's tuples one sub-table ahead.)
template <int W> struct LdsCfg {
  static constexpr int kThreads = 512;
  static constexpr int kMinWaves = 4;
  static constexpr bool kPrefetch = true;
  static constexpr int kBatch = 4;
};
// tuples of the LDS queue: what is left of a CU's 160 KiB beside two 64 KiB (three 48 KiB) slices
template <int W> struct LdsQueue { static constexpr uint32_t kTuples = W == 1 ? 1856 : 288; };

// LDS image of a sub-table.
// One-word keys: the keys of all slots first (32 KiB), then the values (32 KiB).  A bucket's four
// keys are 32 contiguous bytes; its two halves (slots 0-1, slots 2-3) are swapped in buckets with
// bit 3 set, so that the 16-byte read of "slots 0-1" of a random bucket can land on any of the 16
// positions of a 256-byte LDS row (tools/ubench_lds.hip: both halves of a random bucket cost 23.6
// LDS cycles per wave this way, 47.3 with the 64-byte bucket image keys | values that was here
// before -- its key reads could only start at 4 positions -- and 6.6 for a single 8-byte read).
// Slots fill in probe order, so at the load factors a graph is built with (<= 0.75) most keys sit in
// slots 0-1 of their bucket: the second half is only read by the lanes that need it.
// Two-word keys: slot after slot (key word 0, key word 1, value): four strided 8-byte loads per probe.
// (A plane image for two-word keys as well -- first words | second words | values, two 16-byte loads per
// probe -- measured 39.3 ms against 38.0 at C4 in round 3, profiles/r03_experiments.md: the two-word insert
// is not bound by its LDS reads.)
__device__ __forceinline__ uint32_t lds_phys1(uint32_t slot)  // position of logical slot `slot` in a plane
{
  return slot ^ ((slot >> 4) & 2u);  // bit 1 (which half) ^= bit 5 of the slot (= bit 3 of the bucket)
}
constexpr uint32_t kLdsVal1 = 4096;  // word offset of the values in the one-word image

// find-or-insert one occurrence in the LDS-resident sub-table.  Half a bucket (one-word keys) or a
// whole one (two-word keys) is examined per step: the key words are loaded together and compared
// in registers, so nearly every lane is done after one step (walking slot by slot made the wave run
// as many iterations as its unluckiest lane).  Slots fill in probe order and never empty, so "first
// empty slot of the snapshot" + CAS keeps a key from ever being stored twice: a failed CAS re-reads.
// Returns false when the sub-table is full and does not hold the key: the occurrence then belongs
// to the overflow area (mcx_kernels.h, ovf_start), which the caller updates in HBM.
// (LDS-typed pointers: through a generic pointer a volatile vector load becomes a FLAT load followed
// by s_waitcnt vmcnt(0), which also waits for every global load in flight.)
template <int W>
__device__ __forceinline__ bool lds_apply(unsigned long long *lds, const Kmer<W> &key, uint32_t bucket, uint32_t e,
                                          uint32_t &n_novel, uint32_t &full)
{
  const unsigned long long want = key.w[0] | kFlag;
  uint32_t steps = 0;
  if constexpr (W == 1) {
    // A whole bucket per step (both 16-byte halves of its keys, as in lds_try).  Until round 4 this loop looked at two
    // slots per step: at high load factors (hashtest: 800 M keys into 2^30 slots; C2-stress) the general loop is where
    // the insert spends its time -- 66 % / 49 % of a sub-table visit -- and every step is a dependent LDS round trip.
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    typedef MCX_LDS_AS const volatile u64x2 lds_cv;  // (volatile: re-read on every step)
    uint32_t b = bucket;
    for (;;) {
      const uint32_t sw = (b >> 3) & 1u;
      unsigned long long *b0 = lds + b * kBucket;
      unsigned long long *h0 = b0 + 2 * sw, *h1 = b0 + 2 * (sw ^ 1u);  // logical slots 0-1, 2-3
      const u64x2 lo = *(lds_cv *)h0;
      const u64x2 hi = *(lds_cv *)h1;
      unsigned long long *hit = nullptr, *emp = nullptr;
      if (hi.y == 0) emp = h1 + 1;
      if (hi.x == 0) emp = h1;
      if (lo.y == 0) emp = h0 + 1;
      if (lo.x == 0) emp = h0;  // (the first empty slot in probe order)
      if (hi.y == want) hit = h1 + 1;
      if (hi.x == want) hit = h1;
      if (lo.y == want) hit = h0 + 1;
      if (lo.x == want) hit = h0;
      if (hit) {
        unsigned long long *val = hit + kLdsVal1;
        const unsigned long long old = atomicAdd(val, 256ULL);
        if (e & ~(uint32_t)old) atomicOr(val, (unsigned long long)e);
        return true;
      }
      if (emp) {
        if (atomicCAS(emp, 0ULL, want) == 0) {
          unsigned long long *val = emp + kLdsVal1;
          n_novel++;
          atomicAdd(val, 256ULL);
          if (e) atomicOr(val, (unsigned long long)e);
          return true;
        }
        if (++steps > (1u << 22)) { full = 1; return true; }
        continue;  // somebody took the slot: look at the bucket again
      }
      b = (b + 1) & (uint32_t)(Sub<W>::kBuckets - 1);
      if (b == bucket) return false;  // a full sub-table: every slot seen without a hit or a free one
    }
  } else {
    // slot j of bucket b: first key word at sp(j), the second kK1 words further on, the value kV
    constexpr uint32_t kK1 = 1, kV = W;
#define MCX_SP(b_, j_) (lds + (size_t)(b_) * (kBucket * (W + 1)) + (j_) * (W + 1))
    uint32_t b = bucket;
    for (;;) {
      unsigned long long k[kBucket];
#pragma unroll
      for (int j = 0; j < kBucket; j++)
        k[j] = __hip_atomic_load((MCX_LDS_AS unsigned long long *)MCX_SP(b, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      int hit = -1, empty = -1;
      bool retry = false;
#pragma unroll
      for (int j = kBucket - 1; j >= 0; j--) {
        if (k[j] == 0) empty = j;
        if ((k[j] & ~kPending) == want) {
          if (k[j] & kPending) retry = true;  // its owner has not published word 1 yet
          else if (__hip_atomic_load((MCX_LDS_AS unsigned long long *)(MCX_SP(b, j) + kK1), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == key.w[W - 1]) hit = j;
        }
      }
      if (hit >= 0) {
        unsigned long long *val = MCX_SP(b, hit) + kV;
        const unsigned long long old = atomicAdd(val, 256ULL);
        if (e & ~(uint32_t)old) atomicOr(val, (unsigned long long)e);
        return true;
      }
      if (retry) {
        if (++steps > (1u << 22)) { full = 1; return true; }
        continue;
      }
      if (empty >= 0) {
        unsigned long long *r = MCX_SP(b, empty);
        if (atomicCAS(r, 0ULL, want | kPending) == 0) {
          __hip_atomic_store(r + kK1, (unsigned long long)key.w[W - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(r, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          n_novel++;
          atomicAdd(r + kV, 256ULL);
          if (e) atomicOr(r + kV, (unsigned long long)e);
          return true;
        }
        if (++steps > (1u << 22)) { full = 1; return true; }
        continue;  // somebody took the slot: look at the bucket again
      }
      if (++steps > Sub<W>::kBuckets + (1u << 22)) { full = 1; return true; }
      b = (b + 1) & (Sub<W>::kBuckets - 1);
      if (b == bucket) return false;  // a full sub-table: every bucket seen without a hit or a free slot
    }
#undef MCX_SP
  }
}

// One straight-line probe of the start bucket: the common cases -- the key is there, or it is new
// and the bucket has a free slot -- without a loop.  Returns false for everything else (bucket full
// of other keys, slot lost to another lane, two-word key still pending): those occurrences are set
// aside and finished with lds_apply by densely packed lanes.  Why: the insert kernel is bound by
// instruction issue (~150 instructions per occurrence), and in a loop a wave pays a whole extra
// iteration whenever ONE of its 64 lanes has to move on to the next bucket (3 % of the lanes at
// load 0.3: 86 % of the waves).
template <int W>
__device__ __forceinline__ bool lds_try(unsigned long long *lds, const Kmer<W> &key, uint32_t bucket, uint32_t e, uint32_t &n_novel)
{
  const unsigned long long want = key.w[0] | kFlag;
  unsigned long long k[kBucket];
  unsigned long long *kp[kBucket];  // key word 0 of logical slot j; its value is kVal words further on
  constexpr bool kPlanes = W == 1;  // one-word keys: key words and values in planes, buckets half-swizzled
  constexpr uint32_t kVal = W == 1 ? kLdsVal1 : (uint32_t)W;
  constexpr uint32_t kK1 = 1u;  // second key word of a two-word key
  if constexpr (kPlanes) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    typedef MCX_LDS_AS const u64x2 lds_c;
    const uint32_t sw = (bucket >> 3) & 1u;
    unsigned long long *b0 = lds + bucket * kBucket;
    unsigned long long *h0 = b0 + 2 * sw, *h1 = b0 + 2 * (sw ^ 1u);
    const u64x2 a = *(lds_c *)h0;
    const u64x2 b = *(lds_c *)h1;
    k[0] = a.x; k[1] = a.y; k[2] = b.x; k[3] = b.y;
    kp[0] = h0; kp[1] = h0 + 1; kp[2] = h1; kp[3] = h1 + 1;
  } else {
    constexpr int R = W + 1;
    unsigned long long *bp = lds + (size_t)bucket * (kBucket * R);
#pragma unroll
    for (int j = 0; j < kBucket; j++) {
      kp[j] = bp + j * R;
      k[j] = __hip_atomic_load((MCX_LDS_AS unsigned long long *)kp[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  unsigned long long *hk = nullptr;
#pragma unroll
  for (int j = kBucket - 1; j >= 0; j--)
    if (k[j] == want) hk = kp[j];  // (a pending two-word key differs in kPending: no hit)
  if (hk) {
    if (W == 2 && __hip_atomic_load((MCX_LDS_AS unsigned long long *)(hk + kK1), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != key.w[W - 1])
      return false;  // same first word, other second word: the key may still sit further down
    const unsigned long long old = atomicAdd(hk + kVal, 256ULL);
    if (e & ~(uint32_t)old) atomicOr(hk + kVal, (unsigned long long)e);
    return true;
  }
  unsigned long long *ek = nullptr;
#pragma unroll
  for (int j = kBucket - 1; j >= 0; j--) {
    if (k[j] == 0) ek = kp[j];
    if (W == 2 && (k[j] & ~kPending) == want) return false;  // could be this key, not yet published
  }
  if (!ek) return false;
  if (atomicCAS(ek, 0ULL, W == 1 ? want : (want | kPending)) != 0) return false;
  if (W == 2) {
    __hip_atomic_store(ek + kK1, (unsigned long long)key.w[W - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(ek, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  n_novel++;
  atomicAdd(ek + kVal, 256ULL);
  if (e) atomicOr(ek + kVal, (unsigned long long)e);
  return true;
}

// The slice of one sub-table in flight between HBM and LDS: up to 8 16-byte vectors per thread,
// as named members (an indexed array that lives across the sub-table loop is left in scratch
// memory by the compiler, which defeats the purpose of prefetching into registers).
struct SliceRegs { ulonglong2 a, b, c, d, e, f, g, h; };

// HBM -> registers.  One colour: the slice is one contiguous block of [key words, value] records,
// identical to the LDS image.  Several colours: key words and this colour's values are separate
// contiguous arrays; a thread fetches whole 16-byte vectors of both and pairs them up for LDS.
template <int W, bool ONECOL, int T>
__device__ __forceinline__ void slice_load(const TableView &t, uint32_t sub, uint32_t col, int tid, SliceRegs &v)
{
  const uint64_t s0 = (uint64_t)sub * Sub<W>::kSlots;
  if (ONECOL) {
    const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(t.rec + s0 * (W + 1));
    static_assert(Sub<W>::kSlots * (W + 1) * 8 / 16 / T == (W == 1 ? 8 : 6), "vectors per thread");
    // (element-wise: whole-member copies become memcpys of the struct, which then stays in memory)
#define MCX_LD(m, q) { const ulonglong2 x = src[q * T + tid]; v.m.x = x.x; v.m.y = x.y; }
    MCX_LD(a, 0) MCX_LD(b, 1) MCX_LD(c, 2) MCX_LD(d, 3) MCX_LD(e, 4) MCX_LD(f, 5)
    if (W == 1) { MCX_LD(g, 6) MCX_LD(h, 7) }
#undef MCX_LD
  } else if (W == 1) {  // pair p = q * T + tid covers slots 2p, 2p + 1: one key vector, one value vector
    static_assert(W != 1 || Sub<W>::kSlots / 2 == 4 * (uint64_t)T, "4 slot pairs per thread");
    const ulonglong2 *K = reinterpret_cast<const ulonglong2 *>(t.rec + s0);
    const ulonglong2 *V = reinterpret_cast<const ulonglong2 *>(t.val + (uint64_t)col * t.VC + s0);
#define MCX_LD(m, p) { const ulonglong2 x = p; v.m.x = x.x; v.m.y = x.y; }
    MCX_LD(a, K[0 * T + tid]) MCX_LD(b, K[1 * T + tid]) MCX_LD(c, K[2 * T + tid]) MCX_LD(d, K[3 * T + tid])
    MCX_LD(e, V[0 * T + tid]) MCX_LD(f, V[1 * T + tid]) MCX_LD(g, V[2 * T + tid]) MCX_LD(h, V[3 * T + tid])
#undef MCX_LD
  } else {              // pair p = j * T + tid: two key vectors (one per slot), one value vector
    static_assert(W != 2 || Sub<W>::kSlots / 2 == 2 * (uint64_t)T, "2 slot pairs per thread");
    const ulonglong2 *K = reinterpret_cast<const ulonglong2 *>(t.rec + s0 * 2);
    const ulonglong2 *V = reinterpret_cast<const ulonglong2 *>(t.val + (uint64_t)col * t.VC + s0);
#define MCX_LD(m, p) { const ulonglong2 x = p; v.m.x = x.x; v.m.y = x.y; }
    MCX_LD(a, K[2 * (0 * T + tid)]) MCX_LD(b, K[2 * (0 * T + tid) + 1]) MCX_LD(c, V[0 * T + tid])
    MCX_LD(d, K[2 * (1 * T + tid)]) MCX_LD(e, K[2 * (1 * T + tid) + 1]) MCX_LD(f, V[1 * T + tid])
#undef MCX_LD
  }
}

// The same, unless the sub-table still holds only zeros (TableView::touch): then nothing is read.
template <int W, bool ONECOL, int T>
__device__ __forceinline__ void slice_fetch(const TableView &t, uint32_t sub, uint32_t col, int tid, SliceRegs &v, bool zeros_known)
{
  if (zeros_known && !((t.touch[1 + (sub >> 5)] >> (sub & 31u)) & 1u)) {  // (uniform)
#define MCX_Z(m) { v.m.x = 0; v.m.y = 0; }
    MCX_Z(a) MCX_Z(b) MCX_Z(c) MCX_Z(d) MCX_Z(e) MCX_Z(f) MCX_Z(g) MCX_Z(h)
#undef MCX_Z
    return;
  }
  slice_load<W, ONECOL, T>(t, sub, col, tid, v);
}

// registers -> LDS image: slots of W key words + this colour's value word
template <int W, bool ONECOL, int T>
__device__ __forceinline__ void slice_to_lds(unsigned long long *lds, int tid, const SliceRegs &v)
{
  ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(lds);
  if (ONECOL && W == 1) {  // vector q * T + tid = slot (key, value) -> keys | values
#define MCX_ST(m, q) { const uint32_t ph = lds_phys1((uint32_t)(q * T + tid)); lds[ph] = v.m.x; lds[kLdsVal1 + ph] = v.m.y; }
    MCX_ST(a, 0) MCX_ST(b, 1) MCX_ST(c, 2) MCX_ST(d, 3) MCX_ST(e, 4) MCX_ST(f, 5) MCX_ST(g, 6) MCX_ST(h, 7)
#undef MCX_ST
  } else if (ONECOL) {
#define MCX_ST(m, q) dst[q * T + tid] = make_ulonglong2(v.m.x, v.m.y);
    MCX_ST(a, 0) MCX_ST(b, 1) MCX_ST(c, 2) MCX_ST(d, 3) MCX_ST(e, 4) MCX_ST(f, 5)
#undef MCX_ST
  } else if (W == 1) {  // slots 2p, 2p + 1 are one half of a bucket: their keys are one vector, their values another
#define MCX_PUT1(q, kv, vv) { const uint32_t ph = lds_phys1(2u * (uint32_t)(q * T + tid)) >> 1; dst[ph] = make_ulonglong2(kv.x, kv.y); dst[kLdsVal1 / 2 + ph] = make_ulonglong2(vv.x, vv.y); }
    MCX_PUT1(0, v.a, v.e) MCX_PUT1(1, v.b, v.f) MCX_PUT1(2, v.c, v.g) MCX_PUT1(3, v.d, v.h)
#undef MCX_PUT1
  } else {              // slots 2p, 2p + 1 = 6 words = vectors 3p .. 3p + 2: k0a k0b | v0 k1a | k1b v1
#define MCX_PUT2(j, k0, k1, vv) { const int p = j * T + tid; dst[3 * p] = make_ulonglong2(k0.x, k0.y); dst[3 * p + 1] = make_ulonglong2(vv.x, k1.x); dst[3 * p + 2] = make_ulonglong2(k1.y, vv.y); }
    MCX_PUT2(0, v.a, v.b, v.c) MCX_PUT2(1, v.d, v.e, v.f)
#undef MCX_PUT2
  }
}

// LDS image -> HBM
template <int W, bool ONECOL, int T>
__device__ __forceinline__ void slice_store(const TableView &t, uint32_t sub, uint32_t col, int tid, const unsigned long long *lds)
{
  const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(lds);
  const uint64_t s0 = (uint64_t)sub * Sub<W>::kSlots;
  if (ONECOL && W == 1) {
    ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(t.rec + s0 * 2);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t sl = (uint32_t)(q * T + tid), ph = lds_phys1(sl);
      dst[sl] = make_ulonglong2(lds[ph], lds[kLdsVal1 + ph]);
    }
  } else if (ONECOL) {
    ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(t.rec + s0 * (W + 1));
    constexpr int PER = (int)(Sub<W>::kSlots * (W + 1) * 8 / 16 / T);
#pragma unroll
    for (int q = 0; q < PER; q++) {
      dst[q * T + tid] = src[q * T + tid];
    }
  } else if (W == 1) {
    ulonglong2 *K = reinterpret_cast<ulonglong2 *>(t.rec + s0);
    ulonglong2 *V = reinterpret_cast<ulonglong2 *>(t.val + (uint64_t)col * t.VC + s0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t p = (uint32_t)(q * T + tid), ph = lds_phys1(2u * p) >> 1;
      K[p] = src[ph];
      V[p] = src[kLdsVal1 / 2 + ph];
    }
  } else {
    ulonglong2 *K = reinterpret_cast<ulonglong2 *>(t.rec + s0 * 2);
    ulonglong2 *V = reinterpret_cast<ulonglong2 *>(t.val + (uint64_t)col * t.VC + s0);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int p = j * T + tid;
      const ulonglong2 x = src[3 * p], y = src[3 * p + 1], z = src[3 * p + 2];
      K[2 * p] = x;
      K[2 * p + 1] = make_ulonglong2(y.y, z.x);
      V[p] = make_ulonglong2(y.x, z.y);
    }
  }
}

template <int W, bool ONECOL>
__global__ __launch_bounds__(LdsCfg<W>::kThreads, LdsCfg<W>::kMinWaves) void k_lds_insert(TableView t, uint32_t col, BinOut bins,
                                                                       uint32_t sub0, uint32_t nsub, Counters *ctr)
{
  constexpr int kLdsThreads = LdsCfg<W>::kThreads;
  constexpr int kLdsBatch = LdsCfg<W>::kBatch;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  unsigned long long *lds = reinterpret_cast<unsigned long long *>(dyn_lds);
  const int tid = threadIdx.x;
  uint32_t n_novel = 0, full = 0;
  // occurrences that lds_try could not finish (packed tuples, as they came): behind the slice
  unsigned long long *queue = lds + Sub<W>::kSlots * (W + 1);
  constexpr uint32_t kQueueCap = LdsQueue<W>::kTuples;
  __shared__ uint32_t s_nq;
  if (tid == 0) s_nq = 0;

  // sub-tables sub0 .. sub0 + nsub - 1; bin i of `bins` belongs to sub-table sub0 + i and is handed
  // back empty (fill reset) for the next group of regions.
  // The slice of the NEXT sub-table this block will own is fetched into registers while the
  // current one is being updated: load -> apply -> store of one workgroup would otherwise run back
  // to back, and with two workgroups per CU the HBM pipe idles through the apply phases
  // (C2-stress, 2.9 K occurrences per 64 KiB slice: 1.1 TB/s before).
  auto next_bin = [&](uint32_t from) {
    while (from < nsub && bins.counts[from] == 0) from += gridDim.x;
    return from;
  };
  uint32_t bi = next_bin(blockIdx.x);
  SliceRegs v{};
  // sub-tables that nothing has written since the table was zeroed are not read (the first flush
  // of a build reads none of the table: 16 of its 50 bytes per slot)
  const bool zeros_known = t.touch && t.touch[0] == 0;
  constexpr bool kPrefetch = LdsCfg<W>::kPrefetch;
  if (kPrefetch && bi < nsub) slice_fetch<W, ONECOL, kLdsThreads>(t, sub0 + bi, col, tid, v, zeros_known);
  // tuple j0 + idx(q) is the q-th of this thread's batch
  auto idx = [&](int q) -> uint32_t { return (uint32_t)q * kLdsThreads + tid; };
  // Loads are unconditional (an index past the fill reads tuple 0 of the bin, which is then not
  // applied): without branches between them the compiler can count the loads in flight and wait
  // for exactly the batch it is about to apply (s_waitcnt vmcnt(n)) instead of for all of them.
  auto load_batch_of = [&](uint32_t bin, uint64_t nfill, uint64_t j0, Kmer<W> (&tk)[kLdsBatch]) {
    const uint64_t *kin = bins.keys + (uint64_t)bin * bins.cap * W;
#pragma unroll
    for (int q = 0; q < kLdsBatch; q++) {
      uint64_t i = j0 + idx(q);
      i = i < nfill ? i : 0;
      tk[q].w[0] = kin[i * W];
      if (W == 2) tk[q].w[W - 1] = kin[i * W + 1];
    }
  };
  constexpr uint64_t kStep = (uint64_t)kLdsThreads * kLdsBatch;
  MCX_PH_DECL
  while (bi < nsub) {
    const uint32_t sub = sub0 + bi;
    const uint32_t region = sub / t.spb;  // uniform
    MCX_PH(6) MCX_PH_COUNT
    uint64_t n = bins.counts[bi];
    if (n > bins.cap) n = bins.cap;
    const uint32_t nb = next_bin(bi + gridDim.x);
    __syncthreads();  // every thread has read the fills; the previous slice has left LDS
    MCX_PH(0)
    if (tid == 0) bins.counts[bi] = 0;
    if (!kPrefetch) slice_fetch<W, ONECOL, kLdsThreads>(t, sub, col, tid, v, zeros_known);
    slice_to_lds<W, ONECOL, kLdsThreads>(lds, tid, v);
    __syncthreads();
    MCX_PH(1)

    auto load_batch = [&](uint64_t j0, Kmer<W> (&tk)[kLdsBatch]) { load_batch_of(bi, n, j0, tk); };
    // packed tuple + region -> full key, start bucket, edge byte
    auto unpack = [&](const Kmer<W> &tp, Kmer<W> &key, uint32_t &bucket, uint32_t &e) {
      e = (uint32_t)(tp.w[0] >> 56);
      const Kmer<W> qq = tuple_q<W>(tp);
      const uint32_t m = region_mix<W>(qq);
      key = key_unquot<W>(qq, lbq_of(t), ((t.part << t.lb1) | region) ^ mix_g(m, lbq_of(t)));
      bucket = mix_bucket(m, lbq_of(t), Sub<W>::kBuckets);
    };
    auto apply_slow = [&](const Kmer<W> &key, uint32_t bucket, uint32_t e) {
      if (!lds_apply<W>(lds, key, bucket, e, n_novel, full))  // sub-table full: overflow area, in HBM
        probe_insert<W, ONECOL>(t, key, ovf_start<W>(t, key), 0, 0, e, col, n_novel, full, true);
    };
    auto apply_batch = [&](uint64_t j0, const Kmer<W> (&tk)[kLdsBatch]) {
#pragma unroll
      for (int q = 0; q < kLdsBatch; q++)
        if (j0 + idx(q) < n) {
          Kmer<W> key;
          uint32_t bucket, e;
          unpack(tk[q], key, bucket, e);
          if (!lds_try<W>(lds, key, bucket, e, n_novel)) {
            const uint32_t qi = atomicAdd(&s_nq, 1u);
            if (qi < kQueueCap) {
              queue[qi * W] = tk[q].w[0];
              if (W == 2) queue[qi * W + 1] = tk[q].w[W - 1];
            } else {
              apply_slow(key, bucket, e);
            }
          }
        }
    };
    // Two batches in flight: while one is applied (LDS only) the loads of the next are on their way.
    // The next slice is requested after the first two batches (loads return in order, so those do
    // not queue behind its 64 KiB).  Both variants of "is there a next slice" are straight-line code.
    auto run = [&](auto has_next) {
      Kmer<W> ta[kLdsBatch], tb[kLdsBatch];
      load_batch(0, ta);
      load_batch(kStep, tb);
      if (decltype(has_next)::value) slice_fetch<W, ONECOL, kLdsThreads>(t, sub0 + nb, col, tid, v, zeros_known);
      for (uint64_t j0 = 0; j0 < n; j0 += 2 * kStep) {
        apply_batch(j0, ta);
        load_batch(j0 + 2 * kStep, ta);
        apply_batch(j0 + kStep, tb);
        load_batch(j0 + 3 * kStep, tb);
      }
    };
    if (kPrefetch && nb < nsub) run(std::true_type{}); else run(std::false_type{});
    MCX_PH(2)
    __syncthreads();
    MCX_PH(3)
    {  // the occurrences set aside: every lane takes one, all of them run the general probe loop
      const uint32_t nq = min(s_nq, kQueueCap);
      for (uint32_t i = tid; i < nq; i += kLdsThreads) {
        Kmer<W> tp, key;
        uint32_t bucket, e;
        tp.w[0] = queue[i * W];
        if (W == 2) tp.w[W - 1] = queue[i * W + 1];
        unpack(tp, key, bucket, e);
        apply_slow(key, bucket, e);
      }
      __syncthreads();
      if (tid == 0) s_nq = 0;
    }
    MCX_PH(4)
    slice_store<W, ONECOL, kLdsThreads>(t, sub, col, tid, lds);
    if (tid == 0 && t.touch) atomicOr(&t.touch[1 + (sub >> 5)], 1u << (sub & 31u));
    bi = nb;
    MCX_PH(5)
  }
  MCX_PH_DUMP(2)
  block_add(&ctr->novel, n_novel);
  if (full) ctr->full = 1;
}

}  // namespace mcx
