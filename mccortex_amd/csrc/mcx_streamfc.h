// mcx_streamfc.h -- reads -> region bins WITHOUT a sort: k_stream_fc, the round-6 variant of k_stream_bin
// for the hot case (one-word keys, unsharded table, packed tuples, <= 512 regions).
//
// k_stream_bin (mcx_defer.h) counting-sorts a tile's tuples in LDS: histogram, scan, offsets, placement in two
// rounds, write-out -- 9 LDS instructions per tuple (6 of them random), 7-8 barriers per tile, tuples held in
// registers until the tile's offsets are known (profiles/r05p_summary.md: bank-conflict ratio 0.43, waves
// parked 47 % of their cycles).  Here every region has a FIXED segment of CAP tuples in LDS instead:
//
//   k-merise:  tuple -> idx = atomicAdd(cnt[bin]) -> seg[bin][idx]      (idx < CAP; ONE random atomic + ONE
//                                                                         random 8-byte store per tuple)
//              idx >= CAP (a few % of the tuples: Poisson tail)          -> pool of (bin, tuple) entries
//   barrier
//   write-out: every wave owns 128 bins: one reservation per bin (min(cnt, CAP) tuples), then the filled
//              parts of its segments go out 64 / CAP bins per store instruction; pool entries reserve and
//              store one by one; meanwhile the next tile's codes are staged and its counters zeroed
//   barrier
//
// Two barriers per tile of 4096 positions, no scan, no offsets, no second staging round.  What it costs: LDS
// (512 x CAP x 8 bytes -- CAP = 10: 40 KiB, three blocks per CU) and runs of <= CAP tuples per bin and tile.
// A lane whose tuples found the pool full (one region taking hundreds of a tile's tuples: poly-A, satellite
// repeats) keeps them in registers and appends them in extra rounds after the write-out.
//
// MEASURED (round 6, C2, ms per 5.99 G occurrences, isolated; profiles/r06_experiments.md): 29.2-30.3 against
// 22.0-22.3 for the sort (CAP 8: 43.6, CAP 12 / 16 -- two blocks per CU -- 33.7).  Phase clocks: a block spends
// 10.7 us on a tile of 4096 positions (3 blocks per CU = 3.56 us per CU and tile; the sort: 11.3 us per 8192
// positions, 2 blocks per CU = 2.83).  The k-merising phase alone, with its 16 random 8-byte LDS stores per lane
// behind the 16 atomics, takes 2.5 us here against 1.7 us for TWICE the positions in the sort kernel: the
// scatter into LDS is bound by the LDS pipe (ds_write_b64 = 6 cycles per wave instruction), which the sort pays
// as well -- and the sort's longer runs (16 tuples) and 16 waves per CU more than pay for its scan.  With
// private replicas (PRIV: one per block, no global atomic at all, MCX_REP1=768) 33.8 ms: the reservation latency
// was not the bound, and 768 x 512 write fronts no longer fit the L2s.  NOT LAUNCHED unless MCX_STREAM_FC=1.
#pragma once
#include "mcx_defer.h"

namespace mcx {

constexpr int kFcPool = 384;  // pool entries per tile
constexpr int kFcBins = 512;

template <int CAP> struct FcLds {
  uint64_t seg[kFcBins * CAP];
  uint64_t pool_t[kFcPool];
  uint32_t pool_b[kFcPool];
  uint32_t cnt[2][kFcBins + 64];  // [parity of the tile]; [nlocal + 1 + lane % 32] = trash bins
  uint32_t code[2][kChunks + 4];
  uint32_t inv[2][kChunks / 2 + 4];
  uint32_t pool_n[2];
  uint32_t left[2];  // != 0: lanes of this tile hold tuples the pool had no room for
};

// rare path: a tuple whose bin's segment in HBM is full -> per-occurrence insert.  Returns novel | full << 30.
template <bool ONECOL>
__device__ __noinline__ uint32_t fc_direct(const InsertSink<1, ONECOL> &isink, uint32_t region, uint64_t t0)
{
  uint32_t novel = 0, full = 0;
  Kmer<1> tq;
  tq.w[0] = t0;
  const uint32_t e = (uint32_t)(t0 >> 56);
  const Kmer<1> qq = tuple_q<1>(tq);
  table_mark_written(isink.t);
  table_count_fallback(isink.t);
  const Kmer<1> key = key_unquot<1>(qq, lbq_of(isink.t), r_of<1>(isink.t, region, qq));
  const uint64_t slot = key_slot<1>(isink.t, key);
  const uint64_t cur = *key_ptr_t<1, ONECOL>(isink.t, slot);
  probe_insert<1, ONECOL>(isink.t, key, slot, cur, 0, e, isink.col, novel, full);
  return novel | (full << 30);
}

__device__ __forceinline__ uint32_t fc_acc(uint32_t acc, uint32_t r)  // novel counts add, the full flag sticks
{
  return ((acc + (r & 0x3fffffffu)) & 0x3fffffffu) | ((acc | r) & 0xC0000000u);
}

// the pool's entries: one reservation and one 8-byte store each
template <bool ONECOL, int CAP>
__device__ __forceinline__ uint32_t fc_pool_drain(FcLds<CAP> &L, uint32_t pn, const BinOut &out, uint32_t ob0,
                                                  const InsertSink<1, ONECOL> &isink)
{
  uint32_t acc = 0;
  for (uint32_t i = threadIdx.x; i < pn; i += kThreads) {
    const uint32_t b = L.pool_b[i];
    const uint64_t t0 = L.pool_t[i];
    const unsigned long long pos = atomicAdd(&out.counts[ob0 + b], 1ull);
    if (pos < out.cap) out.keys[(uint64_t)(ob0 + b) * out.cap + pos] = t0;
    else acc = fc_acc(acc, fc_direct<ONECOL>(isink, b, t0));
  }
  return acc;
}

// PRIV: every block owns replica blockIdx.x of the bins (the launch has at most bs.rep blocks): the fills live in
// registers of the lanes that own the bins, no reservation is an atomic, and a pool entry carries its arrival index.
template <bool ONECOL, bool PK, int CAP, bool PRIV>
__global__ __launch_bounds__(kThreads, 3) void k_stream_fc(StreamArgs a_arg, BinSpec bs, BinOut out_arg,
                                                           InsertSink<1, ONECOL> isink_arg)
{
  constexpr int W = 1;
  constexpr int G = 64 / CAP;                    // bins per store instruction
  constexpr int ROWS = kFcBins / kThreads;       // rows of 64 bins per wave (2)
  constexpr int ITERS = (64 + G - 1) / G;        // store instructions per row
  struct Cold { StreamArgs a; InsertSink<W, ONECOL> isink; };
  __shared__ Cold cold;
  if (threadIdx.x == 0) { cold.a = a_arg; cold.isink = isink_arg; }
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  FcLds<CAP> &L = *reinterpret_cast<FcLds<CAP> *>(dyn_lds);
  const StreamArgs &a = cold.a;
  const BinOut &out = out_arg;
  const InsertSink<W, ONECOL> &isink = cold.isink;
  const uint64_t a_tile0 = a_arg.tile0, a_ntiles = a_arg.ntiles;
  const uint32_t t_lbq = isink_arg.t.lb1 + isink_arg.t.lbo;
  const int k = a_arg.k;
  uint32_t n_kmers = 0, n_contigs = 0, acc = 0;  // acc: novel | full << 30 of the rare inserts
  const uint32_t ob0 = (blockIdx.x % bs.rep) * bs.nout;
  const uint32_t nlocal = bs.nlocal;
  uint32_t fill[ROWS];  // PRIV: fill of the lane's own bins (this block's replica)
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const uint32_t b = (threadIdx.x >> 6) * (64u * ROWS) + 64u * r + (threadIdx.x & 63u);
    fill[r] = PRIV && b < nlocal ? (uint32_t)min(out_arg.counts[ob0 + b], 0xFFFF0000ull) : 0u;
  }

  // prologue: the first tile's codes and zeroed counters in parity 0, the second tile's chunks in registers
  TileSrc pre;
  pre.a = make_uint4(0, 0, 0, 0); pre.b = make_uint4(0, 0, 0, 0);
  uint64_t tile = a_tile0 + (uint64_t)blockIdx.x;
  {
    const int tid = threadIdx.x;
    if (tile < a_ntiles) tile_fetch<PK>(a_arg, tile, tid, pre);
    tile_stage<PK>(a_arg, pre, tid, L.code[0], L.inv[0]);
    for (uint32_t b = tid; b < kFcBins + 64; b += kThreads) L.cnt[0][b] = 0;
    if (tid < 4) { L.code[0][kChunks + tid] = 0; L.inv[0][kChunks / 2 + tid] = 0xFFFFFFFFu; }
    if (tid < 2) { L.pool_n[tid] = 0; L.left[tid] = 0; }
    if (tile + gridDim.x < a_ntiles) tile_fetch<PK>(a_arg, tile + gridDim.x, tid, pre);
  }
  __syncthreads();
  uint32_t par = 0;
  MCX_PH_DECL
  for (; tile < a_ntiles; tile += gridDim.x, par ^= 1u) {
    const int tid = (int)tid_now();
    const uint32_t *s_code = L.code[par];
    const uint32_t *s_inv = L.inv[par];
    uint32_t *cnt = L.cnt[par];
    MCX_PH(6) MCX_PH_COUNT
    // ---- k-merise: the lane's 16 positions -> segments / pool ----
    const uint32_t pl = 16u * (uint32_t)(tid + 1);
    const uint64_t Vh = inv_win64(s_inv, pl);
    const uint32_t prev_chunk_inv = (s_inv[(pl - 1) >> 5] >> (31 - ((pl - 1) & 31))) & 1u;
    uint32_t range = 0xFFFFu;
    {
      const uint64_t plo = a.pos_lo, phi = a.pos_hi, T0 = tile * kTile;
      if (T0 < plo || T0 + kTile > phi) {
        const uint64_t P0 = T0 + 16ull * (uint64_t)tid;
        const int j_lo = plo > P0 ? (int)min((uint64_t)kPosPerLane, plo - P0) : 0;
        const int j_hi = phi > P0 ? (int)min((uint64_t)kPosPerLane, phi - P0) : 0;
        range = ((0x10000u >> j_lo) - 1u) & ~((0x10000u >> j_hi) - 1u);
      }
    }
    uint32_t ok16, nok16, pok16;  // as in k_stream_bin
    {
      uint64_t Mh = Vh;
      for (int c = 1; c < k;) {  // uniform
        const int s = min(c, k - c);
        Mh |= Mh << s;
        c += s;
      }
      ok16 = ~(uint32_t)(Mh >> 48) & range;
      const int sh = 112 - k;  // 81..109
      const uint64_t nx = Vh >> (sh - 64);
      nok16 = ~(uint32_t)nx & 0xFFFFu;
      pok16 = ~((prev_chunk_inv << 15) | (uint32_t)(Vh >> 49)) & 0xFFFFu;
    }
    n_kmers += __popc(ok16);
    n_contigs += __popc(ok16 & ~pok16);
    const uint32_t trash = nlocal + 1u + ((uint32_t)tid & 31u);
    uint64_t tk[kPosPerLane];   // quotient | edges << 56
    uint32_t tb[kPosPerLane];   // bin (a trash bin for a position without a k-mer)
    uint32_t ovmask = 0;        // positions whose segment was full
    {
      uint32_t arr[kPosPerLane];  // arrival index of the tuple in its bin (moves into tb's upper half)
      const uint32_t own = s_code[pl >> 4];
      const uint32_t before = s_code[(pl - 1) >> 4] & 3u;
      const uint32_t lbq = t_lbq;
      const uint32_t qmask = (1u << lbq) - 1u;
      const LaneWin lw = lane_win(s_code, pl, k);
      const uint32_t feed32 = (uint32_t)(code_win64(s_code, pl + (uint32_t)k) >> 32);
      const uint32_t key_sh = 64u - 2u * (uint32_t)k;
      const uint32_t mix_sh = (32u - lbq) & 31u;
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++) {
        const uint32_t prev_nuc = j == 0 ? before : ((own >> (32 - 2 * j)) & 3u);
        const uint32_t nuc_next = (feed32 >> (30 - 2 * j)) & 3u;
        const uint32_t valid = 0u - ((ok16 >> (15 - j)) & 1u);
        const uint32_t nb = (nok16 >> (15 - j)) & 1u, pb = (pok16 >> (15 - j)) & 1u;
        const uint32_t fh = j ? __builtin_amdgcn_alignbit(lw.a2, lw.a1, 32 - 2 * j) : lw.a2;
        const uint32_t fl = j ? __builtin_amdgcn_alignbit(lw.a1, lw.a0, 32 - 2 * j) : lw.a1;
        const uint32_t rh = j < 15 ? __builtin_amdgcn_alignbit(lw.r2, lw.r1, 2 + 2 * j) : lw.r2;
        const uint32_t rl = j < 15 ? __builtin_amdgcn_alignbit(lw.r1, lw.r0, 2 + 2 * j) : lw.r1;
        const uint64_t f = ((uint64_t)fh << 32) | fl, r = ((uint64_t)rh << 32) | rl;
        const uint32_t o = f < r ? 0u : 1u;
        const uint64_t sel = f < r ? f : r;
        const uint32_t o4 = o << 2;
        const uint32_t e = (nb << (nuc_next | o4)) | (pb << ((prev_nuc ^ 7u) ^ o4));
        const uint64_t kk = sel >> key_sh;
        Kmer<W> q;
        q.w[0] = kk >> lbq;
        const uint32_t Gr = ((uint32_t)kk ^ (region_mix<W>(q) >> mix_sh)) & qmask;
        const uint32_t local = (Gr & valid) | (trash & ~valid);
        tk[j] = q.w[0] | ((uint64_t)e << 56);
        tb[j] = local;
        arr[j] = atomicAdd(&cnt[local], 1u);
        asm volatile("" : "+v"(tb[j]), "+v"(tk[j]));  // one position at a time
      }
      // the atomics return in order: 16 in flight, each store waits for its own
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++) {
        const uint32_t bj = tb[j];
        if (bj < nlocal) {
          if (arr[j] < (uint32_t)CAP) L.seg[bj * CAP + arr[j]] = tk[j];
          else { ovmask |= 1u << j; tb[j] = bj | (arr[j] << 16); }
        }
      }
    }
    MCX_PH(0)
    if (ovmask) {  // Poisson tail: one pool reservation per lane, then its entries
      const uint32_t ov0 = ovmask;
      const uint32_t base = atomicAdd(&L.pool_n[par], (uint32_t)__popc(ov0));
#pragma unroll
      for (int j = 0; j < kPosPerLane; j++) {
        if ((ov0 >> j) & 1u) {
          const uint32_t p = base + __popc(ov0 & ((1u << j) - 1u));
          if (p < (uint32_t)kFcPool) { L.pool_t[p] = tk[j]; L.pool_b[p] = PRIV ? tb[j] : tb[j] & 0xffffu; ovmask &= ~(1u << j); }
        }
      }
      if (ovmask) L.left[par] = 1;  // the pool is full: these wait for the rounds after the write-out
    }
    MCX_PH(1)
    __syncthreads();
    MCX_PH(2)

    // ---- write-out: this wave's 2 x 64 bins; the next tile's codes and counters ----
    const uint32_t lane = (uint32_t)tid & 63u, wbin0 = ((uint32_t)tid >> 6) * (64u * ROWS);
    uint32_t c_row[ROWS], fit_tot[ROWS];  // tuples in the LDS segment; PRIV: arrivals (segment + pool) that fit the bin in HBM
    unsigned long long g_row[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      const uint32_t b = wbin0 + 64u * r + lane;
      const uint32_t call = b < nlocal ? cnt[b] : 0u;
      const uint32_t c = min(call, (uint32_t)CAP);
      c_row[r] = c;
      if (PRIV) {
        g_row[r] = fill[r];
        fit_tot[r] = (uint32_t)min((unsigned long long)call, out.cap > g_row[r] ? out.cap - g_row[r] : 0ull);
        fill[r] = (uint32_t)min((unsigned long long)fill[r] + call, 0xFFFF0000ull);
      } else {
        g_row[r] = c ? atomicAdd(&out.counts[ob0 + b], (unsigned long long)c) : 0ull;
        fit_tot[r] = 0;
      }
    }
    const uint32_t left = L.left[par];
    {  // stage tile + grid (held in `pre`), zero the other parity's counters, fetch tile + 2 grid
      const uint64_t tn = tile + gridDim.x;
      if (tn < a_ntiles) {
        tile_stage<PK>(a_arg, pre, tid, L.code[par ^ 1u], L.inv[par ^ 1u]);
        for (uint32_t b = tid; b < kFcBins + 64; b += kThreads) L.cnt[par ^ 1u][b] = 0;
        if (tid < 4) { L.code[par ^ 1u][kChunks + tid] = 0; L.inv[par ^ 1u][kChunks / 2 + tid] = 0xFFFFFFFFu; }
        if (tn + gridDim.x < a_ntiles) tile_fetch<PK>(a_arg, tn + gridDim.x, tid, pre);
      }
      if (tid == 0) { L.pool_n[par ^ 1u] = 0; L.left[par ^ 1u] = 0; }
    }
    // PRIV: a pool entry (bin | arrival index << 16) is written by the wave that owns its bin: base and room come
    // from the owner lane's registers
    auto drain_priv = [&](uint32_t pn) {
      const uint32_t wv = (uint32_t)tid >> 6;
      for (uint32_t i0 = 0; i0 < pn; i0 += 64u) {  // uniform
        const uint32_t i = i0 + lane;
        const bool have = i < pn;
        const uint32_t ent = have ? L.pool_b[i] : 0u;
        const uint32_t b = ent & 0xffffu, idx = ent >> 16, src = (b & 63u) << 2;
        const uint32_t ga = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)(uint32_t)g_row[0]);
        const uint32_t gb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)(uint32_t)g_row[ROWS - 1]);
        const uint32_t fa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)fit_tot[0]);
        const uint32_t fb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)fit_tot[ROWS - 1]);
        const bool second = (b >> 6) & 1u;
        if (have && (b >> 7) == wv) {
          if (idx < (second ? fb : fa)) out.keys[(uint64_t)(ob0 + b) * out.cap + (second ? gb : ga) + idx] = L.pool_t[i];
          else acc = fc_acc(acc, fc_direct<ONECOL>(isink, b, L.pool_t[i]));
        }
      }
    };
    if (PRIV) drain_priv(min(L.pool_n[par], (uint32_t)kFcPool));
    else acc = fc_acc(acc, fc_pool_drain<ONECOL, CAP>(L, min(L.pool_n[par], (uint32_t)kFcPool), out, ob0, isink));
    MCX_PH(3)
    // lane -> (bin of the group, slot): constant per lane
    const uint32_t grp = lane / (uint32_t)CAP, slot = lane - grp * (uint32_t)CAP;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      // destination of slot 0 of the lane's own bin, and how many of its tuples fit the segment
      const unsigned long long room = out.cap > g_row[r] ? out.cap - g_row[r] : 0ull;
      const uint32_t fit = PRIV ? min(c_row[r], fit_tot[r]) : (uint32_t)min((unsigned long long)c_row[r], room);
      const uint64_t dst0 = (uint64_t)(ob0 + wbin0 + 64u * r + lane) * out.cap + g_row[r];
#pragma unroll
      for (int it = 0; it < ITERS; it++) {
        const uint32_t src = (uint32_t)it * G + grp;   // lane that holds the bin of this group
        const bool act = src < 64u && grp < (uint32_t)G;
        const uint32_t srcb = act ? src : 63u;
        const uint32_t fit_s = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srcb << 2), (int)fit);
        const uint32_t d_lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srcb << 2), (int)(uint32_t)dst0);
        const uint32_t d_hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srcb << 2), (int)(uint32_t)(dst0 >> 32));
        if (act && slot < fit_s)
          out.keys[(((uint64_t)d_hi << 32) | d_lo) + slot] = L.seg[(wbin0 + 64u * r + src) * CAP + slot];
      }
      if (c_row[r] > fit) {  // the bin's segment in HBM is full: the lane that reserved sends the rest through the insert
        const uint32_t b = wbin0 + 64u * r + lane;
        for (uint32_t s = fit; s < c_row[r]; s++) acc = fc_acc(acc, fc_direct<ONECOL>(isink, b, L.seg[b * CAP + s]));
      }
    }
    MCX_PH(4)
    __syncthreads();
    MCX_PH(5)
    if (left) {
      // Rounds for the tuples that found the pool full (one region taking hundreds of a tile's tuples): append what
      // fits, drain, repeat.  pool_n / left of the other parity are free until the next tile's k-merising and are
      // zero again when the rounds end.
      for (;;) {
        if (ovmask) {
          const uint32_t ov0 = ovmask;
          const uint32_t base = atomicAdd(&L.pool_n[par ^ 1u], (uint32_t)__popc(ov0));
#pragma unroll
          for (int j = 0; j < kPosPerLane; j++) {
            if ((ov0 >> j) & 1u) {
              const uint32_t p = base + __popc(ov0 & ((1u << j) - 1u));
              if (p < (uint32_t)kFcPool) { L.pool_t[p] = tk[j]; L.pool_b[p] = PRIV ? tb[j] : tb[j] & 0xffffu; ovmask &= ~(1u << j); }
            }
          }
          if (ovmask) L.left[par ^ 1u] = 1;
        }
        __syncthreads();
        const uint32_t more = L.left[par ^ 1u];
        if (PRIV) drain_priv(min(L.pool_n[par ^ 1u], (uint32_t)kFcPool));
        else acc = fc_acc(acc, fc_pool_drain<ONECOL, CAP>(L, min(L.pool_n[par ^ 1u], (uint32_t)kFcPool), out, ob0, isink));
        __syncthreads();
        if (tid == 0) { L.pool_n[par ^ 1u] = 0; L.left[par ^ 1u] = 0; }
        __syncthreads();
        if (!more) break;
      }
    }
  }
  MCX_PH_DUMP(0)
  if (PRIV) {
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      const uint32_t b = (threadIdx.x >> 6) * (64u * ROWS) + 64u * r + (threadIdx.x & 63u);
      if (b < nlocal) out.counts[ob0 + b] = fill[r];
    }
  }

  block_add(&a.ctr->kmers, n_kmers);
  block_add(&a.ctr->contigs, n_contigs);
  block_add(&a.ctr->novel, acc & 0x3fffffffu);
  if (acc >> 30) a.ctr->full = 1;
  if (a.flag && n_contigs) *a.flag = 1;
}

}  // namespace mcx
