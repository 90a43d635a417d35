// mcx_kernels.h -- gfx950 kernels of the McCortex `build` hot path (included by mcx_api.hip).
//
// One pass over a byte stream of reads in HBM does what the reference does per
// read on the CPU (src/tools/build_graph.c:122-189): split into ACGT contigs,
// roll 2-bit k-mers, canonicalise, Lookup3-hash, find-or-insert into an
// open-addressed table in HBM, coverage +1, edge OR.
//
// Integer / atomic work only: no MFMA.  Bound by random HBM sector accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mcx_kmer.h"

namespace mcx {

// ---------------------------------------------------------------------------
// Table layout in HBM
// ---------------------------------------------------------------------------
// Array of records, S = W + ncols 64-bit words each:
//   word 0      : key word 0 | kFlag            (0 = empty slot)
//   word 1..W-1 : remaining key words
//   word W+c    : colour c value = coverage << 8 | edge byte
// Coverage is a 56-bit count (never wraps) clamped to UINT32_MAX on export,
// which is the reference's saturating +1 (src/graph/db_node.c:139-144); the
// edge byte is only ever OR-ed (src/graph/db_node.h:273-274).  Key, coverage
// and edges of a node share one 64-byte sector, so an occurrence costs one
// random sector instead of the reference's three arrays.
constexpr int kBucket = 4;  // slots per hash bucket (64 B when S == 2)
// The table is an array of independent sub-tables: a key's probe sequence starts at its hash
// bucket and wraps inside its sub-table, so one sub-table can be staged in LDS and owned by a
// single workgroup (mcx_defer.h).  One-word keys: 4096 slots (64 KiB of key + value words, two
// workgroups per CU); two-word keys: 2048 slots (48 KiB, so that two or three workgroups share a
// CU instead of one 96 KiB slice monopolising it).
template <int W> struct Sub {
  static constexpr int kShift = W == 1 ? 12 : 11;
  static constexpr uint64_t kSlots = 1ull << kShift;
  static constexpr uint32_t kBuckets = (uint32_t)(kSlots / kBucket);
};
__host__ __device__ constexpr int sub_shift_for_words(int W) { return W == 1 ? 12 : 11; }

struct TableView {
  uint64_t *rec;
  uint64_t nslots;   // all slots: nmain hash-addressed ones, then the overflow area
  uint64_t nmain;    // = (spb << lb1) sub-tables of Sub<W>::kSlots slots
  uint32_t lb1;      // log2 of the number of top-level regions ("L1 bins") of the table
  uint32_t lbo;      // log2 of the number of shards (GPUs) the global table is split over
  uint32_t part;     // which shard this table is (0 when lbo == 0)
  uint32_t spb;      // sub-tables per region
  uint32_t S;        // words per slot in total (key words + colours)
  uint32_t max_probe;
  // Layout.  One colour: array of records [key words, value] (key and value share a 16-byte /
  // 24-byte record, one sector per occurrence).  Several colours: the key words of all slots
  // first, then one value array per colour -- a flush of one colour then touches the keys and
  // that colour's values only, not every colour's (records of 8 * (W + ncols) bytes made a
  // 4-colour flush move 2.5x the bytes, a 10-colour one 5.5x).
  uint32_t KS;       // words between the keys of consecutive slots (W + 1, or W)
  uint32_t VS;       // words between the values of consecutive slots (W + 1, or 1)
  uint64_t *val;     // value word of (slot 0, colour 0)
  uint64_t VC;       // words between the value arrays of consecutive colours (0, or nslots)
  // touch[0] != 0: some kernel other than the LDS insert may have written hash-addressed slots
  // since the table was zeroed; touch[1 + s / 32] bit s % 32: the LDS insert has written sub-table
  // s.  While touch[0] == 0, a sub-table whose bit is clear still holds only zeros and the LDS
  // insert does not read it (the first flush of a build reads none of the table).
  // In front of touch[0] sit kTouchHdr bytes of slow-path counters (the allocation starts there): occurrences that
  // took the per-tuple direct insert because their partition bin was full (`fallback`), and keys of other shards
  // inserted on the spot (`foreign`).  Correct but slow paths -- "cliffs" -- that would otherwise be silent
  // (mcx_graph_insert_stats; the reference prints its collision histogram, hash_table.c:301-333).  They live here
  // because the slow paths already hold this pointer: the hot kernels pay no register for them.
  uint32_t *touch;
};
constexpr int kTouchHdr = 32;  // bytes
// Called by every kernel that writes hash-addressed slots outside the LDS insert.
__device__ __forceinline__ void table_mark_written(const TableView &t)
{
  if (t.touch) t.touch[0] = 1u;
}
// One atomic per wave, not per lane: these counters sit on ONE 8-byte address, and the paths that bump them are the ones a
// hot k-mer takes hundreds of thousands of times in a row (poly-A: every lane of every wave).  The lanes that are active
// here elect the lowest one, which adds the count of all of them.
__device__ __forceinline__ void table_count_slow(unsigned long long *ctr)
{
  const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
  const uint32_t lane = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
  if (lane == 0) atomicAdd(ctr, (unsigned long long)__popcll(act));  // (mbcnt: active lanes below this one)
}
__device__ __forceinline__ void table_count_fallback(const TableView &t)
{
  if (t.touch) table_count_slow(reinterpret_cast<unsigned long long *>(t.touch) - 4);
}
__device__ __forceinline__ void table_count_foreign(const TableView &t)
{
  if (t.touch) table_count_slow(reinterpret_cast<unsigned long long *>(t.touch) - 3);
}
// "Hash table is full", fail fast (round 5).  A key whose sub-table is full scans the overflow area linearly, the whole
// of it (1 / 32 of the table) before it gives up -- and once that area is full EVERY further new key does: a build whose
// -n is too small for its input (4.8 G distinct k-mers into 1.1 G slots) ground on for a quarter of an hour where the
// reference dies at the first failed insert (hash_table.c:119-123).  The first scan that fails raises touch[-2]; an
// insert that is about to enter the overflow area, or is in the middle of scanning it, gives up as soon as it sees the
// flag (its occurrence is lost, but so is the build: mcx_graph_sync reports MCX_ERR_FULL).
__device__ __forceinline__ bool table_full_flagged(const TableView &t)
{
  return t.touch && __hip_atomic_load(t.touch - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
__device__ __forceinline__ void table_flag_full(const TableView &t)
{
  if (t.touch) __hip_atomic_store(t.touch - 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t *key_ptr(const TableView &t, uint64_t slot) { return t.rec + slot * t.KS; }
__device__ __forceinline__ uint64_t *val_ptr(const TableView &t, uint64_t slot, uint32_t col)
{
  return t.val + slot * t.VS + (uint64_t)col * t.VC;
}
// the same with the layout known at compile time (ONECOL == one colour == records)
template <int W, bool ONECOL> __device__ __forceinline__ uint64_t *key_ptr_t(const TableView &t, uint64_t slot)
{
  return t.rec + slot * (ONECOL ? (uint64_t)(W + 1) : (uint64_t)W);
}
template <int W, bool ONECOL> __device__ __forceinline__ uint64_t *val_ptr_t(const TableView &t, uint64_t slot, uint32_t col)
{
  return ONECOL ? t.rec + slot * (uint64_t)(W + 1) + W : t.val + (uint64_t)col * t.VC + slot;
}

// ---------------------------------------------------------------------------
// Table addressing: a quotient hash; Lookup3 picks the sub-table and the bucket
// ---------------------------------------------------------------------------
// key = (q << lbq) | r with lbq = lb1 + lbo.
//     m       = mix(q)                           folded q times an odd constant (2 instructions)
//     G       = r ^ (m >> (32 - lbq))            one Feistel round: uniform whatever r is
//     owner   = G >> lb1                         shard (GPU) that holds the key: a hash prefix
//     region  = G & (2^lb1 - 1)                  region of that shard's table
//     bucket  = (m >> (22 - lbq)) & 1023         start bucket inside the sub-table
// and with b = sub_hash(q), a second multiplicative hash of q:
//     sub     = region * spb + mulhi(b, spb)     sub-table
// All three build kernels are bound by their instruction counts (60-70 % VALU-busy at 4 waves
// per SIMD), so each hash is as cheap as uniformity allows.
// Given (owner, region), r = G ^ (mix(q) & mask) is recoverable from q alone, so a k-mer occurrence
// that has been binned by region travels as q plus its edge byte in ONE 64-bit word per key
// word (2k - lbq <= 56 bits of q in the top word, edges in bits 56..63) -- mcx_defer.h.
template <int W> __device__ __host__ __forceinline__ uint32_t region_mix(const Kmer<W> &q)
{
  // multiplicative hash of the folded quotient: the quality sits in the HIGH bits of the product
  // (2 instructions for a one-word key; the upper half of the 64-bit product cost 4)
  uint32_t x = (uint32_t)q.w[0] ^ (uint32_t)(q.w[0] >> 32);
  if (W == 2) x ^= ((uint32_t)q.w[W - 1] ^ (uint32_t)(q.w[W - 1] >> 32)) * 0x85EBCA6Bu;
  if (W > 2)  // (three- and four-word keys, k > 63: every further word folded in with its own odd multiplier)
    for (int i = 1; i < W; i++) x ^= ((uint32_t)q.w[i] ^ (uint32_t)(q.w[i] >> 32)) * (0x85EBCA6Bu + 0x9E3779B2u * (uint32_t)(i - 1));
  return x * 0x9E3779B1u;
}
// Second hash of the quotient: picks the sub-table inside the region (through mulhi: its top bits
// count).  A different fold of the words than region_mix's, two multiplications with a shift-xor
// between them: 7 instructions.  (Until round 2 this was the second word of Lookup3 -- 30 of the
// split kernel's 78 instructions per tuple, and that kernel is not the HBM-bound one it was when
// the choice was made.  Lookup3 still decides what it decides in the reference's terms: the shard
// of a key in the hash-prefix exchange, mcx_key_owner, and it addresses the overflow area.)
template <int W> __device__ __host__ __forceinline__ uint32_t sub_hash(const Kmer<W> &q)
{
  uint32_t x = (uint32_t)q.w[0] + (uint32_t)(q.w[0] >> 32) * 0xC2B2AE35u;
  if (W == 2) x += (uint32_t)q.w[W - 1] * 0x27D4EB2Fu + (uint32_t)(q.w[W - 1] >> 32) * 0x165667B1u;
  if (W > 2)
    for (int i = 1; i < W; i++) x += (uint32_t)q.w[i] * (0x27D4EB2Fu + 0x3C6EF372u * (uint32_t)(i - 1)) + (uint32_t)(q.w[i] >> 32) * (0x165667B1u + 0x7F4A7C16u * (uint32_t)(i - 1));
  x *= 0x85EBCA6Bu;
  x ^= x >> 15;
  return x * 0x2C1B3C6Du;
}
// the lbq bits that are XOR-ed onto the remainder: the top ones (0 when lbq == 0)
__device__ __host__ __forceinline__ uint32_t mix_g(uint32_t m, uint32_t lbq) { return (uint32_t)(((uint64_t)m << lbq) >> 32); }
// start bucket inside the sub-table: the bits below them (lbq <= 22 - log2(buckets))
__device__ __host__ __forceinline__ uint32_t mix_bucket(uint32_t m, uint32_t lbq, uint32_t nbuckets)
{
  return (m >> (22u - lbq)) & (nbuckets - 1u);
}
template <int W> __device__ __host__ __forceinline__ Kmer<W> key_quot(const Kmer<W> &key, uint32_t lb1 /* bits to split off */, uint32_t &r)
{
  Kmer<W> q = key;
  if (W == 1) {  // (no special case for lb1 == 0: the mask is empty and the shift is by 0)
    r = (uint32_t)key.w[0] & ((1u << lb1) - 1u);
    q.w[0] = key.w[0] >> lb1;
    return q;
  }
  r = 0;
  if (lb1) {
    r = (uint32_t)key.w[W - 1] & ((1u << lb1) - 1u);
    if (W == 2) q.w[W - 1] = (key.w[W - 1] >> lb1) | (key.w[0] << (64 - lb1));
    if (W > 2)
      for (int i = W - 1; i >= 1; i--) q.w[i] = (key.w[i] >> lb1) | (key.w[i - 1] << (64 - lb1));
    q.w[0] = key.w[0] >> lb1;
  }
  return q;
}
template <int W> __device__ __host__ __forceinline__ Kmer<W> key_unquot(const Kmer<W> &q, uint32_t lb1, uint32_t r)
{
  Kmer<W> key = q;
  if (W == 1) {  // (no special case for lb1 == 0: a shift by 0 and r == 0)
    key.w[0] = (q.w[0] << lb1) | r;
    return key;
  }
  if (lb1) {
    key.w[0] = q.w[0] << lb1;
    if (W == 2) { key.w[0] |= q.w[W - 1] >> (64 - lb1); key.w[W - 1] = q.w[W - 1] << lb1; }
    if (W > 2)
      for (int i = 0; i < W; i++) key.w[i] = (q.w[i] << lb1) | (i + 1 < W ? q.w[i + 1] >> (64 - lb1) : 0ull);
    key.w[W - 1] |= r;
  }
  return key;
}
__device__ __host__ __forceinline__ uint32_t lbq_of(const TableView &t) { return t.lb1 + t.lbo; }

struct TableAddr { uint32_t G, region, sub, bucket; };
// address from the quotient q and remainder r (r = low lbq bits of the key)
template <int W> __device__ __forceinline__ TableAddr addr_of(const TableView &t, const Kmer<W> &q, uint32_t r)
{
  const uint32_t b = sub_hash<W>(q);
  const uint32_t lbq = lbq_of(t), m = region_mix<W>(q);
  TableAddr a;
  a.G = r ^ mix_g(m, lbq);
  a.region = a.G & ((1u << t.lb1) - 1u);
  a.sub = a.region * t.spb + __umulhi(b, t.spb);
  a.bucket = mix_bucket(m, lbq, Sub<W>::kBuckets);
  return a;
}
// shard (GPU) that holds `key` in a table split over 2^lbo shards
template <int W> __device__ __forceinline__ uint32_t key_owner(const TableView &t, const Kmer<W> &key)
{
  uint32_t r;
  const Kmer<W> q = key_quot<W>(key, lbq_of(t), r);
  return (r ^ mix_g(region_mix<W>(q), lbq_of(t))) >> t.lb1;
}
// remainder of a key of THIS shard from its quotient and its region
template <int W> __device__ __forceinline__ uint32_t r_of(const TableView &t, uint32_t region, const Kmer<W> &q)
{
  return ((t.part << t.lb1) | region) ^ mix_g(region_mix<W>(q), lbq_of(t));
}
template <int W> __device__ __forceinline__ uint64_t key_slot(const TableView &t, const Kmer<W> &key)
{
  uint32_t r;
  const Kmer<W> q = key_quot<W>(key, lbq_of(t), r);
  const TableAddr a = addr_of<W>(t, q, r);
  return ((uint64_t)a.sub << Sub<W>::kShift) + (uint64_t)a.bucket * kBucket;
}

// ---------------------------------------------------------------------------
// Who owns a key (a table split over several GPUs)
// ---------------------------------------------------------------------------
// Two ways to deal the keys out.  (1) hash prefix: owner = top bits of the address word G; the table
// itself is sharded (TableView::lbo / part) -- exchange format v2.  (2) minimizer: owner = hash of the
// smallest hashed canonical 13-mer inside the k-mer (mcx_superk.h, exchange format v3: reads travel,
// not occurrences); every shard then holds an ordinary table.  Kernels that walk records or reads on
// every shard and keep what is theirs (k_load_records, k_pcr_starts, k_reads_must_exist) take an
// OwnerSpec.
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

// Host/device reference: owner of a k-mer given as its 2-bit value (w0 = top word, unused for
// k <= 31).  The kernels compute the same thing incrementally; tests compare shard contents
// against this.
MCX_HD uint32_t superk_owner(uint64_t w0, uint64_t w1, int k, uint32_t lbo)
{
  uint32_t best = 0xFFFFFFFFu;
  for (int p = 0; p + kMmer <= k; p++) {
    const int sh = 2 * (k - kMmer - p);  // the m-mer's low bit inside the 2k-bit number w0 : w1
    const uint64_t low = sh >= 64 ? (w0 >> (sh - 64)) : (sh ? (w1 >> sh) | (w0 << (64 - sh)) : w1);
    const uint32_t f = (uint32_t)low & kMmerMask;
    uint32_t r = 0;
    for (int i = 0; i < kMmer; i++) r |= (3u - ((f >> (2 * i)) & 3u)) << (2 * (kMmer - 1 - i));
    const uint32_t h = mmer_hash(f < r ? f : r);
    best = h < best ? h : best;
  }
  return owner_of_minimizer(best, lbo);
}


struct OwnerSpec {
  uint32_t mode;  // 0: one table, everything is mine; 1: hash prefix (the table's lbo / part); 2: minimizer
  uint32_t lbo;   // mode 2: log2 shards
  uint32_t part;  // mode 2: this shard
  int k;
};
template <int W> __device__ __forceinline__ bool owner_is_me(const TableView &t, const OwnerSpec &os, const Kmer<W> &key)
{
  if (os.mode == 2) return W > 2 ? true : superk_owner(W == 1 ? 0ULL : key.w[0], key.w[W - 1], os.k, os.lbo) == os.part;  // (k > 63: one device only)
  if (os.mode == 1) return key_owner<W>(t, key) == t.part;
  return true;
}

struct Counters {  // device-resident, 64-bit each
  unsigned long long novel;     // nodes created (== hash_table num_kmers)
  unsigned long long kmers;     // k-mer occurrences processed
  unsigned long long contigs;   // contigs started
  unsigned long long full;      // != 0: an insert ran out of probes
  unsigned long long bin_over;  // != 0: a partition bin overflowed
  unsigned long long good_reads, bad_reads;
  unsigned long long absent;    // must-exist mode: k-mer occurrences whose k-mer is not in the graph (not "loaded", build_graph.c:175-177)
  unsigned long long binned;    // occurrences the owner side of exchange v3 (k_superk_bin) put into the region bins: what a
                                // reservation made with an upper bound really used (settled launches, mcx_api.hip snap_*)
};

// Add a per-thread tally to a device counter with ONE global atomic per block: all blocks hit
// the same 64-byte counter line, and same-line atomics cost ~12 ns each at the L2 (one per wave
// was 24 K of them per k-merisation launch).  Call from uniform control flow at the end of a kernel.
__device__ __forceinline__ void block_add(unsigned long long *counter, unsigned long long v)
{
  __shared__ unsigned long long s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  if (v) atomicAdd(&s_sum, v);  // (the compiler folds a wave's adds into one LDS atomic)
  __syncthreads();
  if (threadIdx.x == 0 && s_sum) atomicAdd(counter, s_sum);
  __syncthreads();
}

#define MCX_RLX __ATOMIC_RELAXED
#define MCX_AGENT __HIP_MEMORY_SCOPE_AGENT

// ---------------------------------------------------------------------------
// find-or-insert + coverage/edge update for one k-mer occurrence
// ---------------------------------------------------------------------------
// Lock-free replacement of hash_table_find_or_insert_mt (hash_table.c:250-281)
// + db_graph_update_node_mt (db_graph.c:101-105) + this occurrence's share of
// db_graph_add_edge_mt (db_graph.c:152-166).
//
// Visibility argument (per-XCD L2s are not coherent for plain accesses): slot
// state only moves empty -> key, never back.  A stale plain load can therefore
// only claim "empty" for a slot that is taken, and then the agent-scope CAS
// (performed at the coherence point) returns the real occupant.  Value words
// are only touched by agent-scope atomics; the plain read of the value is a
// hint used to skip the edge OR when the bits are already there (bits never
// clear, so a stale hint can only cause a redundant OR).
template <int W>
__device__ __forceinline__ void update_value(uint64_t *val, uint64_t hint, uint32_t e)
{
  __hip_atomic_fetch_add(val, 256ULL, MCX_RLX, MCX_AGENT);
  if (e & ~(uint32_t)hint & 0xffu) __hip_atomic_fetch_or(val, (uint64_t)e, MCX_RLX, MCX_AGENT);
}

// Overflow area.  A probe sequence never leaves its sub-table (that is what lets one workgroup own
// a sub-table in LDS), so a sub-table can fill up while the table as a whole has room: at 95 % load
// about one sub-table in 2000 does.  The reference keeps going until 20 buckets in a row are full
// (hash_table.c:250-281, REHASH_LIMIT hash_table.h:10), i.e. far beyond that load.  Keys that find
// their sub-table full therefore go to a small common area behind the hash-addressed slots
// ([nmain, nslots), 1/32 of the table), addressed by Lookup3 and probed linearly over the whole
// area.  "A key is in the overflow area only if its sub-table is full" is stable -- sub-tables
// never empty -- so find-or-insert stays well defined: scan the sub-table; no hit and no free slot
// -> same protocol in the overflow area; that one full as well -> "Hash table is full".
template <int W> __device__ __forceinline__ uint64_t ovf_start(const TableView &t, const Kmer<W> &key)
{
  const uint32_t h = kmer_hash<W>(key, 0, nullptr);
  const uint64_t nb = (t.nslots - t.nmain) / kBucket;
  return t.nmain + (uint64_t)__umulhi(h, (uint32_t)nb) * kBucket;
}

// `ovf`: start in the overflow area (the caller has seen the key's sub-table full)
template <int W, bool ONECOL>
__device__ __forceinline__ void probe_insert(const TableView &t, const Kmer<W> &key, uint64_t slot,
                                             uint64_t cur, uint64_t hint, uint32_t e, uint32_t col,
                                             uint32_t &novel, uint32_t &full, bool ovf = false)
{
  const uint64_t want = key.w[0] | kFlag;
  uint32_t probes = 0;
  uint64_t limit = ovf ? t.nslots - t.nmain : t.max_probe;
  bool fresh = !ovf;  // `cur`/`hint` were preloaded for this slot
  if (ovf && table_full_flagged(t)) { full = 1; return; }
  for (;;) {
    uint64_t *r = key_ptr_t<W, ONECOL>(t, slot);
    uint64_t *v = val_ptr_t<W, ONECOL>(t, slot, col);
    if (!fresh) {
      if (W == 1 && ONECOL) {
        const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(r);
        cur = kv.x; hint = kv.y;
      } else {
        cur = r[0]; hint = 0;
      }
    }
    fresh = false;
    if (cur == 0) {
      uint64_t expected = 0;
      const uint64_t desired = (W == 1) ? want : (want | kPending);
      if (__hip_atomic_compare_exchange_strong(r, &expected, desired, MCX_RLX, MCX_RLX, MCX_AGENT)) {
        if (W >= 2) {
          // publish the lower word(s) write-through, drain, then clear kPending
          for (int i = 1; i < W; i++) __hip_atomic_store(r + i, key.w[i], MCX_RLX, MCX_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(r, want, MCX_RLX, MCX_AGENT);
        }
        novel++;
        update_value<W>(v, 0, e);
        return;
      }
      cur = expected;  // somebody else took the slot: look at what is there now
      hint = 0;
    }
    if ((cur & ~kPending) == want) {
      if (W >= 2) {
        if (cur & kPending) {  // owner has not published word 1 yet: re-read at agent scope
          cur = __hip_atomic_load(r, MCX_RLX, MCX_AGENT);
          fresh = true; hint = 0;
          if (++probes > limit * 64u) { full = 1; return; }
          continue;
        }
        bool same = true;
        for (int i = 1; i < W; i++) same = same && __hip_atomic_load(r + i, MCX_RLX, MCX_AGENT) == key.w[i];
        if (same) { update_value<W>(v, *v, e); return; }
      } else {
        if (!ONECOL) hint = *v;
        update_value<W>(v, hint, e);
        return;
      }
    }
    if (++probes >= limit) {
      if (ovf || t.nslots == t.nmain) { full = 1; table_flag_full(t); return; }
      ovf = true;  // the whole sub-table was seen without a hit or a free slot
      if (table_full_flagged(t)) { full = 1; return; }
      probes = 0;
      limit = t.nslots - t.nmain;
      slot = ovf_start<W>(t, key);
      continue;
    }
    if (ovf && (probes & 4095u) == 0 && table_full_flagged(t)) { full = 1; return; }
    slot++;
    if (!ovf) { if ((slot & (Sub<W>::kSlots - 1)) == 0) slot -= Sub<W>::kSlots; }  // wrap inside the sub-table
    else if (slot == t.nslots) slot = t.nmain;
  }
}

// ---------------------------------------------------------------------------
// Bulk load of .ctx records (graph_load, src/graph/graphs_load.c:86-214)
// ---------------------------------------------------------------------------
// find (or, unless must_exist, insert) the slot of `key`; kNoSlot = absent / table full
constexpr uint64_t kNoSlot = ~0ULL;
template <int W>
__device__ __forceinline__ uint64_t find_or_insert_rec(const TableView &t, const Kmer<W> &key, bool must_exist,
                                                       uint32_t &novel, uint32_t &full)
{
  const uint64_t want = key.w[0] | kFlag;
  uint64_t slot = key_slot<W>(t, key);
  uint32_t probes = 0;
  uint64_t limit = t.max_probe;
  bool ovf = false;
  for (;;) {
    uint64_t *r = key_ptr(t, slot);
    uint64_t cur = __hip_atomic_load(r, MCX_RLX, MCX_AGENT);
    if (cur == 0) {
      if (must_exist) return kNoSlot;
      uint64_t expected = 0;
      const uint64_t desired = (W == 1) ? want : (want | kPending);
      if (__hip_atomic_compare_exchange_strong(r, &expected, desired, MCX_RLX, MCX_RLX, MCX_AGENT)) {
        if (W >= 2) {
          for (int i = 1; i < W; i++) __hip_atomic_store(r + i, key.w[i], MCX_RLX, MCX_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(r, want, MCX_RLX, MCX_AGENT);
        }
        novel++;
        return slot;
      }
      cur = expected;
    }
    if ((cur & ~kPending) == want) {
      if (W == 1) return slot;
      if (cur & kPending) {  // owner has not published word 1 yet
        if (++probes > limit * 64u) { full = 1; return kNoSlot; }
        continue;
      }
      bool same = true;
      for (int i = 1; i < W; i++) same = same && __hip_atomic_load(r + i, MCX_RLX, MCX_AGENT) == key.w[i];
      if (same) return slot;
    }
    if (++probes >= limit) {  // sub-table full: the key is in the overflow area or nowhere (see ovf_start)
      if (ovf || t.nslots == t.nmain) { if (!must_exist) { full = 1; table_flag_full(t); } return kNoSlot; }
      ovf = true;
      if (!must_exist && table_full_flagged(t)) { full = 1; return kNoSlot; }
      probes = 0;
      limit = t.nslots - t.nmain;
      slot = ovf_start<W>(t, key);
      continue;
    }
    if (ovf && !must_exist && (probes & 4095u) == 0 && table_full_flagged(t)) { full = 1; return kNoSlot; }
    slot++;
    if (!ovf) { if ((slot & (Sub<W>::kSlots - 1)) == 0) slot -= Sub<W>::kSlots; }
    else if (slot == t.nslots) slot = t.nmain;
  }
}

struct RecordStats {  // device-resident
  unsigned long long loaded, novel;
  unsigned long long first_oversized, first_zero_covg, first_edges_no_covg;  // record index or ~0
};

// One thread per record of the .ctx body layout: W key words, file_ncols x u32 coverage,
// file_ncols x u8 edges, byte-packed.  The colour filter is a list of (from, into) pairs
// (file_filter, src/basic/file_filter.c): file colour from[i] is added to colour into[i] of the
// graph (coverage +=, edges |=; graph_file_read, graph_file_reader.c:392-412).  Records whose
// loaded colours all have zero coverage are skipped (graphs_load.c:121-125).  The per-record
// sanity checks of graph_file_read_raw (graph_file_reader.c:358-386) are reported as the index
// of the first offending record.
__device__ __forceinline__ uint32_t load_le32(const uint8_t *p)
{
  return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}

template <int W>
__global__ __launch_bounds__(256) void k_load_records(TableView t, const uint8_t *recs, uint64_t nrecs, uint64_t rec0,
                                                      uint32_t file_ncols, const int32_t *from, const int32_t *into,
                                                      uint32_t nmap, uint32_t must_exist, int mask_col, int kmer_size,
                                                      Counters *ctr, RecordStats *st, OwnerSpec os)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) table_mark_written(t);
  const uint32_t rec_bytes = 8u * W + 5u * file_ncols;
  uint32_t novel = 0, full = 0, loaded = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrecs; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint8_t *p = recs + i * rec_bytes;
    Kmer<W> key;
#pragma unroll
    for (int w = 0; w < W; w++) key.w[w] = (uint64_t)load_le32(p + 8 * w) | (uint64_t)load_le32(p + 8 * w + 4) << 32;
    const uint8_t *pc = p + 8 * W, *pe = pc + 4 * file_ncols;
    const int top_bits = 2 * kmer_size - 64 * (W - 1);
    if (top_bits < 64 && (key.w[0] >> top_bits)) { atomicMin(&st->first_oversized, (unsigned long long)(rec0 + i)); continue; }
    uint32_t any_file = 0, any_loaded = 0;
    bool edges_no_covg = false;
    for (uint32_t c = 0; c < file_ncols; c++) {
      const uint32_t cv = load_le32(pc + 4 * c);
      any_file |= cv;
      if (pe[c] && !cv) edges_no_covg = true;
    }
    for (uint32_t m = 0; m < nmap; m++) any_loaded |= load_le32(pc + 4 * from[m]);
    if (!any_file) atomicMin(&st->first_zero_covg, (unsigned long long)(rec0 + i));
    if (edges_no_covg) atomicMin(&st->first_edges_no_covg, (unsigned long long)(rec0 + i));
    if (!any_loaded) continue;
    if (!owner_is_me<W>(t, os, key)) continue;  // a shard of a multi-GPU table only loads its own keys
    const uint64_t slot = find_or_insert_rec<W>(t, key, must_exist != 0, novel, full);
    if (slot == kNoSlot) continue;
    // must_exist_in_edges (graphs_load.c:166-167): only edges the intersection graph has
    const uint32_t emask = mask_col >= 0 ? (uint32_t)(*val_ptr(t, slot, (uint32_t)mask_col) & 0xffULL) : 0xffu;
    for (uint32_t m = 0; m < nmap; m++) {
      const uint32_t cv = load_le32(pc + 4 * from[m]);
      const uint32_t e = pe[from[m]] & emask;
      uint64_t *val = val_ptr(t, slot, (uint32_t)into[m]);
      if (cv) __hip_atomic_fetch_add(val, (uint64_t)cv << 8, MCX_RLX, MCX_AGENT);
      if (e) __hip_atomic_fetch_or(val, (uint64_t)e, MCX_RLX, MCX_AGENT);
    }
    loaded++;
  }
  if (novel) { atomicAdd(&ctr->novel, (unsigned long long)novel); atomicAdd(&st->novel, (unsigned long long)novel); }
  if (loaded) atomicAdd(&st->loaded, (unsigned long long)loaded);
  if (full) ctr->full = 1;
}

// ---------------------------------------------------------------------------
// `sort` (src/commands/ctx_sort.c:133-152): keys of byte-packed .ctx records, record gather
// ---------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_record_keys(const uint8_t *recs, uint64_t n, uint32_t rec_bytes, uint64_t *k0, uint64_t *k1)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t *p = recs + i * rec_bytes;
  k0[i] = (uint64_t)load_le32(p) | (uint64_t)load_le32(p + 4) << 32;
  if (W == 2) k1[i] = (uint64_t)load_le32(p + 8) | (uint64_t)load_le32(p + 12) << 32;
}

// out record i = in record perm[i]; 16 lanes copy one record (coalesced over its bytes)
__global__ __launch_bounds__(256) void k_gather_records(const uint8_t *in, const uint64_t *perm, uint64_t n, uint32_t rec_bytes, uint8_t *out)
{
  const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const uint32_t l = threadIdx.x & 15u;
  if (i >= n) return;
  const uint8_t *src = in + perm[i] * rec_bytes;
  uint8_t *dst = out + i * rec_bytes;
  for (uint32_t b = l; b < rec_bytes; b += 16) dst[b] = src[b];
}

// strictly increasing keys? (ctx_index.c:136-137 requires it); flags the first violation
template <int W>
__global__ __launch_bounds__(256) void k_check_sorted(const uint64_t *k0, const uint64_t *k1, uint64_t n, unsigned long long *first_bad)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 >= n) return;
  bool lt = k0[i] < k0[i + 1];
  if (W == 2 && k0[i] == k0[i + 1]) lt = k1[i] < k1[i + 1];
  if (!lt) atomicMin(first_bad, (unsigned long long)(i + 1));
}

// ---------------------------------------------------------------------------
// Table scans: db_graph_get_kmer_covg (src/graph/db_graph.c:490-534) and the k-mer coverage
// histogram of clean's pre-pass (src/tools/clean_graph.c:365-377)
// ---------------------------------------------------------------------------
// out[c] = nodes with coverage in colour c, out[ncols + c] = their summed coverage (clamped per
// node to 2^32-1 like the exported value); hist[min(sum over colours (saturating), nbins-1)]++
__global__ __launch_bounds__(256) void k_covg_scan(TableView t, uint32_t ncols, unsigned long long *out,
                                                   unsigned long long *hist, uint32_t nbins, uint32_t lbins)
{
  // LDS: 2 * ncols accumulators, then a block-local copy of the first `lbins` histogram bins
  // (nearly every node lands in the low bins: global atomics on them would serialise)
  extern __shared__ unsigned long long s_acc[];
  unsigned long long *s_hist = s_acc + 2 * ncols;
  for (uint32_t c = threadIdx.x; c < 2 * ncols + lbins; c += blockDim.x) s_acc[c] = 0;
  __syncthreads();
  for (uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; slot < t.nslots; slot += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t *r = key_ptr(t, slot);
    uint64_t w0, v1 = 0;
    if (t.S == 2) {  // one 16-byte load per record (k <= 31, one colour)
      const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(r);
      w0 = kv.x; v1 = kv.y;
    } else {
      w0 = r[0];
    }
    if (!(w0 & kFlag)) continue;
    uint64_t sum = 0;
    for (uint32_t c = 0; c < ncols; c++) {
      uint64_t cv = (t.S == 2 ? v1 : *val_ptr(t, slot, c)) >> 8;
      if (cv > 0xFFFFFFFFull) cv = 0xFFFFFFFFull;
      if (cv) { atomicAdd(&s_acc[c], 1ULL); atomicAdd(&s_acc[ncols + c], (unsigned long long)cv); }
      sum += cv;
    }
    if (sum > 0xFFFFFFFFull) sum = 0xFFFFFFFFull;  // db_node_sum_covg saturates (db_node.h:302-309)
    if (hist) {
      const uint64_t bin = sum < nbins ? sum : nbins - 1;
      if (bin < lbins) atomicAdd(&s_hist[bin], 1ULL);
      else atomicAdd(&hist[bin], 1ULL);
    }
  }
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < 2 * ncols; c += blockDim.x)
    if (s_acc[c]) atomicAdd(&out[c], s_acc[c]);
  for (uint32_t c = threadIdx.x; c < lbins; c += blockDim.x)
    if (s_hist[c]) atomicAdd(&hist[c], s_hist[c]);
}

// Sink of the fused kernel: insert straight into the local table.
template <int W, bool ONECOL> struct InsertSink {
  TableView t;
  uint32_t col;
};

// ---------------------------------------------------------------------------
// Stream front end
// ---------------------------------------------------------------------------
constexpr int kThreads = 256;
constexpr int kPosPerLane = 16;
constexpr int kTile = kThreads * kPosPerLane;  // k-mer start positions per tile
constexpr int kChunks = 272;                   // 16-byte chunks staged per tile: 1 halo + 256 + 15
constexpr int kBatch = 4;              // probes in flight per lane (must divide 16)

struct StreamArgs {
  const uint8_t *stream;    // ASCII stream, or nullptr when the packed form below is given
  // Packed form of the same stream (what the host entry stages: 3 bits per position instead of 8
  // over PCIe, and no SWAR encode in the tile prologue): per 16 positions one code word (2 bits
  // per base, first base on top) and 16 invalid flags (first base = bit 15) -- exactly what
  // encode_words() makes of 16 ASCII bytes.  Positions >= nbytes carry the invalid flag.
  const uint32_t *code;
  const uint16_t *inv;
  uint64_t nbytes;          // positions of context available (outside = separator)
  uint64_t pos_lo, pos_hi;  // k-mer start positions owned by this launch
  uint64_t tile0, ntiles;   // tiles [tile0, ntiles) cover [pos_lo, pos_hi)
  int k;
  Counters *ctr;
  unsigned char *flag;      // optional: set to 1 if any contig starts in this launch
};

// 64 code bits (32 bases) starting at region base index q
__device__ __forceinline__ uint64_t code_win64(const uint32_t *s_code, uint32_t q)
{
  const uint32_t j = q >> 4, sh = (q & 15u) * 2u;
  const uint64_t hi = ((uint64_t)s_code[j] << 32) | s_code[j + 1];
  const uint64_t lo = s_code[j + 2];
  return sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
}
// 64 invalid-flag bits (64 bases) starting at region base index q
__device__ __forceinline__ uint64_t inv_win64(const uint32_t *s_inv, uint32_t q)
{
  const uint32_t j = q >> 5, sh = q & 31u;
  const uint64_t hi = ((uint64_t)s_inv[j] << 32) | s_inv[j + 1];
  const uint64_t lo = s_inv[j + 2];
  return sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
}

// Fetch one 16-byte chunk of the stream at byte offset g (bytes outside [0, nbytes) read as 0,
// which is not ACGT: out-of-range == separator).
__device__ __forceinline__ uint4 load_chunk(const uint8_t *stream, uint64_t nbytes, int64_t g)
{
  if (g >= 0 && (uint64_t)g + 16 <= nbytes) return *reinterpret_cast<const uint4 *>(stream + g);
  uint32_t w[4] = {0, 0, 0, 0};
  if (g + 16 > 0 && g < (int64_t)nbytes)
    for (int i = 0; i < 16; i++) {
      const int64_t p = g + i;
      if (p >= 0 && (uint64_t)p < nbytes) w[i >> 2] |= (uint32_t)stream[p] << (8 * (i & 3));
    }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// 16 bases -> 32 code bits (first base on top) and 16 invalid flags (first base = bit 15).
__device__ __forceinline__ void encode_words(const uint4 v, uint32_t &code, uint32_t &inv)
{
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  code = 0; inv = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t x = w[i];
    // 2-bit codes of 4 bytes gathered with one multiply (byte 0 most significant)
    const uint32_t t = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
    code = (code << 8) | ((t * 0x40100401u) >> 24);
    // exact per-byte "is non-zero" of (folded byte ^ letter): 0x80 where it differs
    const uint32_t u = x & 0xDFDFDFDFu;
#define MCX_NZ(v) ((((v) & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | (v))
    const uint32_t bad = MCX_NZ(u ^ 0x41414141u) & MCX_NZ(u ^ 0x43434343u) &
                         MCX_NZ(u ^ 0x47474747u) & MCX_NZ(u ^ 0x54545454u) & 0x80808080u;
#undef MCX_NZ
    inv = (inv << 4) | ((((bad >> 7) * 0x08040201u) >> 24) & 0xFu);
  }
}

__device__ __forceinline__ void encode_chunk(const uint8_t *stream, uint64_t nbytes, int64_t g,
                                             uint32_t &code, uint32_t &inv)
{
  encode_words(load_chunk(stream, nbytes, g), code, inv);
}

// The 272 chunks of a tile in flight: two per thread (the second one only for the first 16 threads).
// ASCII: the 16 bytes themselves; packed: x = code word, y = invalid flags.
// PK (compile time): the launch reads the packed form (the ASCII kernels are not to pay registers for it).
struct TileSrc { uint4 a, b; };
template <bool PK> __device__ __forceinline__ void tile_fetch(const StreamArgs &a, uint64_t tile, int tid, TileSrc &t)
{
  if (PK) {
    const int64_t c0 = (int64_t)(tile * (kTile / 16)) - 1;  // chunk 0 of the tile's region = the halo
    const int64_t nch = (int64_t)((a.nbytes + 15) / 16);
    const int64_t c = c0 + tid, c2 = c + kThreads;
    const bool in = c >= 0 && c < nch;
    t.a.x = in ? a.code[c] : 0u;
    t.a.y = in ? (uint32_t)a.inv[c] : 0xFFFFu;
    if (tid < kChunks - kThreads) {
      const bool in2 = c2 >= 0 && c2 < nch;
      t.b.x = in2 ? a.code[c2] : 0u;
      t.b.y = in2 ? (uint32_t)a.inv[c2] : 0xFFFFu;
    }
  } else {
    const int64_t r0 = (int64_t)(tile * kTile) - 16;
    t.a = load_chunk(a.stream, a.nbytes, r0 + 16 * (int64_t)tid);
    if (tid < kChunks - kThreads) t.b = load_chunk(a.stream, a.nbytes, r0 + 16 * (int64_t)(tid + kThreads));
  }
}
// ... -> the tile's code words and invalid flags in LDS
template <bool PK> __device__ __forceinline__ void tile_stage(const StreamArgs &a, const TileSrc &t, int tid, uint32_t *s_code, uint32_t *s_inv)
{
  uint32_t code, inv;
  (void)a;
  if (PK) { code = t.a.x; inv = t.a.y; } else encode_words(t.a, code, inv);
  s_code[tid] = code;
  reinterpret_cast<uint16_t *>(s_inv)[tid ^ 1] = (uint16_t)inv;
  if (tid < kChunks - kThreads) {
    if (PK) { code = t.b.x; inv = t.b.y; } else encode_words(t.b, code, inv);
    s_code[tid + kThreads] = code;
    reinterpret_cast<uint16_t *>(s_inv)[(tid + kThreads) ^ 1] = (uint16_t)inv;
  }
}

// ASCII stream -> packed form on the device (one thread per 16 positions; positions >= nbytes invalid)
__global__ void k_pack_stream(const uint8_t *stream, uint64_t nbytes, uint32_t *code_out, uint16_t *inv_out)
{
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c * 16 >= nbytes) return;
  uint32_t code, inv;
  encode_words(load_chunk(stream, nbytes, (int64_t)(c * 16)), code, inv);
  code_out[c] = code;
  inv_out[c] = (uint16_t)inv;
}

// One k-mer occurrence produced by the front end
template <int W> struct Occ {
  Kmer<W> key;
  uint32_t e;
};

template <int W, bool ONECOL>
__device__ __forceinline__ void flush_batch(const InsertSink<W, ONECOL> &sink, const Occ<W> (&occ)[kBatch],
                                            const bool (&ov)[kBatch], uint32_t &novel, uint32_t &full)
{
  uint64_t slot[kBatch], cur[kBatch], hint[kBatch];
#pragma unroll
  for (int i = 0; i < kBatch; i++) {  // issue all first probes before looking at any
    cur[i] = 0; hint[i] = 0; slot[i] = 0;
    if (ov[i]) {
      slot[i] = key_slot<W>(sink.t, occ[i].key);
      const uint64_t *r = key_ptr_t<W, ONECOL>(sink.t, slot[i]);
      if (W == 1 && ONECOL) {
        const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(r);
        cur[i] = kv.x; hint[i] = kv.y;
      } else {
        cur[i] = r[0];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kBatch; i++)
    if (ov[i]) probe_insert<W, ONECOL>(sink.t, occ[i].key, slot[i], cur[i], hint[i], occ[i].e, sink.col, novel, full);
}

__device__ __forceinline__ uint32_t owner_of(uint32_t h2, uint32_t nparts)
{
  return (uint32_t)(((uint64_t)h2 * nparts) >> 32);
}

// Keys of three and four words (k = 65 .. 127: mccortex95 / mccortex127): which of a lane's 16 positions start a k-mer of
// valid bases (ok16, position j at bit 15 - j) and which have a valid base behind that k-mer (nok16), from the invalid
// flags of bases pl .. pl + 191 -- the OR of k shifted copies, built by doubling, as the binning kernels do for one
// and two words (mcx_defer.h).
__device__ __forceinline__ void lane_masks_wide(const uint32_t *s_inv, uint32_t pl, int k, uint32_t &ok16, uint32_t &nok16)
{
  const uint64_t V0 = inv_win64(s_inv, pl), V1 = inv_win64(s_inv, pl + 64), V2 = inv_win64(s_inv, pl + 128);
  uint64_t M0 = V0, M1 = V1, M2 = V2;
  for (int c = 1; c < k;) {  // uniform
    const int s = min(c, k - c);
    if (s < 64) {
      M0 |= (M0 << s) | (M1 >> (64 - s));
      M1 |= (M1 << s) | (M2 >> (64 - s));
      M2 |= M2 << s;
    } else {  // (s = 64 only for k = 127 .. : a whole word, then the rest)
      const int t = s - 64;
      M0 |= t ? (M1 << t) | (M2 >> (64 - t)) : M1;
      M1 |= M2 << t;
    }
    c += s;
  }
  ok16 = ~(uint32_t)(M0 >> 48) & 0xFFFFu;
  // base j + k: bit j + k of the window, counted from the top
  const int wi = k >> 6, sh = k & 63;
  const uint64_t a = wi == 0 ? V0 : V1, b = wi == 0 ? V1 : V2;
  const uint64_t nx = sh ? (a << sh) | (b >> (64 - sh)) : a;
  nok16 = ~(uint32_t)(nx >> 48) & 0xFFFFu;
}

// Fused build kernel: k-merise a stream tile by tile and insert straight into the table
// (direct path; the deferred path of mcx_defer.h shares the front end helpers).
template <int W, bool ONECOL, bool PK>
__global__ __launch_bounds__(kThreads, 1) void k_stream(StreamArgs a, InsertSink<W, ONECOL> isink)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) table_mark_written(isink.t);
  __shared__ uint32_t s_code[kChunks + 4];
  __shared__ uint32_t s_inv[kChunks / 2 + 4];

  const int tid = threadIdx.x;
  const int k = a.k;
  uint32_t n_kmers = 0, n_contigs = 0, n_novel = 0, full = 0;

  const uint64_t top_mask = ~0ULL >> (64 * W - 2 * k);
  const int first_shift = 2 * k - 2 - 64 * (W - 1);  // bit position of base 0 in w[0]

  for (uint64_t tile = a.tile0 + blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();
    {
      TileSrc ts;
      tile_fetch<PK>(a, tile, tid, ts);
      tile_stage<PK>(a, ts, tid, s_code, s_inv);
    }
    if (tid < 4) { s_code[kChunks + tid] = 0; s_inv[kChunks / 2 + tid] = 0xFFFFFFFFu; }
    __syncthreads();

    const uint32_t pl = 16u * (uint32_t)(tid + 1);  // region index of this lane's first position
    const uint64_t Vh = inv_win64(s_inv, pl);
    const uint32_t prev_chunk_inv = (s_inv[(pl - 1) >> 5] >> (31 - ((pl - 1) & 31))) & 1u;

    // positions of this lane owned by the launch: j in [j_lo, j_hi)
    const uint64_t P0 = tile * kTile + 16ull * (uint64_t)tid;
    const int j_lo = a.pos_lo > P0 ? (int)min((uint64_t)kPosPerLane, a.pos_lo - P0) : 0;
    const int j_hi = a.pos_hi > P0 ? (int)min((uint64_t)kPosPerLane, a.pos_hi - P0) : 0;

    if constexpr (W > 2) {
      // three- and four-word keys: the same walk with the validity masks of lane_masks_wide and word loops
      uint32_t ok16, nok16;
      lane_masks_wide(s_inv, pl, k, ok16, nok16);
      ok16 &= ((0x10000u >> j_lo) - 1u) & ~((0x10000u >> j_hi) - 1u);
      const uint32_t pok16 = ~((prev_chunk_inv << 15) | (uint32_t)(Vh >> 49)) & 0xFFFFu;
      Occ<W> occ[kBatch];
      bool ov[kBatch];
      if (ok16) {
        Kmer<W> fw, rc;
        const int topb = k - 32 * (W - 1);  // bases in the top word
        fw.w[0] = code_win64(s_code, pl) >> (64 - 2 * topb);
        for (int i = 1; i < W; i++) fw.w[i] = code_win64(s_code, pl + (uint32_t)(topb + 32 * (i - 1)));
        rc = revcomp<W>(fw, k);
        const uint64_t feed = code_win64(s_code, pl + (uint32_t)k);
        const int fs = 2 * topb - 2;  // bit position of base 0 in w[0]
        uint32_t prev_nuc = s_code[(pl - 1) >> 4] & 3u;
#pragma unroll
        for (int j = 0; j < kPosPerLane; j++) {
          const bool valid = (ok16 >> (15 - j)) & 1u;
          const bool next_ok = (nok16 >> (15 - j)) & 1u, prev_ok = (pok16 >> (15 - j)) & 1u;
          const uint32_t nuc_next = (uint32_t)(feed >> (62 - 2 * j)) & 3u;
          ov[j % kBatch] = valid;
          if (valid) {
            uint32_t o;
            Occ<W> &x = occ[j % kBatch];
            x.key = canonical<W>(fw, rc, o);
            uint32_t e = 0;
            if (next_ok) e |= 1u << (nuc_next + 4u * o);
            if (prev_ok) e |= 1u << ((3u - prev_nuc) + 4u * (1u - o));
            x.e = e;
            n_kmers++;
            n_contigs += prev_ok ? 0u : 1u;
          }
          if (j % kBatch == kBatch - 1) flush_batch<W, ONECOL>(isink, occ, ov, n_novel, full);
          prev_nuc = kmer_first_base<W>(fw, k);
          kmer_push<W>(fw, nuc_next, k);
          for (int i = W - 1; i >= 1; i--) rc.w[i] = (rc.w[i] >> 2) | (rc.w[i - 1] << 62);
          rc.w[0] = (rc.w[0] >> 2) | ((uint64_t)(3u - nuc_next) << fs);
        }
      }
      continue;
    }
    const uint64_t Vl = (W == 2) ? inv_win64(s_inv, pl + 64) : 0;
    // any valid k-mer among this lane's 16 positions?
    bool any = false;
#pragma unroll
    for (int j = 0; j < kPosPerLane; j++) {
      const uint64_t Th = (W == 2 && j) ? ((Vh << j) | (Vl >> (64 - j))) : (Vh << j);
      any |= ((Th >> (64 - k)) == 0) & (j >= j_lo) & (j < j_hi);
    }

    Occ<W> occ[kBatch];
    bool ov[kBatch];
    // (all indices into occ/ov are compile-time constants after unrolling: no scratch)
    {
      if (any) {
        Kmer<W> fw, rc;
        if (W == 1) {
          fw.w[0] = code_win64(s_code, pl) >> (64 - 2 * k);
        } else {
          const uint64_t hi = code_win64(s_code, pl), lo = code_win64(s_code, pl + 32);
          const int s = 128 - 2 * k;
          fw.w[0] = hi >> s;
          fw.w[W - 1] = (lo >> s) | (hi << (64 - s));
        }
        rc = revcomp<W>(fw, k);
        const uint64_t feed = code_win64(s_code, pl + (uint32_t)k);
        uint32_t prev_nuc = s_code[(pl - 1) >> 4] & 3u;
#pragma unroll
        for (int j = 0; j < kPosPerLane; j++) {
          const uint64_t Th = (W == 2 && j) ? ((Vh << j) | (Vl >> (64 - j))) : (Vh << j);
          const bool valid = ((Th >> (64 - k)) == 0) & (j >= j_lo) & (j < j_hi);
          const bool next_ok = ((Th >> (63 - k)) & 1ULL) == 0;
          const bool prev_ok = (j == 0) ? (prev_chunk_inv == 0) : (((Vh >> (64 - j)) & 1ULL) == 0);
          const uint32_t nuc_next = (uint32_t)(feed >> (62 - 2 * j)) & 3u;
          ov[j % kBatch] = valid;
          if (valid) {
            uint32_t o;
            Occ<W> &x = occ[j % kBatch];
            x.key = canonical<W>(fw, rc, o);
            uint32_t e = 0;
            if (next_ok) e |= 1u << (nuc_next + 4u * o);
            if (prev_ok) e |= 1u << ((3u - prev_nuc) + 4u * (1u - o));
            x.e = e;
            n_kmers++;
            n_contigs += prev_ok ? 0u : 1u;
          }
          if (j % kBatch == kBatch - 1) flush_batch<W, ONECOL>(isink, occ, ov, n_novel, full);
          // roll to position j+1
          prev_nuc = (uint32_t)(fw.w[0] >> first_shift) & 3u;
          if (W == 1) {
            fw.w[0] = ((fw.w[0] << 2) | nuc_next) & top_mask;
            rc.w[0] = (rc.w[0] >> 2) | ((uint64_t)(3u - nuc_next) << first_shift);
          } else {
            fw.w[0] = ((fw.w[0] << 2) | (fw.w[W - 1] >> 62)) & top_mask;
            fw.w[W - 1] = (fw.w[W - 1] << 2) | nuc_next;
            rc.w[W - 1] = (rc.w[W - 1] >> 2) | (rc.w[0] << 62);
            rc.w[0] = (rc.w[0] >> 2) | ((uint64_t)(3u - nuc_next) << first_shift);
          }
        }
      }
    }
  }

  // block-level reduction of the statistics: one atomic per counter per wave
  // (hipcc folds the per-lane atomicAdd into a wave reduction)
  if (n_kmers) atomicAdd(&a.ctr->kmers, (unsigned long long)n_kmers);
  if (n_contigs) atomicAdd(&a.ctr->contigs, (unsigned long long)n_contigs);
  if (n_novel) atomicAdd(&a.ctr->novel, (unsigned long long)n_novel);
  if (full == 1) a.ctr->full = 1;
  if (a.flag && n_contigs) *a.flag = 1;
}

// ---------------------------------------------------------------------------
// Tuple insert (owner side of the sharded build)
// ---------------------------------------------------------------------------
template <int W, bool ONECOL>
__global__ __launch_bounds__(kThreads) void k_insert_tuples(InsertSink<W, ONECOL> sink, const uint64_t *keys,
                                                            const uint8_t *edges, uint64_t n, Counters *ctr,
                                                            uint32_t only_own /* skip the keys of other shards */)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) table_mark_written(sink.t);
  uint32_t n_novel = 0, full = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kThreads;
  for (uint64_t i0 = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i0 < n; i0 += stride * kBatch) {
    Occ<W> occ[kBatch];
    bool ov[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; b++) {
      const uint64_t i = i0 + (uint64_t)b * stride;
      ov[b] = i < n;
      if (i < n) {
        Occ<W> &x = occ[b];
        for (int w = 0; w < W; w++) x.key.w[w] = keys[i * W + w];
        x.e = edges[i];
        if (only_own && key_owner<W>(sink.t, x.key) != sink.t.part) ov[b] = false;
      }
    }
    flush_batch<W, ONECOL>(sink, occ, ov, n_novel, full);
  }
  if (n_novel) atomicAdd(&ctr->novel, (unsigned long long)n_novel);
  if (full) ctr->full = 1;
}

// ---------------------------------------------------------------------------
// Per-read statistics (good/bad reads: build_graph.c:186-188)
// ---------------------------------------------------------------------------
// One lane per read; a read is good iff it holds an ACGT run of >= k bases.
// stream_off[i] is the start of read i in the separator-delimited stream and
// stream_off[i+1]-1 its separator.
__global__ void k_read_flags(const uint8_t *stream, const uint64_t *stream_off, uint64_t nreads, int k,
                             unsigned char *flags)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nreads) return;
  const uint64_t b = stream_off[i], e = stream_off[i + 1] - 1;
  int run = 0;
  bool ok = false;
  for (uint64_t p = b; p < e && !ok; p++) {
    run = base_valid(stream[p]) ? run + 1 : 0;
    ok = run >= k;
  }
  flags[i] = ok ? 1 : 0;
}

// the same on the packed form of the stream: invalid flag of position p = bit 15 - (p & 15) of inv[p >> 4]
__global__ void k_read_flags_packed(const uint16_t *inv, const uint64_t *stream_off, uint64_t nreads, int k, unsigned char *flags)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nreads) return;
  const uint64_t b = stream_off[i], e = stream_off[i + 1] - 1;
  int run = 0;
  bool ok = false;
  for (uint64_t p = b; p < e && !ok; p++) {
    run = ((inv[p >> 4] >> (15 - (p & 15))) & 1) ? 0 : run + 1;
    ok = run >= k;
  }
  flags[i] = ok ? 1 : 0;
}

__global__ void k_count_flags(const unsigned char *flags, uint64_t n, Counters *ctr)
{
  unsigned long long good = 0, bad = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (flags[i]) good++; else bad++;
  }
  if (good) atomicAdd(&ctr->good_reads, good);
  if (bad) atomicAdd(&ctr->bad_reads, bad);
}

// -Q/-H path: a read is good iff it produced at least one contig byte
__global__ void k_count_sizes(const uint64_t *sizes, const uint8_t *keep, uint64_t n, Counters *ctr)
{
  unsigned long long good = 0, bad = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (keep && !keep[i]) continue;  // a filtered duplicate is neither (load_read never sees it)
    if (sizes[i]) good++; else bad++;
  }
  if (good) atomicAdd(&ctr->good_reads, good);
  if (bad) atomicAdd(&ctr->bad_reads, bad);
}

// ---------------------------------------------------------------------------
// Quality / homopolymer contig splitting (seq_reader.c:61-172), -Q / -H only
// ---------------------------------------------------------------------------
// One lane per read walks the read exactly as load_read does
// (build_graph.c:154-189) and emits its contigs, '\n'-separated, into a new
// stream that the ordinary front end then consumes.  pass 0 sizes, pass 1 writes.
__device__ inline uint64_t qh_contig_start(const uint8_t *seq, uint64_t len, const uint8_t *qual,
                                           uint64_t offset, uint64_t k, uint32_t qcut, uint32_t hcut)
{
  uint64_t pos = offset, kend;
  while ((kend = pos + k) <= len) {
    uint64_t i = kend;
    while (i > pos && base_valid(seq[i - 1])) i--;
    if (i > pos) { pos = i; continue; }
    if (qual && qcut > 0) {
      i = kend;
      while (i > pos && (int)(int8_t)qual[i - 1] > (int)qcut) i--;
      if (i > pos) { pos = i; continue; }
    }
    if (hcut > 0) {
      uint64_t run = 1;
      for (i = kend - 1; i > pos; i--) {
        if (seq[i - 1] == seq[i]) { run++; if (run == hcut) break; }
        else run = 1;
      }
      if (i > pos) { pos = i; continue; }
    }
    return pos;
  }
  return len;
}

__device__ inline uint64_t qh_contig_end(const uint8_t *seq, uint64_t len, const uint8_t *qual,
                                         uint64_t cstart, uint64_t k, uint32_t qcut, uint32_t hcut,
                                         uint64_t *search)
{
  uint64_t end = cstart + k, hp = 1;
  if (hcut > 0) while (hp < end && seq[end - 1 - hp] == seq[end - 1]) hp++;
  for (; end < len; end++) {
    if (!base_valid(seq[end]) || (qual && (int)(int8_t)qual[end] < (int)qcut)) break;
    if (hcut > 0) {
      if (seq[end] == seq[end - 1]) { hp++; if (hp >= hcut) break; }
      else hp = 1;
    }
  }
  *search = (hcut > 0 && hp >= hcut) ? end - hcut + 1 : end;
  return end;
}

// (--remove-pcr: mates 2i / 2i + 1 carry their own cutoff, pmask = 1; reads with keep[r] == 0 were
// filtered out and emit nothing)
__global__ void k_qh_contigs(const uint8_t *bases, const uint8_t *quals, const uint64_t *off, uint64_t nreads,
                             int k, uint32_t qcut1, uint32_t qcut2, uint32_t pmask, uint32_t hcut,
                             const uint8_t *keep, int pass, uint64_t *out_sizes,
                             const uint64_t *out_off, uint8_t *out_stream)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nreads) return;
  if (keep && !keep[r]) { if (!pass) out_sizes[r] = 0; return; }
  const uint32_t qcut = ((uint32_t)r & pmask) ? qcut2 : qcut1;
  const uint8_t *seq = bases + off[r];
  const uint8_t *qual = (quals && qcut > 0) ? quals + off[r] : nullptr;
  const uint64_t len = off[r + 1] - off[r];
  uint64_t cs, ce, search = 0, w = pass ? out_off[r] : 0;
  while ((cs = qh_contig_start(seq, len, qual, search, (uint64_t)k, qcut, hcut)) < len) {
    ce = qh_contig_end(seq, len, qual, cs, (uint64_t)k, qcut, hcut, &search);
    if (pass) {
      for (uint64_t i = cs; i < ce; i++) out_stream[w + (i - cs)] = seq[i];
      out_stream[w + (ce - cs)] = '\n';
    }
    w += ce - cs + 1;
  }
  if (!pass) out_sizes[r] = w;
}

// ---------------------------------------------------------------------------
// build --remove-pcr: seq_reads_are_novel (src/tools/build_graph.c:28-92)
// ---------------------------------------------------------------------------
// The reference keeps a "read starts here" bit per (node, orientation) and drops a read (pair)
// whose first k-mer(s) all carry the bit already; otherwise it sets the bits and loads the
// read(s).  A read that meets a node whose bit is clear is always kept, so the bit of a node is
// set by the FIRST read (pair) of the input that starts at it: with T(n) = the index of that
// read, read i is a duplicate iff T(n) < i for each of its start nodes.  T is an atomicMin, which
// makes the filter a parallel one that gives exactly the result of the reference walking the
// reads in input order on one thread.  `first[2 * slot + orient]`: 0 = set by an earlier batch,
// 0xFFFFFFFF = clear, else 1 + the index (in this batch) of the first read (pair) that starts there.

// seq_reader_orient_mp_FF (src/basic/seq_reader.c:506-510): r1 is reverse-complemented when
// matedir & 2, r2 when matedir & 1 (cortex_types.h:18-25) -- in place, qualities reversed.
__global__ void k_pcr_orient(uint8_t *bases, uint8_t *quals, const uint64_t *off, uint64_t nreads, uint32_t pmask, uint32_t matedir)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nreads) return;
  const bool flip = ((uint32_t)r & pmask) ? (matedir & 1u) : (matedir & 2u);
  if (!flip) return;
  uint8_t *s = bases + off[r];
  uint8_t *q = quals ? quals + off[r] : nullptr;
  const uint64_t len = off[r + 1] - off[r];
  auto comp = [](uint8_t c) -> uint8_t {  // A<->T, C<->G keeping the case; anything else stays
    return base_valid(c) ? (uint8_t)(c ^ (((c & 0x1f) == 1 || (c & 0x1f) == 20) ? 0x15 : 0x04)) : c;
  };
  for (uint64_t i = 0; i < len - i; i++) {
    const uint64_t j = len - 1 - i;
    const uint8_t a = comp(s[i]), b = comp(s[j]);
    s[i] = b; s[j] = a;  // (i == j: complemented once)
    if (q) { const uint8_t t = q[i]; q[i] = q[j]; q[j] = t; }
  }
}

constexpr uint64_t kNoNode = ~0ULL;
constexpr uint64_t kForeignNode = ~0ULL - 1;  // the read's start k-mer belongs to another shard of a multi-GPU table
// one lane per read: first k-mer of its first contig -> node (created if new, coverage untouched:
// db_graph_find_or_add_node_mt, build_graph.c:66-76) and T(node) = min(T(node), 1 + read (pair) index)
template <int W>
__global__ void k_pcr_starts(TableView t, const uint8_t *bases, const uint8_t *quals, const uint64_t *off, uint64_t nreads,
                             int k, uint32_t qcut1, uint32_t qcut2, uint32_t pmask, uint32_t hcut,
                             uint32_t *first, uint64_t *node_of, Counters *ctr, OwnerSpec os)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) table_mark_written(t);
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nreads) return;
  const uint32_t qcut = ((uint32_t)r & pmask) ? qcut2 : qcut1;
  const uint8_t *seq = bases + off[r];
  const uint8_t *qual = (quals && qcut > 0) ? quals + off[r] : nullptr;
  const uint64_t len = off[r + 1] - off[r];
  const uint64_t cs = qh_contig_start(seq, len, qual, 0, (uint64_t)k, qcut, hcut);
  uint64_t node = kNoNode;
  if (cs < len) {
    Kmer<W> fw;
    for (int w = 0; w < W; w++) fw.w[w] = 0;
    for (uint64_t i = cs; i < cs + (uint64_t)k; i++) kmer_push<W>(fw, ((seq[i] >> 1) ^ (seq[i] >> 2)) & 3u, k);
    const Kmer<W> rc = revcomp<W>(fw, k);
    uint32_t o, novel = 0, full = 0;
    const Kmer<W> key = canonical<W>(fw, rc, o);
    if (!owner_is_me<W>(t, os, key)) { node_of[r] = kForeignNode; return; }
    const uint64_t slot = find_or_insert_rec<W>(t, key, false, novel, full);
    if (novel) atomicAdd(&ctr->novel, 1ULL);
    if (full) atomicAdd(&ctr->full, 1ULL);
    if (slot != kNoSlot) {
      node = 2 * slot + o;
      atomicMin(first + node, (uint32_t)(r >> (pmask ? 1 : 0)) + 1u);
    }
  }
  node_of[r] = node;
}

// one lane per read (pair): duplicate iff every start node it has was claimed by an earlier read
// (pair) (build_graph.c:78-84; a read (pair) without any k-mer counts as a duplicate there too)
__global__ void k_pcr_decide(const uint64_t *node_of, const uint32_t *first, uint64_t nunits, uint32_t pmask,
                             uint8_t *keep, unsigned long long *ndup)
{
  const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long dup = 0;
  if (u < nunits) {
    const uint32_t me = (uint32_t)u + 1u;
    const uint64_t r0 = pmask ? 2 * u : u;
    const uint64_t n1 = node_of[r0], n2 = pmask ? node_of[r0 + 1] : kNoNode;
    dup = (n1 == kNoNode || first[n1] < me) && (n2 == kNoNode || first[n2] < me);
    keep[r0] = (uint8_t)!dup;
    if (pmask) keep[r0 + 1] = (uint8_t)!dup;
  }
  block_add(ndup, dup);
}

// the start nodes of this batch now carry the bit for every later batch
__global__ void k_pcr_commit(const uint64_t *node_of, uint64_t nreads, uint32_t *first)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nreads && node_of[r] < kForeignNode) first[node_of[r]] = 0;
}

// Multi-GPU table: every shard runs k_pcr_starts over ALL reads of the batch and answers for the
// start nodes it owns: claim[r] = 1 "read r's start node was claimed by an earlier read (pair), or
// r has no k-mer", 0 "r is the first to start there", 2 "not my node".  Exactly one shard answers
// 0 / 1 for a read with a k-mer; a read without one gets 1 from all of them.
__global__ void k_pcr_claim(const uint64_t *node_of, const uint32_t *first, uint64_t nreads, uint32_t pmask, uint8_t *claim)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nreads) return;
  const uint32_t me = (uint32_t)(r >> (pmask ? 1 : 0)) + 1u;
  const uint64_t n = node_of[r];
  claim[r] = n == kForeignNode ? 2 : (uint8_t)(n == kNoNode || first[n] < me);
}
// ... and, with the claims of all shards side by side (claims[shard * nreads + r]), decides like
// k_pcr_decide.  Every shard reaches the same verdict; it KEEPS the units [u_lo, u_hi) only: the
// kept reads of a batch are dealt out to the shards for the ordinary (sharded) insert.
__global__ void k_pcr_decide_claims(const uint8_t *claims, uint32_t nshards, uint64_t nreads, uint64_t nunits, uint32_t pmask,
                                    uint64_t u_lo, uint64_t u_hi, uint8_t *keep, unsigned long long *ndup)
{
  const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long dup = 0;
  if (u < nunits) {
    const uint64_t r0 = pmask ? 2 * u : u;
    uint32_t c1 = 2, c2 = pmask ? 2 : 1;
    for (uint32_t s = 0; s < nshards; s++) {
      c1 = min(c1, (uint32_t)claims[(uint64_t)s * nreads + r0]);
      if (pmask) c2 = min(c2, (uint32_t)claims[(uint64_t)s * nreads + r0 + 1]);
    }
    dup = c1 && c2;
    const uint8_t k = (uint8_t)(!dup && u >= u_lo && u < u_hi);
    keep[r0] = k;
    if (pmask) keep[r0 + 1] = k;
  }
  block_add(ndup, dup);
}


// ---------------------------------------------------------------------------
// Order-independent checksum of the graph: sum over nodes of a 64-bit mix of the exported record
// (key words, per colour min(coverage, 2^32-1) and edges).  Two tables hold the same graph iff
// their exports are equal as sets; the checksum lets that be tested at sizes where comparing
// exports byte by byte is impractical (tests/test_gpu_fullsize.py).
// ---------------------------------------------------------------------------
__host__ __device__ inline uint64_t mix64(uint64_t x)
{  // splitmix64 finaliser
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27; x *= 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t record_hash(const uint64_t *key_words, int W, const uint32_t *covgs, const uint8_t *edges, uint32_t ncols)
{
  uint64_t h = 0x9E3779B97F4A7C15ULL;
  for (int w = 0; w < W; w++) h = mix64(h ^ key_words[w]);
  for (uint32_t c = 0; c < ncols; c++) h = mix64(h ^ (((uint64_t)covgs[c] << 8) | edges[c]) ^ ((uint64_t)(c + 1) << 48));
  return h;
}

__global__ __launch_bounds__(256) void k_checksum(TableView t, uint32_t W, uint32_t ncols, unsigned long long *out)
{
  unsigned long long acc = 0, cnt = 0;
  for (uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; slot < t.nslots; slot += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t *r = key_ptr(t, slot);
    if (!(r[0] & kFlag)) continue;
    uint64_t h = 0x9E3779B97F4A7C15ULL;
    h = mix64(h ^ (r[0] & kKeyMask));
    for (uint32_t w = 1; w < W; w++) h = mix64(h ^ r[w]);
    for (uint32_t c = 0; c < ncols; c++) {
      const uint64_t v = *val_ptr(t, slot, c);
      uint64_t cv = v >> 8;
      if (cv > 0xFFFFFFFFull) cv = 0xFFFFFFFFull;
      h = mix64(h ^ ((cv << 8) | (v & 0xff)) ^ ((uint64_t)(c + 1) << 48));
    }
    acc += h; cnt++;
  }
  block_add(&out[0], acc);
  block_add(&out[1], cnt);
}

// ---------------------------------------------------------------------------
// build --intersect: reads only update k-mers that are already in the graph
// ---------------------------------------------------------------------------
// One lane per read walks it exactly as load_read + build_graph_from_str_mt do with
// must_exist_in_graph (src/tools/build_graph.c:99-150,154-189): every k-mer of every contig is
// looked up; a found k-mer gets coverage +1; an edge is added between consecutive k-mers only
// when BOTH were found.  Not a fast path (intersection builds are small and rare): clarity first.
// Table split over several GPUs (mcx_multi.h: grp_add_reads_must_exist): consecutive k-mers of a read
// live on different shards, and the edge rule needs to know whether BOTH were found.  Every shard
// walks all reads twice.  phase 1: for the k-mers it owns, present[index of the k-mer's last base] =
// "found in my table" (no updates); the shards' arrays are OR-ed together (each byte is written by
// exactly one shard).  phase 2, with the merged array: the reference's rule, each shard updating
// only the nodes it owns; shard 0 keeps the read / contig / k-mer statistics.  phase 0: one table.
template <int W>
__global__ void k_reads_must_exist(TableView t, const uint8_t *bases, const uint8_t *quals, const uint64_t *off,
                                   uint64_t nreads, int k, uint32_t qcut, uint32_t hcut, uint32_t col, Counters *ctr,
                                   uint8_t *present, uint32_t phase, OwnerSpec os)
{
  if (phase != 1 && blockIdx.x == 0 && threadIdx.x == 0) table_mark_written(t);
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nreads) return;
  const uint8_t *seq = bases + off[r];
  const uint8_t *qual = (quals && qcut > 0) ? quals + off[r] : nullptr;
  const uint64_t len = off[r + 1] - off[r];
  unsigned long long n_kmers = 0, n_contigs = 0, n_absent = 0;
  uint32_t dummy_novel = 0, full = 0;
  uint64_t cs, ce, search = 0;
  while ((cs = qh_contig_start(seq, len, qual, search, (uint64_t)k, qcut, hcut)) < len) {
    ce = qh_contig_end(seq, len, qual, cs, (uint64_t)k, qcut, hcut, &search);
    Kmer<W> fw;
    for (int w = 0; w < W; w++) fw.w[w] = 0;
    uint64_t prev_slot = kNoSlot;
    uint32_t prev_o = 0, prev_first = 0;
    for (uint64_t i = cs; i < ce; i++) {
      const uint32_t nuc = ((seq[i] >> 1) ^ (seq[i] >> 2)) & 3u;  // A C G T (either case) -> 0 1 2 3
      kmer_push<W>(fw, nuc, k);
      if (i + 1 < cs + (uint64_t)k) continue;  // first k - 1 bases
      const Kmer<W> rc = revcomp<W>(fw, k);
      uint32_t o;
      const Kmer<W> key = canonical<W>(fw, rc, o);
      const uint32_t first = kmer_first_base<W>(fw, k);  // first base of this k-mer, read strand
      if (phase == 0) {
        const uint64_t slot = find_or_insert_rec<W>(t, key, true, dummy_novel, full);
        if (slot != kNoSlot) {
          __hip_atomic_fetch_add(val_ptr(t, slot, col), 256ULL, MCX_RLX, MCX_AGENT);
          if (prev_slot != kNoSlot) {  // db_graph_add_edge_mt(prev, curr): db_graph.c:152-166
            __hip_atomic_fetch_or(val_ptr(t, prev_slot, col), (uint64_t)(1u << (nuc + 4u * prev_o)), MCX_RLX, MCX_AGENT);
            __hip_atomic_fetch_or(val_ptr(t, slot, col), (uint64_t)(1u << ((3u - prev_first) + 4u * (1u - o))), MCX_RLX, MCX_AGENT);
          }
        }
        prev_slot = slot; prev_o = o; prev_first = first;
        n_kmers++;
        n_absent += slot == kNoSlot;
      } else {
        const uint64_t gi = off[r] + i;  // the k-mer's last base: one byte of `present` per k-mer occurrence
        const bool mine = owner_is_me<W>(t, os, key);
        if (phase == 1) {
          if (mine) present[gi] = find_or_insert_rec<W>(t, key, true, dummy_novel, full) != kNoSlot;
        } else {
          const bool found = present[gi] != 0;
          // prev_slot: kNoSlot = the previous k-mer was not found; kNoSlot - 1 = found, on another shard
          uint64_t slot = kNoSlot;
          if (found) slot = mine ? find_or_insert_rec<W>(t, key, true, dummy_novel, full) : kNoSlot - 1;
          if (found && prev_slot != kNoSlot) {
            if (prev_slot != kNoSlot - 1)
              __hip_atomic_fetch_or(val_ptr(t, prev_slot, col), (uint64_t)(1u << (nuc + 4u * prev_o)), MCX_RLX, MCX_AGENT);
            if (mine && slot != kNoSlot)
              __hip_atomic_fetch_or(val_ptr(t, slot, col), (uint64_t)(1u << ((3u - prev_first) + 4u * (1u - o))), MCX_RLX, MCX_AGENT);
          }
          if (found && mine && slot != kNoSlot) __hip_atomic_fetch_add(val_ptr(t, slot, col), 256ULL, MCX_RLX, MCX_AGENT);
          prev_slot = slot; prev_o = o; prev_first = first;
          n_kmers++;
          n_absent += !found;
        }
      }
    }
    n_contigs++;
  }
  if (phase == 1 || (phase == 2 && (os.mode == 2 ? os.part : t.part) != 0)) return;  // (statistics: once)
  if (n_kmers) atomicAdd(&ctr->kmers, n_kmers);
  if (n_absent) atomicAdd(&ctr->absent, n_absent);
  if (n_contigs) atomicAdd(&ctr->contigs, n_contigs);
  atomicAdd(n_contigs ? &ctr->good_reads : &ctr->bad_reads, 1ULL);
}

// dst |= src, byte arrays (the shards' `present` arrays of k_reads_must_exist)
__global__ void k_or_bytes(uint8_t *dst, const uint8_t *src, uint64_t n)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] |= src[i];
}

// db_graph_remove_no_covg_kmers + db_graph_intersect_edges (src/graph/db_graph.c:630-673): drop
// k-mers without coverage in any visible colour (the slot becomes a tombstone: non-zero, flag
// clear) and AND every colour's edges with the intersection graph's edges, kept in colour `hidden`
__global__ __launch_bounds__(256) void k_intersect_finish(TableView t, uint32_t ncols_vis, uint32_t hidden, Counters *ctr,
                                                          unsigned long long *removed)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) table_mark_written(t);
  unsigned long long gone = 0;
  for (uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; slot < t.nslots; slot += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t *r = key_ptr(t, slot);
    if (!(r[0] & kFlag)) continue;
    const uint64_t mask = *val_ptr(t, slot, hidden) & 0xffULL;
    uint64_t any = 0;
    for (uint32_t c = 0; c < ncols_vis; c++) {
      uint64_t *v = val_ptr(t, slot, c);
      const uint64_t x = *v;
      any |= x >> 8;
      *v = (x & ~0xffULL) | (x & mask);
    }
    if (!any) { r[0] = kPending; gone++; }
  }
  if (gone) { atomicAdd(removed, gone); atomicAdd(&ctr->novel, 0ULL - gone); }
}

// ---------------------------------------------------------------------------
// Export: compact occupied slots
// ---------------------------------------------------------------------------
// Pass over the table; every occupied slot appends (key words, slot index).  A lane looks at 16
// consecutive slots and the wave reserves its output range with ONE atomic on the cursor (all
// waves hit the same address: ~12 ns each, so one atomic per 64 slots made the 2^29-slot scan take
// 100 ms; per 1024 slots it is 6 ms).
template <int W>
// (prefix, pbits): only keys whose top `pbits` bits (of the 2k-bit k-mer) equal `prefix` -- the export
// walks the key space in ranges when its scratch arrays would not fit beside the table.
__global__ __launch_bounds__(kThreads) void k_compact(TableView t, uint64_t *key0, uint64_t *key1, uint64_t *slot_out,
                                                      unsigned long long *cursor, uint64_t cap_out, int kmer_size,
                                                      uint32_t prefix, uint32_t pbits)
{
  constexpr int PER = 16;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t stride = (uint64_t)gridDim.x * kThreads * PER;
  // nslots is a multiple of 2048 (sub-table size) and of the 4096 slots a block covers per round
  // only when it is a multiple of kThreads * PER: the tail is guarded per slot
  for (uint64_t s0 = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) * PER; s0 - (uint64_t)lane * PER < t.nslots; s0 += stride) {
    uint64_t w0[PER];
    uint32_t occ = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
      w0[i] = (s0 + i < t.nslots) ? *key_ptr(t, s0 + i) : 0;
      bool take = (w0[i] & kFlag) != 0;
      if (take && pbits) {
        const uint64_t kw = w0[i] & kKeyMask;
        const int nb0 = 2 * kmer_size - 64 * (W - 1);  // key bits in the top word
        uint64_t top;
        if ((int)pbits <= nb0) top = kw >> (nb0 - (int)pbits);
        else top = (kw << ((int)pbits - nb0)) | (key_ptr(t, s0 + i)[1] >> (64 - ((int)pbits - nb0)));  // (pbits <= 30: the top word and the next)
        take = top == prefix;
      }
      occ |= (uint32_t)take << i;
    }
    const uint32_t cnt = __popc(occ);
    uint32_t incl = cnt;  // inclusive prefix sum over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d) incl += y;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    if (total == 0) continue;  // uniform
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(cursor, (unsigned long long)total);
    base = __shfl(base, 0, 64);
    unsigned long long pos = base + (incl - cnt);
#pragma unroll
    for (int i = 0; i < PER; i++)
      if (occ >> i & 1u) {
        if (pos < cap_out) {
          key0[pos] = w0[i] & kKeyMask;  // hash_table_fetch masks the top two bits (hash_table.h:41-46)
          if (W == 2) key1[pos] = key_ptr(t, s0 + i)[1];
          slot_out[pos] = s0 + i;
        }
        pos++;
      }
  }
}

__global__ void k_gather_u64(const uint64_t *src, const uint64_t *idx, uint64_t *dst, uint64_t n)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_iota(uint64_t *dst, uint64_t n)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = i;
}

// Keys of more than two words (k > 63): the LSD passes of a sort fetch one key word at a time, through the permutation
// so far, from the table (export) or from byte-packed records (`sort`).
__global__ void k_gather_keyword(TableView t, const uint64_t *slot_of, const uint64_t *perm, uint32_t w, uint64_t *dst, uint64_t n)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint64_t x = key_ptr(t, slot_of[perm[i]])[w]; dst[i] = w ? x : (x & kKeyMask); }
}
__global__ void k_gather_record_word(const uint8_t *recs, uint32_t rec_bytes, const uint64_t *perm, uint32_t w, uint64_t *dst, uint64_t n)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t *p = recs + perm[i] * rec_bytes + 8 * w;
  dst[i] = (uint64_t)load_le32(p) | (uint64_t)load_le32(p + 4) << 32;
}
// strictly increasing W-word keys of byte-packed records?  flags the first violation
__global__ void k_check_sorted_records(const uint8_t *recs, uint32_t rec_bytes, uint32_t W, uint64_t n, unsigned long long *first_bad)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 >= n) return;
  const uint8_t *a = recs + i * rec_bytes, *b = a + rec_bytes;
  bool lt = false;
  for (uint32_t w = 0; w < W; w++) {
    const uint64_t x = (uint64_t)load_le32(a + 8 * w) | (uint64_t)load_le32(a + 8 * w + 4) << 32;
    const uint64_t y = (uint64_t)load_le32(b + 8 * w) | (uint64_t)load_le32(b + 8 * w + 4) << 32;
    if (x != y) { lt = x < y; break; }
  }
  if (!lt) atomicMin(first_bad, (unsigned long long)(i + 1));
}

// Final gather in sorted order: record i comes from compact index perm[i].
// Writes .ctx body records (W*8 key bytes, ncols u32 covg, ncols u8 edges;
// graph_writer.c:116-127) for records [first, first+count).
template <int W>
__global__ void k_emit_records(TableView t, const uint64_t *slot_of, const uint64_t *perm, uint64_t first,
                               uint64_t count, uint32_t ncols, uint8_t *out)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint64_t s = slot_of[perm ? perm[first + i] : first + i];
  const uint64_t *r = key_ptr(t, s);
  const uint32_t recsz = 8u * W + 5u * ncols;
  uint8_t *o = out + i * recsz;
  uint64_t kw[W];
  kw[0] = r[0] & kKeyMask;
  for (int w = 1; w < W; w++) kw[w] = r[w];
  for (int w = 0; w < W; w++)
    for (int b = 0; b < 8; b++) o[w * 8 + b] = (uint8_t)(kw[w] >> (8 * b));
  for (uint32_t c = 0; c < ncols; c++) {
    const uint64_t v = *val_ptr(t, s, c);
    const uint64_t cv = v >> 8;
    const uint32_t covg = cv > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (uint32_t)cv;  // COVG_MAX saturation
    for (int b = 0; b < 4; b++) o[8 * W + 4 * c + b] = (uint8_t)(covg >> (8 * b));
    o[8 * W + 4 * ncols + c] = (uint8_t)(v & 0xff);
  }
}

}  // namespace mcx
