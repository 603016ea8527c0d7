// Categorify.fit groupby-size WITHOUT global atomics: key column -> dense
// (key, count) list.  Replaces categorify.py:955-1051 (_top_level_groupby,
// size only) and, with weights, the concat + re-groupby of :1054-1070.
//
// Why: the first version (nvt_count.hip: LDS front table + one global
// open-addressing table) spends its time in device-scope atomics -- 512-1024
// workgroups flushing the same hot keys serialise at the memory-side atomic unit
// (36 distinct keys: 290 us; 1000 keys: 1 ms for a 45 M-row column whose stream
// takes 35 us), and every row of a high-cardinality column is 2 random atomics.
// LDS atomics, by contrast, run at near stream speed (micro-benchmark
// tools/micro/lds_count_probe.hip: 45 M keys, 36 distinct: 58 us = 3.1 TB/s).
//
// Path S ("small", <= ~6000 distinct keys)
//   stage 1  512 workgroups, each counts its grid-stride share into a private
//            8192-slot LDS table and appends its occupied slots to a partials
//            list (ONE reservation atomic per workgroup);
//   stage 2  16 workgroups re-count the (key, weight) partials the same way;
//   stage 3  1 workgroup produces the final dense list.
//   No global hash table, no contended atomics.  A workgroup whose table fills
//   up raises OVERFLOW and the caller reruns the column on path P.
//
// Path P ("partitioned", anything larger)
//   P0  per-workgroup LDS histogram of the top hash bits  -> bucket sizes
//   P0b scan -> exact bucket starts (no over-allocation, no overflow)
//   P1  scatter rows to 64 coarse buckets   (LDS-staged, 512 B contiguous runs)
//   P2  scatter each coarse bucket to 64/256 fine buckets
//   P3  one workgroup per fine bucket: LDS table count -> dense output
//   All occurrences of a key land in one fine bucket, so counts are exact and the
//   only atomics left are LDS ones plus one reservation per tile / workgroup.
//   HBM traffic: 6 x 4 B per row (3 reads + 2 writes + hist read) against 4 B
//   algorithmic -- the price of removing 90 M random device atomics.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "nvt_common.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"
#include "nvt_range.hpp"
#include "nvt_scan.hpp"

// key vectors per lane and batch of lds_stage_kernel.  4 was best while the misses of a batch were
// walked key position by key position; with every lane walking its own misses a batch costs as
// many probe chains as its unluckiest lane has misses, and 8 keys per lane beat 16 (p0 0.774 ->
// 0.738 ms, p6 0.505 -> 0.465 for the 13 LDS-resident Criteo columns; 1 vector: 0.765 / 0.464)
#ifndef NVT_STAGE_U
#define NVT_STAGE_U 2
#endif
#ifndef NVT_SMALL_DIV
#define NVT_SMALL_DIV 4
#endif

namespace nvt {

template <typename K>
struct DKey;
template <>
struct DKey<int32_t> {
  static constexpr int32_t empty = INT32_MIN;
  static constexpr int vec = 4;
  using cas_t = int;
};
template <>
struct DKey<int64_t> {
  static constexpr int64_t empty = INT64_MIN;
  static constexpr int vec = 2;
  using cas_t = unsigned long long;
};

// state words (uint64) written by these kernels
constexpr int DS_NULLS = NVT_ST_NULLS, DS_SENT = NVT_ST_SENTINEL, DS_OUT = NVT_ST_OCCUPIED,
              DS_OVF = NVT_ST_OVERFLOW, DS_ROWS = NVT_ST_ROWS;

constexpr int kLdsSlots = 8192;     // weighted stages / per-bucket tables (u64 or u32 counts)
constexpr int kLdsSlotsBig = 16384; // unweighted path S: int32 key + u32 count = 128 KiB, 1 WG / CU
constexpr int kLdsProbe = 512;  // linear-probing clusters reach ~25 slots at 37 % load; the real
                                // "table full" signal is lfill > max_fill, not the chain length

__host__ __device__ constexpr int max_fill(int slots) { return slots / 4 * 3; }

template <typename K>
__device__ __forceinline__ K lds_cas(K *addr, K expect, K val) {
  using C = typename DKey<K>::cas_t;
  return (K)atomicCAS(reinterpret_cast<C *>(addr), (C)expect, (C)val);
}

// (A wave-level "aggregate the lanes that share the first lane's key" pre-pass was tried to
// relieve same-address LDS atomics on hot keys; it cost more issue slots than it saved on
// every cardinality measured, see profiles/r01_notes.md.)
// Insert into a workgroup-private LDS table.  Returns false when no slot was found.
// `h` must be independent of whatever selected the rows that reach this table: path S
// uses the upper bits of slot_hash, path P the LOW bits of part_hash (its top bits chose
// the bucket; reusing slot_hash there clustered and overflowed 24-probe chains at 37 % load).
template <typename K, typename C, int SLOTS = kLdsSlots>
__device__ __forceinline__ bool lds_add(K *lkeys, C *lcnt, unsigned *lfill, K key, C w,
                                        uint32_t h) {
  constexpr K EMPTY = DKey<K>::empty;
#ifndef NVT_PROBE_UNROLL
#define NVT_PROBE_UNROLL 4
#endif
#pragma unroll NVT_PROBE_UNROLL
  for (int p = 0; p < kLdsProbe; ++p) {
    uint32_t s = (h + p) & (SLOTS - 1);
    K cur = lkeys[s];
    if (cur == EMPTY) {
      cur = lds_cas<K>(&lkeys[s], EMPTY, key);
      if (cur == EMPTY) {
        cur = key;
        atomicAdd(lfill, 1u);
      }
    }
    if (cur == key) {
      atomicAdd(&lcnt[s], w);
      return true;
    }
  }
  return false;
}

// Append the occupied LDS slots to (out_keys, out_cnt) at a range reserved with one
// atomic on *cursor.  All threads of the block must call this.
template <typename K, typename C, int BS, int SLOTS = kLdsSlots>
__device__ __forceinline__ void lds_flush(const K *lkeys, const C *lcnt, K *out_keys,
                                          int64_t *out_cnt, uint64_t out_cap,
                                          unsigned long long *cursor, uint64_t *state) {
  constexpr K EMPTY = DKey<K>::empty;
  __shared__ unsigned wsum[BS / kWave];
  __shared__ unsigned long long base_s;
  constexpr int PER = SLOTS / BS;
  const unsigned lane = lane_id(), w = threadIdx.x / kWave;
  unsigned mine = 0;
  const int first = threadIdx.x * PER;
#pragma unroll 8
  for (int j = 0; j < PER; ++j) mine += (lkeys[first + j] != EMPTY);
  unsigned inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned o = __shfl_up(inc, off, 64);
    if (lane >= (unsigned)off) inc += o;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  unsigned wbase = 0, total = 0;
  for (unsigned i = 0; i < BS / kWave; ++i) {
    if (i < w) wbase += wsum[i];
    total += wsum[i];
  }
  if (threadIdx.x == 0) base_s = total ? atomicAdd(cursor, (unsigned long long)total) : 0ull;
  __syncthreads();
  uint64_t pos = base_s + wbase + inc - mine;
  if (base_s + total > out_cap) {
    if (threadIdx.x == 0) atomicOr((unsigned long long *)&state[DS_OVF], 2ull);
    return;
  }
  unsigned long long mx = 0;
#pragma unroll 8
  for (int j = 0; j < PER; ++j) {
    K k = lkeys[first + j];
    if (k != EMPTY) {
      unsigned long long c = (unsigned long long)lcnt[first + j];
      out_keys[pos] = k;
      out_cnt[pos] = (int64_t)c;
      mx = c > mx ? c : mx;
      ++pos;
    }
  }
  // final list only: largest count, so the host can size the vocabulary sort without a
  // second round trip (one relaxed read, an atomic only when this block raises the max)
  if (cursor == reinterpret_cast<unsigned long long *>(&state[DS_OUT])) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      unsigned long long o = __shfl_down(mx, off, 64);
      mx = o > mx ? o : mx;
    }
    if (lane == 0 && mx > 0) {
      unsigned long long *gm = reinterpret_cast<unsigned long long *>(&state[NVT_ST_MAXCOUNT]);
      if (mx > __hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(gm, mx);
    }
  }
}

// Same, but into a caller-assigned region (no cursor): used for the per-chunk partial lists
// of split (skewed) buckets and for P3's atomic-free staging of its results.  *out_len
// receives the entry count; with `state` the largest count is folded into
// state[NVT_ST_MAXCOUNT].
template <typename K, typename C, int BS, int SLOTS = kLdsSlots>
__device__ __forceinline__ void lds_flush_region(const K *lkeys, const C *lcnt, K *out_keys,
                                                 int64_t *out_cnt, unsigned *out_len,
                                                 uint64_t *state = nullptr) {
  constexpr K EMPTY = DKey<K>::empty;
  __shared__ unsigned wsum2[BS / kWave];
  constexpr int PER = SLOTS / BS;
  const unsigned lane = lane_id(), w = threadIdx.x / kWave;
  unsigned mine = 0;
  const int first = threadIdx.x * PER;
#pragma unroll 8
  for (int j = 0; j < PER; ++j) mine += (lkeys[first + j] != EMPTY);
  unsigned inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned o = __shfl_up(inc, off, 64);
    if (lane >= (unsigned)off) inc += o;
  }
  if (lane == 63) wsum2[w] = inc;
  __syncthreads();
  unsigned wbase = 0, total = 0;
  for (unsigned i = 0; i < BS / kWave; ++i) {
    if (i < w) wbase += wsum2[i];
    total += wsum2[i];
  }
  if (threadIdx.x == 0) *out_len = total;
  unsigned pos = wbase + inc - mine;
  unsigned long long mx = 0;
#pragma unroll 8
  for (int j = 0; j < PER; ++j) {
    K k = lkeys[first + j];
    if (k != EMPTY) {
      unsigned long long c = (unsigned long long)lcnt[first + j];
      out_keys[pos] = k;
      out_cnt[pos] = (int64_t)c;
      mx = c > mx ? c : mx;
      ++pos;
    }
  }
  if (state != nullptr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      unsigned long long o = __shfl_down(mx, off, 64);
      mx = o > mx ? o : mx;
    }
    if (lane == 0 && mx > 0) {
      unsigned long long *gm = reinterpret_cast<unsigned long long *>(&state[NVT_ST_MAXCOUNT]);
      if (mx > __hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(gm, mx);
    }
  }
}

// ---------------------------------------------------------------------------
// Path S.  Two launches, no tree:
//
//   stage 1  (256 << split_bits) workgroups.  Workgroup b owns row slab `slab` (grid-stride
//            over the column, 256 slabs) and key class q = low split_bits bits of slot_hash;
//            it counts the rows of its slab whose key is in its class into a private LDS
//            table.  With split_bits = 0 that is every row (<= ~11 k distinct keys); with
//            2 / 3 bits the column is read 4 / 8 times but tables hold a quarter / an
//            eighth of the vocabulary each (<= ~43 k / ~86 k distinct), which is still far
//            cheaper than one partition pass.  The 8 * SPLIT workgroups that share slabs
//            8g .. 8g+7 are consecutive block ids: block b runs on XCD b % 8, so all SPLIT
//            readers of a slab sit on ONE XCD and the re-reads are L2 hits.
//            The table is flushed GROUPED BY HOME RANGE (top 8 bits of the home slot) into
//            a fixed region per workgroup, with a 257-entry offset row -- no cursor atomics.
//   stage 2  one workgroup per (class q, range r): gathers segment r of the 256 partial
//            lists of class q (a wave per list) into a 512-slot LDS table and appends the
//            result to the output (one reservation atomic per workgroup).  Every key has
//            exactly one (q, r), so the merge is embarrassingly parallel: 1.9 M partial
//            entries (7 k-key column) merge in ~10 us instead of ~190 us for the former
//            32 -> 4 -> 1 workgroup tree.
// ---------------------------------------------------------------------------
constexpr int kStageBS = 1024;  // 16 waves per workgroup, one 96-128 KiB LDS table per CU
constexpr int kSlabs = 256;     // row slabs of stage 1 (= workgroups per key class)
constexpr int kRanges = 256;    // home ranges per table
constexpr int kMergeBS = 256, kMergeSlots = 512;
constexpr unsigned kRepFill = 256;  // replicate hot keys per lane group while fill <= this

// hash of stage 1: home slot from bits >= kStageHomeShift, key class from the (up to 3) bits
// at kStageClassShift
#ifndef NVT_STAGE_FMIX
__device__ __forceinline__ uint32_t stage_hash(int32_t key) { return mul24_hash(key); }
template <typename K>
struct StageBits {
  static constexpr int home = sizeof(K) == 4 ? 18 : 17, cls = sizeof(K) == 4 ? 15 : 0;
};
#else
__device__ __forceinline__ uint32_t stage_hash(int32_t key) { return slot_hash(key); }
template <typename K>
struct StageBits {
  static constexpr int home = 17, cls = 0;
};
#endif
__device__ __forceinline__ uint64_t stage_hash(int64_t key) { return slot_hash(key); }
template <typename K, int SLOTS>
__device__ __forceinline__ uint32_t home_slot(K key) {
  return (uint32_t)(stage_hash(key) >> StageBits<K>::home) & (SLOTS - 1);
}
template <typename K>
__device__ __forceinline__ uint32_t key_class(K key, unsigned split_mask) {
  return (uint32_t)(stage_hash(key) >> StageBits<K>::cls) & split_mask;
}

template <typename K, typename C, int SLOTS>
__global__ __launch_bounds__(kStageBS) void lds_stage_kernel(
    const K *__restrict__ keys, const uint8_t *__restrict__ valid,
    const int64_t *__restrict__ weights, uint64_t n, int split_bits, int tiny, K *part_keys,
    int64_t *part_cnt, unsigned *seg_off, uint64_t *state) {
  constexpr K EMPTY = DKey<K>::empty;
  constexpr int VEC = DKey<K>::vec;
  __shared__ K lkeys[SLOTS];
  __shared__ C lcnt[SLOTS + kWave];  // + one scratch word per lane (see the unconditional add)
  __shared__ unsigned rcnt[kRanges], wtot[kRanges / kWave];
  __shared__ unsigned lfill, lovf, s_next;
  __shared__ unsigned long long s_nulls, s_sent;
#ifdef NVT_STAGE_TIMING
  long long tm[8];
  int tmi = 0;
#define NVT_STM() do { if (threadIdx.x == 0) tm[tmi++] = clock64(); } while (0)
#else
#define NVT_STM() do {} while (0)
#endif
  NVT_STM();
  for (int i = threadIdx.x; i < SLOTS; i += kStageBS) {
    lkeys[i] = EMPTY;
    lcnt[i] = 0;
  }
  if (threadIdx.x < kRanges) rcnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    lfill = 0;
    lovf = 0;
    s_next = 0;
    s_nulls = 0;
    s_sent = 0;
  }
  __syncthreads();
  NVT_STM();
  const unsigned split = 1u << split_bits, split_mask = split - 1;
  const unsigned q = (blockIdx.x >> 3) & split_mask;
  const unsigned slab = ((blockIdx.x >> (3 + split_bits)) << 3) | (blockIdx.x & 7);
  const unsigned nlists = gridDim.x;
  unsigned long long my_nulls = 0, my_sent = 0;
  bool failed = false;
  // A column with a handful of keys (Criteo has five with <= 14) makes every lane of a wave
  // hit the same 1-3 LDS words, and same-address LDS atomics serialise (106 us for 3 keys vs
  // 55 us for 36, tools/micro/lds_cfg_probe.hip).  On the `tiny` path (the caller expects
  // <= 64 distinct keys) each group of 8 lanes probes from its own offset: up to 8 copies
  // of a key, merged for free by stage 2 (duplicates within a partial list are legal).
  // A wrong expectation only costs duplicates: past kRepFill entries replication stops.
  // (Counting a sampled hot key in registers instead -- what P3 does for split buckets --
  // was tried here too: the extra compare per key costs more than the conflicts it removes,
  // +20 us per column; this loop is issue-bound, not LDS-bound.)
  uint32_t rep = tiny ? (lane_id() & 7u) * 2053u : 0u;
  auto add = [&](K key, unsigned long long w) {
    if (key == EMPTY) {
      if (q == 0) my_sent += w;
      return;
    }
    const auto h = stage_hash(key);
    if (((uint32_t)(h >> StageBits<K>::cls) & split_mask) != q) return;
    if (!lds_add<K, C, SLOTS>(lkeys, lcnt, &lfill, key, (C)w, (uint32_t)(h >> StageBits<K>::home) + rep))
      failed = true;
  };
  const uint64_t stride = (uint64_t)kSlabs * kStageBS;
  const uint64_t first = (uint64_t)slab * kStageBS + threadIdx.x;
  if (weights == nullptr) {
    const uint64_t nvec = n / VEC;
    using VecT = typename std::conditional<sizeof(K) == 4, int4, longlong2>::type;
    const VecT *vkeys = reinterpret_cast<const VecT *>(keys);
    // software pipeline: the U vectors of iteration i+1 are requested before iteration i is
    // pushed through the LDS table (one workgroup per CU: latency is covered by ILP, not TLP)
    constexpr int U = NVT_STAGE_U;
    VecT npack[U];
    unsigned nvb[U];
    // Each slab is a CONTIGUOUS range of the column, walked front to back in 16 KiB steps
    // (a grid-stride walk had every workgroup jump 4 MiB between consecutive loads: 1024
    // widely separated 16 KiB windows live at any time).
    const uint64_t per_slab = (nvec + kSlabs - 1) / kSlabs;
    const uint64_t slab_lo = (uint64_t)slab * per_slab;
    const uint64_t slab_hi = slab_lo + per_slab < nvec ? slab_lo + per_slab : nvec;
    // Round 6: a wave takes its batches of U x 64 consecutive vectors from a counter in LDS.  With a
    // fixed share per wave the oldest wave of a SIMD (it wins the issue arbitration) was through
    // with its share at 55 % of the loop's duration and the workgroup waited for the youngest one
    // with one wave per SIMD left to hide its LDS round trips (phase timers: 34-41 % of the kernel
    // between the first wave's last batch and the last wave's).
    constexpr uint64_t vstride = kWave;  // distance between the U vectors of one batch
    const uint64_t nbatch = (slab_hi > slab_lo ? slab_hi - slab_lo + vstride * U - 1 : 0) / (vstride * U);
    auto grab = [&]() -> uint64_t {
      unsigned c = 0;
      if (lane_id() == 0) c = atomicAdd(&s_next, 1u);
      return (uint64_t)__builtin_amdgcn_readfirstlane((int)c);
    };
    auto issue = [&](uint64_t v0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint64_t v = v0 + (uint64_t)u * vstride;
        nvb[u] = 0x10000;  // out of range
        if (v < slab_hi) {
          npack[u] = vkeys[v];
          nvb[u] = valid ? (unsigned)valid[(v * VEC) >> 3] : 0xFFu;  // raw byte, shifted later
        }
      }
    };
    uint64_t batch = grab();
    issue(slab_lo + batch * (vstride * U) + lane_id());
    unsigned fill_now = 0;  // refreshed with the batched home-slot reads below: a separate read
                            // here would drain every queued LDS atomic of the previous batch
    while (batch < nbatch) {
      const uint64_t v0 = slab_lo + batch * (vstride * U) + lane_id();
      if (fill_now > (unsigned)max_fill(SLOTS)) break;  // filling up: the column needs a larger path
      if (fill_now > kRepFill) rep = 0;
      VecT pack[U];
      unsigned vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        pack[u] = npack[u];
        vb[u] = nvb[u];
      }
      batch = grab();
      issue(slab_lo + batch * (vstride * U) + lane_id());   // (past the slab: no loads)
      // Probe in two sweeps.  Sweep 1 reads the HOME slot of every key of the batch -- U * VEC
      // independent LDS reads behind one wait; a key already sitting there (the common case
      // once the table is warm) only needs a fire-and-forget ds_add.  Sweep 2 walks the
      // probe chain for the rest.  One key at a time, each read -> compare -> add chain was
      // a full LDS round trip exposed to a workgroup with only 4 waves per SIMD: the loop
      // was latency-bound (which is also why masking 3/4 of the lanes never made it faster).
      constexpr int NKB = U * VEC;
      K kq[NKB];
      uint32_t hq[NKB];
      unsigned live = 0;  // bit q: key q is valid, of this class, not the sentinel
      // Branch-free classification (PMC: the per-key if / else ladders cost as many SALU
      // exec-mask instructions as there were VALU instructions, 37 + 36 per key).
      unsigned nnull = 0, nsent = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool inrange = !(vb[u] & 0x10000);
        const unsigned bits =
            inrange ? (vb[u] >> (((v0 + (uint64_t)u * vstride) * VEC) & 7)) & ((1u << VEC) - 1u) : 0u;
        if constexpr (sizeof(K) == 4) {
          kq[u * VEC + 0] = pack[u].x;
          kq[u * VEC + 1] = pack[u].y;
          kq[u * VEC + 2] = pack[u].z;
          kq[u * VEC + 3] = pack[u].w;
        } else {
          kq[u * VEC + 0] = pack[u].x;
          kq[u * VEC + 1] = pack[u].y;
        }
        nnull += inrange ? (unsigned)VEC - (unsigned)__popc(bits) : 0u;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int qi = u * VEC + j;
          const bool v = (bits >> j) & 1;
          const bool is_sent = v & (kq[qi] == EMPTY);
          const auto h = stage_hash(kq[qi]);
          const bool lv = v & !is_sent & (((uint32_t)(h >> StageBits<K>::cls) & split_mask) == q);
          nsent += is_sent ? 1u : 0u;
          hq[qi] = lv ? (((uint32_t)(h >> StageBits<K>::home) + rep) & (SLOTS - 1)) : 0u;
          live |= (lv ? 1u : 0u) << qi;
        }
      }
      my_nulls += nnull;
      if (q == 0) my_sent += nsent;
      K cur[NKB];
#pragma unroll
      for (int qi = 0; qi < NKB; ++qi) cur[qi] = lkeys[hq[qi]];
      fill_now = lfill;
      unsigned missbits = 0;
#pragma unroll
      for (int qi = 0; qi < NKB; ++qi) {
        const bool lv = (live >> qi) & 1;
        const bool hit = lv & (cur[qi] == kq[qi]);
        // unconditional add: lanes without a hit bump a per-lane scratch word past the table
        atomicAdd(&lcnt[hit ? hq[qi] : (uint32_t)SLOTS + lane_id()], (C)1);
        missbits |= ((lv & !hit) ? 1u : 0u) << qi;
      }
#ifdef NVT_STAGE_SPARSE_MISS
      if (missbits) {
#pragma unroll
        for (int qi = 0; qi < NKB; ++qi)
          if (((missbits >> qi) & 1) &&
              !lds_add<K, C, SLOTS>(lkeys, lcnt, &lfill, kq[qi], (C)1, hq[qi]))
            failed = true;
      }
#else
      // The probe chain of a miss is a loop of dependent LDS round trips, and with a few percent
      // of misses SOME lane misses at every one of the NKB key positions: walked position by
      // position the wave paid NKB chains per batch with a handful of lanes active in each.
      // Every lane walks ITS next miss instead: as many chains as the unluckiest lane has
      // misses (2-3 of 8 at a 7 % miss rate).
      while (__any(missbits != 0)) {
        if (missbits) {
          const int qm = (int)__ffs((int)missbits) - 1;
          K mk = kq[0];
          uint32_t mh = hq[0];
#pragma unroll
          for (int qi = 1; qi < NKB; ++qi) {
            mk = qm == qi ? kq[qi] : mk;
            mh = qm == qi ? hq[qi] : mh;
          }
          missbits &= missbits - 1u;
          if (!lds_add<K, C, SLOTS>(lkeys, lcnt, &lfill, mk, (C)1, mh)) failed = true;
        }
      }
#endif
    }
    for (uint64_t i = nvec * VEC + first; i < n; i += stride) {
      if (bit_valid(valid, i))
        add(keys[i], 1ull);
      else
        ++my_nulls;
    }
  } else {
    constexpr int UW = 4;
    for (uint64_t i0 = first; i0 < n; i0 += stride * UW) {
      const unsigned fill_now = lfill;
      if (fill_now > (unsigned)max_fill(SLOTS)) break;
      if (fill_now > kRepFill) rep = 0;
      K kk[UW];
      unsigned long long ww[UW];
      int st[UW];  // 0 = out of range, 1 = key, 2 = null row
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        uint64_t i = i0 + (uint64_t)u * stride;
        st[u] = 0;
        if (i < n) {
          ww[u] = (unsigned long long)weights[i];
          st[u] = !bit_valid(valid, i) ? 2 : 1;
          if (st[u] == 1) kk[u] = keys[i];
        }
      }
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        if (st[u] == 1)
          add(kk[u], ww[u]);
        else if (st[u] == 2)
          my_nulls += ww[u];
      }
    }
  }
  NVT_STM();
  if (failed) atomicOr(&lovf, 1u);
  if (q == 0 && my_nulls) atomicAdd(&s_nulls, my_nulls);
  if (my_sent) atomicAdd(&s_sent, my_sent);
  __syncthreads();
  NVT_STM();
  if (lovf || lfill > (unsigned)max_fill(SLOTS)) {
    if (threadIdx.x == 0) atomicOr((unsigned long long *)&state[DS_OVF], 1ull);
    // stage 2 must not read stale offsets from this list
    for (int r = threadIdx.x; r <= kRanges; r += kStageBS) seg_off[(uint64_t)r * nlists + blockIdx.x] = 0;
    return;
  }
  if (threadIdx.x == 0) {
    if (s_nulls) atomicAdd((unsigned long long *)&state[DS_NULLS], s_nulls);
    if (s_sent) atomicAdd((unsigned long long *)&state[DS_SENT], s_sent);
    if (blockIdx.x == 0) atomicAdd((unsigned long long *)&state[DS_ROWS], (unsigned long long)n);
  }
#ifdef NVT_EXP_NOFLUSH
  for (int r = threadIdx.x; r <= kRanges; r += kStageBS) seg_off[(uint64_t)r * nlists + blockIdx.x] = 0;
  return;
#endif
  // ---- flush grouped by home range: LDS histogram -> scan -> ranked scatter ----
  constexpr int RSHIFT = (SLOTS == 16384 ? 14 : SLOTS == 8192 ? 13 : 12) - 8;
  for (int i = threadIdx.x; i < SLOTS; i += kStageBS) {
    K k = lkeys[i];
    if (k != EMPTY) atomicAdd(&rcnt[home_slot<K, SLOTS>(k) >> RSHIFT], 1u);
  }
  __syncthreads();
  unsigned mine = 0, inc = 0;
  if (threadIdx.x < kRanges) {
    mine = rcnt[threadIdx.x];
    inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o = __shfl_up(inc, off, 64);
      if (lane_id() >= (unsigned)off) inc += o;
    }
    if (lane_id() == 63) wtot[threadIdx.x / kWave] = inc;
  }
  __syncthreads();
  if (threadIdx.x < kRanges) {
    unsigned wbase = 0;
    for (unsigned i = 0; i < threadIdx.x / kWave; ++i) wbase += wtot[i];
    const unsigned startv = wbase + inc - mine;
    rcnt[threadIdx.x] = startv;  // becomes the range's write cursor
    seg_off[(uint64_t)threadIdx.x * nlists + blockIdx.x] = startv;
    if (threadIdx.x == kRanges - 1)
      seg_off[(uint64_t)kRanges * nlists + blockIdx.x] = startv + mine;
  }
  __syncthreads();
  K *ok = part_keys + (uint64_t)blockIdx.x * max_fill(SLOTS);
  int64_t *oc = part_cnt + (uint64_t)blockIdx.x * max_fill(SLOTS);
  for (int i = threadIdx.x; i < SLOTS; i += kStageBS) {
    K k = lkeys[i];
    if (k != EMPTY) {
      unsigned pos = atomicAdd(&rcnt[home_slot<K, SLOTS>(k) >> RSHIFT], 1u);
      ok[pos] = k;
      oc[pos] = (int64_t)lcnt[i];
    }
  }
#ifdef NVT_STAGE_TIMING
  __syncthreads();
  NVT_STM();
  if (threadIdx.x == 0)
    for (int t = 1; t < tmi; ++t)
      atomicAdd((unsigned long long *)&state[9 + t], (unsigned long long)(tm[t] - tm[t - 1]));
#endif
}

// Path S, stage 2: workgroup (q, r) merges segment r of the kSlabs partial lists of class q.
template <typename K>
__global__ __launch_bounds__(kMergeBS) void range_merge_kernel(
    const K *__restrict__ part_keys, const int64_t *__restrict__ part_cnt,
    const unsigned *__restrict__ seg_off, int split_bits, uint64_t region, K *out_keys,
    int64_t *out_cnt, uint64_t out_cap, uint64_t *state) {
  constexpr K EMPTY = DKey<K>::empty;
  using C = unsigned long long;
  __shared__ K lkeys[kMergeSlots];
  __shared__ C lcnt[kMergeSlots];
  __shared__ unsigned lfill, lovf, wsum[kMergeBS / kWave];
  __shared__ unsigned long long base_s;
  __shared__ int s_skip;
  for (int i = threadIdx.x; i < kMergeSlots; i += kMergeBS) {
    lkeys[i] = EMPTY;
    lcnt[i] = 0;
  }
  if (threadIdx.x == 0) {
    lfill = 0;
    lovf = 0;
    // stage 1 already overflowed: the result is discarded anyway
    s_skip = (int)(__hip_atomic_load(&state[DS_OVF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1);
  }
  __syncthreads();
  if (s_skip) return;  // workgroup-uniform (read once by thread 0)
  const unsigned r = blockIdx.x & (kRanges - 1), q = blockIdx.x >> 8;
  const unsigned nlists = (unsigned)kSlabs << split_bits;
  const unsigned lane = lane_id(), w = threadIdx.x / kWave;
  // thread t owns list t of this class: segment bounds -> LDS, exclusive scan of the lengths
  // gives a flat index space over all 256 segments, so the loads below are independent and
  // balanced (a wave-per-list loop here was a 64-deep chain of dependent global loads).
  static_assert(kMergeBS == kSlabs, "one thread per partial list");
  __shared__ unsigned seg_lo[kSlabs], seg_start[kSlabs + 1];
  {
    const unsigned li = threadIdx.x;
    const unsigned b1 = ((((li >> 3) << split_bits) | q) << 3) | (li & 7);
    const unsigned lo = seg_off[(uint64_t)r * nlists + b1];
    const unsigned hi = seg_off[(uint64_t)(r + 1) * nlists + b1];
    const unsigned len = hi - lo;
    unsigned inc = len;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o = __shfl_up(inc, off, 64);
      if (lane >= (unsigned)off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned wbase = 0;
    for (unsigned i = 0; i < w; ++i) wbase += wsum[i];
    seg_lo[li] = lo;
    seg_start[li] = wbase + inc - len;
    if (li == kSlabs - 1) seg_start[kSlabs] = wbase + inc;
  }
  __syncthreads();
  const unsigned total_in = seg_start[kSlabs];
  bool failed = false;
  constexpr int UM = 4;
  for (unsigned j0 = threadIdx.x; j0 < total_in; j0 += kMergeBS * UM) {
    if (lfill > (unsigned)max_fill(kMergeSlots)) break;  // too many keys for path 0
    K kk[UM];
    int64_t cc[UM];
    bool ok[UM];
#pragma unroll
    for (int u = 0; u < UM; ++u) {
      const unsigned j = j0 + u * kMergeBS;
      ok[u] = j < total_in;
      if (ok[u]) {
        unsigned a = 0, bnd = kSlabs;  // largest li with seg_start[li] <= j
        while (bnd - a > 1) {
          const unsigned m = (a + bnd) >> 1;
          if (seg_start[m] <= j) a = m; else bnd = m;
        }
        const unsigned b1 = ((((a >> 3) << split_bits) | q) << 3) | (a & 7);
        const uint64_t idx = (uint64_t)b1 * region + seg_lo[a] + (j - seg_start[a]);
        kk[u] = part_keys[idx];
        cc[u] = part_cnt[idx];
      }
    }
#pragma unroll
    for (int u = 0; u < UM; ++u)
      if (ok[u] && !lds_add<K, C, kMergeSlots>(lkeys, lcnt, &lfill, kk[u], (C)cc[u],
                                               (uint32_t)(slot_hash(kk[u]) >> 4)))
        failed = true;
  }
  if (failed) atomicOr(&lovf, 1u);
  __syncthreads();
  if (lovf || lfill > (unsigned)max_fill(kMergeSlots)) {
    if (threadIdx.x == 0) atomicOr((unsigned long long *)&state[DS_OVF], 1ull);
    return;
  }
  // compact (2 slots per thread) and append with one reservation
  constexpr int PER = kMergeSlots / kMergeBS;
  unsigned mine = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) mine += (lkeys[threadIdx.x * PER + j] != EMPTY);
  unsigned inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned o = __shfl_up(inc, off, 64);
    if (lane >= (unsigned)off) inc += o;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  unsigned wbase = 0, total = 0;
  for (unsigned i = 0; i < kMergeBS / kWave; ++i) {
    if (i < w) wbase += wsum[i];
    total += wsum[i];
  }
  if (total == 0) return;
  if (threadIdx.x == 0)
    base_s = atomicAdd(reinterpret_cast<unsigned long long *>(&state[DS_OUT]),
                       (unsigned long long)total);
  __syncthreads();
  if (base_s + total > out_cap) {
    if (threadIdx.x == 0) atomicOr((unsigned long long *)&state[DS_OVF], 2ull);
    return;
  }
  uint64_t pos = base_s + wbase + inc - mine;
  unsigned long long mx = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    K k = lkeys[threadIdx.x * PER + j];
    if (k != EMPTY) {
      unsigned long long c = lcnt[threadIdx.x * PER + j];
      out_keys[pos] = k;
      out_cnt[pos] = (int64_t)c;
      mx = c > mx ? c : mx;
      ++pos;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if (lane == 0 && mx > 0) {
    unsigned long long *gm = reinterpret_cast<unsigned long long *>(&state[NVT_ST_MAXCOUNT]);
    if (mx > __hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(gm, mx);
  }
}

// ---------------------------------------------------------------------------
// Path P
// ---------------------------------------------------------------------------
constexpr int kTile = 8192;        // rows per scatter tile (32 rows per thread)
constexpr int kChunk = 65536;      // rows one P3 workgroup counts
#ifndef NVT_COUNT_BS
#define NVT_COUNT_BS 512
#endif
#ifndef NVT_P1_SLOTS
#define NVT_P1_SLOTS 4096
#endif
constexpr int kCountBS = NVT_COUNT_BS;      // P3 workgroup size
constexpr int kMaxFine = 1 << 14;  // up to 6 + 8 hash bits
constexpr int kHistBlocks = 512;

template <typename K>
__device__ __forceinline__ uint32_t part_hash(K key) {
  // independent of the LDS-table hash (which uses bits >= 17 of slot_hash)
  return fmix32((uint32_t)slot_hash(key) * 0x9E3779B1u + 0x7F4A7C15u);
}

// P0, tile by tile (same kTile-row tiles as the P1 scatter).  Per tile: the histogram over
// the COARSE bucket (top b1 hash bits) goes to tile_hist[bucket * ntiles + tile]; after a
// device-wide exclusive scan of that bucket-major array every (tile, bucket) pair knows
// exactly where it writes, so P1 needs no cursor atomics (they cost half its time: 5.5 k
// tiles bumping the same 64-256 words) and the partition is deterministic.  Per workgroup:
// the histogram over the FINE bucket (top b1+b2 bits) for the bucket boundaries.
template <typename K>
__global__ __launch_bounds__(1024) void part_hist_kernel(const K *__restrict__ keys,
                                                           const uint8_t *__restrict__ valid,
                                                           const int64_t *__restrict__ weights,
                                                           uint64_t n, int b1, int bits,
                                                           unsigned *block_hist, unsigned *tile_hist,
                                                           uint64_t ntiles, uint64_t *state) {
  __shared__ unsigned h[kMaxFine];
  __shared__ unsigned ht[256];
  __shared__ unsigned long long s_nulls;
  const int nb = 1 << bits, nc = 1 << b1;
  for (int i = threadIdx.x; i < nb; i += 1024) h[i] = 0;
  if (threadIdx.x == 0) s_nulls = 0;
  unsigned long long nulls = 0;
  constexpr int VEC = DKey<K>::vec;
  constexpr int NV = kTile / VEC / 1024;  // 16-byte vectors per thread per tile (2 or 4)
  using VecT = typename std::conditional<sizeof(K) == 4, int4, longlong2>::type;
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (threadIdx.x < 256) ht[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t row0 = tile * kTile;
    VecT pack[NV];
    unsigned vb[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const uint64_t i0 = row0 + ((uint64_t)u * 1024 + threadIdx.x) * VEC;
      vb[u] = 0x10000;  // not a full in-range vector
      if (i0 + VEC <= n) {
        pack[u] = *reinterpret_cast<const VecT *>(keys + i0);
        vb[u] = valid ? (unsigned)valid[i0 >> 3] : 0xFFu;  // raw bitmap byte, shifted later
      }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const uint64_t i0 = row0 + ((uint64_t)u * 1024 + threadIdx.x) * VEC;
      K kv[VEC];
      unsigned bits_ok = 0, in_range = 0;
      if (!(vb[u] & 0x10000)) {
        if constexpr (sizeof(K) == 4) {
          kv[0] = pack[u].x;
          kv[1] = pack[u].y;
          kv[2] = pack[u].z;
          kv[3] = pack[u].w;
        } else {
          kv[0] = pack[u].x;
          kv[1] = pack[u].y;
        }
        bits_ok = (vb[u] >> (i0 & 7)) & ((1u << VEC) - 1u);
        in_range = (1u << VEC) - 1u;
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          kv[j] = 0;
          if (i0 + j < n) {
            in_range |= 1u << j;
            if (bit_valid(valid, i0 + j)) {
              kv[j] = keys[i0 + j];
              bits_ok |= 1u << j;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if ((bits_ok >> j) & 1) {
          const unsigned fine = part_hash<K>(kv[j]) >> (32 - bits);
          atomicAdd(&ht[fine >> (bits - b1)], 1u);
          if (bits > b1) atomicAdd(&h[fine], 1u);
        }
      }
      const unsigned nmask = in_range & ~bits_ok;
      if (weights == nullptr) {
        nulls += __popc(nmask);
      } else if (nmask) {
        for (int j = 0; j < VEC; ++j)
          if ((nmask >> j) & 1) nulls += (unsigned long long)weights[i0 + j];
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < nc) {
      const unsigned c = ht[threadIdx.x];
      tile_hist[(uint64_t)threadIdx.x * ntiles + tile] = c;
      if (bits == b1) h[threadIdx.x] += c;  // one level: fine == coarse
    }
    __syncthreads();
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 1024) block_hist[(uint64_t)blockIdx.x * nb + i] = h[i];
  // one device atomic per WORKGROUP (an atomic per wave on this single word serialised at the
  // memory side: +85 us on every column that has nulls)
  if (nulls) atomicAdd(&s_nulls, nulls);
  __syncthreads();
  if (threadIdx.x == 0 && s_nulls) atomicAdd((unsigned long long *)&state[DS_NULLS], s_nulls);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd((unsigned long long *)&state[DS_ROWS], (unsigned long long)n);
}

// ---- hot filter in front of paths 1 / 2 / 3 (int32 keys, unweighted) ----------------------------
// A power-law column sends 60-95 % of its rows to a few thousand keys (Criteo C1: 65 % of the
// rows carry one of the 14 k most frequent of 6.2 M keys).  Partitioning those rows is wasted
// work: they only need counters.  So the histogram pass also looks every key up in a read-only
// LDS table of "hot" keys; a hit is ONE LDS atomic and the row is switched off in the bitmap
// that the scatter / count stages see (they already skip null rows), a miss goes through the
// partition as before.  Everything downstream of the histogram then handles only the cold
// rows (measured with an ideal hot set, tools/coldfrac_probe.py: C1 595 -> 368 us, C11
// 495 -> 240 us incl. the plain histogram pass).
//   hot_sample_kernel   one workgroup picks the hot set from up to 64 blocks of 1024 rows
//                       spread over the column: keys seen twice first, then first come while
//                       there is room.  The table image is written once, so that every
//                       workgroup of the histogram pass holds the SAME slot layout and the
//                       per-workgroup counters can be summed slot by slot (no hash merge).
//                       A sample that the table would serve badly (< 1/8 of its rows) empties
//                       the image: the column then behaves exactly as without the filter.
//   part_hist_hot_kernel  part_hist_kernel + lookup + cold bitmap + per-workgroup hot counters
//   hot_reduce_kernel   column sums of the counters -> (key, count) entries appended to the
//                       output list behind the partition's entries
// The hot set is a heuristic; the result is exact for ANY hot set because a key is either in
// the image (all of its rows are counted by the counters) or not (all of them are partitioned).
// The table: buckets of NVT_HOT_WIDTH slots, ONE candidate bucket per key, so a lookup is one
// 8- or 16-byte LDS read.  A key whose bucket is full is simply not hot (2 choices x 2 slots
// kept ~8 % more keys and cost a second read per row: 2.37 vs 2.22 ms for the nine filtered
// path-1 columns).
#ifndef NVT_HOT_WIDTH
#define NVT_HOT_WIDTH 2
#endif
constexpr int kHotSlots = NVT_HOT_IMAGE_WORDS;  // 32 KiB of keys + 32 KiB of counters
constexpr int kHotWidth = NVT_HOT_WIDTH;
constexpr int kHotBuckets = kHotSlots / kHotWidth;
constexpr int kHotBlocks = 256;         // histogram workgroups (one per CU: 130 KiB of LDS each)
constexpr int kHotSampleBlocks = 64;    // x 1024 rows

__device__ __forceinline__ uint32_t hot_bucket(int32_t key) { return hot_image_bucket(key, kHotBuckets - 1); }
// slot of `key` in its bucket (already loaded), or -1
__device__ __forceinline__ int hot_find(const int2 &b, int32_t key, uint32_t base) {
  int slot = -1;
  slot = b.x == key ? (int)base : slot;
  slot = b.y == key ? (int)base + 1 : slot;
  return slot;
}
__device__ __forceinline__ int hot_find(const int4 &b, int32_t key, uint32_t base) {
  int slot = -1;
  slot = b.x == key ? (int)base : slot;
  slot = b.y == key ? (int)base + 1 : slot;
  slot = b.z == key ? (int)base + 2 : slot;
  slot = b.w == key ? (int)base + 3 : slot;
  return slot;
}

struct HotSampleCol {
  const int32_t *keys;
  const uint8_t *valid;
  uint64_t n;
  int32_t *image;
  int nb_log2;  // > 0: also derive the key ranges of the range path (image[NVT_RANGE_AUX_*])
  int pieces;   // range path: also decide on the piecewise map (NVT_PATH_PIECES: after an overflow)
};
#ifdef NVT_NO_PIECEWISE
constexpr bool kPiecewise = false;
#else
constexpr bool kPiecewise = true;
#endif
constexpr int kHotBatch = 32;
struct HotSampleBatch {
  HotSampleCol c[kHotBatch];
};
// one workgroup per column (the LDS work of a sample is ~80 us on one CU: the columns of a
// call are sampled side by side, ahead of their pipelines)
__global__ __launch_bounds__(1024) void hot_sample_kernel(const HotSampleBatch batch) {
  constexpr int32_t EMPTY = DKey<int32_t>::empty;
  const int32_t *__restrict__ keys = batch.c[blockIdx.x].keys;
  const uint8_t *__restrict__ valid = batch.c[blockIdx.x].valid;
  const uint64_t n = batch.c[blockIdx.x].n;
  int32_t *image = batch.c[blockIdx.x].image;
  __shared__ int32_t tk[kHotSlots];
  __shared__ unsigned seen[2048];  // 64 K-bit "seen once" filter
  __shared__ unsigned s_hits, s_rows, s_umin, s_umax;
  // range path only: how often the sample shows every image slot's key, and a one-row sketch of
  // the keys that found their bucket full -- a FREQUENT key that lost the race for its bucket
  // would flood one region of the partition pass (see the rescue below)
  __shared__ unsigned tcnt[kHotSlots];
  __shared__ unsigned msk[2048];
  __shared__ int32_t mk[256];
  __shared__ unsigned mc[256];
  __shared__ unsigned s_missed;
  const bool rescue = batch.c[blockIdx.x].nb_log2 > 0;
  __shared__ uint64_t s_map[5];
  unsigned pmb[kHotSlots / 1024], pmr[kHotSlots / 1024];
  bool pieces_done = false;
  for (int i = threadIdx.x; i < kHotSlots; i += 1024) tk[i] = EMPTY;
  for (int i = threadIdx.x; i < 2048; i += 1024) seen[i] = 0;
  if (rescue) {
    for (int i = threadIdx.x; i < kHotSlots; i += 1024) tcnt[i] = 0;
    for (int i = threadIdx.x; i < 2048; i += 1024) msk[i] = 0;
    if (threadIdx.x < 256) {
      mk[threadIdx.x] = EMPTY;
      mc[threadIdx.x] = 0;
    }
  }
  if (threadIdx.x == 0) {
    s_missed = 0;
    s_hits = s_rows = 0;
    s_umin = 0xFFFFFFFFu;
    s_umax = 0u;
  }
  unsigned umin = 0xFFFFFFFFu, umax = 0u;  // order-preserving unsigned images of the sampled keys
  __syncthreads();
  const uint64_t nblk = (n + 1023) / 1024;
  const unsigned S = (unsigned)(nblk < (uint64_t)kHotSampleBlocks ? nblk : kHotSampleBlocks);
  const uint64_t step = (S ? nblk / S : 1) * 1024;  // rows between the starts of sampled blocks
  auto insert = [&](int32_t key, uint32_t) -> bool {
    const uint32_t b = hot_bucket(key) * kHotWidth;
#pragma unroll
    for (int c = 0; c < kHotWidth; ++c) {
      const int32_t prev = atomicCAS(&tk[b + c], EMPTY, key);
      if (prev == EMPTY || prev == key) return true;
    }
    return false;
  };
  // the sample is read in batches of kBatch rows per thread, every load of a batch in flight
  // before the first is used (one workgroup: a dependent load per row would pay the memory
  // latency 2 x 64 times -- 200 us; holding all 64 rows per thread in registers spills)
  constexpr int kBatch = 16;
  int32_t kreg[kBatch];
  auto load_batch = [&](unsigned it0) {
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      const unsigned it = it0 + q;
      const uint64_t i = (uint64_t)it * step + threadIdx.x;
      kreg[q] = (it < S && i < n) ? keys[i] : EMPTY;
    }
    if (valid) {  // null rows are skipped like the sentinel key
      unsigned vm = 0;
#pragma unroll
      for (int q = 0; q < kBatch; ++q) {
        const unsigned it = it0 + q;
        const uint64_t i = (uint64_t)it * step + threadIdx.x;
        const unsigned byte = (it < S && i < n) ? valid[i >> 3] : 0u;
        vm |= ((byte >> (i & 7)) & 1u) << q;
      }
#pragma unroll
      for (int q = 0; q < kBatch; ++q)
        if (!((vm >> q) & 1u)) kreg[q] = EMPTY;
    }
  };
  // sweep 1: a key enters the table when the sample shows it for the second time
  for (unsigned it0 = 0; it0 < S; it0 += kBatch) {
    load_batch(it0);
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      const int32_t key = kreg[q];
      if (key != EMPTY) {
        const uint32_t h = slot_hash(key);
        const uint32_t bit = (h * 0x9E3779B1u) >> 16;
        const unsigned m = 1u << (bit & 31);
        if (atomicOr(&seen[bit >> 5], m) & m) insert(key, h);
      }
    }
  }
  __syncthreads();
  // sweep 2: the remaining keys, first come, while their buckets have room; the share of
  // sampled rows that find their key estimates what the table will absorb
  unsigned hits = 0, rows = 0;
  for (unsigned it0 = 0; it0 < S; it0 += kBatch) {
    load_batch(it0);
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      const int32_t key = kreg[q];
      if (key != EMPTY) {
        const uint32_t h = slot_hash(key);  // (the sketch of the rescue below)
        const uint32_t b = hot_bucket(key) * kHotWidth;
        bool found = false;
#pragma unroll
        for (int c = 0; c < kHotWidth; ++c) found = found || tk[b + c] == key;
        bool in = found;
        if (!found) in = insert(key, h);
        if (rescue) {
          if (in) {
#pragma unroll
            for (int c = 0; c < kHotWidth; ++c)
              if (tk[b + c] == key) atomicAdd(&tcnt[b + c], 1u);
          } else {
            atomicAdd(&msk[(h * 0x85EBCA6Bu) >> 21], 1u);
            atomicAdd(&s_missed, 1u);
          }
        }
        hits += found;
        rows += 1;
        const unsigned u = (unsigned)key ^ 0x80000000u;
        umin = u < umin ? u : umin;
        umax = u > umax ? u : umax;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    hits += __shfl_down(hits, off, 64);
    rows += __shfl_down(rows, off, 64);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned a = __shfl_down(umin, off, 64), b = __shfl_down(umax, off, 64);
    umin = a < umin ? a : umin;
    umax = b > umax ? b : umax;
  }
  if (lane_id() == 0) {
    atomicAdd(&s_hits, hits);
    atomicAdd(&s_rows, rows);
    atomicMin(&s_umin, umin);
    atomicMax(&s_umax, umax);
  }
  __syncthreads();
  if (rescue) {
    // Rescue of frequent keys that are NOT in the image.  The image takes keys first come, so a
    // key as frequent as 1 % of the rows occasionally finds both slots of its bucket taken by
    // two rarer keys; all its rows then go through ONE bin of the partition pass and overflow a
    // (bucket, workgroup) region (2 x the average rows + 64): about one 45 M-row Criteo partition
    // in a hundred had to be recounted on the sort path for that.  A key whose sample count
    // reaches rows / (2 * buckets) replaces the rarer occupant of its bucket.  The sketch makes
    // the common case (nothing to rescue) free: a third sweep of the sample runs only when some
    // sketch counter stands out from the noise of the one-off keys.
    const unsigned nbk = 1u << batch.c[blockIdx.x].nb_log2;
    const unsigned T = max(16u, s_rows / (2u * nbk));
    const unsigned thr = T + 2u * (s_missed / 2048u);
    const int any = __syncthreads_or(msk[threadIdx.x] >= thr || msk[threadIdx.x + 1024] >= thr);
    if (any) {
      for (unsigned it0 = 0; it0 < S; it0 += kBatch) {
        load_batch(it0);
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
          const int32_t key = kreg[q];
          if (key == EMPTY) continue;
          const uint32_t h = slot_hash(key);
          if (msk[(h * 0x85EBCA6Bu) >> 21] < thr) continue;
          const uint32_t b = hot_bucket(key) * kHotWidth;
          bool found = false;
#pragma unroll
          for (int c = 0; c < kHotWidth; ++c) found = found || tk[b + c] == key;
          if (found) continue;
          uint32_t m = (h >> 3) & 255u;
          for (int step = 0; step < 256; ++step, m = (m + 1) & 255u) {  // exact count of the candidates
            const int32_t prev = atomicCAS(&mk[m], EMPTY, key);
            if (prev == EMPTY || prev == key) {
              atomicAdd(&mc[m], 1u);
              break;
            }
          }
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int m = 0; m < 256; ++m) {
          const int32_t key = mk[m];
          const unsigned c = mc[m];
          if (key == EMPTY || c < T) continue;
          const uint32_t b = hot_bucket(key) * kHotWidth;
          int worst = 0;
#pragma unroll
          for (int w = 1; w < kHotWidth; ++w)
            if (tcnt[b + w] < tcnt[b + worst]) worst = w;
          if (tcnt[b + worst] < c) {  // the rarer occupant leaves the image (and goes through the bins)
            tk[b + worst] = key;
            tcnt[b + worst] = c;
          }
        }
      }
      __syncthreads();
    }
  }
  const bool useful = (uint64_t)s_hits * 8 >= (uint64_t)s_rows && s_rows > 0;
  for (int i = threadIdx.x; i < kHotSlots; i += 1024) image[i] = useful ? tk[i] : EMPTY;
  const int nb_log2 = batch.c[blockIdx.x].nb_log2;
  if (nb_log2 > 0 && threadIdx.x == 0) {
    // range path: fine slot = (min(u - ulo, span) * mul) >> sh over the sampled span padded by
    // 1/64 on either side (keys outside land in the edge slots: monotone, just unbalanced);
    // see RangeMap in nvt_range_count.hip
    uint64_t lo = s_umin, hi = s_umax;
    if (s_rows == 0) {
      lo = 0;
      hi = 0xFFFFFFFFull;
    }
    const uint64_t pad = ((hi - lo) >> 6) + 1;
    lo = lo > pad ? lo - pad : 0;
    hi = hi + pad < 0xFFFFFFFFull ? hi + pad : 0xFFFFFFFFull;
    const uint64_t span = hi - lo, F = 1ull << (nb_log2 + 14);
    uint32_t mul;
    int sh;
    range_map_params(span, F, &mul, &sh);
    image[NVT_RANGE_AUX_LO] = (int32_t)(uint32_t)lo;
    image[NVT_RANGE_AUX_LO + 1] = (int32_t)(uint32_t)span;
    image[NVT_RANGE_AUX_LO + 2] = (int32_t)mul;
    image[NVT_RANGE_AUX_LO + 3] = 0;
    image[NVT_RANGE_AUX_LO + 4] = sh;
    image[NVT_RANGE_AUX_LO + 5] = 0;  // bucket-region table layout (nvt_range.hpp)
    image[NVT_RANGE_AUX_LO + 7] = 0;  // linear map (the piecewise form is decided below)
    s_map[0] = lo;
    s_map[1] = span;
    s_map[2] = mul;
    s_map[3] = (uint64_t)sh;
  }
  if (nb_log2 >= 6 && kPiecewise && batch.c[blockIdx.x].pieces) {
    // ---- piecewise map: the caller put kRpPieces + 1 splitters (order-preserving u32 images,
    // strictly increasing) into the aux block -- taken from an EXACT key-ordered (key, count)
    // list of an earlier pass over this column (kernels.range_splitters: rows and distinct keys
    // blended, so that no piece holds more than ~2x the average of either).  A sample of a few
    // thousand rows cannot do this: nearly every cold key is a singleton in it, so it sees rows,
    // not distinct keys, and the tail pieces of a dense-id column came out with 3.5x the average
    // number of distinct keys (tools/pieces_probe.py).  Here: multipliers + the CSR of the hot keys.
    __shared__ uint32_t s_pw[2 * kRpPieces + 3];
    for (int p = threadIdx.x; p <= kRpPieces; p += 1024)
      s_pw[p] = (uint32_t)image[NVT_RANGE_AUX_PW + p];
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t S = 1u << (nb_log2 + 8);   // fine slots per piece = buckets / 64 x 16384
      bool ok = true;
      for (int p = 0; p < kRpPieces; ++p) ok = ok && s_pw[p + 1] > s_pw[p];
      if (ok) {
        uint32_t flags[2] = {0u, 0u};
        for (int p = 0; p < kRpPieces; ++p) {
          const uint32_t w = s_pw[p + 1] - s_pw[p];
          uint32_t mulp;
          if (w > S) {
            mulp = (uint32_t)((((uint64_t)S) << 32) / w);
            flags[p >> 5] |= 1u << (p & 31);
          } else {  // fewer keys than slots: 16-bit fixed point (an integer factor S / w would leave
                    // up to half of the piece's slots unused and its first buckets overfull)
            const uint64_t m16 = (((uint64_t)S) << 16) / w;
            mulp = (uint32_t)(m16 < 0xFFFFFFFFull ? m16 : 0xFFFFFFFFull);
          }
          s_pw[kRpPwMul + p] = mulp;
        }
        s_pw[kRpPwSh] = flags[0];
        s_pw[kRpPwSh + 1] = flags[1];
        s_map[4] = S;
      } else {
        s_map[4] = 0;   // (malformed splitters: the linear map)
      }
    }
    __syncthreads();
    if (s_map[4]) {
      for (int p = threadIdx.x; p < 2 * kRpPieces + 3; p += 1024) image[NVT_RANGE_AUX_PW + p] = (int32_t)s_pw[p];
      if (threadIdx.x == 0) image[NVT_RANGE_AUX_LO + 7] = (int32_t)s_map[4];
    }
    __syncthreads();
    // (the CSR below maps the hot keys with the map that was just decided)
    if (s_map[4]) {
      RangeMap pm;
      pm.ulo = 0; pm.span = 0; pm.mul = 0; pm.sh = 0; pm.flat = 0;
      pm.piece_slots = (uint32_t)s_map[4];
      pm.pw = s_pw;
      pm.lpw = (const __attribute__((address_space(3))) uint32_t *)s_pw;
      unsigned *bcnt = seen;
      for (int i = threadIdx.x; i < 1025; i += 1024) bcnt[i] = 0;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kHotSlots / 1024; ++q) {
        const int i = q * 1024 + threadIdx.x;
        const int32_t key = useful ? tk[i] : EMPTY;
        pmb[q] = 0xFFFFFFFFu;
        if (key != EMPTY) {
          pmb[q] = pm.fine(key) >> 14;
          pmr[q] = atomicAdd(&bcnt[pmb[q]], 1u);
        }
      }
      pieces_done = true;
    }
  } else if (nb_log2 > 0 && threadIdx.x == 0) {
    s_map[4] = 0;
  }
  if (nb_log2 > 0) {
    // the image slots indexed by range bucket (counting sort): the per-bucket count workgroup
    // of the range path picks up its hot keys without scanning the whole image
    unsigned *bcnt = seen;  // 2048 words, free again
    unsigned myb[kHotSlots / 1024], myr[kHotSlots / 1024];
    if (!pieces_done) {
    for (int i = threadIdx.x; i < 1025; i += 1024) bcnt[i] = 0;
    __syncthreads();
    const uint64_t lo = s_map[0], span = s_map[1], mul = s_map[2];
    const int sh = (int)s_map[3];
#pragma unroll
    for (int q = 0; q < kHotSlots / 1024; ++q) {
      const int i = q * 1024 + threadIdx.x;
      const int32_t key = useful ? tk[i] : EMPTY;
      myb[q] = 0xFFFFFFFFu;
      if (key != EMPTY) {
        const uint64_t u = (uint32_t)key ^ 0x80000000u;
        uint64_t d = u > lo ? u - lo : 0;
        d = d < span ? d : span;
        myb[q] = (unsigned)((((d << sh) * mul) >> 32) >> 14);  // RangeMap::fine
        myr[q] = atomicAdd(&bcnt[myb[q]], 1u);
      }
    }
    } else {
#pragma unroll
      for (int q = 0; q < kHotSlots / 1024; ++q) {
        myb[q] = pmb[q];
        myr[q] = pmr[q];
      }
    }
    __syncthreads();
    // exclusive scan of the 1024 bucket counts (one per thread)
    __shared__ unsigned swt[16];
    const unsigned v = bcnt[threadIdx.x];
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(inc, off, 64);
      if (lane_id() >= (unsigned)off) inc += o;
    }
    if (lane_id() == 63) swt[threadIdx.x / 64] = inc;
    __syncthreads();
    unsigned wb = 0;
    for (unsigned q = 0; q < threadIdx.x / 64; ++q) wb += swt[q];
    const unsigned start = wb + inc - v;
    __syncthreads();
    bcnt[threadIdx.x] = start;
    image[NVT_RANGE_AUX_HOTSTART + threadIdx.x] = (int32_t)start;
    if (threadIdx.x == 1023) image[NVT_RANGE_AUX_HOTSTART + 1024] = (int32_t)(start + v);
    __syncthreads();
    unsigned short *order = reinterpret_cast<unsigned short *>(image + NVT_RANGE_AUX_HOTORDER);
#pragma unroll
    for (int q = 0; q < kHotSlots / 1024; ++q)
      if (myb[q] != 0xFFFFFFFFu) order[bcnt[myb[q]] + myr[q]] = (unsigned short)(q * 1024 + threadIdx.x);
  }
}

// part_hist_kernel for int32 keys without weights, with the hot-key lookup (see above).
// cold[] is an Arrow bitmap over whole tiles: bit = row valid AND key not hot.
__global__ __launch_bounds__(1024) void part_hist_hot_kernel(
    const int32_t *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n, int b1,
    int bits, const int32_t *__restrict__ image, unsigned *block_hist, unsigned *tile_hist,
    uint64_t ntiles, uint8_t *cold, unsigned *hot_cnt, uint64_t *state) {
  using K = int32_t;
  constexpr K EMPTY = DKey<K>::empty;
  __shared__ unsigned h[kMaxFine];
  using BucketT = std::conditional<kHotWidth == 4, int4, int2>::type;
  __shared__ BucketT tk[kHotBuckets];
  __shared__ unsigned tc[kHotSlots];
  __shared__ unsigned ht[256];
  __shared__ unsigned long long s_nulls;
  const int nb = 1 << bits, nc = 1 << b1;
  for (int i = threadIdx.x; i < nb; i += 1024) h[i] = 0;
  for (int i = threadIdx.x; i < kHotBuckets; i += 1024)
    tk[i] = reinterpret_cast<const BucketT *>(image)[i];
  for (int i = threadIdx.x; i < kHotSlots; i += 1024) tc[i] = 0;
  if (threadIdx.x == 0) s_nulls = 0;
  unsigned long long nulls = 0;
  constexpr int VEC = 4;
  constexpr int NV = kTile / VEC / 1024;
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (threadIdx.x < 256) ht[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t row0 = tile * kTile;
    int4 pack[NV];
    unsigned vb[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const uint64_t i0 = row0 + ((uint64_t)u * 1024 + threadIdx.x) * VEC;
      vb[u] = 0x10000;  // not a full in-range vector
      if (i0 + VEC <= n) {
        pack[u] = *reinterpret_cast<const int4 *>(keys + i0);
        vb[u] = valid ? (unsigned)valid[i0 >> 3] : 0xFFu;
      }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const uint64_t i0 = row0 + ((uint64_t)u * 1024 + threadIdx.x) * VEC;
      K kv[VEC];
      unsigned bits_ok = 0, in_range = 0;
      if (!(vb[u] & 0x10000)) {
        kv[0] = pack[u].x;
        kv[1] = pack[u].y;
        kv[2] = pack[u].z;
        kv[3] = pack[u].w;
        bits_ok = (vb[u] >> (i0 & 7)) & 0xFu;
        in_range = 0xFu;
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          kv[j] = 0;
          if (i0 + j < n) {
            in_range |= 1u << j;
            if (bit_valid(valid, i0 + j)) {
              kv[j] = keys[i0 + j];
              bits_ok |= 1u << j;
            }
          }
        }
      }
      // the buckets of all keys of the vector are requested before any of them is used
      BucketT bk[VEC];
      uint32_t sa[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        sa[j] = hot_bucket(kv[j]);
        bk[j] = tk[sa[j]];
      }
      unsigned cold_bits = 0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if ((bits_ok >> j) & 1) {
          const K key = kv[j];
          const int slot = hot_find(bk[j], key, kHotWidth * sa[j]);
          if (slot >= 0 && key != EMPTY) {
            atomicAdd(&tc[slot], 1u);
          } else {
            cold_bits |= 1u << j;
            const unsigned fine = part_hash<K>(key) >> (32 - bits);
            atomicAdd(&ht[fine >> (bits - b1)], 1u);
            if (bits > b1) atomicAdd(&h[fine], 1u);
          }
        }
      }
      nulls += __popc(in_range & ~bits_ok);
      // one bitmap byte = the vectors of two neighbouring lanes (i0 is a multiple of 4)
      const unsigned other = __shfl_xor(cold_bits, 1, 64);
      if ((threadIdx.x & 1) == 0) cold[i0 >> 3] = (uint8_t)(cold_bits | (other << 4));
    }
    __syncthreads();
    if ((int)threadIdx.x < nc) {
      const unsigned c = ht[threadIdx.x];
      tile_hist[(uint64_t)threadIdx.x * ntiles + tile] = c;
      if (bits == b1) h[threadIdx.x] += c;
    }
    __syncthreads();
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 1024) block_hist[(uint64_t)blockIdx.x * nb + i] = h[i];
  for (int i = threadIdx.x; i < kHotSlots; i += 1024)
    hot_cnt[(uint64_t)blockIdx.x * kHotSlots + i] = tc[i];
  if (nulls) atomicAdd(&s_nulls, nulls);
  __syncthreads();
  if (threadIdx.x == 0 && s_nulls) atomicAdd((unsigned long long *)&state[DS_NULLS], s_nulls);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd((unsigned long long *)&state[DS_ROWS], (unsigned long long)n);
}

// column sums of the per-workgroup hot counters -> entries appended to the output list.
// 64 slots per workgroup x 16 groups of counter rows: every thread sums nblocks / 16 values
// with all loads in flight (one thread per slot walking 256 rows took 100 us).
constexpr int kHotRedGroups = 16;
__global__ __launch_bounds__(64 * kHotRedGroups) void hot_reduce_kernel(
    const int32_t *__restrict__ image, const unsigned *__restrict__ hot_cnt, int nblocks,
    int32_t *out_keys, int64_t *out_cnt, uint64_t out_cap, uint64_t *state) {
  constexpr int32_t EMPTY = DKey<int32_t>::empty;
  __shared__ unsigned long long part[kHotRedGroups][64];
  const unsigned l = threadIdx.x & 63, g = threadIdx.x >> 6;
  const unsigned slot = blockIdx.x * 64 + l;
  unsigned long long t = 0;
#pragma unroll 16
  for (int b = (int)g; b < nblocks; b += kHotRedGroups) t += hot_cnt[(uint64_t)b * kHotSlots + slot];
  part[g][l] = t;
  __syncthreads();
  if (g != 0) return;  // one wave finishes the 64 slots
  unsigned long long tot = 0;
#pragma unroll
  for (int q = 0; q < kHotRedGroups; ++q) tot += part[q][l];
  const int32_t key = image[slot];
  if (key == EMPTY) tot = 0;
  const unsigned long long peers = __ballot(tot > 0);
  const unsigned total = (unsigned)__popcll(peers);
  if (total == 0) return;
  unsigned long long b0 = 0;
  if (l == 0) b0 = atomicAdd((unsigned long long *)&state[DS_OUT], (unsigned long long)total);
  const unsigned long long base = __shfl(b0, 0, 64);
  if (base + total > out_cap) {
    if (l == 0) atomicOr((unsigned long long *)&state[DS_OVF], 2ull);
    return;
  }
  if (tot > 0) {
    const uint64_t pos = base + (unsigned)__popcll(peers & ((1ull << l) - 1ull));
    out_keys[pos] = key;
    out_cnt[pos] = (int64_t)tot;
  }
  unsigned long long mx = tot;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if (l == 0 && mx > 0) {
    unsigned long long *gm = reinterpret_cast<unsigned long long *>(&state[NVT_ST_MAXCOUNT]);
    if (mx > __hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(gm, mx);
  }
}

// P0b-1: bucket totals = column sums of the per-block histograms.  64 bins x 16 row groups
// per workgroup; loads are coalesced across bins, 16 in flight per lane, 2 batches per lane
// (with 4 row groups the 128-deep per-lane chain made this 31 us for 512 KB of input).
constexpr int kReduceGroups = 16;
__global__ __launch_bounds__(64 * kReduceGroups) void part_reduce_kernel(
    const unsigned *__restrict__ block_hist, int nblocks, int nb, unsigned long long *totals) {
  __shared__ unsigned long long part[kReduceGroups][64];
  const int f = blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  unsigned long long t = 0;
  if (f < nb) {
#pragma unroll 16
    for (int b = g; b < nblocks; b += kReduceGroups) t += block_hist[(uint64_t)b * nb + f];
  }
  part[g][threadIdx.x & 63] = t;
  __syncthreads();
  if (g == 0 && f < nb) {
    unsigned long long tot = 0;
#pragma unroll
    for (int k = 0; k < kReduceGroups; ++k) tot += part[k][threadIdx.x];
    totals[f] = tot;
  }
}

// P0b-2: exclusive scan -> exact bucket starts, cursors, per-coarse tile starts. One block.
__global__ __launch_bounds__(1024) void part_scan_kernel(const unsigned long long *__restrict__ totals,
                                                         int bits, int b1,
                                                         unsigned long long *fine_start,
                                                         unsigned long long *fine_cursor,
                                                         unsigned long long *coarse_cursor,
                                                         unsigned *tile_start,
                                                         unsigned *chunk_start,
                                                         unsigned *pchunk_start,
                                                         unsigned long long chunk_rows,
                                                         unsigned long long small_rows) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long carry;
  const int nb = 1 << bits;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    int f = base + threadIdx.x;
    unsigned long long v = f < nb ? totals[f] : 0, inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned long long o = __shfl_up(inc, off, 64);
      if (lane_id() >= (unsigned)off) inc += o;
    }
    const unsigned w = threadIdx.x / kWave;
    if (lane_id() == 63) wsum[w] = inc;
    __syncthreads();
    unsigned long long wb = carry;
    for (unsigned k = 0; k < w; ++k) wb += wsum[k];
    if (f < nb) {
      fine_start[f] = wb + inc - v;
      fine_cursor[f] = wb + inc - v;
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = wb + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) fine_start[nb] = carry;
  __syncthreads();
  const int nc = 1 << b1, sub = nb >> b1;
  if ((int)threadIdx.x < nc) coarse_cursor[threadIdx.x] = fine_start[threadIdx.x * sub];
  {  // per-coarse-bucket tile counts -> exclusive scan (nc <= 256: one value per thread; a
     // serial loop over dependent global loads here cost 25 us per column)
    __shared__ unsigned tcnt[256];
    if ((int)threadIdx.x < nc) {
      const unsigned long long sz =
          fine_start[(threadIdx.x + 1) * sub] - fine_start[threadIdx.x * sub];
      tcnt[threadIdx.x] = (unsigned)((sz + kTile - 1) / kTile);
    }
    __syncthreads();
    if (threadIdx.x < kWave) {  // one wave scans the <= 256 counts, 4 per lane
      unsigned v[4], tot = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (int)threadIdx.x * 4 + j;
        v[j] = c < nc ? tcnt[c] : 0;
        tot += v[j];
      }
      unsigned inc = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        unsigned o = __shfl_up(inc, off, 64);
        if (lane_id() >= (unsigned)off) inc += o;
      }
      unsigned run = inc - tot;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (int)threadIdx.x * 4 + j;
        if (c < nc) tile_start[c] = run;
        run += v[j];
      }
      if (threadIdx.x == kWave - 1) tile_start[nc] = inc;
    }
  }
  // P3 work list: a fine bucket is processed as one primary chunk of chunk_rows plus, when a
  // hot key drags its whole bucket (skew), excess chunks of small_rows; such "split" buckets
  // get one partial-list region per chunk, merged per bucket by P4.  Two more block scans.
  __syncthreads();
  __shared__ unsigned long long wsum2[16][2];
  __shared__ unsigned long long carry2[2];
  if (threadIdx.x == 0) carry2[0] = carry2[1] = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    int f = base + threadIdx.x;
    unsigned long long sz = f < nb ? fine_start[f + 1] - fine_start[f] : 0;
    // primary chunk of chunk_rows, the excess (skew) in chunks of small_rows
    unsigned long long k = sz <= chunk_rows ? (sz > 0)
                                            : 1 + (sz - chunk_rows + small_rows - 1) / small_rows;
    unsigned long long v0 = k, v1 = (k > 1) ? k : 0, i0 = v0, i1 = v1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned long long o0 = __shfl_up(i0, off, 64), o1 = __shfl_up(i1, off, 64);
      if (lane_id() >= (unsigned)off) {
        i0 += o0;
        i1 += o1;
      }
    }
    const unsigned w = threadIdx.x / kWave;
    if (lane_id() == 63) {
      wsum2[w][0] = i0;
      wsum2[w][1] = i1;
    }
    __syncthreads();
    unsigned long long b0 = carry2[0], b1c = carry2[1];
    for (unsigned q = 0; q < w; ++q) {
      b0 += wsum2[q][0];
      b1c += wsum2[q][1];
    }
    if (f < nb) {
      chunk_start[f] = (unsigned)(b0 + i0 - v0);
      pchunk_start[f] = (unsigned)(b1c + i1 - v1);
    }
    __syncthreads();
    if (threadIdx.x == 1023) {
      carry2[0] = b0 + i0;
      carry2[1] = b1c + i1;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    chunk_start[nb] = (unsigned)carry2[0];
    pchunk_start[nb] = (unsigned)carry2[1];
  }
}

// P1 / P2: LDS-staged scatter of one tile of rows into 2^nbits buckets.
//   LEVEL 1: tile t covers input rows [t*kTile, ...); bucket = top b1 bits of the hash.
//   LEVEL 2: tiles are laid out per coarse bucket (tile_start); bucket = the next nbits.
template <typename K, int LEVEL, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void part_scatter_kernel(
    const K *__restrict__ keys, const uint8_t *__restrict__ valid,
    const int64_t *__restrict__ weights, uint64_t n, int b1, int nbits,
    const unsigned long long *__restrict__ fine_start, unsigned long long *cursor,
    const unsigned *__restrict__ tile_start, const unsigned *__restrict__ tile_off,
    const unsigned long long *__restrict__ tile_off_base, K *__restrict__ out_keys,
    int64_t *__restrict__ out_w) {
  constexpr int ROWS = kTile / kBlock;
  __shared__ K stage[kTile];
  __shared__ unsigned lcnt[256], loff[256];
  __shared__ unsigned long long gbase[256];
  __shared__ uint64_t seg_lo, seg_hi;
  __shared__ int coarse_s;
  const int nbk = 1 << nbits;
  if (LEVEL == 1) {
    if (threadIdx.x == 0) {
      seg_lo = (uint64_t)blockIdx.x * kTile;
      seg_hi = seg_lo + kTile < n ? seg_lo + kTile : n;
      coarse_s = 0;
    }
  } else {
    if (threadIdx.x == 0) {
      const int nc = 1 << b1;
      int c = -1;
      if (blockIdx.x < tile_start[nc]) {
        int lo = 0, hi = nc - 1;  // last c with tile_start[c] <= blockIdx.x
        while (lo < hi) {
          int mid = (lo + hi + 1) >> 1;
          if (tile_start[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
        }
        c = lo;
      }
      coarse_s = c;
      if (c >= 0) {
        const int sub = nbk;
        uint64_t cs = fine_start[(uint64_t)c * sub], ce = fine_start[(uint64_t)(c + 1) * sub];
        seg_lo = cs + (uint64_t)(blockIdx.x - tile_start[c]) * kTile;
        seg_hi = seg_lo + kTile < ce ? seg_lo + kTile : ce;
      }
    }
  }
  if (threadIdx.x < 256) lcnt[threadIdx.x] = 0;
  __syncthreads();
  if (LEVEL == 2 && coarse_s < 0) return;
  const uint64_t lo = seg_lo, hi = seg_hi;
  const int shift = (LEVEL == 1) ? (32 - b1) : (32 - b1 - nbits);
  const uint32_t mask = (uint32_t)nbk - 1;

  constexpr int RSLOTS = ROWS + (LEVEL == 2 ? 1 : 0);  // LEVEL 2: + one row of the overhang
  K k[RSLOTS];
  unsigned pos[RSLOTS];
  unsigned short bk[RSLOTS];
  // row handled by register slot r.  LEVEL 1 reads the (16-byte aligned) input column with
  // one 16-byte load per lane and takes the VEC validity bits from a single bitmap byte;
  // LEVEL 2 segments start anywhere, so they are read element-wise.
  constexpr int VEC = DKey<K>::vec;
  // LEVEL 2 segments start anywhere: they are read with 16-byte loads from the aligned
  // address below `lo` (rows outside [lo, hi) masked off); the up to VEC - 1 rows this pushes
  // past the last full vector are the "overhang", one per thread 0 .. VEC-2, in slot ROWS.
  // (Element-wise loads issued 4x the load instructions: 174 us against 94 us for LEVEL 1.)
  const uint64_t a0 = LEVEL == 1 ? lo : (lo & ~(uint64_t)(VEC - 1));
  auto row_of = [&](int r) -> uint64_t {
    if (r == ROWS) return a0 + (uint64_t)kTile + threadIdx.x;  // overhang (LEVEL 2 only)
    return a0 + ((uint64_t)(r / VEC) * kBlock + threadIdx.x) * VEC + (r % VEC);
  };
  if (LEVEL == 1) {
    using VecT = typename std::conditional<sizeof(K) == 4, int4, longlong2>::type;
    constexpr int NV = ROWS / VEC;
    VecT pack[NV];
    unsigned vraw[NV];
    // phase 1: issue every load of the tile (keys + raw bitmap bytes), no dependent math
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const uint64_t i0 = row_of(u * VEC);
      vraw[u] = 0x10000;  // not a full in-range vector
      if (i0 + VEC <= hi) {
        pack[u] = *reinterpret_cast<const VecT *>(keys + i0);
        vraw[u] = valid ? (unsigned)valid[i0 >> 3] : 0xFFu;
      }
    }
    // phase 2: bucket + rank
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const uint64_t i0 = row_of(u * VEC);
      unsigned vb = 0;
      K kv[VEC];
      if (!(vraw[u] & 0x10000)) {
        if constexpr (sizeof(K) == 4) {
          kv[0] = pack[u].x;
          kv[1] = pack[u].y;
          kv[2] = pack[u].z;
          kv[3] = pack[u].w;
        } else {
          kv[0] = pack[u].x;
          kv[1] = pack[u].y;
        }
        vb = (vraw[u] >> (i0 & 7)) & ((1u << VEC) - 1u);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          kv[j] = 0;
          if (i0 + j < hi && bit_valid(valid, i0 + j)) {
            kv[j] = keys[i0 + j];
            vb |= 1u << j;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int r = u * VEC + j;
        bk[r] = 0xFFFF;
        k[r] = kv[j];
        if ((vb >> j) & 1) {
          unsigned b = (part_hash<K>(kv[j]) >> shift) & mask;
          bk[r] = (unsigned short)b;
          pos[r] = atomicAdd(&lcnt[b], 1u);
        }
      }
    }
  } else {
    using VecT = typename std::conditional<sizeof(K) == 4, int4, longlong2>::type;
    constexpr int NV = ROWS / VEC;
    VecT pack[NV];
    bool full[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const uint64_t i0 = row_of(u * VEC);
      full[u] = i0 < hi && i0 + VEC <= n;  // the 16 bytes exist (n = length of the buffer)
      if (full[u]) pack[u] = *reinterpret_cast<const VecT *>(keys + i0);
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const uint64_t i0 = row_of(u * VEC);
      K kv[VEC];
      if (full[u]) {
        if constexpr (sizeof(K) == 4) {
          kv[0] = pack[u].x;
          kv[1] = pack[u].y;
          kv[2] = pack[u].z;
          kv[3] = pack[u].w;
        } else {
          kv[0] = pack[u].x;
          kv[1] = pack[u].y;
        }
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) kv[j] = (i0 + j >= lo && i0 + j < hi) ? keys[i0 + j] : (K)0;
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int r = u * VEC + j;
        bk[r] = 0xFFFF;
        k[r] = kv[j];
        if (i0 + j >= lo && i0 + j < hi) {
          unsigned b = (part_hash<K>(kv[j]) >> shift) & mask;
          bk[r] = (unsigned short)b;
          pos[r] = atomicAdd(&lcnt[b], 1u);
        }
      }
    }
    {  // overhang rows a0 + kTile .. a0 + kTile + VEC - 2
      const uint64_t i = row_of(ROWS);
      bk[ROWS] = 0xFFFF;
      k[ROWS] = (K)0;
      if (threadIdx.x < VEC - 1 && i >= lo && i < hi) {
        k[ROWS] = keys[i];
        unsigned b = (part_hash<K>(k[ROWS]) >> shift) & mask;
        bk[ROWS] = (unsigned short)b;
        pos[ROWS] = atomicAdd(&lcnt[b], 1u);
      }
    }
  }
  __syncthreads();
  // exclusive scan of lcnt over the block (kBlock == 256 >= buckets) + global reservation
  {
    __shared__ unsigned ws4[kBlock / kWave];
    const unsigned v = (int)threadIdx.x < nbk ? lcnt[threadIdx.x] : 0;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o = __shfl_up(inc, off, 64);
      if (lane_id() >= (unsigned)off) inc += o;
    }
    const unsigned w = threadIdx.x / kWave;
    if (lane_id() == 63) ws4[w] = inc;
    __syncthreads();
    unsigned add = 0;
    for (unsigned q = 0; q < w; ++q) add += ws4[q];
    loff[threadIdx.x] = add + inc - v;
    if ((int)threadIdx.x < nbk && v) {
      if (LEVEL == 1) {
        // exact offset of this (tile, bucket) from the scanned per-tile histograms
        gbase[threadIdx.x] =
            scan_lookup(tile_off, tile_off_base, (uint64_t)threadIdx.x * gridDim.x + blockIdx.x);
      } else {
        unsigned long long *cur = cursor + (uint64_t)coarse_s * nbk;
        gbase[threadIdx.x] = atomicAdd(&cur[threadIdx.x], (unsigned long long)v);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RSLOTS; ++r)
    if (bk[r] != 0xFFFF) stage[loff[bk[r]] + pos[r]] = k[r];
  __syncthreads();
  const unsigned total = loff[nbk - 1] + lcnt[nbk - 1];
  for (unsigned i = threadIdx.x; i < total; i += kBlock) {
    K key = stage[i];
    unsigned b = (part_hash<K>(key) >> shift) & mask;
    out_keys[gbase[b] + (i - loff[b])] = key;
  }
  if (WEIGHTED) {
    // weights ride along: same destination, recomputed from (bucket, pos)
#pragma unroll
    for (int r = 0; r < RSLOTS; ++r) {
      if (bk[r] != 0xFFFF) out_w[gbase[bk[r]] + pos[r]] = weights[row_of(r)];
    }
  }
}

// P3: one workgroup per (fine bucket, chunk of kChunk rows).  Single-chunk buckets go
// straight to the output list; chunks of split buckets write partial lists for P4.
template <typename K, bool WEIGHTED, int SLOTS, int BS>
__global__ __launch_bounds__(BS) void part_count_kernel(
    const K *__restrict__ keys, const int64_t *__restrict__ weights,
    const unsigned long long *__restrict__ fine_start, const unsigned *__restrict__ chunk_start,
    const unsigned *__restrict__ pchunk_start, int nb, uint64_t chunk_rows, uint64_t small_rows,
    K *part_keys, int64_t *part_cnt, unsigned *part_len, K *tmp_keys, int64_t *tmp_cnt, unsigned *blk_cnt,
    unsigned long long *blk_lo, uint64_t *state) {
  constexpr K EMPTY = DKey<K>::empty;
  using C = typename std::conditional<WEIGHTED, unsigned long long, unsigned>::type;
  __shared__ K lkeys[SLOTS];
  __shared__ C lcnt[SLOTS];
  __shared__ unsigned lfill, lovf;
  __shared__ unsigned long long s_sent;
  __shared__ int s_f;
  __shared__ unsigned s_j;
  // Unit order = dispatch order: the nb primary chunks first (one per bucket), then the small
  // excess chunks of split buckets, which fill the tail.  (With equal-size chunks a hot
  // bucket's ~30 extra units started a whole second round on the 256 CUs: +130 us per column.)
  if (threadIdx.x == 0) {
    int f = -1;
    unsigned j = 0;
    if ((int)blockIdx.x < nb) {
      if (chunk_start[blockIdx.x + 1] > chunk_start[blockIdx.x]) f = (int)blockIdx.x;
    } else {
      const unsigned p = blockIdx.x - (unsigned)nb;  // index into the split buckets' regions
      if (p < pchunk_start[nb]) {
        int lo = 0, hi = nb - 1;  // last f with pchunk_start[f] <= p (skips unsplit buckets)
        while (lo < hi) {
          int mid = (lo + hi + 1) >> 1;
          if (pchunk_start[mid] <= p) lo = mid; else hi = mid - 1;
        }
        j = p - pchunk_start[lo];
        if (j > 0) f = lo;  // j == 0 is the primary chunk, already a unit of its own
      }
    }
    s_f = f;
    s_j = j;
    lfill = 0;
    lovf = 0;
    s_sent = 0;
  }
  __syncthreads();
  const int f = s_f;
  if (f < 0) {
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = 0;
    return;
  }
  for (int i = threadIdx.x; i < SLOTS; i += BS) {
    lkeys[i] = EMPTY;
    lcnt[i] = 0;
  }
  __syncthreads();
  const unsigned j = s_j;
  const unsigned nchunks = chunk_start[f + 1] - chunk_start[f];
  const uint64_t end = fine_start[f + 1];
  const uint64_t lo = fine_start[f] + (j == 0 ? 0 : chunk_rows + (uint64_t)(j - 1) * small_rows);
  const uint64_t span = j == 0 ? chunk_rows : small_rows;
  const uint64_t hi = lo + span < end ? lo + span : end;
  bool failed = false;
  unsigned long long my_sent = 0;
  // Split buckets exist because of a hot key, and in their chunks most lanes of every wave
  // would add to the SAME LDS word (a 64-way same-address conflict serialises the atomic:
  // such chunks ran ~4x slower per row).  Sample 64 rows of the chunk; a key holding >= 25 %
  // of the sample is counted in a per-lane register instead and added once per wave.
  __shared__ K s_hk;
  __shared__ int s_has_hk;
  K hk = EMPTY;
  bool has_hk = false;
  if (nchunks > 1) {  // workgroup-uniform
    if (threadIdx.x < kWave) {
      const K smp = keys[lo + ((hi - lo) * threadIdx.x) / kWave];
      K best = EMPTY;
      int bestc = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const K cand = __shfl(smp, c * 16 + 5, 64);
        const int m = __popcll(__ballot(smp == cand));
        if (m > bestc) {
          bestc = m;
          best = cand;
        }
      }
      if (threadIdx.x == 0) {
        s_has_hk = (bestc >= 16 && best != EMPTY) ? 1 : 0;
        s_hk = best;
      }
    }
    __syncthreads();
    has_hk = s_has_hk != 0;
    hk = s_hk;
  }
  unsigned long long my_hot = 0;
  auto add_one = [&](K key, unsigned long long w) {
    if (key == EMPTY) {
      my_sent += w;
      return;
    }
    if (has_hk && key == hk) {
      my_hot += w;
      return;
    }
    if (!lds_add<K, C, SLOTS>(lkeys, lcnt, &lfill, key, (C)w, part_hash<K>(key))) failed = true;
  };
  if constexpr (!WEIGHTED) {
    // bucket segments start anywhere: peel to a 16-byte boundary, then 16-byte loads (the
    // element-wise version issued 4x the load instructions and ran at half the speed of the
    // stage-1 kernel on the same number of rows per CU)
    constexpr int VEC = DKey<K>::vec;
    using VecT = typename std::conditional<sizeof(K) == 4, int4, longlong2>::type;
    const uint64_t head = (lo + VEC - 1) / VEC * VEC < hi ? (lo + VEC - 1) / VEC * VEC : hi;
    const uint64_t body_end = head + (hi - head) / VEC * VEC;
    for (uint64_t i = lo + threadIdx.x; i < head; i += BS) add_one(keys[i], 1ull);
    for (uint64_t i = body_end + threadIdx.x; i < hi; i += BS) add_one(keys[i], 1ull);
    const VecT *vk = reinterpret_cast<const VecT *>(keys + head);
    const uint64_t nvec = (body_end - head) / VEC;
    constexpr int U = 4;
    for (uint64_t v0 = threadIdx.x; v0 < nvec; v0 += (uint64_t)BS * U) {
      if (lfill > (unsigned)max_fill(SLOTS)) break;
      VecT pack[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t v = v0 + (uint64_t)u * BS;
        if (v < nvec) pack[u] = vk[v];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (v0 + (uint64_t)u * BS >= nvec) continue;
        if constexpr (sizeof(K) == 4) {
          add_one(pack[u].x, 1ull);
          add_one(pack[u].y, 1ull);
          add_one(pack[u].z, 1ull);
          add_one(pack[u].w, 1ull);
        } else {
          add_one(pack[u].x, 1ull);
          add_one(pack[u].y, 1ull);
        }
      }
    }
  } else {
    constexpr int U = 8;
    for (uint64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (uint64_t)BS * U) {
      if (lfill > (unsigned)max_fill(SLOTS)) break;
      K kk[U];
      unsigned long long ww[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint64_t i = i0 + (uint64_t)u * BS;
        ww[u] = 0;
        if (i < hi) {
          kk[u] = keys[i];
          ww[u] = (unsigned long long)weights[i];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + (uint64_t)u * BS < hi) add_one(kk[u], ww[u]);
    }
  }
  if (has_hk) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_hot += __shfl_down(my_hot, off, 64);
    if (lane_id() == 0 && my_hot > 0 &&
        !lds_add<K, C, SLOTS>(lkeys, lcnt, &lfill, hk, (C)my_hot, part_hash<K>(hk)))
      failed = true;
  }
  if (failed) atomicOr(&lovf, 1u);
  if (my_sent) atomicAdd(&s_sent, my_sent);
  __syncthreads();
  if (lovf || lfill > (unsigned)max_fill(SLOTS)) {
    if (threadIdx.x == 0) {
      atomicOr((unsigned long long *)&state[DS_OVF], 1ull);
      blk_cnt[blockIdx.x] = 0;
    }
    return;
  }
  if (threadIdx.x == 0 && s_sent) atomicAdd((unsigned long long *)&state[DS_SENT], s_sent);
  if (nchunks == 1) {
    // No output cursor here: thousands of workgroups bumping one word serialise at the
    // memory side and made this kernel 2x slower.  The distinct keys of rows [lo, hi) fit in
    // tmp[lo, hi); part_offsets_kernel / part_copy_kernel pack the pieces afterwards.
    if (threadIdx.x == 0) blk_lo[blockIdx.x] = lo;
    lds_flush_region<K, C, BS, SLOTS>(lkeys, lcnt, tmp_keys + lo, tmp_cnt + lo,
                                      &blk_cnt[blockIdx.x], state);
  } else {
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = 0;
    const uint64_t region = (uint64_t)(pchunk_start[f] + j);
    lds_flush_region<K, C, BS, SLOTS>(lkeys, lcnt, part_keys + region * max_fill(SLOTS),
                                      part_cnt + region * max_fill(SLOTS), &part_len[region]);
  }
}

// P3b: exclusive scan of the per-workgroup result counts -> packed offsets; the total seeds
// the output cursor that P4 continues from.  One workgroup.
__global__ __launch_bounds__(1024) void part_offsets_kernel(const unsigned *__restrict__ blk_cnt,
                                                            unsigned nblk,
                                                            unsigned long long *blk_off,
                                                            uint64_t out_cap, uint64_t *state) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (unsigned base = 0; base < nblk; base += 1024) {
    unsigned i = base + threadIdx.x;
    unsigned long long v = i < nblk ? blk_cnt[i] : 0, inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned long long o = __shfl_up(inc, off, 64);
      if (lane_id() >= (unsigned)off) inc += o;
    }
    const unsigned w = threadIdx.x / kWave;
    if (lane_id() == 63) wsum[w] = inc;
    __syncthreads();
    unsigned long long wb = carry;
    for (unsigned k = 0; k < w; ++k) wb += wsum[k];
    if (i < nblk) blk_off[i] = wb + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = wb + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (carry > out_cap) atomicOr((unsigned long long *)&state[DS_OVF], 2ull);
    state[DS_OUT] = carry > out_cap ? 0 : carry;
  }
}

// P3c: pack every workgroup's staged result into the output list (coalesced copies)
template <typename K>
__global__ __launch_bounds__(kBlock) void part_copy_kernel(
    const K *__restrict__ tmp_keys, const int64_t *__restrict__ tmp_cnt,
    const unsigned *__restrict__ blk_cnt, const unsigned long long *__restrict__ blk_off,
    const unsigned long long *__restrict__ blk_lo, K *out_keys, int64_t *out_cnt,
    const uint64_t *__restrict__ state) {
  const unsigned cnt = blk_cnt[blockIdx.x];
  if (cnt == 0 || (state[DS_OVF] & 2)) return;
  const unsigned long long src = blk_lo[blockIdx.x], dst = blk_off[blockIdx.x];
  for (unsigned i = threadIdx.x; i < cnt; i += kBlock) {
    out_keys[dst + i] = tmp_keys[src + i];
    out_cnt[dst + i] = tmp_cnt[src + i];
  }
}

// P4: one workgroup per split bucket merges that bucket's per-chunk partial lists (same
// table geometry as P3, so whatever fitted there fits here).
template <typename K, typename C, int SLOTS>
__global__ __launch_bounds__(kStageBS) void part_merge_kernel(
    const unsigned *__restrict__ chunk_start, const unsigned *__restrict__ pchunk_start,
    const K *__restrict__ part_keys, const int64_t *__restrict__ part_cnt,
    const unsigned *__restrict__ part_len, K *out_keys, int64_t *out_cnt, uint64_t out_cap,
    unsigned long long *cursor, uint64_t *state) {
  constexpr K EMPTY = DKey<K>::empty;
  const int f = blockIdx.x;
  const unsigned nchunks = chunk_start[f + 1] - chunk_start[f];
  if (nchunks <= 1) return;
  // an earlier kernel of this call already overflowed: the result is discarded anyway
  // (and a full table would make every insert below walk kLdsProbe slots: 8 ms per launch)
  __shared__ K lkeys[SLOTS];
  __shared__ C lcnt[SLOTS];
  __shared__ unsigned lfill, lovf;
  __shared__ int s_skip;
  if (threadIdx.x == 0) {
    lfill = 0;
    lovf = 0;
    s_skip = (int)(__hip_atomic_load(&state[DS_OVF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1);
  }
  __syncthreads();
  if (s_skip) return;  // workgroup-uniform
  for (int i = threadIdx.x; i < SLOTS; i += kStageBS) {
    lkeys[i] = EMPTY;
    lcnt[i] = 0;
  }
  __syncthreads();
  bool failed = false;
  // a wave per region, 4 independent loads in flight per lane: a hot bucket has ~200 short
  // regions, and walking them one after the other with the whole workgroup was a chain of
  // ~400 dependent global-load latencies (140 us)
  const unsigned lane = lane_id(), wv = threadIdx.x / kWave;
  constexpr int UM = 4;
  for (unsigned j = wv; j < nchunks; j += kStageBS / kWave) {
    const uint64_t region = (uint64_t)(pchunk_start[f] + j);
    const unsigned len = part_len[region];
    const K *pk = part_keys + region * max_fill(SLOTS);
    const int64_t *pc = part_cnt + region * max_fill(SLOTS);
    for (unsigned i0 = lane; i0 < len; i0 += kWave * UM) {
      if (lfill > (unsigned)max_fill(SLOTS)) break;  // filling up: the call fails below
      K kk[UM];
      int64_t cc[UM];
#pragma unroll
      for (int u = 0; u < UM; ++u) {
        const unsigned i = i0 + u * kWave;
        if (i < len) {
          kk[u] = pk[i];
          cc[u] = pc[i];
        }
      }
#pragma unroll
      for (int u = 0; u < UM; ++u)
        if (i0 + u * kWave < len &&
            !lds_add<K, C, SLOTS>(lkeys, lcnt, &lfill, kk[u], (C)cc[u], part_hash<K>(kk[u])))
          failed = true;
    }
  }
  if (failed) atomicOr(&lovf, 1u);
  __syncthreads();
  if (lovf || lfill > (unsigned)max_fill(SLOTS)) {
    if (threadIdx.x == 0) atomicOr((unsigned long long *)&state[DS_OVF], 1ull);
    return;
  }
  lds_flush<K, C, kStageBS, SLOTS>(lkeys, lcnt, out_keys, out_cnt, out_cap, cursor, state);
}

inline uint64_t align16(uint64_t x) { return (x + 15) & ~15ull; }

// Partitioned paths:
//   1: ONE level, 256 buckets, 16384-slot tables (int32 keys, unweighted; 8192 otherwise):
//      up to ~2.5 M distinct keys with a single scatter pass;
//   2: 64 x 64 buckets, 4096-slot tables (8192 when weighted)      up to ~9 M distinct;
//   3: 64 x 256 buckets, 8192-slot tables                          up to ~32 M distinct.
struct PathCfg {
  int b1, b2, slots;
  uint64_t chunk_rows;  // primary chunk of a bucket
  uint64_t small_rows;  // chunk size for a bucket's excess rows (skew)
};
// upper bounds on the partial-list regions of split buckets and on P3 work units
inline uint64_t max_regions_of(const PathCfg &c, uint64_t n) {
  return n / c.small_rows + n / c.chunk_rows + 2;  // excess chunks + one primary per split bucket
}
inline uint64_t max_units_of(const PathCfg &c, uint64_t n, int nb) {
  return (uint64_t)nb + max_regions_of(c, n);
}
inline PathCfg path_cfg(int path, int key_bytes, int weighted, uint64_t n) {
  const bool small = weighted || key_bytes == 8;
  if (path == 1) {
    // one workgroup per bucket in the common case: chunk = average bucket + 15 %, so only
    // buckets inflated by a hot key are split (and merged by P4)
    uint64_t chunk = (n / 256) + (n / 256) / 7 + 1;
    chunk = chunk < 65536 ? 65536 : (chunk > (1ull << 20) ? (1ull << 20) : chunk);
    // the excess of a bucket inflated by a hot key is cut into eighths, dispatched after all
    // primary chunks, so it fills the tail instead of starting a second round
    uint64_t small_rows = chunk / NVT_SMALL_DIV < 16384 ? 16384 : chunk / NVT_SMALL_DIV;
    return {8, 0, small ? kLdsSlots : kLdsSlotsBig, chunk, small_rows};
  }
  if (path == 2) return {6, 6, weighted ? kLdsSlots : 4096, (uint64_t)kChunk, (uint64_t)kChunk};
  return {6, 8, kLdsSlots, (uint64_t)kChunk, (uint64_t)kChunk};
}

struct DenseWs {
  // path S
  char *p1_keys;
  int64_t *p1_cnt;
  unsigned *seg_off;
  // path P
  char *bufA, *bufB;
  int64_t *wA, *wB;
  unsigned *block_hist, *tile_start, *tile_hist;
  unsigned long long *scan_tot;
  unsigned long long *fine_start, *fine_cursor, *coarse_cursor, *totals;
  unsigned *chunk_start, *pchunk_start, *part_len;
  char *part_keys;
  int64_t *part_cnt;
  uint64_t max_regions;
  char *tmp_keys;      // [n] staged P3 results (row-range addressed)
  int64_t *tmp_cnt;    // [n]
  unsigned *blk_cnt;   // [t3 max]
  unsigned long long *blk_off, *blk_lo;
  // hot filter (path | NVT_PATH_HOT)
  int32_t *hot_image;   // [kHotSlots] 2-choice table image of the hot keys
  unsigned *hot_cnt;    // [kHotBlocks][kHotSlots] per-workgroup counters
  uint8_t *cold_bits;   // [ntiles * kTile / 8] valid AND not hot
};

// path argument -> stage-1 key-class bits (-1: a partitioned path)
inline int split_bits_of(int path) {
  return (path == 0 || path == 6) ? 0 : path == 7 ? 1 : -1;
}
inline int stage_slots(int key_bytes, int weighted) {
  return (weighted || key_bytes == 8) ? kLdsSlots : kLdsSlotsBig;
}

inline uint64_t dense_ws_layout(int key_bytes, uint64_t n, int path, int weighted, char *base,
                                DenseWs *ws) {
  const bool hot = (path & NVT_PATH_HOT) != 0;
  path &= ~NVT_PATH_HOT;
  uint64_t off = 0;
  auto take = [&](uint64_t bytes) {
    char *p = base ? base + off : nullptr;
    off += align16(bytes);
    return p;
  };
  DenseWs w;
  memset(&w, 0, sizeof(w));
  if (split_bits_of(path) >= 0) {
    const uint64_t nlists = (uint64_t)kSlabs << split_bits_of(path);
    const uint64_t cap = nlists * max_fill(stage_slots(key_bytes, weighted));
    w.p1_keys = take(cap * key_bytes);
    w.p1_cnt = (int64_t *)take(cap * 8);
    w.seg_off = (unsigned *)take((kRanges + 1) * nlists * 4);
  } else {
    w.bufA = take(n * key_bytes);
    w.bufB = take(n * key_bytes);
    if (weighted) {
      w.wA = (int64_t *)take(n * 8);
      w.wB = (int64_t *)take(n * 8);
    }
    w.block_hist = (unsigned *)take((uint64_t)kHistBlocks * kMaxFine * 4);
    w.tile_start = (unsigned *)take(260 * 4);
    {
      const uint64_t ntiles = (n + kTile - 1) / kTile, len = 256 * ntiles;
      w.tile_hist = (unsigned *)take(len * 4);
      w.scan_tot = (unsigned long long *)take(scan_chunks(len) * 8 + 8);
    }
    w.fine_start = (unsigned long long *)take((kMaxFine + 1) * 8);
    w.fine_cursor = (unsigned long long *)take((kMaxFine + 1) * 8);
    w.coarse_cursor = (unsigned long long *)take(256 * 8);
    w.totals = (unsigned long long *)take((uint64_t)kMaxFine * 8);
    w.chunk_start = (unsigned *)take((kMaxFine + 1) * 4);
    w.pchunk_start = (unsigned *)take((kMaxFine + 1) * 4);
    const PathCfg cfg = path_cfg(path, key_bytes, weighted, n);
    w.max_regions = max_regions_of(cfg, n);
    w.part_len = (unsigned *)take(w.max_regions * 4);
    w.part_keys = take(w.max_regions * max_fill(cfg.slots) * key_bytes);
    w.part_cnt = (int64_t *)take(w.max_regions * max_fill(cfg.slots) * 8);
    const uint64_t t3max = max_units_of(cfg, n, kMaxFine) + 1;
    w.tmp_keys = take(n * key_bytes);
    w.tmp_cnt = (int64_t *)take(n * 8);
    w.blk_cnt = (unsigned *)take(t3max * 4);
    w.blk_off = (unsigned long long *)take(t3max * 8);
    w.blk_lo = (unsigned long long *)take(t3max * 8);
    if (hot) {
      w.hot_image = (int32_t *)take(kHotSlots * 4);
      w.hot_cnt = (unsigned *)take((uint64_t)kHotBlocks * kHotSlots * 4);
      w.cold_bits = (uint8_t *)take((n + kTile - 1) / kTile * (kTile / 8));
    }
  }
  if (ws) *ws = w;
  return off;
}

template <typename K>
int dense_count(const K *keys, const uint8_t *valid, const int64_t *weights, uint64_t n, int path,
                void *wsp, K *out_keys, int64_t *out_cnt, uint64_t out_cap, uint64_t *state,
                hipStream_t s, bool clear_state = true, int32_t *hot_image_ext = nullptr,
                void *range_table = nullptr) {
  NVT_CHECK_ARG(state && wsp, "null state/workspace");
  if ((path & 0xFF) == NVT_PATH_SORT) {
    if constexpr (sizeof(K) == 4) {
      NVT_CHECK_ARG(weights == nullptr, "the sort path takes int32 keys without weights");
      NVT_CHECK_ARG(hot_image_ext != nullptr, "the sort path needs the column's histogram block");
      NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(keys) & 15) == 0, "keys must be 16-byte aligned");
      NVT_CHECK_ARG(n == 0 || (keys && out_keys && out_cnt), "null keys/out");
      if (clear_state) NVT_CHECK_HIP(hipMemsetAsync(state, 0, NVT_STATE_WORDS * 8, s));
      if (n == 0) return NVT_OK;
      return sort_count_i32((const int32_t *)keys, valid, n, wsp, (unsigned *)hot_image_ext,
                            (int32_t *)out_keys, out_cnt, out_cap, state, s);
    } else {
      set_error("dense_count: the sort path takes int32 keys");
      return NVT_EINVAL;
    }
  }
  if ((path & 0xFF) == NVT_PATH_RANGE) {
    if constexpr (sizeof(K) == 4) {
      NVT_CHECK_ARG(weights == nullptr, "the range path takes int32 keys without weights");
      NVT_CHECK_ARG(hot_image_ext != nullptr,
                    "the range path needs the column's aux block (nvt_dense_count_many, hot_image)");
      NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(keys) & 15) == 0, "keys must be 16-byte aligned");
      NVT_CHECK_ARG(n == 0 || (keys && out_keys && out_cnt), "null keys/out");
      NVT_CHECK_ARG(n < (1ull << 32), "at most 2^32-1 rows per call (32-bit LDS counters)");
      if (clear_state) NVT_CHECK_HIP(hipMemsetAsync(state, 0, NVT_STATE_WORDS * 8, s));
      if (n == 0) return NVT_OK;
      return range_count_i32((const int32_t *)keys, valid, n, (path >> 8) & 0xFF, wsp, hot_image_ext,
                             (int32_t *)out_keys, out_cnt, out_cap, range_table, state, s,
                             (path & NVT_PATH_PIECES) != 0);
    } else {
      set_error("dense_count: the range path takes int32 keys");
      return NVT_EINVAL;
    }
  }
  const int path_arg = path;
  const bool hot = (path & NVT_PATH_HOT) != 0;
  path &= ~NVT_PATH_HOT;
  NVT_CHECK_ARG(path == 0 || path == 6 || path == 7 || (path >= 1 && path <= 3),
                "path must be 0 / 6 / 7 (LDS tables: one key class / tiny / two key classes) or "
                "1 / 2 / 3 (partitioned)");
  NVT_CHECK_ARG(!hot || (path >= 1 && path <= 3 && sizeof(K) == 4 && weights == nullptr),
                "the hot filter takes int32 keys without weights on paths 1 / 2 / 3");
  NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(keys) & 15) == 0, "keys must be 16-byte aligned");
  NVT_CHECK_ARG(n == 0 || (keys && out_keys && out_cnt), "null keys/out");
  NVT_CHECK_ARG(n < (1ull << 32), "at most 2^32-1 rows per call (32-bit LDS counters)");
  static const char *const kPathName[12] = {"dense_count_p0", "dense_count_p1", "dense_count_p2",
                                            "dense_count_p3", "dense_count_p4", "dense_count_p5",
                                            "dense_count_p6", "dense_count_p7", "dense_count_p8",
                                            "dense_count_h1", "dense_count_h2", "dense_count_h3"};
  NVT_PROF(kPathName[hot ? 8 + path : path], n * sizeof(K), s);
  if (clear_state) NVT_CHECK_HIP(hipMemsetAsync(state, 0, NVT_STATE_WORDS * 8, s));
  DenseWs w;
  dense_ws_layout((int)sizeof(K), n, path_arg, weights != nullptr, (char *)wsp, &w);
  unsigned long long *cur = reinterpret_cast<unsigned long long *>(state);
  if (n == 0) return NVT_OK;
  const int sbits = split_bits_of(path);
  if (sbits >= 0) {
    // unweighted: every partial sum is < 2^32 (n is), so u32 counts and (int32 keys)
    // 16384-slot tables; weighted merges need u64 counts and use 8192 slots
    const unsigned nlists = (unsigned)kSlabs << sbits;
#define NVT_STAGE(C, S)                                                                          \
  do {                                                                                           \
    lds_stage_kernel<K, C, S><<<nlists, kStageBS, 0, s>>>(keys, valid, weights, n, sbits,        \
                                                          path == 6, (K *)w.p1_keys, w.p1_cnt,   \
                                                          w.seg_off, state);                     \
    NVT_CHECK_LAUNCH();                                                                          \
    range_merge_kernel<K><<<(unsigned)kRanges << sbits, kMergeBS, 0, s>>>(                       \
        (const K *)w.p1_keys, w.p1_cnt, w.seg_off, sbits, (uint64_t)max_fill(S), out_keys,       \
        out_cnt, out_cap, state);                                                                \
    NVT_CHECK_LAUNCH();                                                                          \
  } while (0)
    if (weights) {
      NVT_STAGE(unsigned long long, kLdsSlots);
    } else if constexpr (sizeof(K) == 4) {
      NVT_STAGE(unsigned, kLdsSlotsBig);
    } else {
      NVT_STAGE(unsigned, kLdsSlots);
    }
#undef NVT_STAGE
  } else {
    const PathCfg cfg = path_cfg(path, (int)sizeof(K), weights != nullptr, n);
    const int b1 = cfg.b1, b2 = cfg.b2, bits = b1 + b2;
    const uint64_t chunk_rows = cfg.chunk_rows;
    const unsigned t1 = (unsigned)((n + kTile - 1) / kTile);
    const K *fine_keys = nullptr;
    const int64_t *fine_w = nullptr;
    const unsigned t3 = (unsigned)max_units_of(cfg, n, 1 << bits);  // upper bound on P3 units
    int hist_blocks = kHistBlocks;
    if (hot) {
      if constexpr (sizeof(K) == 4) {
        if (hot_image_ext) {
          w.hot_image = hot_image_ext;  // sampled by nvt_dense_count_many ahead of the pipelines
        } else {
          HotSampleBatch hb;
          hb.c[0] = {(const int32_t *)keys, valid, n, w.hot_image, 0, 0};
          hot_sample_kernel<<<1, 1024, 0, s>>>(hb);
          NVT_CHECK_LAUNCH();
        }
        hist_blocks = kHotBlocks;
        part_hist_hot_kernel<<<kHotBlocks, 1024, 0, s>>>((const int32_t *)keys, valid, n, b1, bits,
                                                         w.hot_image, w.block_hist, w.tile_hist,
                                                         t1, w.cold_bits, w.hot_cnt, state);
        NVT_CHECK_LAUNCH();
        valid = w.cold_bits;  // the scatter sees the cold rows only
      }
    } else {
      part_hist_kernel<K><<<kHistBlocks, 1024, 0, s>>>(keys, valid, weights, n, b1, bits,
                                                         w.block_hist, w.tile_hist, t1, state);
      NVT_CHECK_LAUNCH();
    }
    const unsigned long long *tile_base = nullptr;  // last scan step is done by the P1 scatter
    {
      int rc = exclusive_scan_u32_deferred(w.tile_hist, ((uint64_t)1 << b1) * t1, w.scan_tot,
                                           &tile_base, s);
      if (rc) return rc;
    }
    part_reduce_kernel<<<((1 << bits) + 63) / 64, 64 * kReduceGroups, 0, s>>>(w.block_hist, hist_blocks,
                                                                  1 << bits, w.totals);
    NVT_CHECK_LAUNCH();
    part_scan_kernel<<<1, 1024, 0, s>>>(w.totals, bits, b1, w.fine_start, w.fine_cursor,
                                        w.coarse_cursor, w.tile_start, w.chunk_start,
                                        w.pchunk_start, chunk_rows, cfg.small_rows);
    NVT_CHECK_LAUNCH();
    const unsigned t2 = t1 + (1u << b1);  // upper bound: every coarse bucket rounds up once
    if (weights) {
      part_scatter_kernel<K, 1, true><<<t1, kBlock, 0, s>>>(keys, valid, weights, n, b1, b1,
                                                            w.fine_start, w.coarse_cursor,
                                                            w.tile_start, w.tile_hist, tile_base,
                                                            (K *)w.bufA, w.wA);
      NVT_CHECK_LAUNCH();
      fine_keys = (const K *)w.bufA;
      fine_w = w.wA;
      if (b2) {
        part_scatter_kernel<K, 2, true><<<t2, kBlock, 0, s>>>((const K *)w.bufA, nullptr, w.wA, n,
                                                              b1, b2, w.fine_start, w.fine_cursor,
                                                              w.tile_start, nullptr, nullptr,
                                                              (K *)w.bufB, w.wB);
        NVT_CHECK_LAUNCH();
        fine_keys = (const K *)w.bufB;
        fine_w = w.wB;
      }
    } else {
      part_scatter_kernel<K, 1, false><<<t1, kBlock, 0, s>>>(keys, valid, nullptr, n, b1, b1,
                                                             w.fine_start, w.coarse_cursor,
                                                             w.tile_start, w.tile_hist, tile_base,
                                                             (K *)w.bufA, nullptr);
      NVT_CHECK_LAUNCH();
      fine_keys = (const K *)w.bufA;
      if (b2) {
        part_scatter_kernel<K, 2, false><<<t2, kBlock, 0, s>>>((const K *)w.bufA, nullptr, nullptr,
                                                               n, b1, b2, w.fine_start,
                                                               w.fine_cursor, w.tile_start, nullptr,
                                                               nullptr, (K *)w.bufB, nullptr);
        NVT_CHECK_LAUNCH();
        fine_keys = (const K *)w.bufB;
      }
    }
#define NVT_P3P4(WEIGHTED, C, SLOTS, BS)                                                          \
  do {                                                                                            \
    part_count_kernel<K, WEIGHTED, SLOTS, BS><<<t3, BS, 0, s>>>(                                  \
        fine_keys, fine_w, w.fine_start, w.chunk_start, w.pchunk_start, 1 << bits, chunk_rows,    \
        cfg.small_rows,                                                                           \
        (K *)w.part_keys, w.part_cnt, w.part_len, (K *)w.tmp_keys, w.tmp_cnt, w.blk_cnt,         \
        w.blk_lo, state);                                                                         \
    NVT_CHECK_LAUNCH();                                                                           \
    part_offsets_kernel<<<1, 1024, 0, s>>>(w.blk_cnt, t3, w.blk_off, out_cap, state);             \
    NVT_CHECK_LAUNCH();                                                                           \
    part_copy_kernel<K><<<t3, kBlock, 0, s>>>((const K *)w.tmp_keys, w.tmp_cnt, w.blk_cnt,        \
                                              w.blk_off, w.blk_lo, out_keys, out_cnt, state);     \
    NVT_CHECK_LAUNCH();                                                                           \
    part_merge_kernel<K, C, SLOTS><<<1u << bits, kStageBS, 0, s>>>(                               \
        w.chunk_start, w.pchunk_start, (const K *)w.part_keys, w.part_cnt, w.part_len, out_keys,  \
        out_cnt, out_cap, &cur[DS_OUT], state);                                                   \
    NVT_CHECK_LAUNCH();                                                                           \
  } while (0)
    if (weights) {
      NVT_P3P4(true, unsigned long long, kLdsSlots, kCountBS);
    } else if (cfg.slots == kLdsSlotsBig) {
      if constexpr (sizeof(K) == 4) NVT_P3P4(false, unsigned, kLdsSlotsBig, 1024);
    } else if (cfg.slots == 4096) {
      NVT_P3P4(false, unsigned, 4096, kCountBS);
    } else {
      NVT_P3P4(false, unsigned, kLdsSlots, kCountBS);
    }
#undef NVT_P3P4
    if (hot) {
      if constexpr (sizeof(K) == 4) {
        hot_reduce_kernel<<<kHotSlots / 64, 64 * kHotRedGroups, 0, s>>>(w.hot_image, w.hot_cnt, kHotBlocks,
                                                          (int32_t *)out_keys, out_cnt, out_cap,
                                                          state);
        NVT_CHECK_LAUNCH();
      }
    }
  }
  return NVT_OK;
}

}  // namespace nvt

using namespace nvt;

extern "C" {

int nvt_range_table_bytes(int nb_log2, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes && nb_log2 >= 6 && nb_log2 <= 10, "64 .. 1024 buckets");
  *bytes = (((uint64_t)1 << nb_log2) * kRpRegion + kRpGuard) * 8;
  return NVT_OK;
}
int nvt_dense_count_ws_bytes(int key_bytes, uint64_t n, int path, int weighted, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes && (key_bytes == 4 || key_bytes == 8), "key_bytes must be 4 or 8");
  if ((path & 0xFF) == NVT_PATH_SORT) {
    NVT_CHECK_ARG(key_bytes == 4 && !weighted, "the sort path takes int32 keys without weights");
    *bytes = sort_count_ws_bytes(n) + 64;
    return NVT_OK;
  }
  if ((path & 0xFF) == NVT_PATH_RANGE) {
    const int nb_log2 = (path >> 8) & 0xFF;
    NVT_CHECK_ARG(key_bytes == 4 && !weighted && nb_log2 >= 6 && nb_log2 <= 10,
                  "the range path takes int32 keys without weights, 64 .. 1024 buckets");
    *bytes = range_count_ws_bytes(n, nb_log2) + 64;
    return NVT_OK;
  }
  NVT_CHECK_ARG((path & ~NVT_PATH_HOT) >= 0 && (path & ~NVT_PATH_HOT) <= 7, "path must be 0..7");
  NVT_CHECK_ARG(!(path & NVT_PATH_HOT) || ((path & ~NVT_PATH_HOT) >= 1 && (path & ~NVT_PATH_HOT) <= 3 &&
                                          key_bytes == 4 && !weighted),
                "the hot filter takes int32 keys without weights on paths 1 / 2 / 3");
  *bytes = dense_ws_layout(key_bytes, n, path, weighted, nullptr, nullptr) + 64;
  return NVT_OK;
}
int nvt_dense_count_many(const nvt_count_col *cols, int ncols, void *stream) {
  NVT_CHECK_ARG(ncols == 0 || cols, "null descriptors");
  // state blocks laid out back to back (the usual case: one tensor, one row per column) are
  // cleared by ONE memset instead of one tiny fill kernel per column
  bool contiguous = ncols > 1;
  for (int i = 0; i < ncols && contiguous; ++i)
    contiguous = cols[i].state != nullptr && cols[i].state == cols[0].state + (uint64_t)i * NVT_STATE_WORDS;
  if (contiguous)
    NVT_CHECK_HIP(hipMemsetAsync(cols[0].state, 0, (uint64_t)ncols * NVT_STATE_WORDS * 8,
                                 (hipStream_t)stream));
  // columns that were given DIFFERENT workspaces may run concurrently: each distinct ws pointer
  // (up to kSideStreams of them) gets an internal stream forked from / joined into `stream`, so
  // one column's short serial kernels (reduce / scan / offsets) hide under another's wide ones.
  // Columns sharing a workspace stay ordered on one stream.
  hipStream_t main_s = (hipStream_t)stream;
  std::vector<void *> wss;
  for (int i = 0; i < ncols; ++i)
    if (std::find(wss.begin(), wss.end(), cols[i].ws) == wss.end()) wss.push_back(cols[i].ws);
  SidePool *pool = nullptr;
  const bool fork = wss.size() > 1 && wss.size() <= (size_t)kSideStreams;
  if (wss.size() > (size_t)kSideStreams) {
    set_error("nvt_dense_count_many: at most %d distinct workspaces per call", kSideStreams);
    return NVT_EINVAL;
  }
  if (fork) {
    int rc = side_pool(1, &pool);
    if (rc) return rc;
    NVT_CHECK_HIP(hipEventRecord(pool->fork, main_s));
    for (size_t k = 0; k < wss.size(); ++k) NVT_CHECK_HIP(hipStreamWaitEvent(pool->s[k], pool->fork, 0));
  }
  // hot-key samples of every filtered column: ONE launch (a workgroup per column) on the
  // caller's stream.  The internal streams were forked before it: columns that need no sample
  // start at once, a stream waits for the samples only in front of its first filtered column.
  bool sampled = false;
  {
    HotSampleBatch hb;
    int nh = 0;
    auto flush = [&]() -> int {
      if (nh) {
        NVT_PROF("dense_count_sample", 0, main_s);
        hot_sample_kernel<<<nh, 1024, 0, main_s>>>(hb);
        NVT_CHECK_LAUNCH();
        sampled = true;
      }
      nh = 0;
      return NVT_OK;
    };
    for (int i = 0; i < ncols; ++i) {
      const nvt_count_col &c = cols[i];
      const bool range = (c.path & 0xFF) == NVT_PATH_RANGE;
      if (!((c.path & NVT_PATH_HOT) || range) || c.key_bytes != 4 || c.weights || c.n == 0 ||
          !c.hot_image)
        continue;
      hb.c[nh++] = {(const int32_t *)c.keys, c.valid, c.n, c.hot_image, range ? (c.path >> 8) & 0xFF : 0,
                    (range && (c.path & NVT_PATH_PIECES)) ? 1 : 0};
      if (nh == kHotBatch) {
        int rc = flush();
        if (rc) return rc;
      }
    }
    int rc = flush();
    if (rc) return rc;
  }
  bool waited[kSideStreams] = {false, false, false};
  int rc_all = NVT_OK;
  if (fork && sampled) NVT_CHECK_HIP(hipEventRecord(pool->aux, main_s));
  for (int i = 0; i < ncols; ++i) {
    const nvt_count_col &c = cols[i];
    hipStream_t cs = main_s;
    if (fork) {
      const size_t k = std::find(wss.begin(), wss.end(), c.ws) - wss.begin();
      cs = pool->s[k];
      if (sampled && ((c.path & NVT_PATH_HOT) || (c.path & 0xFF) == NVT_PATH_RANGE) && c.hot_image &&
          !waited[k]) {
        NVT_CHECK_HIP(hipStreamWaitEvent(cs, pool->aux, 0));
        waited[k] = true;
      }
    }
    int rc;
    if (c.key_bytes == 4)
      rc = dense_count<int32_t>((const int32_t *)c.keys, c.valid, c.weights, c.n, c.path, c.ws,
                                (int32_t *)c.out_keys, c.out_counts, c.out_capacity, c.state, cs,
                                !contiguous,
                                ((c.path & NVT_PATH_HOT) || (c.path & 0xFF) == NVT_PATH_RANGE ||
                                 (c.path & 0xFF) == NVT_PATH_SORT)
                                    ? c.hot_image
                                    : nullptr,
                                c.range_table);
    else if (c.key_bytes == 8)
      rc = dense_count<int64_t>((const int64_t *)c.keys, c.valid, c.weights, c.n, c.path, c.ws,
                                (int64_t *)c.out_keys, c.out_counts, c.out_capacity, c.state, cs,
                                !contiguous);
    else {
      set_error("nvt_dense_count_many: key_bytes must be 4 or 8 (column %d)", i);
      rc = NVT_EINVAL;
    }
    if (rc) {
      rc_all = rc;  // the columns launched so far keep running on the internal streams: they
      break;        // are joined below all the same, so the caller may free / reuse its buffers
    }
  }
  if (fork)
    for (size_t k = 0; k < wss.size(); ++k) {
      NVT_CHECK_HIP(hipEventRecord(pool->join[k], pool->s[k]));
      NVT_CHECK_HIP(hipStreamWaitEvent(main_s, pool->join[k], 0));
    }
  return rc_all;
}
int nvt_dense_count_i32(const int32_t *keys, const uint8_t *valid, const int64_t *weights,
                        uint64_t n, int path, void *ws, int32_t *out_keys, int64_t *out_counts,
                        uint64_t out_capacity, uint64_t *state, void *stream) {
  return dense_count<int32_t>(keys, valid, weights, n, path, ws, out_keys, out_counts, out_capacity,
                              state, (hipStream_t)stream);
}
int nvt_dense_count_i64(const int64_t *keys, const uint8_t *valid, const int64_t *weights,
                        uint64_t n, int path, void *ws, int64_t *out_keys, int64_t *out_counts,
                        uint64_t out_capacity, uint64_t *state, void *stream) {
  return dense_count<int64_t>(keys, valid, weights, n, path, ws, out_keys, out_counts, out_capacity,
                              state, (hipStream_t)stream);
}

}  // extern "C"
