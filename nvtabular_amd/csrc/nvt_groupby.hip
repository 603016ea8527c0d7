// Multi-key groupby-aggregate tables: JoinGroupby / TargetEncoding fit and
// "combo" Categorify (categorify.py:955-1137 with agg columns), plus the
// row -> group lookup their transforms need (join_groupby.py:198-203,
// target_encoding.py:350-371: a left merge on the key columns followed by a
// re-sort; here a read-only probe that keeps row order).
//
// Slot protocol (keys are 1..3 int64 words + a null mask, too wide for one CAS):
//   state word 0 = empty, 1 = being written, 2 = ready.
//   writer: CAS 0->1, write-through (sc1) stores of the key words, drain
//           (s_waitcnt vmcnt(0)), write-through store of state = 2;
//   reader: L1-bypassing (sc1) load of state, then of the key words.
// Per-CU L1s are never refreshed by other CUs' stores and the per-XCD L2s are
// not coherent, so every protocol word goes through agent-scope atomics;
// accumulators are only ever touched by atomic RMWs, which execute at the
// memory side.  (MI355X_MICROARCH.md, "Workgroup dispatch ... visibility".)
#include <cstdlib>
#include <limits>

#include "nvt_common.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"

// One 16-byte header per slot: a probe of a single-key table (the common JoinGroupby /
// TargetEncoding group) touches ONE 64-byte sector instead of four arrays (state, null mask,
// key, group id: gb_lookup 1.78 -> ms per 20 M rows, profiles/r02_notes.md).
struct nvt_gb_head {
  long long key0;
  unsigned meta;   // bits 0-7 slot state, bits 8-15 null mask of the key tuple
  unsigned index;  // slot -> compact group id (lookup tables), 0xFFFFFFFF = none
};

struct nvt_gb_table {
  int nkeys;
  int nvals;
  int flags;
  uint64_t capacity;
  struct nvt_gb_head *head;    // [cap] {first key component, state | null mask, group id}
  long long *keys;             // [nkeys - 1][cap] the further key components
  unsigned long long *size;    // [cap]
  unsigned long long *count;   // [cap]
  double *sum;                 // [nvals][cap]
  double *sumsq;               // [nvals][cap] or null
  double *vmin;                // [nvals][cap] or null
  double *vmax;                // [nvals][cap] or null
  uint64_t *state;             // [NVT_STATE_WORDS]
  void *ptr_scratch;           // device copy of per-call pointer tables
  void *scratch;               // sort workspace of nvt_gb_update (grown on demand)
  uint64_t scratch_bytes;
  int external;                // arrays live in caller-provided memory (nvt_gb_create_in)
  int external_scratch;        // scratch set by nvt_gb_set_workspace
};

namespace nvt {

constexpr int kMaxKeys = 3;
constexpr int kMaxVals = 8;
constexpr unsigned ST_EMPTY = 0, ST_LOCKED = 1, ST_READY = 2;
constexpr int kGbMaxProbe = 1024;

struct GbView {
  int nkeys, nvals, flags;
  uint64_t mask, cap;
  nvt_gb_head *head;
  long long *keys;  // components 1 .. nkeys-1
  unsigned long long *size, *count;
  double *sum, *sumsq, *vmin, *vmax;
  uint64_t *state;
  unsigned long long *vcount;  // [nvals][cap] non-null values per column (nvt_seg_aggregate only)
};

struct GbRowArgs {
  const int64_t *keys[kMaxKeys];
  const uint8_t *key_valid[kMaxKeys];
  const void *vals[kMaxVals];
  const uint8_t *val_valid[kMaxVals];
  int vdtype[kMaxVals];
  // sort path (gb_segreduce_kernel): a value column in the order of the words -- written by the
  // first aggregate of a pass that gathers it by row (sorted_out), read back streaming by the
  // next aggregate on the same words (sorted_in) instead of a second random gather
  const void *sorted_in[kMaxVals];
  void *sorted_out[kMaxVals];
};

struct GbMergeArgs {
  const int64_t *keys[kMaxKeys];
  const uint8_t *null_mask;
  const int64_t *size, *count;
  const double *sum[kMaxVals], *sumsq[kMaxVals], *vmin[kMaxVals], *vmax[kMaxVals];
};

struct GbOutArgs {
  int64_t *keys[kMaxKeys];
  uint8_t *null_mask;
  int64_t *size, *count;
  double *sum[kMaxVals], *sumsq[kMaxVals], *vmin[kMaxVals], *vmax[kMaxVals];
};

__device__ __forceinline__ uint64_t tuple_hash(const long long (&k)[kMaxKeys], unsigned nm,
                                               int nkeys) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ nm;
  for (int j = 0; j < nkeys; ++j) h = fmix64(h ^ (uint64_t)k[j]) + 0x9E3779B97F4A7C15ull * (j + 1);
  return fmix64(h);
}

// Find (or, when `insert`, create) the slot of a key tuple.  Returns the slot
// index or -1 (absent / overflow).
__device__ __forceinline__ int64_t find_slot(const GbView &t, const long long (&k)[kMaxKeys],
                                             unsigned nm, bool insert, unsigned *n_new,
                                             unsigned *ovf) {
  uint64_t slot = tuple_hash(k, nm, t.nkeys) & t.mask;
  int probe = 0;
  unsigned spins = 0;  // bounded wait on a LOCKED slot: report overflow rather than hang the GPU
  while (probe < kGbMaxProbe) {
    nvt_gb_head *hd = &t.head[slot];
    unsigned st = __hip_atomic_load(&hd->meta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st == ST_EMPTY) {
      if (!insert) return -1;
      unsigned prev = atomicCAS(&hd->meta, ST_EMPTY, ST_LOCKED);
      if (prev == ST_EMPTY) {
        __hip_atomic_store(&hd->key0, k[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int j = 1; j < t.nkeys; ++j)
          __hip_atomic_store(&t.keys[(uint64_t)(j - 1) * t.cap + slot], k[j], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&hd->meta, ST_READY | (nm << 8), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        *n_new += 1;
        return (int64_t)slot;
      }
      st = prev;  // somebody else got it: fall through with its state
    }
    if (st == ST_LOCKED) {  // writer is a few instructions from READY: re-read
      if (++spins < (1u << 22)) continue;
      *ovf = 1;
      return -1;
    }
    // READY: compare (the null mask travels with the state word)
    bool same = (st >> 8) == nm &&
                __hip_atomic_load(&hd->key0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == k[0];
    for (int j = 1; same && j < t.nkeys; ++j)
      same = __hip_atomic_load(&t.keys[(uint64_t)(j - 1) * t.cap + slot], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT) == k[j];
    if (same) return (int64_t)slot;
    slot = (slot + 1) & t.mask;
    ++probe;
  }
  *ovf = 1;
  return -1;
}

__device__ __forceinline__ void atomic_min_f64(double *addr, double v) {
  unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
  unsigned long long old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (v < __longlong_as_double((long long)old)) {
    unsigned long long prev = atomicCAS(a, old, (unsigned long long)__double_as_longlong(v));
    if (prev == old) break;
    old = prev;
  }
}
__device__ __forceinline__ void atomic_max_f64(double *addr, double v) {
  unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
  unsigned long long old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (v > __longlong_as_double((long long)old)) {
    unsigned long long prev = atomicCAS(a, old, (unsigned long long)__double_as_longlong(v));
    if (prev == old) break;
    old = prev;
  }
}

__device__ __forceinline__ bool load_val(const void *p, int dtype, uint64_t i, double *out) {
  switch (dtype) {
    case NVT_F32: {
      float v = reinterpret_cast<const float *>(p)[i];
      *out = (double)v;
      return v == v;
    }
    case NVT_F64: {
      double v = reinterpret_cast<const double *>(p)[i];
      *out = v;
      return v == v;
    }
    case NVT_I32:
      *out = (double)reinterpret_cast<const int32_t *>(p)[i];
      return true;
    case NVT_I64:
      *out = (double)reinterpret_cast<const int64_t *>(p)[i];
      return true;
    case NVT_U8:
      *out = (double)reinterpret_cast<const uint8_t *>(p)[i];
      return true;
  }
  return false;
}

// the value as load_val read it, back in its own dtype (exact: v came from that dtype; NaN stays NaN)
__device__ __forceinline__ void store_val(void *p, int dtype, uint64_t i, double v) {
  switch (dtype) {
    case NVT_F32: reinterpret_cast<float *>(p)[i] = (float)v; break;
    case NVT_F64: reinterpret_cast<double *>(p)[i] = v; break;
    case NVT_I32: reinterpret_cast<int32_t *>(p)[i] = (int32_t)v; break;
    case NVT_I64: reinterpret_cast<int64_t *>(p)[i] = (int64_t)v; break;
    case NVT_U8: reinterpret_cast<uint8_t *>(p)[i] = (uint8_t)v; break;
  }
}

__global__ __launch_bounds__(kBlock) void gb_clear_kernel(GbView t) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  const double inf = std::numeric_limits<double>::infinity();
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < t.cap; i += stride) {
    nvt_gb_head e;
    e.key0 = 0;
    e.meta = ST_EMPTY;
    e.index = 0xFFFFFFFFu;
    t.head[i] = e;
    t.size[i] = 0;
    t.count[i] = 0;
    for (int j = 1; j < t.nkeys; ++j) t.keys[(uint64_t)(j - 1) * t.cap + i] = 0;
    for (int j = 0; j < t.nvals; ++j) {
      t.sum[(uint64_t)j * t.cap + i] = 0.0;
      if (t.sumsq) t.sumsq[(uint64_t)j * t.cap + i] = 0.0;
      if (t.vmin) {
        t.vmin[(uint64_t)j * t.cap + i] = inf;
        t.vmax[(uint64_t)j * t.cap + i] = -inf;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < NVT_STATE_WORDS) t.state[threadIdx.x] = 0;
}

__device__ __forceinline__ unsigned read_tuple(const GbRowArgs &a, int nkeys, uint64_t i,
                                               long long (&k)[kMaxKeys]) {
  unsigned nm = 0;
  for (int j = 0; j < kMaxKeys; ++j) k[j] = 0;
  for (int j = 0; j < nkeys; ++j) {
    if (bit_valid(a.key_valid[j], i))
      k[j] = a.keys[j][i];
    else
      nm |= 1u << j;
  }
  return nm;
}

__global__ __launch_bounds__(kBlock) void gb_update_kernel(GbView t, GbRowArgs a, uint64_t n) {
  unsigned n_new = 0, ovf = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    long long k[kMaxKeys];
    unsigned nm = read_tuple(a, t.nkeys, i, k);
    int64_t slot = find_slot(t, k, nm, true, &n_new, &ovf);
    if (slot < 0) continue;
    atomicAdd(&t.size[slot], 1ull);
    if (!(nm & 1u)) atomicAdd(&t.count[slot], 1ull);
    for (int j = 0; j < t.nvals; ++j) {
      double v;
      bool ok = load_val(a.vals[j], a.vdtype[j], i, &v) && bit_valid(a.val_valid[j], i);
      if (!ok) continue;
      uint64_t o = (uint64_t)j * t.cap + slot;
      atomicAdd(&t.sum[o], v);
      if (t.sumsq) atomicAdd(&t.sumsq[o], v * v);
      if (t.vmin) {
        atomic_min_f64(&t.vmin[o], v);
        atomic_max_f64(&t.vmax[o], v);
      }
    }
  }
  if (n_new) atomicAdd((unsigned long long *)&t.state[NVT_ST_OCCUPIED], (unsigned long long)n_new);
  if (ovf) atomicOr((unsigned long long *)&t.state[NVT_ST_OVERFLOW], 1ull);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd((unsigned long long *)&t.state[NVT_ST_ROWS], (unsigned long long)n);
}

// ---- update without per-row accumulator atomics (JoinGroupby / TargetEncoding fit) -----------
// The kernel above does up to 2 + 4 * nvals device atomics per row on the row's group; with the
// skewed keys these operators see, thousands of rows hit the same few slots and serialise at
// the memory side (20 M rows, 5 M keys, one float target: 5.5 s per fit + transform).  Instead:
//   gb_assign_kernel     row -> slot (find-or-insert: reads only once a key exists), written as
//                        the packed word (slot << 32 | row)
//   sort_words_bits      onesweep LSD radix on the slot bits: every group's rows become ONE run,
//                        in row order (stable)
//   gb_segreduce_kernel  wave-level segmented reduction over the sorted words (values gathered
//                        by row); one add per (wave, run) into the group's accumulators -- a hot
//                        group of a million rows costs 16 k adds instead of a million, and runs of
//                        cold groups touch distinct addresses.
__global__ __launch_bounds__(kBlock) void gb_assign_kernel(GbView t, GbRowArgs a, uint64_t n,
                                                           uint64_t *__restrict__ words) {
  unsigned n_new = 0, ovf = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    long long k[kMaxKeys];
    unsigned nm = read_tuple(a, t.nkeys, i, k);
    int64_t slot = find_slot(t, k, nm, true, &n_new, &ovf);
    // rows that found no slot (overflow: the call fails) sort behind everything else
    words[i] = ((uint64_t)(slot < 0 ? t.cap : (uint64_t)slot) << 32) | (uint64_t)(uint32_t)i;
  }
  if (n_new) atomicAdd((unsigned long long *)&t.state[NVT_ST_OCCUPIED], (unsigned long long)n_new);
  if (ovf) atomicOr((unsigned long long *)&t.state[NVT_ST_OVERFLOW], 1ull);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd((unsigned long long *)&t.state[NVT_ST_ROWS], (unsigned long long)n);
}

struct SegAcc {
  double cnt, sum, sq, mn, mx;
};

// EXCL: every slot's rows are ONE run of the words and its accumulators still hold their initial
// values (nvt_sgb_reduce): a run that begins and ends inside a wave's 64 words belongs to that
// lane alone and is written with plain stores -- only the runs that touch the row's first or
// last word (they may continue in the neighbouring rows) need atomics.  With ~4 rows per group
// that is 14 of 16 runs.  slot_div: the words' slots are g * slot_div + fold and this
// aggregate wants g (words regrouped for another aggregate's folds).
// What this kernel waits for (experiment modes, 20 M rows / 5 M groups, profiles/r05_notes.md):
// the scan itself is ~100 us; gathering a value by row ~300 us (random sectors); and the
// accumulator writes were ~250 us as long as every row of 64 words sent its first and last run to
// the dense arrays with ATOMICS next to the plain stores of the runs in between (an atomic on a
// line with pending partial writes is slow, whatever the key skew).  So a wave now CARRIES the
// run that reaches the end of a row into the next row of its chunk: only the first and the last
// run of a wave's whole chunk (~2400 words) can continue in another wave's chunk and use
// atomics; everything else is a plain store by the lane that ends the run.
//   * run lengths, valid counts and run ownership come from BALLOTS of the run heads (scalar bit
//     arithmetic), only the statistics the table carries are scanned (SQ / MM compile-time):
//     13 crossbar trips per 64 rows for TargetEncoding's reduction where the first version had 96;
//   * U rows per trip: their loads are issued together; the carry is wave-uniform and touches a
//     row only after its scan, so the scan chains of the U rows still interleave;
//   * one walk over the chunk per value column (the carry of several columns would have to be
//     indexed dynamically, i.e. live in scratch); sizes / counts ride on the first walk.
__device__ __forceinline__ double bcast_lane_f64(double x, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned long long bcast_lane_u64(unsigned long long x, int src) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}

template <bool EXCL, bool SQ, bool MM>
__global__ __launch_bounds__(kBlock) void gb_segreduce_kernel(GbView t, GbRowArgs a, uint64_t n,
                                                              const uint64_t *__restrict__ words,
                                                              uint32_t slot_div) {
  const double inf = std::numeric_limits<double>::infinity();
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  const unsigned lane = lane_id();
  const uint64_t nwaves = (uint64_t)gridDim.x * (kBlock / kWave);
  const uint64_t wave0 = (uint64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  // every wave walks ONE contiguous chunk of the sorted words in rows of 64.  The slots of a chunk
  // are a contiguous range when the words are ordered by slot: the accumulators a wave writes
  // are its own region, far from the other waves' (a grid-stride walk put the whole GPU's stores
  // into the same ~1 MB window of the dense arrays at any time: 596 us instead of 443)
  const uint64_t per_wave = ((n + nwaves - 1) / nwaves + kWave - 1) / kWave * kWave;
  const uint64_t chunk_begin = wave0 * per_wave;
  const uint64_t chunk_end = (wave0 + 1) * per_wave < n ? (wave0 + 1) * per_wave : n;
  const unsigned long long below = (2ull << lane) - 1ull;  // lanes 0 .. lane
  constexpr int U = 4;
  const int npass = t.nvals > 0 ? t.nvals : 1;
  for (int j = 0; j < npass; ++j) {
    const bool with_val = t.nvals > 0, with_size = j == 0;
    // the carried run (wave-uniform): its slot, whether it began at the chunk's first word (then
    // it may continue a run of the previous wave's chunk), and its partial statistics
    uint32_t c_slot = kNone;
    bool c_first = false;
    unsigned long long c_sz = 0, c_ct = 0, c_any = 0;
    double c_sum = 0.0, c_sq = 0.0, c_mn = inf, c_mx = -inf;
    // writes one finished run (called by ONE lane): own = nobody else adds to this slot
    auto emit = [&](uint32_t sl, bool own, unsigned long long sz, unsigned long long ct,
                    unsigned long long any, double sum, double sq, double mn, double mx) {
      if (with_size) {
        if (own) {
          if (t.size) t.size[sl] = sz;
          if (t.count && ct > 0) t.count[sl] = ct;
        } else {
          if (t.size) atomicAdd(&t.size[sl], sz);
          if (t.count && ct > 0) atomicAdd(&t.count[sl], ct);
        }
      }
      if (with_val && any > 0) {
        const uint64_t o = (uint64_t)j * t.cap + sl;
        if (own) {
          t.sum[o] = sum;
          if (t.vcount) t.vcount[o] = any;
          if constexpr (SQ) t.sumsq[o] = sq;
          if constexpr (MM) {
            t.vmin[o] = mn;
            t.vmax[o] = mx;
          }
        } else {
          atomicAdd(&t.sum[o], sum);
          if (t.vcount) atomicAdd(&t.vcount[o], any);
          if constexpr (SQ) atomicAdd(&t.sumsq[o], sq);
          if constexpr (MM) {
            atomic_min_f64(&t.vmin[o], mn);
            atomic_max_f64(&t.vmax[o], mx);
          }
        }
      }
    };
    for (uint64_t base = chunk_begin; base < chunk_end; base += (uint64_t)kWave * U) {
      uint64_t idx[U], w[U];
      bool act[U];
      double v[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        idx[u] = base + (uint64_t)u * kWave + lane;
        act[u] = idx[u] < chunk_end;
        w[u] = act[u] ? words[idx[u]] : ~0ull;
      }
      uint32_t slot[U], row[U];
      bool live[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t raw = (uint32_t)(w[u] >> 32);
        slot[u] = (slot_div > 1 && raw != kNone) ? raw / slot_div : raw;
        if (!(act[u] && raw != kNone && slot[u] < t.cap)) slot[u] = kNone;  // rows nobody counts
        live[u] = slot[u] != kNone;
        row[u] = (uint32_t)w[u];
        v[u] = 0;
        ok[u] = false;
        if (with_val) {
          if (a.sorted_in[j]) {  // (no validity bitmap on this path: the caller checked)
            ok[u] = act[u] && load_val(a.sorted_in[j], a.vdtype[j], idx[u], &v[u]) && live[u];
          } else {
            ok[u] = live[u] && load_val(a.vals[j], a.vdtype[j], row[u], &v[u]) &&
                    bit_valid(a.val_valid[j], row[u]);
          }
        }
      }
      if (with_val && a.sorted_out[j] && !a.sorted_in[j]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (act[u]) store_val(a.sorted_out[j], a.vdtype[j], idx[u], live[u] ? v[u] : 0.0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (base + (uint64_t)u * kWave >= chunk_end) break;  // (uniform)
        const uint32_t prev = __shfl_up(slot[u], 1, 64);
        // run heads of this row (lane 0 starts a run as far as the row is concerned)
        const unsigned long long heads = __ballot(lane == 0 || prev != slot[u]);
        const unsigned start = 63u - (unsigned)__clzll(heads & below);   // first lane of this lane's run
        const unsigned long long run = below & ~((1ull << start) - 1ull);  // lanes start .. lane
        const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
        unsigned long long sz = (unsigned long long)(lane - start + 1), ct = sz;
        if (with_size && a.key_valid[0])
          ct = (unsigned long long)__popcll(__ballot(live[u] && bit_valid(a.key_valid[0], row[u])) & run);
        double sum = ok[u] ? v[u] : 0.0, sq = ok[u] ? v[u] * v[u] : 0.0;
        double mn = ok[u] ? v[u] : inf, mx = ok[u] ? v[u] : -inf;
        unsigned long long any = (unsigned long long)__popcll(__ballot(ok[u]) & run);
        if (with_val) {
          // inclusive segmented scan in row order: the lane `off` below belongs to this run when
          // it is not in front of the run's first lane
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const bool take = lane >= start + (unsigned)off;
            const double osum = __shfl_up(sum, off, 64);
            if (take) sum = osum + sum;  // earlier rows first
            if constexpr (SQ) {
              const double osq = __shfl_up(sq, off, 64);
              if (take) sq = osq + sq;
            }
            if constexpr (MM) {
              const double omn = __shfl_up(mn, off, 64), omx = __shfl_up(mx, off, 64);
              if (take) {
                mn = omn < mn ? omn : mn;
                mx = omx > mx ? omx : mx;
              }
            }
          }
        }
        // the carried run continues in this row's first run, or is finished
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot[u]);
        const bool cont = c_slot != kNone && c_slot == s0;
        if (c_slot != kNone && !cont && lane == 0)
          emit(c_slot, EXCL && !c_first, c_sz, c_ct, c_any, c_sum, c_sq, c_mn, c_mx);
        // does this row's first run begin at the chunk's first word?
        const bool row_first = base + (uint64_t)u * kWave == chunk_begin;
        const bool first_run_first = cont ? c_first : row_first;
        if (cont && start == 0) {
          sz += c_sz;
          ct += c_ct;
          any += c_any;
          sum = c_sum + sum;
          if constexpr (SQ) sq = c_sq + sq;
          if constexpr (MM) {
            mn = c_mn < mn ? c_mn : mn;
            mx = c_mx > mx ? c_mx : mx;
          }
        }
        // runs that end inside the row: written by their last lane
        if (live[u] && tail && lane < 63)
          emit(slot[u], EXCL && (start > 0 || !first_run_first), sz, ct, any, sum, sq, mn, mx);
        // the run that reaches lane 63 is carried on
        const unsigned start63 = (unsigned)__builtin_amdgcn_readlane((int)start, 63);
        c_slot = (uint32_t)__builtin_amdgcn_readlane((int)slot[u], 63);
        c_first = start63 == 0 ? first_run_first : false;
        c_sz = bcast_lane_u64(sz, 63);
        c_ct = bcast_lane_u64(ct, 63);
        c_any = bcast_lane_u64(any, 63);
        c_sum = bcast_lane_f64(sum, 63);
        if constexpr (SQ) c_sq = bcast_lane_f64(sq, 63);
        if constexpr (MM) {
          c_mn = bcast_lane_f64(mn, 63);
          c_mx = bcast_lane_f64(mx, 63);
        }
      }
    }
    // the chunk's last run may continue in the next wave's chunk
    if (c_slot != kNone && lane == 0) emit(c_slot, false, c_sz, c_ct, c_any, c_sum, c_sq, c_mn, c_mx);
  }
}

template <bool EXCL>
static void launch_segreduce(const GbView &t, const GbRowArgs &a, uint64_t n, const uint64_t *words,
                             uint32_t slot_div, hipStream_t s) {
  const unsigned grid = stream_grid(n, kBlock * 4, 8);
  const bool sq = t.nvals > 0 && t.sumsq != nullptr, mm = t.nvals > 0 && t.vmin != nullptr;
  if (sq && mm) gb_segreduce_kernel<EXCL, true, true><<<grid, kBlock, 0, s>>>(t, a, n, words, slot_div);
  else if (sq) gb_segreduce_kernel<EXCL, true, false><<<grid, kBlock, 0, s>>>(t, a, n, words, slot_div);
  else if (mm) gb_segreduce_kernel<EXCL, false, true><<<grid, kBlock, 0, s>>>(t, a, n, words, slot_div);
  else gb_segreduce_kernel<EXCL, false, false><<<grid, kBlock, 0, s>>>(t, a, n, words, slot_div);
}

// ---- Groupby operator (groupby.py:236-263): row order + per-group aggregates ----------------
// rows are ordered by (group id, sort columns) with the stable radix sort of nvt_sort.hip over
// packed (key32 << 32 | row) words -- a 64-bit sort key takes two passes of 32 bits -- and every
// conventional aggregate comes from ONE wave-level segmented reduction over the ordered rows
// (gb_segreduce_kernel above, writing straight into the output arrays).
template <typename T>
__device__ __forceinline__ uint64_t sortable_bits(T v);
template <>
__device__ __forceinline__ uint64_t sortable_bits<int64_t>(int64_t v) {
  return (uint64_t)v ^ 0x8000000000000000ull;
}
template <>
__device__ __forceinline__ uint64_t sortable_bits<int32_t>(int32_t v) {
  return sortable_bits<int64_t>((int64_t)v);
}
template <>
__device__ __forceinline__ uint64_t sortable_bits<uint8_t>(uint8_t v) {
  return sortable_bits<int64_t>((int64_t)v);
}
template <>
__device__ __forceinline__ uint64_t sortable_bits<double>(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
template <>
__device__ __forceinline__ uint64_t sortable_bits<float>(float v) {
  return sortable_bits<double>((double)v);
}

// out[i] = order-preserving 64-bit image of x[i]; nulls / NaN sort LAST for either direction
// (pandas sort_values na_position="last"); descending = complemented image
template <typename T>
__global__ __launch_bounds__(kBlock) void sort_key_kernel(const T *__restrict__ x,
                                                          const uint8_t *__restrict__ valid,
                                                          uint64_t n, int ascending,
                                                          uint64_t *__restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const T v = x[i];
    const bool ok = bit_valid(valid, i) && !is_nan(v);
    uint64_t b = sortable_bits<T>(v);
    if (!ascending) b = ~b;
    out[i] = ok ? (b == ~0ull ? b - 1 : b) : ~0ull;
  }
}

// words[i] = (hi32(i) << 32) | row(i) with row(i) = perm ? low 32 bits of perm[i] : i and
// hi32 = 32-bit chunk `chunk` (0 = low, 1 = high) of src64[row] -- or, for group ids, the id
// itself with -1 (null key) mapped to `null_hi` so that those rows sort last
__global__ __launch_bounds__(kBlock) void pack_words_kernel(const uint64_t *__restrict__ src64,
                                                            const int64_t *__restrict__ gid,
                                                            const uint64_t *__restrict__ perm,
                                                            uint64_t n, int chunk, uint32_t null_hi,
                                                            uint64_t *__restrict__ words) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const uint32_t row = perm ? (uint32_t)perm[i] : (uint32_t)i;
    uint32_t hi;
    if (gid) {
      const int64_t g = gid[row];
      hi = g < 0 ? null_hi : (uint32_t)g;
    } else {
      hi = (uint32_t)(src64[row] >> (32 * chunk));
    }
    words[i] = ((uint64_t)hi << 32) | row;
  }
}

__global__ __launch_bounds__(kBlock) void gb_merge_kernel(GbView t, GbMergeArgs a, uint64_t n) {
  unsigned n_new = 0, ovf = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    long long k[kMaxKeys] = {0, 0, 0};
    unsigned nm = a.null_mask ? a.null_mask[i] : 0;
    for (int j = 0; j < t.nkeys; ++j) k[j] = ((nm >> j) & 1) ? 0 : a.keys[j][i];
    int64_t slot = find_slot(t, k, nm, true, &n_new, &ovf);
    if (slot < 0) continue;
    if (a.size) atomicAdd(&t.size[slot], (unsigned long long)a.size[i]);
    if (a.count) atomicAdd(&t.count[slot], (unsigned long long)a.count[i]);
    for (int j = 0; j < t.nvals; ++j) {
      uint64_t o = (uint64_t)j * t.cap + slot;
      if (a.sum[j]) atomicAdd(&t.sum[o], a.sum[j][i]);
      if (t.sumsq && a.sumsq[j]) atomicAdd(&t.sumsq[o], a.sumsq[j][i]);
      if (t.vmin && a.vmin[j]) {
        double lo = a.vmin[j][i], hi = a.vmax[j][i];
        if (lo == lo) atomic_min_f64(&t.vmin[o], lo);
        if (hi == hi) atomic_max_f64(&t.vmax[o], hi);
      }
    }
  }
  if (n_new) atomicAdd((unsigned long long *)&t.state[NVT_ST_OCCUPIED], (unsigned long long)n_new);
  if (ovf) atomicOr((unsigned long long *)&t.state[NVT_ST_OVERFLOW], 1ull);
}

// table -> dense group arrays, arbitrary order.  Each workgroup takes tiles of kBlock *
// kGbCompactItems consecutive slots, ranks the READY ones with a wave scan + LDS and reserves
// its output range with ONE atomic per tile (an atomic per wave on the single cursor
// serialises at the memory side: 4.5 ms for a 16 M-slot table, 39 % of the cfg4 step).
constexpr int kGbCompactItems = 8;
__global__ __launch_bounds__(kBlock) void gb_compact_kernel(GbView t, GbOutArgs o,
                                                            uint64_t *out_n) {
  constexpr uint64_t TILE = (uint64_t)kBlock * kGbCompactItems;
  __shared__ unsigned wsum[kBlock / kWave];
  __shared__ unsigned long long tile_base;
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  const unsigned lane = lane_id(), w = threadIdx.x / kWave;
  const uint64_t ntiles = (t.cap + TILE - 1) / TILE;
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // item q of the tile's round r is slot tile * TILE + r * kBlock + threadIdx.x (coalesced)
    unsigned occ = 0, mine = 0;
#pragma unroll
    for (int r = 0; r < kGbCompactItems; ++r) {
      const uint64_t i = tile * TILE + (uint64_t)r * kBlock + threadIdx.x;
      if (i < t.cap && (t.head[i].meta & 0xFFu) == ST_READY) {
        occ |= 1u << r;
        ++mine;
      }
    }
    unsigned inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o2 = __shfl_up(inc, off, 64);
      if (lane >= (unsigned)off) inc += o2;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned wbase = 0, total = 0;
    for (unsigned q = 0; q < kBlock / kWave; ++q) {
      if (q < w) wbase += wsum[q];
      total += wsum[q];
    }
    if (threadIdx.x == 0)
      tile_base = total ? atomicAdd((unsigned long long *)out_n, (unsigned long long)total) : 0;
    __syncthreads();
    uint64_t g = tile_base + wbase + inc - mine;
#pragma unroll
    for (int r = 0; r < kGbCompactItems; ++r) {
      if (!((occ >> r) & 1)) continue;
      const uint64_t i = tile * TILE + (uint64_t)r * kBlock + threadIdx.x;
      const nvt_gb_head hd = t.head[i];
      // the table now also maps its keys to the compact group ids (nvt_gb_lookup works on it
      // without a second table built by nvt_gb_index_build)
      t.head[i].index = (unsigned)g;
      if (o.keys[0]) o.keys[0][g] = hd.key0;
      for (int j = 1; j < t.nkeys; ++j)
        if (o.keys[j]) o.keys[j][g] = t.keys[(uint64_t)(j - 1) * t.cap + i];
      if (o.null_mask) o.null_mask[g] = (uint8_t)(hd.meta >> 8);
      if (o.size) o.size[g] = (int64_t)t.size[i];
      if (o.count) o.count[g] = (int64_t)t.count[i];
      for (int j = 0; j < t.nvals; ++j) {
        uint64_t s = (uint64_t)j * t.cap + i;
        if (o.sum[j]) o.sum[j][g] = t.sum[s];
        if (o.sumsq[j]) o.sumsq[j][g] = t.sumsq ? t.sumsq[s] : qnan;
        if (o.vmin[j]) {
          double v = t.vmin ? t.vmin[s] : qnan;
          o.vmin[j][g] = (v == std::numeric_limits<double>::infinity()) ? qnan : v;
        }
        if (o.vmax[j]) {
          double v = t.vmax ? t.vmax[s] : qnan;
          o.vmax[j][g] = (v == -std::numeric_limits<double>::infinity()) ? qnan : v;
        }
      }
      ++g;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void gb_index_build_kernel(GbView t, GbMergeArgs a,
                                                                uint64_t n) {
  unsigned n_new = 0, ovf = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    long long k[kMaxKeys] = {0, 0, 0};
    unsigned nm = a.null_mask ? a.null_mask[i] : 0;
    for (int j = 0; j < t.nkeys; ++j) k[j] = ((nm >> j) & 1) ? 0 : a.keys[j][i];
    int64_t slot = find_slot(t, k, nm, true, &n_new, &ovf);
    if (slot >= 0) t.head[slot].index = (unsigned)i;
  }
  if (n_new) atomicAdd((unsigned long long *)&t.state[NVT_ST_OCCUPIED], (unsigned long long)n_new);
  if (ovf) atomicOr((unsigned long long *)&t.state[NVT_ST_OVERFLOW], 1ull);
}

__global__ __launch_bounds__(kBlock) void gb_lookup_kernel(GbView t, GbRowArgs a, uint64_t n,
                                                           int64_t *__restrict__ out) {
  unsigned n_new = 0, ovf = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    long long k[kMaxKeys];
    unsigned nm = read_tuple(a, t.nkeys, i, k);
    int64_t slot = find_slot(t, k, nm, false, &n_new, &ovf);
    unsigned g = 0xFFFFFFFFu;
    if (slot >= 0) g = t.head[slot].index;
    out[i] = g == 0xFFFFFFFFu ? -1 : (int64_t)g;
  }
}

// ---- one int32 key column: groupby by SORTING (JoinGroupby / TargetEncoding fit) --------------
// The hash update above costs a random 16-byte probe per row (gb_assign 1.28 ms for 20 M rows),
// a sort of (slot, row) words, and a compaction of the sparse table afterwards.  With ONE key
// column of at most 32 bits none of the three is needed: the rows are sorted by the key itself,
//   sort_packed_keys     words = (key image << 32 | fold << rb | row), packed from the column by
//                        the histogram and the first scatter pass (never written out unsorted);
//                        onesweep LSD radix on bits [rb, 64): every key's rows are one run, the
//                        folds of a key (TargetEncoding's kfold) consecutive sub-runs
//   sgb_rle_kernel       run heads -> group index g (decoupled look-back over the tiles): the
//                        groups come out DENSE and ordered by key, the words are rewritten as
//                        ((g * kfold + fold) << 32 | row)
//   gb_segreduce_kernel  the same wave-level segmented reduction as above, into dense arrays
//                        indexed g * kfold + fold (monotone in the sorted order: atomics of
//                        neighbouring waves fall into neighbouring sectors)
//   sgb_fold_total_kernel  per-key totals as the sum over the key's folds (TargetEncoding needs
//                        both tables, categorify.py:1344-1540 run twice in the reference)
// and the key -> group index for the transform side is a flat range table laid out from the
// sorted keys (flat_build_kernel, nvt_sort.hip) instead of a second hash table.
#ifndef NVT_SGB_ROWS
#define NVT_SGB_ROWS 16
#endif
constexpr int kSgbRows = NVT_SGB_ROWS;
constexpr int kSgbTile = kBlock * kSgbRows;
constexpr unsigned long long kSgbAgg = 1ull << 62, kSgbPrefix = 2ull << 62,
                             kSgbMask = (1ull << 62) - 1ull;

// smallest / largest key of a column (int64 columns take the sort path when their keys span
// less than 2^32): out2 = {min, max}, initialised by the caller to {INT64_MAX, INT64_MIN}
template <typename K>
__global__ __launch_bounds__(kBlock) void key_minmax_kernel(const K *__restrict__ keys, uint64_t n,
                                                            long long *out2) {
  long long mn = INT64_MAX, mx = INT64_MIN;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const long long k = (long long)keys[i];
    mn = k < mn ? k : mn;
    mx = k > mx ? k : mx;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const long long a = __shfl_down(mn, off, 64), b = __shfl_down(mx, off, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if (lane_id() == 0) {
    atomicMin(&out2[0], mn);
    atomicMax(&out2[1], mx);
  }
}

// MERGE (owner-side merge of (key, count) rows, nvt_count_merge_sorted): the words are
// (column << (rb + 32) | key image << rb | row), a group is a (column, key) pair; out_keys
// receives the group's COLUMN, out_keys32 its key, the rewritten words carry slot = group.
template <bool MERGE>
__global__ __launch_bounds__(kBlock) void sgb_rle_kernel(
    const uint64_t *__restrict__ in, uint64_t n, int rb, unsigned kfold, uint64_t cap,
    unsigned long long *status, unsigned *ticket, int64_t *__restrict__ out_keys,
    int32_t *__restrict__ out_keys32, uint64_t *__restrict__ out_words, uint64_t *state,
    int64_t bias) {
  constexpr int NW = kBlock / kWave;
  __shared__ unsigned wtot[NW];
  __shared__ unsigned long long s_base;
  __shared__ unsigned s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const unsigned tile = s_tile, w = threadIdx.x / kWave, l = lane_id();
  const uint64_t wave0 = (uint64_t)tile * kSgbTile + (uint64_t)w * (kSgbRows * kWave);
  uint64_t W[kSgbRows];
  unsigned long long hb[kSgbRows];  // ballot of the run heads of every 64-word row
  // group field (key image; MERGE: column + key image) in front of this wave's run (one
  // address: a broadcast load)
  const int fsh = MERGE ? rb : 32;
  uint64_t prev_hi = (wave0 > 0 && wave0 < n) ? in[wave0 - 1] >> fsh : 0ull;
  unsigned wheads = 0;
#pragma unroll
  for (int r = 0; r < kSgbRows; ++r) {
    const uint64_t i = wave0 + (uint64_t)r * kWave + l;
    const bool act = i < n;
    W[r] = act ? in[i] : ~0ull;
    const uint64_t hi = W[r] >> fsh;
    uint64_t up = __shfl_up(hi, 1, 64);
    if (l == 0) up = prev_hi;
    const bool head = act && (i == 0 || hi != up);
    hb[r] = __ballot(head);
    prev_hi = __shfl(hi, 63, 64);
    wheads += (unsigned)__popcll(hb[r]);
  }
  if (l == 0) wtot[w] = wheads;
  __syncthreads();
  unsigned wbase = 0, ttot = 0;
  for (int q = 0; q < NW; ++q) {
    if (q < (int)w) wbase += wtot[q];
    ttot += wtot[q];
  }
  if (threadIdx.x == 0) {
    unsigned long long *my = status + tile;
    __hip_atomic_store(my, (tile == 0 ? kSgbPrefix : kSgbAgg) | (unsigned long long)ttot,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long carry = 0;
    if (tile > 0) {
      unsigned tb = tile - 1;
      while (true) {
        const unsigned long long v =
            __hip_atomic_load(status + tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned f = (unsigned)(v >> 62);
        if (f == 0) {
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        carry += v & kSgbMask;
        if (f == 2) break;
        --tb;
      }
      __hip_atomic_store(my, kSgbPrefix | (carry + ttot), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    s_base = carry;
    if ((uint64_t)(tile + 1) * kSgbTile >= n) {  // the tile that holds the last row
      const unsigned long long g = carry + ttot;
      state[NVT_ST_OCCUPIED] = g;
      state[NVT_ST_ROWS] = n;
      if (g > cap) state[NVT_ST_NEED] = g;
    }
  }
  __syncthreads();
  uint64_t g0 = s_base + wbase;  // run heads in front of this wave's rows
  const uint64_t rowmask = rb >= 32 ? 0xFFFFFFFFull : ((1ull << rb) - 1ull);
#pragma unroll
  for (int r = 0; r < kSgbRows; ++r) {
    const uint64_t i = wave0 + (uint64_t)r * kWave + l;
    const unsigned incl = (unsigned)__popcll(hb[r] & ((2ull << l) - 1ull));
    if (i < n) {
      const uint64_t g = g0 + incl - 1;  // >= 0: row 0 is a head
      const bool fits = g < cap;
      if constexpr (MERGE) {
        if (((hb[r] >> l) & 1ull) && fits) {
          out_keys[g] = (int64_t)(W[r] >> (rb + 32));                             // column
          out_keys32[g] = (int32_t)((uint32_t)(W[r] >> rb) ^ 0x80000000u);        // key
        }
        out_words[i] = ((fits ? g : 0xFFFFFFFFull) << 32) | (W[r] & rowmask);
      } else {
        const uint32_t hi = (uint32_t)(W[r] >> 32);
        if (((hb[r] >> l) & 1ull) && fits) {
          out_keys[g] = (int64_t)((uint64_t)hi + (uint64_t)bias);  // the key itself
          out_keys32[g] = (int32_t)(hi ^ 0x80000000u);              // key - bias - 2^31: the flat index's key
        }
        const uint64_t low = W[r] & 0xFFFFFFFFull;
        const uint64_t fold = (rb >= 32 || kfold == 1) ? 0ull : (low >> rb);
        const uint64_t slot = fits ? g * kfold + fold : 0xFFFFFFFFull;
        out_words[i] = (slot << 32) | (low & rowmask);
      }
    }
    g0 += (unsigned)__popcll(hb[r]);
  }
}

__global__ __launch_bounds__(kBlock) void sgb_fill_kernel(double *__restrict__ p, uint64_t n,
                                                          double v) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) p[i] = v;
}

// per-key totals over the folds (+ the one-probe records of the transform).  A workgroup takes
// kBlock consecutive groups: their kfold-entry runs of fsize / fsum are contiguous and are read
// coalesced, the records (2 * (kfold + 1) doubles per group) are assembled in LDS and leave as
// one contiguous block (a thread per group wrote 12 scattered 8-byte words: 462 us for 5 M groups).
constexpr int kSgbMaxFoldLds = 16;
__global__ __launch_bounds__(kBlock) void sgb_fold_total_kernel(
    const uint64_t *__restrict__ state, unsigned kfold, uint64_t cap, int nvals,
    const unsigned long long *__restrict__ fsize, const double *__restrict__ fsum,
    unsigned long long *__restrict__ tsize, double *__restrict__ tsum,
    double *__restrict__ records) {
  // (sized by the launch for THIS kfold: the 16-fold maximum as a static array held 70 KB, two
  // workgroups per CU; 25 KB for five folds lets six of them hide each other's loads)
  extern __shared__ double lrec[];  // [kBlock][2 * (kfold + 1)]
  uint64_t ng = state[NVT_ST_OCCUPIED];
  ng = ng < cap ? ng : cap;
  const unsigned rs = 2 * (kfold + 1);
  const uint64_t nblocks = (ng + kBlock - 1) / kBlock;
  for (uint64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const uint64_t g0 = b * kBlock;
    const unsigned cnt = (unsigned)((ng - g0) < (uint64_t)kBlock ? (ng - g0) : (uint64_t)kBlock);
    for (int j = 0; j < (nvals > 0 ? nvals : 1); ++j) {
      // stage {sum_f, count_f} of the block's groups: element e = (group, fold) in memory order
      for (unsigned e = threadIdx.x; e < cnt * kfold; e += kBlock) {
        const unsigned g = e / kfold, f = e - g * kfold;
        lrec[g * rs + 3 + 2 * f] = (double)fsize[g0 * kfold + e];
        if (nvals) lrec[g * rs + 2 + 2 * f] = fsum[(uint64_t)j * cap * kfold + g0 * kfold + e];
      }
      __syncthreads();
      if (threadIdx.x < cnt) {
        const unsigned g = threadIdx.x;
        double d = 0.0, c = 0.0;
        for (unsigned f = 0; f < kfold; ++f) {
          if (nvals) d += lrec[g * rs + 2 + 2 * f];
          c += lrec[g * rs + 3 + 2 * f];
        }
        lrec[g * rs] = d;
        lrec[g * rs + 1] = c;
        if (j == 0) tsize[g0 + g] = (unsigned long long)c;  // counts < 2^53: exact
        if (nvals) tsum[(uint64_t)j * cap + g0 + g] = d;
      }
      __syncthreads();
      if (records && nvals) {
        double *dst = records + ((uint64_t)j * cap + g0) * rs;
        for (unsigned e = threadIdx.x; e < cnt * rs; e += kBlock) dst[e] = lrec[e];
      }
      __syncthreads();
    }
  }
}

// owner-side merge of the (key, count) rows a rank receives in the multi-GPU exchange
// (categorify.py:1054-1070 _mid_level_groupby on the owner): rows = (count << 32 | key) words in
// `nseg` segments (source-major, column-minor: segment s belongs to column s % ncol).  The rows
// are tagged with their column and sorted by (column, key): equal keys of a column become one
// run whose counts are summed by the segmented reduction -- the merged lists come out ordered
// by key (what the vocabulary ordering wants) without a hash table and without a second sort.
constexpr int kMergeRowBits = 26;  // rows per call < 2^26; 32 key bits; 6 column bits
__global__ __launch_bounds__(kBlock) void merge_pack_kernel(const int64_t *__restrict__ rows,
                                                            uint64_t n,
                                                            const uint64_t *__restrict__ seg_off,
                                                            int nseg, int ncol,
                                                            uint64_t *__restrict__ words,
                                                            int32_t *__restrict__ cnt32) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int lo = 0, hi = nseg;  // the segment with seg_off[s] <= i < seg_off[s + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (seg_off[mid] <= i) lo = mid; else hi = mid;
    }
    const uint64_t col = (uint64_t)(lo % ncol);
    const uint64_t w = (uint64_t)rows[i];
    const uint64_t img = (uint32_t)w ^ 0x80000000u;
    words[i] = (col << (kMergeRowBits + 32)) | (img << kMergeRowBits) | i;
    cnt32[i] = (int32_t)(w >> 32);
  }
}

inline GbView view_of(nvt_gb_table *t) {
  GbView v;
  v.nkeys = t->nkeys;
  v.nvals = t->nvals;
  v.flags = t->flags;
  v.cap = t->capacity;
  v.mask = t->capacity - 1;
  v.head = t->head;
  v.keys = t->keys;
  v.size = t->size;
  v.count = t->count;
  v.sum = t->sum;
  v.sumsq = t->sumsq;
  v.vmin = t->vmin;
  v.vmax = t->vmax;
  v.state = t->state;
  v.vcount = nullptr;
  return v;
}

}  // namespace nvt

using namespace nvt;

extern "C" {

void nvt_gb_destroy(nvt_gb_table *t) {
  if (!t) return;
  if (!t->external) {
    void *ptrs[] = {t->head, t->keys, t->size, t->count, t->sum,
                    t->sumsq, t->vmin, t->vmax, t->state};
    for (void *p : ptrs)
      if (p) (void)hipFree(p);
  }
  if (t->scratch && !t->external_scratch) (void)hipFree(t->scratch);
  delete t;
}

static uint64_t gb_align(uint64_t x) { return (x + 255) & ~255ull; }

int nvt_gb_table_bytes(int nkeys, int nvals, int flags, uint64_t capacity, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes, "null out");
  NVT_CHECK_ARG(nkeys >= 1 && nkeys <= kMaxKeys && nvals >= 0 && nvals <= kMaxVals, "bad shape");
  uint64_t b = gb_align(capacity * 16) + gb_align(capacity * 8 * (nkeys - 1)) +
               gb_align(capacity * 8) * 2 + gb_align(capacity * 8 * nvals) +
               gb_align(NVT_STATE_WORDS * 8);
  if (nvals && (flags & NVT_GB_SUMSQ)) b += gb_align(capacity * 8 * nvals);
  if (nvals && (flags & NVT_GB_MINMAX)) b += 2 * gb_align(capacity * 8 * nvals);
  *bytes = b + 256;
  return NVT_OK;
}

// Table whose arrays live in caller-provided device memory (nvt_gb_table_bytes() bytes, 256-byte
// aligned): no hipMalloc / hipFree -- both synchronise the device -- on the fit path.
int nvt_gb_create_in(int nkeys, int nvals, int flags, uint64_t capacity, void *memory,
                     uint64_t bytes, nvt_gb_table **out) {
  NVT_CHECK_ARG(out && memory, "null out/memory");
  NVT_CHECK_ARG(nkeys >= 1 && nkeys <= kMaxKeys, "nkeys must be 1..3");
  NVT_CHECK_ARG(nvals >= 0 && nvals <= kMaxVals, "nvals must be 0..8");
  NVT_CHECK_ARG(capacity >= 64 && (capacity & (capacity - 1)) == 0, "capacity must be 2^k >= 64");
  uint64_t need = 0;
  nvt_gb_table_bytes(nkeys, nvals, flags, capacity, &need);
  NVT_CHECK_ARG(bytes >= need, "memory block too small (nvt_gb_table_bytes)");
  nvt_gb_table *t = new nvt_gb_table();
  memset(t, 0, sizeof(*t));
  t->nkeys = nkeys;
  t->nvals = nvals;
  t->flags = flags;
  t->capacity = capacity;
  t->external = 1;
  char *p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(memory) + 255) & ~(uintptr_t)255);
  auto take = [&](uint64_t b) {
    char *r = p;
    p += gb_align(b);
    return (void *)r;
  };
  t->head = (nvt_gb_head *)take(capacity * 16);
  if (nkeys > 1) t->keys = (long long *)take(capacity * 8 * (nkeys - 1));
  t->size = (unsigned long long *)take(capacity * 8);
  t->count = (unsigned long long *)take(capacity * 8);
  if (nvals) t->sum = (double *)take(capacity * 8 * nvals);
  t->state = (uint64_t *)take(NVT_STATE_WORDS * 8);
  if (nvals && (flags & NVT_GB_SUMSQ)) t->sumsq = (double *)take(capacity * 8 * nvals);
  if (nvals && (flags & NVT_GB_MINMAX)) {
    t->vmin = (double *)take(capacity * 8 * nvals);
    t->vmax = (double *)take(capacity * 8 * nvals);
  }
  *out = t;
  return NVT_OK;
}

// device address of the table's uint64[NVT_STATE_WORDS] state block (for nvt_mailbox_post)
uint64_t *nvt_gb_state_ptr(nvt_gb_table *t) { return t ? t->state : nullptr; }

int nvt_gb_update_ws_bytes(uint64_t n, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes, "null out");
  *bytes = ((n * 8 + 255) & ~255ull) + sort_words_tmp_bytes(n) + 512;
  return NVT_OK;
}
// caller-owned scratch for nvt_gb_update (nvt_gb_update_ws_bytes(n)); NULL returns to the
// table-owned, grown-on-demand buffer
int nvt_gb_set_workspace(nvt_gb_table *t, void *ws, uint64_t bytes) {
  NVT_CHECK_ARG(t, "null table");
  if (t->scratch && !t->external_scratch) (void)hipFree(t->scratch);
  t->scratch = ws;
  t->scratch_bytes = ws ? bytes : 0;
  t->external_scratch = ws ? 1 : 0;
  return NVT_OK;
}

int nvt_gb_create(int nkeys, int nvals, int flags, uint64_t capacity, nvt_gb_table **out) {
  NVT_CHECK_ARG(out, "null out");
  NVT_CHECK_ARG(nkeys >= 1 && nkeys <= kMaxKeys, "nkeys must be 1..3");
  NVT_CHECK_ARG(nvals >= 0 && nvals <= kMaxVals, "nvals must be 0..8");
  NVT_CHECK_ARG(capacity >= 64 && (capacity & (capacity - 1)) == 0, "capacity must be 2^k >= 64");
  nvt_gb_table *t = new nvt_gb_table();
  memset(t, 0, sizeof(*t));
  t->nkeys = nkeys;
  t->nvals = nvals;
  t->flags = flags;
  t->capacity = capacity;
  auto alloc = [&](void **p, uint64_t bytes) -> bool {
    if (bytes == 0) return true;
    return hipMalloc(p, bytes) == hipSuccess;
  };
  bool ok = alloc((void **)&t->head, capacity * 16) &&
            alloc((void **)&t->keys, capacity * 8 * (nkeys - 1)) &&
            alloc((void **)&t->size, capacity * 8) && alloc((void **)&t->count, capacity * 8) &&
            alloc((void **)&t->sum, capacity * 8 * nvals) &&
            alloc((void **)&t->state, NVT_STATE_WORDS * 8);
  if (ok && nvals && (flags & NVT_GB_SUMSQ)) ok = alloc((void **)&t->sumsq, capacity * 8 * nvals);
  if (ok && nvals && (flags & NVT_GB_MINMAX))
    ok = alloc((void **)&t->vmin, capacity * 8 * nvals) &&
         alloc((void **)&t->vmax, capacity * 8 * nvals);
  if (!ok) {
    nvt_gb_destroy(t);
    set_error("nvt_gb_create: hipMalloc failed for capacity %llu", (unsigned long long)capacity);
    return NVT_ENOMEM;
  }
  *out = t;
  return NVT_OK;
}

int nvt_gb_clear(nvt_gb_table *t, void *stream) {
  NVT_CHECK_ARG(t, "null table");
  NVT_PROF("groupby_clear", 0, (hipStream_t)stream);
  gb_clear_kernel<<<stream_grid(t->capacity, kBlock * 2), kBlock, 0, (hipStream_t)stream>>>(
      view_of(t));
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_gb_update(nvt_gb_table *t, const int64_t *const *keys, const uint8_t *const *key_valid,
                  const void *const *vals, const int *vdtypes, const uint8_t *const *val_valid,
                  uint64_t n, void *stream) {
  NVT_CHECK_ARG(t && keys, "null table/keys");
  NVT_CHECK_ARG(t->nvals == 0 || (vals && vdtypes), "null vals");
  if (n == 0) return NVT_OK;
  GbRowArgs a;
  memset(&a, 0, sizeof(a));
  for (int j = 0; j < t->nkeys; ++j) {
    NVT_CHECK_ARG(keys[j], "null key column");
    a.keys[j] = keys[j];
    a.key_valid[j] = key_valid ? key_valid[j] : nullptr;
  }
  for (int j = 0; j < t->nvals; ++j) {
    NVT_CHECK_ARG(vals[j], "null value column");
    a.vals[j] = vals[j];
    a.vdtype[j] = vdtypes[j];
    a.val_valid[j] = val_valid ? val_valid[j] : nullptr;
  }
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_update", 0, s);
  if (n < (1ull << 15) || n >= (1ull << 30) || t->capacity > (1ull << 31) || ab_env("NVT_GB_ATOMIC")) {
    // tiny inputs (the sort would be launch latency) and > 2^30 rows per call: per-row atomics
    gb_update_kernel<<<stream_grid(n, kBlock * 2), kBlock, 0, s>>>(view_of(t), a, n);
    NVT_CHECK_LAUNCH();
    return NVT_OK;
  }
  // scratch: words[n] + sort workspace
  const uint64_t need = ((n * 8 + 255) & ~255ull) + sort_words_tmp_bytes(n) + 256;
  if (t->scratch_bytes < need) {
    NVT_CHECK_ARG(!t->external_scratch, "workspace set by nvt_gb_set_workspace is too small");
    if (t->scratch) {
      NVT_CHECK_HIP(hipStreamSynchronize(s));
      NVT_CHECK_HIP(hipFree(t->scratch));
      t->scratch = nullptr;
      t->scratch_bytes = 0;
    }
    const uint64_t grow = need + need / 4;
    if (hipMalloc(&t->scratch, grow) != hipSuccess) {
      set_error("nvt_gb_update: hipMalloc of %llu scratch bytes failed", (unsigned long long)grow);
      return NVT_ENOMEM;
    }
    t->scratch_bytes = grow;
  }
  uint64_t *words = reinterpret_cast<uint64_t *>(t->scratch);
  void *sort_tmp = reinterpret_cast<char *>(t->scratch) + ((n * 8 + 255) & ~255ull);
  gb_assign_kernel<<<stream_grid(n, kBlock * 2), kBlock, 0, s>>>(view_of(t), a, n, words);
  NVT_CHECK_LAUNCH();
  int cap_bits = 1;  // slots are < capacity; the overflow marker `capacity` needs one more bit
  while ((1ull << cap_bits) <= t->capacity) ++cap_bits;
  uint64_t *sorted = nullptr;
  int rc = sort_words_bits(words, n, 32, 32 + cap_bits, sort_tmp, &sorted, s);
  if (rc) return rc;
  launch_segreduce<false>(view_of(t), a, n, sorted, 1u, s);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

static int fill_merge_args(nvt_gb_table *t, GbMergeArgs &a, const int64_t *const *keys,
                           const uint8_t *null_mask, const int64_t *size, const int64_t *count,
                           const double *const *sum, const double *const *sumsq,
                           const double *const *vmin, const double *const *vmax) {
  memset(&a, 0, sizeof(a));
  for (int j = 0; j < t->nkeys; ++j) {
    NVT_CHECK_ARG(keys && keys[j], "null key column");
    a.keys[j] = keys[j];
  }
  a.null_mask = null_mask;
  a.size = size;
  a.count = count;
  for (int j = 0; j < t->nvals; ++j) {
    a.sum[j] = sum ? sum[j] : nullptr;
    a.sumsq[j] = sumsq ? sumsq[j] : nullptr;
    a.vmin[j] = vmin ? vmin[j] : nullptr;
    a.vmax[j] = vmax ? vmax[j] : nullptr;
    NVT_CHECK_ARG((a.vmin[j] == nullptr) == (a.vmax[j] == nullptr), "min/max come in pairs");
  }
  return NVT_OK;
}

int nvt_gb_merge(nvt_gb_table *t, const int64_t *const *keys, const uint8_t *key_null_mask,
                 const int64_t *size, const int64_t *count, const double *const *sum,
                 const double *const *sumsq, const double *const *vmin,
                 const double *const *vmax, uint64_t n, void *stream) {
  NVT_CHECK_ARG(t, "null table");
  if (n == 0) return NVT_OK;
  GbMergeArgs a;
  int rc = fill_merge_args(t, a, keys, key_null_mask, size, count, sum, sumsq, vmin, vmax);
  if (rc) return rc;
  gb_merge_kernel<<<stream_grid(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(view_of(t), a, n);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_sort_key_u64(const void *x, int dtype, const uint8_t *valid, uint64_t n, int ascending,
                     uint64_t *out, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(x && out, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = stream_grid(n, kBlock * 4);
  switch (dtype) {
    case NVT_F32: sort_key_kernel<float><<<grid, kBlock, 0, s>>>((const float *)x, valid, n, ascending, out); break;
    case NVT_F64: sort_key_kernel<double><<<grid, kBlock, 0, s>>>((const double *)x, valid, n, ascending, out); break;
    case NVT_I32: sort_key_kernel<int32_t><<<grid, kBlock, 0, s>>>((const int32_t *)x, valid, n, ascending, out); break;
    case NVT_I64: sort_key_kernel<int64_t><<<grid, kBlock, 0, s>>>((const int64_t *)x, valid, n, ascending, out); break;
    case NVT_U8: sort_key_kernel<uint8_t><<<grid, kBlock, 0, s>>>((const uint8_t *)x, valid, n, ascending, out); break;
    default:
      set_error("nvt_sort_key_u64: unsupported dtype %d", dtype);
      return NVT_EINVAL;
  }
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_order_rows_ws_bytes(uint64_t n, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes, "null out");
  *bytes = ((n * 8 + 255) & ~255ull) + sort_words_tmp_bytes(n) + 512;
  return NVT_OK;
}

// Stable refinement of a row order: perm_out = rows of perm_in (NULL: 0..n-1) re-ordered by
//   key64 (both 32-bit halves, two radix sorts)                  when key64 != NULL
//   group id (ids in [0, ngroups), -1 = null key -> sorted last)  when gid != NULL
// perm_in / perm_out: uint64 row indices (may alias).  ws: nvt_order_rows_ws_bytes(n).
int nvt_order_rows(const uint64_t *key64, const int64_t *gid, uint64_t ngroups,
                   const uint64_t *perm_in, uint64_t n, uint64_t *perm_out, void *ws, void *stream) {
  NVT_CHECK_ARG((key64 != nullptr) != (gid != nullptr), "exactly one of key64 / gid");
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(perm_out && ws, "null pointer");
  NVT_CHECK_ARG(n < (1ull << 30) && ngroups < (1ull << 31), "at most 2^30-1 rows");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_order", 0, s);
  uint64_t *words = reinterpret_cast<uint64_t *>(ws);
  void *sort_tmp = reinterpret_cast<char *>(ws) + ((n * 8 + 255) & ~255ull);
  const unsigned grid = stream_grid(n, kBlock * 4);
  const uint64_t *cur = perm_in;
  const int rounds = key64 ? 2 : 1;
  for (int r = 0; r < rounds; ++r) {
    int hi_bits = 32;
    if (gid) {
      hi_bits = 1;
      while ((1ull << hi_bits) <= ngroups) ++hi_bits;  // ids 0..ngroups (ngroups = null marker)
    }
    pack_words_kernel<<<grid, kBlock, 0, s>>>(key64, gid, cur, n, r, (uint32_t)ngroups, words);
    NVT_CHECK_LAUNCH();
    uint64_t *sorted = nullptr;
    int rc = sort_words_bits(words, n, 32, 32 + hi_bits, sort_tmp, &sorted, s);
    if (rc) return rc;
    // the low halves are the new row order (kept as the 64-bit words: consumers mask them)
    NVT_CHECK_HIP(hipMemcpyAsync(perm_out, sorted, n * 8, hipMemcpyDeviceToDevice, s));
    cur = perm_out;
  }
  return NVT_OK;
}

// Per-group aggregates over rows ordered by group: words[i] = (group << 32 | row), groups
// ascending (the output of nvt_order_rows with gid; rows of null keys carry group >= ngroups and
// are ignored).  out_size / out_count: uint64[ngroups] (rows / non-null values of column 0);
// out_sum / out_sumsq / out_min / out_max: double[nvals][ngroups] or NULL.  All outputs must be
// initialised by the caller (0, 0, 0, 0, +inf, -inf).
int nvt_seg_aggregate(const uint64_t *words, uint64_t n, uint64_t ngroups, const void *const *vals,
                      const int *vdtypes, const uint8_t *const *val_valid, int nvals,
                      uint64_t *out_size, uint64_t *out_count, double *out_sum, double *out_sumsq,
                      double *out_min, double *out_max, void *stream) {
  if (n == 0 || ngroups == 0) return NVT_OK;
  NVT_CHECK_ARG(words && out_size && (out_count || nvals == 0), "null pointer");
  NVT_CHECK_ARG(nvals >= 0 && nvals <= kMaxVals, "nvals must be 0..8");
  NVT_CHECK_ARG((out_min == nullptr) == (out_max == nullptr), "min/max come in pairs");
  GbView t;
  memset(&t, 0, sizeof(t));
  t.nkeys = 1;
  t.nvals = nvals;
  t.cap = ngroups;
  t.size = reinterpret_cast<unsigned long long *>(out_size);
  t.count = nullptr;
  t.vcount = reinterpret_cast<unsigned long long *>(out_count);
  t.sum = out_sum;
  t.sumsq = out_sumsq;
  t.vmin = out_min;
  t.vmax = out_max;
  GbRowArgs a;
  memset(&a, 0, sizeof(a));
  for (int j = 0; j < nvals; ++j) {
    NVT_CHECK_ARG(vals && vals[j] && vdtypes, "null value column");
    a.vals[j] = vals[j];
    a.vdtype[j] = vdtypes[j];
    a.val_valid[j] = val_valid ? val_valid[j] : nullptr;
  }
  NVT_CHECK_ARG(nvals == 0 || out_sum, "null out_sum");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_aggregate", 0, s);
  launch_segreduce<false>(t, a, n, words, 1u, s);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

static uint64_t sgb_pad(uint64_t x) { return (x + 255) & ~255ull; }

int nvt_sgb_sort_ws_bytes(uint64_t n, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes, "null out");
  *bytes = sgb_pad(n * 8) + sgb_pad(sort_words_tmp_bytes(n)) + 512;
  return NVT_OK;
}

int nvt_key_minmax(const void *keys, int key_dtype, uint64_t n, int64_t *out2, void *stream) {
  NVT_CHECK_ARG(keys && out2, "null pointer");
  NVT_CHECK_ARG(key_dtype == NVT_I32 || key_dtype == NVT_I64, "key dtype must be int32 / int64");
  hipStream_t s = (hipStream_t)stream;
  const int64_t init[2] = {INT64_MAX, INT64_MIN};
  NVT_CHECK_HIP(hipMemcpyAsync(out2, init, sizeof(init), hipMemcpyHostToDevice, s));
  if (n == 0) return NVT_OK;
  NVT_PROF("groupby_sort", n * (key_dtype == NVT_I64 ? 8ull : 4ull), s);
  const unsigned grid = stream_grid(n, kBlock * 8);
  if (key_dtype == NVT_I32)
    key_minmax_kernel<int32_t><<<grid, kBlock, 0, s>>>((const int32_t *)keys, n, (long long *)out2);
  else
    key_minmax_kernel<int64_t><<<grid, kBlock, 0, s>>>((const int64_t *)keys, n, (long long *)out2);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_sgb_sort(const void *keys, int key_dtype, int64_t key_bias, const uint8_t *fold, int kfold,
                 uint64_t n, void *ws, uint64_t **sorted_out, int *row_bits_out, void *stream) {
  NVT_CHECK_ARG(keys && ws && sorted_out && row_bits_out, "null pointer");
  NVT_CHECK_ARG(key_dtype == NVT_I32 || key_dtype == NVT_I64, "key dtype must be int32 / int64");
  NVT_CHECK_ARG(kfold >= 1 && kfold <= 256, "kfold must be 1..256");
  NVT_CHECK_ARG((kfold > 1) == (fold != nullptr), "fold ids come with kfold > 1");
  int fb = 0;
  while ((1 << fb) < kfold) ++fb;
  const int rb = 32 - fb;
  NVT_CHECK_ARG(n >= 1 && n < (1ull << 30) && n <= (1ull << rb), "row index does not fit next to the fold");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_sort", n * (key_dtype == NVT_I64 ? 8ull : 4ull), s);
  // (the unsorted words are never written out: the histogram and the first scatter pass pack them
  // from the column -- sgb_pack_kernel's 8 bytes per row written + read twice are gone)
  void *sort_tmp = reinterpret_cast<char *>(ws) + sgb_pad(n * 8);
  uint64_t *sorted = nullptr;
  int rc = sort_packed_keys(keys, key_dtype, key_bias, fold, rb, n, sort_tmp, &sorted, s);
  if (rc) return rc;
  *sorted_out = sorted;
  *row_bits_out = rb;
  return NVT_OK;
}

int nvt_sgb_regroup_ws_bytes(uint64_t n, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes, "null out");
  const uint64_t ntiles = (n + kSgbTile - 1) / kSgbTile;
  *bytes = sgb_pad(ntiles * 8 + 64) + 256;
  return NVT_OK;
}

int nvt_sgb_regroup(const uint64_t *sorted, int row_bits, int kfold, int64_t key_bias, uint64_t n,
                    uint64_t cap, int64_t *out_keys, int32_t *out_keys32, uint64_t *regrouped,
                    uint64_t *state, void *ws, void *stream) {
  NVT_CHECK_ARG(sorted && out_keys && out_keys32 && regrouped && state && ws, "null pointer");
  NVT_CHECK_ARG(regrouped != sorted, "regrouped must not alias the sorted words");
  NVT_CHECK_ARG(kfold >= 1 && kfold <= 256, "kfold must be 1..256");
  NVT_CHECK_ARG(row_bits >= 24 && row_bits <= 32, "row_bits must be 24..32");
  NVT_CHECK_ARG(kfold == 1 || (1 << (32 - row_bits)) >= kfold, "the sorted words carry fewer fold bits");
  NVT_CHECK_ARG(n >= 1 && n < (1ull << 30), "1 .. 2^30-1 rows");
  NVT_CHECK_ARG(cap >= 1 && cap * (uint64_t)kfold < 0xFFFFFFFFull, "capacity * kfold must be < 2^32 - 1");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_sorted", n * 4ull, s);
  const uint64_t ntiles = (n + kSgbTile - 1) / kSgbTile;
  unsigned long long *status = reinterpret_cast<unsigned long long *>(ws);
  unsigned *ticket = reinterpret_cast<unsigned *>(status + ntiles);
  NVT_CHECK_HIP(hipMemsetAsync(status, 0, ntiles * 8 + 64, s));
  NVT_CHECK_HIP(hipMemsetAsync(state, 0, NVT_STATE_WORDS * 8, s));
  sgb_rle_kernel<false><<<(unsigned)ntiles, kBlock, 0, s>>>(sorted, n, row_bits, (unsigned)kfold, cap,
                                                            status, ticket, out_keys, out_keys32,
                                                            regrouped, state, key_bias);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_sgb_reduce(const uint64_t *regrouped, int words_kfold, int kfold, const void *const *vals,
                   const int *vdtypes, const uint8_t *const *val_valid, int nvals, int flags,
                   uint64_t n, uint64_t cap, uint64_t *out_size, double *out_sum, double *out_sumsq,
                   double *out_min, double *out_max, uint64_t *tot_size, double *tot_sum,
                   double *te_records, const uint64_t *state, const void *const *sorted_in,
                   void *const *sorted_out, void *stream) {
  NVT_CHECK_ARG(regrouped && out_size && state, "null pointer");
  NVT_CHECK_ARG(nvals >= 0 && nvals <= kMaxVals, "nvals must be 0..8");
  NVT_CHECK_ARG(nvals == 0 || (vals && vdtypes && out_sum), "null vals / out_sum");
  NVT_CHECK_ARG(words_kfold >= 1 && words_kfold <= 256, "words_kfold must be 1..256");
  NVT_CHECK_ARG(kfold == words_kfold || kfold == 1, "kfold must be 1 or the words' kfold");
  NVT_CHECK_ARG(kfold == 1 || kfold <= kSgbMaxFoldLds, "at most 16 folds");
  NVT_CHECK_ARG(kfold == 1 || (tot_size && (nvals == 0 || tot_sum)), "null totals");
  NVT_CHECK_ARG(te_records == nullptr || (kfold > 1 && nvals > 0), "records come with folds and values");
  NVT_CHECK_ARG(((flags & NVT_GB_SUMSQ) != 0) == (out_sumsq != nullptr) || nvals == 0, "sumsq flag / array");
  NVT_CHECK_ARG(((flags & NVT_GB_MINMAX) != 0) == (out_min != nullptr && out_max != nullptr) || nvals == 0,
                "min/max flag / arrays");
  NVT_CHECK_ARG(n >= 1 && n < (1ull << 30), "1 .. 2^30-1 rows");
  NVT_CHECK_ARG(cap >= 1 && cap * (uint64_t)kfold < 0xFFFFFFFFull, "capacity * kfold must be < 2^32 - 1");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_sorted", n * 4ull, s);
  const uint64_t slots = cap * (uint64_t)kfold;
  NVT_CHECK_HIP(hipMemsetAsync(out_size, 0, slots * 8, s));
  if (nvals) NVT_CHECK_HIP(hipMemsetAsync(out_sum, 0, slots * 8 * nvals, s));
  if (nvals && out_sumsq) NVT_CHECK_HIP(hipMemsetAsync(out_sumsq, 0, slots * 8 * nvals, s));
  if (nvals && out_min) {
    const double inf = std::numeric_limits<double>::infinity();
    sgb_fill_kernel<<<stream_grid(slots * nvals, kBlock * 4), kBlock, 0, s>>>(out_min, slots * nvals, inf);
    sgb_fill_kernel<<<stream_grid(slots * nvals, kBlock * 4), kBlock, 0, s>>>(out_max, slots * nvals, -inf);
    NVT_CHECK_LAUNCH();
  }
  GbView t;
  memset(&t, 0, sizeof(t));
  t.nkeys = 1;
  t.nvals = nvals;
  t.cap = slots;
  t.size = reinterpret_cast<unsigned long long *>(out_size);
  t.sum = out_sum;
  t.sumsq = nvals ? out_sumsq : nullptr;
  t.vmin = nvals ? out_min : nullptr;
  t.vmax = nvals ? out_max : nullptr;
  GbRowArgs a;
  memset(&a, 0, sizeof(a));
  for (int j = 0; j < nvals; ++j) {
    NVT_CHECK_ARG(vals[j], "null value column");
    a.vals[j] = vals[j];
    a.vdtype[j] = vdtypes[j];
    a.val_valid[j] = val_valid ? val_valid[j] : nullptr;
    a.sorted_in[j] = sorted_in ? sorted_in[j] : nullptr;
    a.sorted_out[j] = sorted_out ? sorted_out[j] : nullptr;
    NVT_CHECK_ARG(!(a.sorted_in[j] && a.val_valid[j]), "sorted values come without a validity bitmap");
    NVT_CHECK_ARG(vdtypes[j] != NVT_I64 || (!a.sorted_in[j] && !a.sorted_out[j]),
                  "int64 values are not carried in sorted order (not exact in a double)");
  }
  const uint32_t div = kfold == words_kfold ? 1u : (uint32_t)words_kfold;
  launch_segreduce<true>(t, a, n, regrouped, div, s);
  NVT_CHECK_LAUNCH();
  if (kfold > 1) {
    const size_t lds = (size_t)kBlock * 2 * ((size_t)kfold + 1) * sizeof(double);
    if (lds > 48 * 1024)  // (13-16 folds: beyond the default dynamic limit)
      NVT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sgb_fold_total_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    sgb_fold_total_kernel<<<stream_grid(cap, kBlock, 8), kBlock, lds, s>>>(
        state, (unsigned)kfold, cap, nvals, reinterpret_cast<const unsigned long long *>(out_size),
        out_sum, reinterpret_cast<unsigned long long *>(tot_size), tot_sum, te_records);
    NVT_CHECK_LAUNCH();
  }
  return NVT_OK;
}

int nvt_count_merge_sorted_ws_bytes(uint64_t n, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes, "null out");
  const uint64_t ntiles = (n + kSgbTile - 1) / kSgbTile;
  *bytes = sgb_pad(n * 8) + sgb_pad(sort_words_tmp_bytes(n)) + sgb_pad(n * 4) + sgb_pad(n * 8) +
           sgb_pad(ntiles * 8 + 64) + 512;
  return NVT_OK;
}

int nvt_count_merge_sorted(const int64_t *rows, uint64_t n, const uint64_t *seg_off, int nseg,
                           int ncol, int32_t *out_keys, int64_t *out_col, double *out_sum,
                           uint64_t *state, void *ws, void *stream) {
  NVT_CHECK_ARG(rows && seg_off && out_keys && out_col && out_sum && state && ws, "null pointer");
  NVT_CHECK_ARG(n >= 1 && n < (1ull << kMergeRowBits), "1 .. 2^26-1 rows per call");
  NVT_CHECK_ARG(nseg >= 1 && ncol >= 1 && ncol <= (1 << (64 - 32 - kMergeRowBits)) && nseg % ncol == 0,
                "1..64 columns, whole groups of segments");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("count_merge_sorted", n * 8ull, s);
  const uint64_t ntiles = (n + kSgbTile - 1) / kSgbTile;
  char *p = reinterpret_cast<char *>(ws);
  uint64_t *words = reinterpret_cast<uint64_t *>(p);
  p += sgb_pad(n * 8);
  void *sort_tmp = p;
  p += sgb_pad(sort_words_tmp_bytes(n));
  int32_t *cnt32 = reinterpret_cast<int32_t *>(p);
  p += sgb_pad(n * 4);
  uint64_t *regrouped = reinterpret_cast<uint64_t *>(p);
  p += sgb_pad(n * 8);
  unsigned long long *status = reinterpret_cast<unsigned long long *>(p);
  unsigned *ticket = reinterpret_cast<unsigned *>(status + ntiles);
  NVT_CHECK_HIP(hipMemsetAsync(status, 0, ntiles * 8 + 64, s));
  NVT_CHECK_HIP(hipMemsetAsync(state, 0, NVT_STATE_WORDS * 8, s));
  NVT_CHECK_HIP(hipMemsetAsync(out_sum, 0, n * 8, s));
  merge_pack_kernel<<<stream_grid(n, kBlock * 4), kBlock, 0, s>>>(rows, n, seg_off, nseg, ncol, words,
                                                                  cnt32);
  NVT_CHECK_LAUNCH();
  int cb = 1;
  while ((1 << cb) < ncol) ++cb;
  uint64_t *sorted = nullptr;
  int rc = sort_words_bits(words, n, kMergeRowBits, kMergeRowBits + 32 + cb, sort_tmp, &sorted, s);
  if (rc) return rc;
  sgb_rle_kernel<true><<<(unsigned)ntiles, kBlock, 0, s>>>(sorted, n, kMergeRowBits, 1u, n, status,
                                                           ticket, out_col, out_keys, regrouped, state,
                                                           0);
  NVT_CHECK_LAUNCH();
  GbView t;
  memset(&t, 0, sizeof(t));
  t.nkeys = 1;
  t.nvals = 1;
  t.cap = n;
  t.sum = out_sum;
  GbRowArgs a;
  memset(&a, 0, sizeof(a));
  a.vals[0] = cnt32;
  a.vdtype[0] = NVT_I32;
  launch_segreduce<true>(t, a, n, regrouped, 1u, s);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_gb_state(nvt_gb_table *t, uint64_t *host_state, void *stream) {
  NVT_CHECK_ARG(t && host_state, "null pointer");
  NVT_CHECK_HIP(hipMemcpyAsync(host_state, t->state, NVT_STATE_WORDS * 8, hipMemcpyDeviceToHost,
                               (hipStream_t)stream));
  NVT_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return NVT_OK;
}

int nvt_gb_compact(nvt_gb_table *t, int64_t *const *out_keys, uint8_t *out_null_mask,
                   int64_t *out_size, int64_t *out_count, double *const *out_sum,
                   double *const *out_sumsq, double *const *out_min, double *const *out_max,
                   uint64_t *out_n, void *stream) {
  NVT_CHECK_ARG(t && out_n, "null pointer");
  GbOutArgs o;
  memset(&o, 0, sizeof(o));
  for (int j = 0; j < t->nkeys; ++j) o.keys[j] = out_keys ? out_keys[j] : nullptr;
  o.null_mask = out_null_mask;
  o.size = out_size;
  o.count = out_count;
  for (int j = 0; j < t->nvals; ++j) {
    o.sum[j] = out_sum ? out_sum[j] : nullptr;
    o.sumsq[j] = out_sumsq ? out_sumsq[j] : nullptr;
    o.vmin[j] = out_min ? out_min[j] : nullptr;
    o.vmax[j] = out_max ? out_max[j] : nullptr;
  }
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_compact", 0, s);
  NVT_CHECK_HIP(hipMemsetAsync(out_n, 0, sizeof(uint64_t), s));
  gb_compact_kernel<<<stream_grid(t->capacity, kBlock * kGbCompactItems), kBlock, 0, s>>>(view_of(t), o, out_n);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_gb_index_build(nvt_gb_table *t, const int64_t *const *keys, const uint8_t *key_null_mask,
                       uint64_t n_groups, void *stream) {
  NVT_CHECK_ARG(t, "null table");
  NVT_CHECK_ARG(n_groups < t->capacity, "capacity must exceed the number of groups");
  int rc = nvt_gb_clear(t, stream);
  if (rc) return rc;
  if (n_groups == 0) return NVT_OK;
  GbMergeArgs a;
  rc = fill_merge_args(t, a, keys, key_null_mask, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr);
  if (rc) return rc;
  gb_index_build_kernel<<<stream_grid(n_groups, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      view_of(t), a, n_groups);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_gb_lookup(nvt_gb_table *t, const int64_t *const *keys, const uint8_t *const *key_valid,
                  uint64_t n, int64_t *out_group, void *stream) {
  NVT_CHECK_ARG(t && keys && out_group, "null pointer");
  if (n == 0) return NVT_OK;
  GbRowArgs a;
  memset(&a, 0, sizeof(a));
  for (int j = 0; j < t->nkeys; ++j) {
    NVT_CHECK_ARG(keys[j], "null key column");
    a.keys[j] = keys[j];
    a.key_valid[j] = key_valid ? key_valid[j] : nullptr;
  }
  NVT_PROF("groupby_lookup", n * 8ull * (uint64_t)t->nkeys, (hipStream_t)stream);
  gb_lookup_kernel<<<stream_grid(n, kBlock * 2), kBlock, 0, (hipStream_t)stream>>>(view_of(t), a,
                                                                                    n, out_group);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // extern "C"
