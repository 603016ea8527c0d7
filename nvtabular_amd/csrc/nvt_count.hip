// Categorify.fit on MI355X: groupby-size as open-addressing hash tables.
//
// Replaces the pandas/libcudf groupby of categorify.py:955-1051
// (_top_level_groupby, agg_list == ["size"]) and the concat + re-groupby of
// categorify.py:1054-1070 (_mid_level_groupby).
//
// Kernel shape (HBM-bound, integer work -- no MFMA):
//   * each lane streams 16 B of keys per iteration (4 x int32 / 2 x int64),
//     grid-stride, 1024 workgroups so all 256 CUs hold 4 resident blocks;
//   * every workgroup owns a 4096-slot table in LDS that absorbs the hot keys
//     of a skewed (Zipf) column with ds atomics, so a low-cardinality column
//     never touches HBM except for its input stream;
//   * keys that do not fit the LDS table go to the global table (linear
//     probing, 32/64-bit CAS on the key word, atomic add on the count word);
//   * at the end each block flushes its LDS table into the global one.
#include "nvt_common.hpp"

namespace nvt {

template <typename K>
struct CountSlot;
template <>
struct CountSlot<int32_t> {
  int32_t key;
  uint32_t cnt;
};
template <>
struct CountSlot<int64_t> {
  int64_t key;
  unsigned long long cnt;
};

template <typename K>
struct KeyTraits;
template <>
struct KeyTraits<int32_t> {
  static constexpr int32_t empty = INT32_MIN;
  static constexpr int vec = 4;
  using cnt_t = uint32_t;
  using cas_t = int;
};
template <>
struct KeyTraits<int64_t> {
  static constexpr int64_t empty = INT64_MIN;
  static constexpr int vec = 2;
  using cnt_t = unsigned long long;
  using cas_t = unsigned long long;
};

constexpr int kMaxProbe = 128;  // longer chains mean the table is too full -> overflow

template <typename K>
__device__ __forceinline__ K cas_key(K *addr, K expect, K val) {
  using C = typename KeyTraits<K>::cas_t;
  return (K)atomicCAS(reinterpret_cast<C *>(addr), (C)expect, (C)val);
}

// Insert `key` with weight `w` into the global table.  Returns 1 when a new
// slot was claimed.  Sets *ovf when the probe chain is too long.
template <typename K>
__device__ __forceinline__ int global_insert(CountSlot<K> *table, uint64_t mask, K key,
                                             typename KeyTraits<K>::cnt_t w, unsigned *ovf) {
  constexpr K EMPTY = KeyTraits<K>::empty;
  uint64_t slot = (uint64_t)slot_hash(key) & mask;
  for (int probe = 0; probe < kMaxProbe; ++probe) {
    K cur = __hip_atomic_load(&table[slot].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int claimed = 0;
    if (cur == EMPTY) {
      cur = cas_key<K>(&table[slot].key, EMPTY, key);
      claimed = (cur == EMPTY);
      if (claimed) cur = key;
    }
    if (cur == key) {
      atomicAdd(&table[slot].cnt, w);
      return claimed;
    }
    slot = (slot + 1) & mask;
  }
  *ovf = 1;
  return 0;
}

template <typename K, int LDS_SLOTS>
__global__ __launch_bounds__(kBlock) void count_kernel(const K *__restrict__ keys,
                                                       const uint8_t *__restrict__ valid,
                                                       uint64_t n, CountSlot<K> *table,
                                                       uint64_t mask, uint64_t *state) {
  constexpr K EMPTY = KeyTraits<K>::empty;
  constexpr int VEC = KeyTraits<K>::vec;
  __shared__ K lkeys[LDS_SLOTS];
  __shared__ uint32_t lcnt[LDS_SLOTS];
  __shared__ unsigned s_nulls, s_sent, s_new, s_ovf;

  for (int i = threadIdx.x; i < LDS_SLOTS; i += kBlock) {
    lkeys[i] = EMPTY;
    lcnt[i] = 0;
  }
  if (threadIdx.x == 0) {
    s_nulls = 0;
    s_sent = 0;
    s_new = 0;
    s_ovf = 0;
  }
  __syncthreads();

  unsigned my_nulls = 0, my_sent = 0, my_new = 0;
  unsigned my_ovf = 0;

  auto add_key = [&](K key) {
    if (key == EMPTY) {
      ++my_sent;
      return;
    }
    // LDS front table: upper hash bits so it is decorrelated from the global slot
    uint32_t h = (uint32_t)(slot_hash(key) >> 17);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      uint32_t s = (h + p) & (LDS_SLOTS - 1);
      K cur = lkeys[s];
      if (cur == EMPTY) {
        cur = cas_key<K>(&lkeys[s], EMPTY, key);
        if (cur == EMPTY) cur = key;
      }
      if (cur == key) {
        atomicAdd(&lcnt[s], 1u);
        return;
      }
    }
    my_new += global_insert<K>(table, mask, key, 1, &my_ovf);
  };

  const uint64_t nvec = n / VEC;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  using VecT = typename std::conditional<sizeof(K) == 4, int4, longlong2>::type;
  const VecT *vkeys = reinterpret_cast<const VecT *>(keys);
  for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
    // a table that already overflowed is going to be thrown away: stop early
    if (my_ovf || __hip_atomic_load(&state[NVT_ST_OVERFLOW], __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT))
      break;
    VecT pack = vkeys[v];
    unsigned vbits = 0xF;
    if (valid != nullptr) {
      uint64_t row = v * VEC;  // VEC divides 8, so the VEC bits sit inside one byte
      vbits = (valid[row >> 3] >> (row & 7));
    }
    K k[VEC];
    if constexpr (sizeof(K) == 4) {
      k[0] = pack.x;
      k[1] = pack.y;
      k[2] = pack.z;
      k[3] = pack.w;
    } else {
      k[0] = pack.x;
      k[1] = pack.y;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if ((vbits >> j) & 1)
        add_key(k[j]);
      else
        ++my_nulls;
    }
  }
  // scalar tail
  for (uint64_t i = nvec * VEC + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    if (bit_valid(valid, i))
      add_key(keys[i]);
    else
      ++my_nulls;
  }
  __syncthreads();

  // flush the LDS table
  for (int s = threadIdx.x; s < LDS_SLOTS; s += kBlock) {
    K key = lkeys[s];
    if (key != EMPTY)
      my_new += global_insert<K>(table, mask, key, (typename KeyTraits<K>::cnt_t)lcnt[s], &my_ovf);
  }

  if (my_nulls) atomicAdd(&s_nulls, my_nulls);
  if (my_sent) atomicAdd(&s_sent, my_sent);
  if (my_new) atomicAdd(&s_new, my_new);
  if (my_ovf) atomicOr(&s_ovf, 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_nulls) atomicAdd((unsigned long long *)&state[NVT_ST_NULLS], (unsigned long long)s_nulls);
    if (s_sent)
      atomicAdd((unsigned long long *)&state[NVT_ST_SENTINEL], (unsigned long long)s_sent);
    if (s_new) atomicAdd((unsigned long long *)&state[NVT_ST_OCCUPIED], (unsigned long long)s_new);
    if (s_ovf) atomicOr((unsigned long long *)&state[NVT_ST_OVERFLOW], 1ull);
    if (blockIdx.x == 0)
      atomicAdd((unsigned long long *)&state[NVT_ST_ROWS], (unsigned long long)n);
  }
}

template <typename K>
__global__ __launch_bounds__(kBlock) void merge_kernel(const K *__restrict__ keys,
                                                       const int64_t *__restrict__ counts,
                                                       uint64_t n, CountSlot<K> *table,
                                                       uint64_t mask, uint64_t *state) {
  constexpr K EMPTY = KeyTraits<K>::empty;
  unsigned my_new = 0, my_ovf = 0;
  unsigned long long my_sent = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    K key = keys[i];
    int64_t c = counts[i];
    if (key == EMPTY) {
      my_sent += (unsigned long long)c;
      continue;
    }
    my_new += global_insert<K>(table, mask, key, (typename KeyTraits<K>::cnt_t)c, &my_ovf);
  }
  if (my_sent) atomicAdd((unsigned long long *)&state[NVT_ST_SENTINEL], my_sent);
  if (my_new) atomicAdd((unsigned long long *)&state[NVT_ST_OCCUPIED], (unsigned long long)my_new);
  if (my_ovf) atomicOr((unsigned long long *)&state[NVT_ST_OVERFLOW], 1ull);
}

template <typename K>
__global__ __launch_bounds__(kBlock) void clear_kernel(CountSlot<K> *table, uint64_t capacity) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  CountSlot<K> e;
  e.key = KeyTraits<K>::empty;
  e.cnt = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < capacity; i += stride)
    table[i] = e;
}

// table -> dense arrays, arbitrary order (the vocabulary sort fixes it afterwards).
// Each block takes tiles of kBlock * kCompactItems slots, ranks the occupied ones with
// a wave scan + LDS, and reserves its output range with ONE atomic per tile -- a
// per-wave atomic on the single cursor serialises at ~90 atomics/us and was 16 ms
// per step in the first profile (profiles/r01_baseline_kernel_stats.csv).
constexpr int kCompactItems = 8;
template <typename K>
__global__ __launch_bounds__(kBlock) void compact_kernel(const CountSlot<K> *__restrict__ table,
                                                         uint64_t capacity, K *out_keys,
                                                         int64_t *out_counts, uint64_t *out_n) {
  constexpr K EMPTY = KeyTraits<K>::empty;
  constexpr uint64_t TILE = (uint64_t)kBlock * kCompactItems;
  __shared__ unsigned wsum[kBlock / kWave];
  __shared__ unsigned long long tile_base;
  const unsigned lane = lane_id();
  const unsigned w = threadIdx.x / kWave;
  const uint64_t ntiles = (capacity + TILE - 1) / TILE;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    // thread owns kCompactItems consecutive slots (coalesced enough: 64-128 B per lane)
    const uint64_t first = t * TILE + (uint64_t)threadIdx.x * kCompactItems;
    CountSlot<K> s[kCompactItems];
    unsigned mine = 0;
#pragma unroll
    for (int j = 0; j < kCompactItems; ++j) {
      s[j].key = EMPTY;
      s[j].cnt = 0;
      if (first + j < capacity) s[j] = table[first + j];
      mine += (s[j].key != EMPTY);
    }
    unsigned inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o = __shfl_up(inc, off, 64);
      if (lane >= (unsigned)off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned wbase = 0, total = 0;
    for (unsigned i = 0; i < kBlock / kWave; ++i) {
      if (i < w) wbase += wsum[i];
      total += wsum[i];
    }
    if (threadIdx.x == 0)
      tile_base = total ? atomicAdd((unsigned long long *)out_n, (unsigned long long)total) : 0;
    __syncthreads();
    uint64_t pos = tile_base + wbase + inc - mine;
#pragma unroll
    for (int j = 0; j < kCompactItems; ++j) {
      if (s[j].key != EMPTY) {
        out_keys[pos] = s[j].key;
        out_counts[pos] = (int64_t)s[j].cnt;
        ++pos;
      }
    }
    __syncthreads();
  }
}

template <typename K>
int count_launch(const K *keys, const uint8_t *valid, uint64_t n, void *table, uint64_t capacity,
                 uint64_t *state, hipStream_t stream) {
  NVT_CHECK_ARG(table && state, "null table/state");
  NVT_CHECK_ARG(capacity >= 64 && (capacity & (capacity - 1)) == 0, "capacity must be 2^k >= 64");
  NVT_CHECK_ARG(n == 0 || keys, "null keys");
  NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(keys) & 15) == 0, "keys must be 16-byte aligned");
  if (n == 0) return NVT_OK;
  constexpr int VEC = KeyTraits<K>::vec;
  // 2 resident blocks per CU, each with an 8192-slot LDS table (64/96 KiB): fewer,
  // larger private tables halve the number of end-of-kernel flushes into the global table
  unsigned grid = stream_grid(n / VEC + 1, kBlock * 4, 2);
  count_kernel<K, 8192><<<grid, kBlock, 0, stream>>>(keys, valid, n,
                                                     reinterpret_cast<CountSlot<K> *>(table),
                                                     capacity - 1, state);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

template <typename K>
int merge_launch(const K *keys, const int64_t *counts, uint64_t n, void *table, uint64_t capacity,
                 uint64_t *state, hipStream_t stream) {
  NVT_CHECK_ARG(table && state, "null table/state");
  NVT_CHECK_ARG(capacity >= 64 && (capacity & (capacity - 1)) == 0, "capacity must be 2^k >= 64");
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(keys && counts, "null keys/counts");
  merge_kernel<K><<<stream_grid(n, kBlock), kBlock, 0, stream>>>(
      keys, counts, n, reinterpret_cast<CountSlot<K> *>(table), capacity - 1, state);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

template <typename K>
int compact_launch(const void *table, uint64_t capacity, K *out_keys, int64_t *out_counts,
                   uint64_t *out_n, hipStream_t stream) {
  NVT_CHECK_ARG(table && out_keys && out_counts && out_n, "null pointer");
  NVT_CHECK_HIP(hipMemsetAsync(out_n, 0, sizeof(uint64_t), stream));
  compact_kernel<K><<<stream_grid(capacity, kBlock * kCompactItems), kBlock, 0, stream>>>(
      reinterpret_cast<const CountSlot<K> *>(table), capacity, out_keys, out_counts, out_n);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // namespace nvt

using namespace nvt;

extern "C" {

int nvt_count_table_bytes(int key_bytes, uint64_t capacity, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes && (key_bytes == 4 || key_bytes == 8), "key_bytes must be 4 or 8");
  *bytes = capacity * (key_bytes == 4 ? sizeof(CountSlot<int32_t>) : sizeof(CountSlot<int64_t>));
  return NVT_OK;
}

int nvt_count_clear(void *table, int key_bytes, uint64_t capacity, uint64_t *state, void *stream) {
  NVT_CHECK_ARG(table && (key_bytes == 4 || key_bytes == 8), "bad table/key_bytes");
  hipStream_t s = (hipStream_t)stream;
  if (key_bytes == 4)
    clear_kernel<int32_t><<<stream_grid(capacity, kBlock * 4), kBlock, 0, s>>>(
        reinterpret_cast<CountSlot<int32_t> *>(table), capacity);
  else
    clear_kernel<int64_t><<<stream_grid(capacity, kBlock * 4), kBlock, 0, s>>>(
        reinterpret_cast<CountSlot<int64_t> *>(table), capacity);
  NVT_CHECK_LAUNCH();
  if (state) NVT_CHECK_HIP(hipMemsetAsync(state, 0, NVT_STATE_WORDS * sizeof(uint64_t), s));
  return NVT_OK;
}

int nvt_count_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, void *table,
                  uint64_t capacity, uint64_t *state, void *stream) {
  return count_launch<int32_t>(keys, valid, n, table, capacity, state, (hipStream_t)stream);
}
int nvt_count_i64(const int64_t *keys, const uint8_t *valid, uint64_t n, void *table,
                  uint64_t capacity, uint64_t *state, void *stream) {
  return count_launch<int64_t>(keys, valid, n, table, capacity, state, (hipStream_t)stream);
}
int nvt_count_merge_i32(const int32_t *keys, const int64_t *counts, uint64_t n, void *table,
                        uint64_t capacity, uint64_t *state, void *stream) {
  return merge_launch<int32_t>(keys, counts, n, table, capacity, state, (hipStream_t)stream);
}
int nvt_count_merge_i64(const int64_t *keys, const int64_t *counts, uint64_t n, void *table,
                        uint64_t capacity, uint64_t *state, void *stream) {
  return merge_launch<int64_t>(keys, counts, n, table, capacity, state, (hipStream_t)stream);
}
int nvt_count_compact_i32(const void *table, uint64_t capacity, int32_t *out_keys,
                          int64_t *out_counts, uint64_t *out_n, void *stream) {
  return compact_launch<int32_t>(table, capacity, out_keys, out_counts, out_n,
                                 (hipStream_t)stream);
}
int nvt_count_compact_i64(const void *table, uint64_t capacity, int64_t *out_keys,
                          int64_t *out_counts, uint64_t *out_n, void *stream) {
  return compact_launch<int64_t>(table, capacity, out_keys, out_counts, out_n,
                                 (hipStream_t)stream);
}

}  // extern "C"
