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

// ---------------------------------------------------------------------------
// Vocabulary order: (count descending, key ascending) -- the two sort_values
// calls of categorify.py:1300,1316 with the stable tie rule.
// LSD radix sort, 8-bit digits, one wave per 1024-element tile; stability
// inside a tile comes from ballot-matching equal digits in lane order.
// Passes whose digit is constant over the whole array are skipped.
// ---------------------------------------------------------------------------
constexpr int kSortRows = 16;
constexpr int kSortTile = kWave * kSortRows;

template <typename K>
__device__ __forceinline__ unsigned sort_digit(K key, int64_t cnt, int pass) {
  constexpr int KB = (int)sizeof(K);
  if (pass < KB) {
    using U = typename std::make_unsigned<K>::type;
    U u = (U)key ^ ((U)1 << (8 * KB - 1));  // signed order
    return (unsigned)((u >> (8 * pass)) & 0xFF);
  }
  uint64_t inv = ~(uint64_t)cnt;  // descending counts
  return (unsigned)((inv >> (8 * (pass - KB))) & 0xFF);
}

template <typename K>
__global__ __launch_bounds__(kBlock) void sort_pass_hist_kernel(const K *__restrict__ keys,
                                                                const int64_t *__restrict__ cnts,
                                                                uint64_t n,
                                                                unsigned long long *pass_hist) {
  constexpr int NP = (int)sizeof(K) + 8;
  __shared__ unsigned h[NP * 256];
  for (int i = threadIdx.x; i < NP * 256; i += kBlock) h[i] = 0;
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    K k = keys[i];
    int64_t c = cnts[i];
#pragma unroll
    for (int p = 0; p < NP; ++p) atomicAdd(&h[p * 256 + sort_digit<K>(k, c, p)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NP * 256; i += kBlock)
    if (h[i]) atomicAdd(&pass_hist[i], (unsigned long long)h[i]);
}

// peers = lanes of this wave holding the same digit (inactive lanes excluded)
__device__ __forceinline__ unsigned long long match_digit(unsigned digit, bool active) {
  unsigned long long peers = __ballot(active);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    unsigned long long m = __ballot((digit >> b) & 1);
    peers &= ((digit >> b) & 1) ? m : ~m;
  }
  return peers;
}

template <typename K>
__global__ __launch_bounds__(kWave) void sort_tile_hist_kernel(const K *__restrict__ keys,
                                                               const int64_t *__restrict__ cnts,
                                                               uint64_t n, int pass,
                                                               unsigned *tile_hist,
                                                               uint64_t ntiles) {
  __shared__ unsigned h[256];
  const unsigned lane = threadIdx.x;
  for (int i = lane; i < 256; i += kWave) h[i] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
#pragma unroll 4
  for (int r = 0; r < kSortRows; ++r) {
    uint64_t i = base + (uint64_t)r * kWave + lane;
    if (i < n) atomicAdd(&h[sort_digit<K>(keys[i], cnts[i], pass)], 1u);
  }
  __syncthreads();
  for (int d = lane; d < 256; d += kWave) tile_hist[(uint64_t)d * ntiles + blockIdx.x] = h[d];
}

// Exclusive scan of `len` uint32 in three steps (chunk scan, chunk-total scan, add).
constexpr int kScanChunk = 2048;  // 256 threads x 8
__global__ __launch_bounds__(kBlock) void scan_chunk_kernel(unsigned *data, uint64_t len,
                                                            unsigned long long *chunk_tot) {
  __shared__ unsigned wsum[kBlock / kWave];
  const uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * 8;
  unsigned v[8];
  unsigned tot = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v[j] = (base + j < len) ? data[base + j] : 0;
    tot += v[j];
  }
  // wave inclusive scan of tot
  unsigned inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned o = __shfl_up(inc, off, 64);
    if (lane_id() >= (unsigned)off) inc += o;
  }
  const unsigned w = threadIdx.x / kWave;
  if (lane_id() == 63) wsum[w] = inc;
  __syncthreads();
  unsigned wbase = 0;
  for (unsigned i = 0; i < w; ++i) wbase += wsum[i];
  unsigned run = wbase + inc - tot;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (base + j < len) data[base + j] = run;
    run += v[j];
  }
  if (threadIdx.x == kBlock - 1) chunk_tot[blockIdx.x] = (unsigned long long)(wbase + inc);
}
__global__ void scan_totals_kernel(unsigned long long *chunk_tot, uint64_t nchunks) {
  // single block; sequential over tiles of 256 with a wave/block scan
  __shared__ unsigned long long carry;
  __shared__ unsigned long long wsum[kBlock / kWave];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint64_t b = 0; b < nchunks; b += kBlock) {
    uint64_t i = b + threadIdx.x;
    unsigned long long v = i < nchunks ? chunk_tot[i] : 0;
    unsigned long long inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned long long o = __shfl_up(inc, off, 64);
      if (lane_id() >= (unsigned)off) inc += o;
    }
    const unsigned w = threadIdx.x / kWave;
    if (lane_id() == 63) wsum[w] = inc;
    __syncthreads();
    unsigned long long wbase = carry;
    for (unsigned k = 0; k < w; ++k) wbase += wsum[k];
    if (i < nchunks) chunk_tot[i] = wbase + inc - v;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry = wbase + inc;
    __syncthreads();
  }
}
__global__ __launch_bounds__(kBlock) void scan_add_kernel(unsigned *data, uint64_t len,
                                                          const unsigned long long *chunk_tot) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * 8;
  const unsigned add = (unsigned)chunk_tot[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (base + j < len) data[base + j] += add;
}

template <typename K>
__global__ __launch_bounds__(kWave) void sort_scatter_kernel(
    const K *__restrict__ keys, const int64_t *__restrict__ cnts, uint64_t n, int pass,
    const unsigned *__restrict__ tile_off, uint64_t ntiles, K *out_keys, int64_t *out_cnts) {
  __shared__ unsigned run[256];
  const unsigned lane = threadIdx.x;
  for (int d = lane; d < 256; d += kWave) run[d] = tile_off[(uint64_t)d * ntiles + blockIdx.x];
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
  for (int r = 0; r < kSortRows; ++r) {
    uint64_t i = base + (uint64_t)r * kWave + lane;
    bool active = i < n;
    K k = active ? keys[i] : (K)0;
    int64_t c = active ? cnts[i] : 0;
    unsigned d = sort_digit<K>(k, c, pass);
    unsigned long long peers = match_digit(d, active);
    unsigned rank = __popcll(peers & ((1ull << lane) - 1ull));
    unsigned dst = 0;
    if (active) dst = run[d] + rank;
    __syncthreads();  // all lanes read run[] before leaders bump it
    if (active && rank == 0) run[d] += (unsigned)__popcll(peers);
    __syncthreads();
    if (active) {
      out_keys[dst] = k;
      out_cnts[dst] = c;
    }
  }
}

template <typename K>
int vocab_sort(K *keys, int64_t *counts, uint64_t n, void *tmp, hipStream_t stream) {
  constexpr int NP = (int)sizeof(K) + 8;
  if (n <= 1) return NVT_OK;
  NVT_CHECK_ARG(n < (1ull << 32), "at most 2^32-1 vocabulary entries");
  const uint64_t ntiles = (n + kSortTile - 1) / kSortTile;
  // tmp layout: keys2 | counts2 | tile_hist | chunk_tot | pass_hist
  char *p = reinterpret_cast<char *>(tmp);
  int64_t *counts2 = reinterpret_cast<int64_t *>(p);
  p += n * sizeof(int64_t);
  K *keys2 = reinterpret_cast<K *>(p);
  p += ((n * sizeof(K) + 15) / 16) * 16;
  unsigned *tile_hist = reinterpret_cast<unsigned *>(p);
  const uint64_t hist_len = 256 * ntiles;
  p += ((hist_len * sizeof(unsigned) + 15) / 16) * 16;
  const uint64_t nchunks = (hist_len + kScanChunk - 1) / kScanChunk;
  unsigned long long *chunk_tot = reinterpret_cast<unsigned long long *>(p);
  p += nchunks * sizeof(unsigned long long);
  unsigned long long *pass_hist = reinterpret_cast<unsigned long long *>(p);

  NVT_CHECK_HIP(hipMemsetAsync(pass_hist, 0, NP * 256 * sizeof(unsigned long long), stream));
  sort_pass_hist_kernel<K><<<stream_grid(n, kBlock * 8, 4), kBlock, 0, stream>>>(keys, counts, n,
                                                                                   pass_hist);
  NVT_CHECK_LAUNCH();
  unsigned long long host_hist[NP * 256];
  NVT_CHECK_HIP(hipMemcpyAsync(host_hist, pass_hist, sizeof(host_hist), hipMemcpyDeviceToHost,
                               stream));
  NVT_CHECK_HIP(hipStreamSynchronize(stream));

  K *src_k = keys, *dst_k = keys2;
  int64_t *src_c = counts, *dst_c = counts2;
  for (int pass = 0; pass < NP; ++pass) {
    bool trivial = false;
    for (int d = 0; d < 256; ++d)
      if (host_hist[pass * 256 + d] == n) trivial = true;
    if (trivial) continue;
    sort_tile_hist_kernel<K><<<(unsigned)ntiles, kWave, 0, stream>>>(src_k, src_c, n, pass,
                                                                     tile_hist, ntiles);
    NVT_CHECK_LAUNCH();
    scan_chunk_kernel<<<(unsigned)nchunks, kBlock, 0, stream>>>(tile_hist, hist_len, chunk_tot);
    NVT_CHECK_LAUNCH();
    scan_totals_kernel<<<1, kBlock, 0, stream>>>(chunk_tot, nchunks);
    NVT_CHECK_LAUNCH();
    scan_add_kernel<<<(unsigned)nchunks, kBlock, 0, stream>>>(tile_hist, hist_len, chunk_tot);
    NVT_CHECK_LAUNCH();
    sort_scatter_kernel<K><<<(unsigned)ntiles, kWave, 0, stream>>>(src_k, src_c, n, pass,
                                                                   tile_hist, ntiles, dst_k, dst_c);
    NVT_CHECK_LAUNCH();
    K *tk = src_k;
    src_k = dst_k;
    dst_k = tk;
    int64_t *tc = src_c;
    src_c = dst_c;
    dst_c = tc;
  }
  if (src_k != keys) {
    NVT_CHECK_HIP(hipMemcpyAsync(keys, src_k, n * sizeof(K), hipMemcpyDeviceToDevice, stream));
    NVT_CHECK_HIP(
        hipMemcpyAsync(counts, src_c, n * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
  }
  return NVT_OK;
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

int nvt_vocab_sort_tmp_bytes(int key_bytes, uint64_t n, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes && (key_bytes == 4 || key_bytes == 8), "key_bytes must be 4 or 8");
  const uint64_t ntiles = (n + kSortTile - 1) / kSortTile;
  const uint64_t hist_len = 256 * ntiles;
  const uint64_t nchunks = (hist_len + kScanChunk - 1) / kScanChunk;
  uint64_t b = n * 8 + ((n * key_bytes + 15) / 16) * 16 + ((hist_len * 4 + 15) / 16) * 16 +
               nchunks * 8 + (uint64_t)(key_bytes + 8) * 256 * 8 + 64;
  *bytes = b;
  return NVT_OK;
}
int nvt_vocab_sort_i32(int32_t *keys, int64_t *counts, uint64_t n, void *tmp, void *stream) {
  NVT_CHECK_ARG(n <= 1 || (keys && counts && tmp), "null pointer");
  return vocab_sort<int32_t>(keys, counts, n, tmp, (hipStream_t)stream);
}
int nvt_vocab_sort_i64(int64_t *keys, int64_t *counts, uint64_t n, void *tmp, void *stream) {
  NVT_CHECK_ARG(n <= 1 || (keys && counts && tmp), "null pointer");
  return vocab_sort<int64_t>(keys, counts, n, tmp, (hipStream_t)stream);
}

}  // extern "C"
