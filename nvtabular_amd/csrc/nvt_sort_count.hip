// Categorify.fit groupby-size, path 10 ("sort path"): int32 keys, unweighted, more distinct keys
// than the range path's 1024 LDS buckets hold (> ~6.5 M: Criteo-1TB's 38-40 M-unique columns,
// bench/examples/dask-nvtabular-criteo-benchmark.py:360-366).  Replaces categorify.py:955-1051
// (_top_level_groupby, size only) for columns in which nearly every row carries its own key:
// counting such rows in hash tables means two partition passes plus one LDS table per 3 k keys
// (path 3) and leaves an unordered list of tens of millions of entries to a 7-pass sort.  Here:
//
//   sc_pack_kernel    row -> 64-bit word (order-preserving key image << 32); null rows -> ~0
//   sort_words_bits   onesweep LSD radix sort on the 32 key bits (nvt_sort.hip, 4 passes)
//   sc_rle_kernel     run heads of the sorted words: rank by a decoupled look-back over the
//                     tiles, out_keys[rank] = key, head_pos[rank] = first row of the run
//   sc_counts_kernel  counts = differences of consecutive run starts; histogram of
//                     min(count, 255), largest count
//
// The (key, count) list leaves this path SORTED BY KEY like the range path's, so the vocabulary
// is ordered by the one-pass class scatter (nvt_sort.hip: cls_scatter_kernel).
#include "nvt_common.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"

namespace nvt {

namespace {

constexpr int kScBS = 256, kScRows = 16, kScTile = kScBS * kScRows;  // 4096 words per tile
constexpr unsigned kScAgg = 1u << 30, kScPrefix = 2u << 30, kScMask = (1u << 30) - 1u;
constexpr uint64_t kInvalidWord = ~0ull;

__global__ __launch_bounds__(kBlock) void sc_pack_kernel(const int32_t *__restrict__ keys,
                                                         const uint8_t *__restrict__ valid,
                                                         uint64_t n, uint64_t *__restrict__ words,
                                                         uint64_t *state) {
  __shared__ unsigned long long s_nulls;
  if (threadIdx.x == 0) s_nulls = 0;
  __syncthreads();
  const uint64_t nvec = n / 4;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  unsigned long long nulls = 0;
  for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
    const int4 p = reinterpret_cast<const int4 *>(keys)[v];
    const unsigned vb = valid ? ((unsigned)valid[(v * 4) >> 3] >> ((v * 4) & 7)) & 0xFu : 0xFu;
    const int32_t k[4] = {p.x, p.y, p.z, p.w};
    uint64_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w[j] = ((vb >> j) & 1) ? ((uint64_t)((uint32_t)k[j] ^ 0x80000000u) << 32) : kInvalidWord;
    nulls += 4 - __popc(vb);
    ulonglong2 a, b;
    a.x = w[0];
    a.y = w[1];
    b.x = w[2];
    b.y = w[3];
    reinterpret_cast<ulonglong2 *>(words)[2 * v] = a;
    reinterpret_cast<ulonglong2 *>(words)[2 * v + 1] = b;
  }
  for (uint64_t i = nvec * 4 + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const bool ok = bit_valid(valid, i);
    words[i] = ok ? ((uint64_t)((uint32_t)keys[i] ^ 0x80000000u) << 32) : kInvalidWord;
    nulls += ok ? 0 : 1;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) nulls += __shfl_down(nulls, off, 64);
  if (lane_id() == 0 && nulls) atomicAdd(&s_nulls, nulls);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_nulls) atomicAdd((unsigned long long *)&state[NVT_ST_NULLS], s_nulls);
    if (blockIdx.x == 0) atomicAdd((unsigned long long *)&state[NVT_ST_ROWS], (unsigned long long)n);
  }
}

// run heads of the sorted words.  Tile = 4096 consecutive words staged in LDS; thread t owns 16
// consecutive words; tile offsets by a decoupled look-back (ticketed tile ids)
__global__ __launch_bounds__(kScBS) void sc_rle_kernel(const uint64_t *__restrict__ sorted,
                                                       uint64_t n, unsigned *status,
                                                       unsigned *ticket, int32_t *__restrict__ out_keys,
                                                       unsigned *__restrict__ head_pos,
                                                       uint64_t out_cap, uint64_t *state) {
  // key halves of the tile, slot 0 = the word in front of it; one pad word per 32 so that the
  // 16-word runs of neighbouring threads do not start on the same bank
  __shared__ uint32_t hi[kScTile + kScTile / 32 + 2];
  auto at = [](unsigned j) { return j + (j >> 5); };  // j = 1 + position in the tile (0 = word in front)
  __shared__ unsigned wtot[kScBS / kWave];
  __shared__ unsigned s_tile, s_base;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const unsigned tile = s_tile, lane = lane_id(), w = threadIdx.x / kWave;
  const uint64_t base_i = (uint64_t)tile * kScTile;
  const uint64_t ntiles = (n + kScTile - 1) / kScTile;
  for (int r = 0; r < kScRows; ++r) {
    const uint64_t i = base_i + (uint64_t)r * kScBS + threadIdx.x;
    uint64_t v = kInvalidWord;
    if (i < n) v = sorted[i];
    hi[at(1 + r * kScBS + threadIdx.x)] = (uint32_t)(v >> 32);
  }
  if (threadIdx.x == 0) {
    uint64_t pv = kInvalidWord;
    if (base_i > 0) pv = sorted[base_i - 1];
    hi[at(0)] = (uint32_t)(pv >> 32);
  }
  __syncthreads();
  // validity cannot be told from the key half alone (0xFFFFFFFF is the image of INT32_MAX):
  // nulls sort last, so "valid" = index < n_valid, n_valid = rows - nulls (the pack kernel is done)
  const uint64_t n_valid = state[NVT_ST_ROWS] - state[NVT_ST_NULLS];
  unsigned flags = 0, mine = 0;
#pragma unroll
  for (int e = 0; e < kScRows; ++e) {
    const unsigned j = threadIdx.x * kScRows + e;
    const uint64_t i = base_i + j;
    const bool in = i < n_valid;
    const bool head = in && (i == 0 || hi[at(1 + j)] != hi[at(j)]);
    flags |= (head ? 1u : 0u) << e;
    mine += head ? 1u : 0u;
  }
  unsigned inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(inc, off, 64);
    if (lane >= (unsigned)off) inc += o;
  }
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  unsigned wb = 0, tot = 0;
  for (unsigned q = 0; q < kScBS / kWave; ++q) {
    if (q < w) wb += wtot[q];
    tot += wtot[q];
  }
  if (threadIdx.x == 0) {
    unsigned *my = status + tile;
    __hip_atomic_store(my, (tile == 0 ? kScPrefix : kScAgg) | tot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    unsigned excl = 0;
    if (tile > 0) {
      unsigned tb = tile - 1;
      while (true) {
        const unsigned v = __hip_atomic_load(status + tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned f = v >> 30;
        if (f == 0) {
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        excl += v & kScMask;
        if (f == 2) break;
        --tb;
      }
      __hip_atomic_store(my, kScPrefix | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_base = excl;
    if ((uint64_t)tile == ntiles - 1) {
      if ((uint64_t)excl + tot > out_cap) {
        atomicOr((unsigned long long *)&state[NVT_ST_OVERFLOW], 2ull);
        state[NVT_ST_NEED] = (uint64_t)excl + tot;
      } else {
        state[NVT_ST_OCCUPIED] = (uint64_t)excl + tot;
      }
    }
  }
  __syncthreads();
  unsigned rank = s_base + wb + inc - mine;
  if ((uint64_t)s_base + tot > out_cap) return;  // (every later tile sees it too: ranks only grow)
#pragma unroll
  for (int e = 0; e < kScRows; ++e) {
    if ((flags >> e) & 1) {
      const unsigned j = threadIdx.x * kScRows + e;
      if ((uint64_t)rank < out_cap) {
        out_keys[rank] = (int32_t)(hi[at(1 + j)] ^ 0x80000000u);
        head_pos[rank] = (unsigned)(base_i + j);
      }
      ++rank;
    }
  }
}

// counts of the runs + what the ordering pass needs
__global__ __launch_bounds__(kBlock) void sc_counts_kernel(const unsigned *__restrict__ head_pos,
                                                           int64_t *__restrict__ out_cnt,
                                                           unsigned *cls_hist, uint64_t *state) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  if (state[NVT_ST_OVERFLOW] & 2ull) return;
  const uint64_t U = state[NVT_ST_OCCUPIED];
  const uint64_t n_valid = state[NVT_ST_ROWS] - state[NVT_ST_NULLS];
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  unsigned mx = 0, ones = 0;
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < U; r += stride) {
    const uint64_t nxt = r + 1 < U ? (uint64_t)head_pos[r + 1] : n_valid;
    const unsigned c = (unsigned)(nxt - head_pos[r]);
    out_cnt[r] = (int64_t)c;
    mx = c > mx ? c : mx;
    if (c == 1)
      ++ones;
    else
      atomicAdd(&h[c < 255u ? c : 255u], 1u);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ones += __shfl_down(ones, off, 64);
    const unsigned o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if (lane_id() == 0) {
    if (ones) atomicAdd(&h[1], ones);
    if (mx) {
      unsigned long long *gm = reinterpret_cast<unsigned long long *>(&state[NVT_ST_MAXCOUNT]);
      if (mx > __hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(gm, (unsigned long long)mx);
    }
  }
  __syncthreads();
  const unsigned v = h[threadIdx.x];
  if (v) {
    atomicAdd(&cls_hist[threadIdx.x], v);
    if (threadIdx.x == 255) atomicAdd((unsigned long long *)&state[NVT_ST_BIG], (unsigned long long)v);
  }
}

inline uint64_t al256(uint64_t x) { return (x + 255) & ~255ull; }

}  // namespace

uint64_t sort_count_ws_bytes(uint64_t n) {
  const uint64_t ntiles = (n + kScTile - 1) / kScTile;
  return al256(n * 8) + al256(sort_words_tmp_bytes(n)) + al256(ntiles * 4 + 64) + 256;
}

// hist = device uint32[256] (cleared here)
int sort_count_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, void *wsp, unsigned *hist,
                   int32_t *out_keys, int64_t *out_cnt, uint64_t out_cap, uint64_t *state,
                   hipStream_t s) {
  NVT_CHECK_ARG(hist != nullptr, "sort path: the column needs its histogram block (hot_image)");
  NVT_CHECK_ARG(n < (1ull << 30), "sort path: at most 2^30-1 rows per call");
  NVT_PROF("dense_count_s10", n * 4, s);
  char *p = reinterpret_cast<char *>(wsp);
  uint64_t *words = reinterpret_cast<uint64_t *>(p);
  p += al256(n * 8);
  void *sort_tmp = p;
  p += al256(sort_words_tmp_bytes(n));
  const uint64_t ntiles = (n + kScTile - 1) / kScTile;
  unsigned *status = reinterpret_cast<unsigned *>(p);
  unsigned *ticket = status + ntiles;
  NVT_CHECK_HIP(hipMemsetAsync(status, 0, ntiles * 4 + 64, s));
  NVT_CHECK_HIP(hipMemsetAsync(hist, 0, 256 * 4, s));
  sc_pack_kernel<<<stream_grid(n / 4 + 1, kBlock, 8), kBlock, 0, s>>>(keys, valid, n, words, state);
  NVT_CHECK_LAUNCH();
  uint64_t *sorted = nullptr;
  int rc = sort_words_bits(words, n, 32, 64, sort_tmp, &sorted, s);
  if (rc) return rc;
  // the unsorted words are dead after the first pass: their buffer takes the run starts
  // (unless the sort did nothing and returned it: n <= 1)
  unsigned *head_pos = reinterpret_cast<unsigned *>(sorted == words ? sort_tmp : (void *)words);
  sc_rle_kernel<<<(unsigned)ntiles, kScBS, 0, s>>>(sorted, n, status, ticket, out_keys, head_pos,
                                                   out_cap, state);
  NVT_CHECK_LAUNCH();
  sc_counts_kernel<<<stream_grid(n / 2 + 1, kBlock, 8), kBlock, 0, s>>>(head_pos, out_cnt, hist, state);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // namespace nvt
