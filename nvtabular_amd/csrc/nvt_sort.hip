// Vocabulary order: (count descending, key ascending) -- the two sort_values calls of
// categorify.py:1300,1316 with the deterministic tie rule.
//
//  * n <= 8192: one workgroup, bitonic network in LDS over the composite
//    (inverted count, key) -- most Criteo vocabularies are this small and a
//    multi-pass radix sort would be pure launch latency.
//  * larger: LSD radix sort, 8-bit digits over (key bytes, then inverted-count bytes);
//    passes whose digit is constant over the whole array are skipped (the high count
//    bytes almost always are).  Per pass: tile histogram -> scan -> stable scatter.
//    A tile is 2048 elements = 4 waves x 8 rows x 64 lanes; stability inside a tile
//    comes from ballot-matching equal digits in lane order (rank within a row), a
//    per-wave running digit counter in LDS (rows), and a 4-entry prefix over the waves.
#include <type_traits>

#include <cstdlib>
#include <vector>

#include "nvt_common.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"
#include "nvt_range.hpp"
#include "nvt_image.hpp"
#include "nvt_scan.hpp"

extern "C" int nvt_vocab_sort_tmp_bytes(int key_bytes, uint64_t n, uint64_t *bytes);

namespace nvt {

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

// ---- small: single-workgroup bitonic sort -------------------------------------
constexpr int kSmallMax = 8192;
constexpr int kSmallBS = 1024;

template <typename K>
__device__ __forceinline__ bool vocab_before(int64_t ca, K ka, int64_t cb, K kb) {
  return ca > cb || (ca == cb && ka < kb);
}

template <typename K>
__global__ __launch_bounds__(kSmallBS) void sort_small_kernel(K *keys, int64_t *counts, unsigned n) {
  __shared__ K sk[kSmallMax];
  __shared__ int64_t sc[kSmallMax];
  unsigned m = 1;
  while (m < n) m <<= 1;
  for (unsigned i = threadIdx.x; i < m; i += kSmallBS) {
    if (i < n) {
      sk[i] = keys[i];
      sc[i] = counts[i];
    } else {  // padding sorts last: count = INT64_MIN
      sk[i] = std::numeric_limits<K>::max();
      sc[i] = INT64_MIN;
    }
  }
  __syncthreads();
  for (unsigned size = 2; size <= m; size <<= 1) {
    for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
      for (unsigned t = threadIdx.x; t < m / 2; t += kSmallBS) {
        unsigned lo = 2 * t - (t & (stride - 1));
        unsigned hi = lo + stride;
        bool up = (lo & size) == 0;
        K ka = sk[lo], kb = sk[hi];
        int64_t ca = sc[lo], cb = sc[hi];
        bool swap = up ? vocab_before<K>(cb, kb, ca, ka) : vocab_before<K>(ca, ka, cb, kb);
        if (swap) {
          sk[lo] = kb;
          sk[hi] = ka;
          sc[lo] = cb;
          sc[hi] = ca;
        }
      }
      __syncthreads();
    }
  }
  for (unsigned i = threadIdx.x; i < n; i += kSmallBS) {
    keys[i] = sk[i];
    counts[i] = sc[i];
  }
}

// ---- large: LSD radix ------------------------------------------------------------
constexpr int kRows = 8;
constexpr int kTileSort = kBlock * kRows;  // 2048 elements

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

// element index of (wave w, row r, lane l) inside a tile: waves own contiguous 512-element runs
__device__ __forceinline__ uint64_t tile_elem(uint64_t tile, unsigned w, unsigned r, unsigned l) {
  return tile * kTileSort + (uint64_t)w * (kRows * kWave) + (uint64_t)r * kWave + l;
}

template <typename K>
__global__ __launch_bounds__(kBlock) void sort_tile_hist_kernel(const K *__restrict__ keys,
                                                                const int64_t *__restrict__ cnts,
                                                                uint64_t n, int pass,
                                                                unsigned *tile_hist,
                                                                uint64_t ntiles) {
  constexpr int KB = (int)sizeof(K);
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned w = threadIdx.x / kWave, l = lane_id();
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    uint64_t i = tile_elem(blockIdx.x, w, r, l);
    if (i < n) {
      // only the array that holds this pass's digit is read
      unsigned d = pass < KB ? sort_digit<K>(keys[i], 0, pass) : sort_digit<K>((K)0, cnts[i], pass);
      atomicAdd(&h[d], 1u);
    }
  }
  __syncthreads();
  tile_hist[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
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
__global__ __launch_bounds__(kBlock) void sort_scatter_kernel(
    const K *__restrict__ keys, const int64_t *__restrict__ cnts, uint64_t n, int pass,
    const unsigned *__restrict__ tile_off, uint64_t ntiles, K *out_keys, int64_t *out_cnts) {
  __shared__ unsigned wcnt[kBlock / kWave][256];
  const unsigned w = threadIdx.x / kWave, l = lane_id();
  for (int i = threadIdx.x; i < (kBlock / kWave) * 256; i += kBlock) (&wcnt[0][0])[i] = 0;
  __syncthreads();
  K k[kRows];
  int64_t c[kRows];
  unsigned dig[kRows], local[kRows];
  bool act[kRows];
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    uint64_t i = tile_elem(blockIdx.x, w, r, l);
    act[r] = i < n;
    k[r] = act[r] ? keys[i] : (K)0;
    c[r] = act[r] ? cnts[i] : 0;
    dig[r] = sort_digit<K>(k[r], c[r], pass);
  }
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    unsigned long long peers = match_digit(dig[r], act[r]);
    unsigned rank = __popcll(peers & ((1ull << l) - 1ull));
    unsigned before = act[r] ? wcnt[w][dig[r]] : 0;  // digits seen in earlier rows of this wave
    __builtin_amdgcn_wave_barrier();
    if (act[r] && rank == 0) wcnt[w][dig[r]] = before + (unsigned)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    local[r] = before + rank;
  }
  __syncthreads();
  {  // per digit: exclusive prefix over the 4 waves, seeded with the tile's global offset
    const unsigned d = threadIdx.x;
    unsigned run = tile_off[(uint64_t)d * ntiles + blockIdx.x];
#pragma unroll
    for (int q = 0; q < kBlock / kWave; ++q) {
      unsigned t = wcnt[q][d];
      wcnt[q][d] = run;
      run += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    if (act[r]) {
      unsigned dst = wcnt[w][dig[r]] + local[r];
      out_keys[dst] = k[r];
      out_cnts[dst] = c[r];
    }
  }
}

// ---- int32 keys with a known max count: LSD radix over ONE packed 64-bit word -------------
//   comp = (~count32 << 32) | (key ^ 0x80000000)     ascending comp == (count desc, key asc)
// (counts fit 32 bits: n rows < 2^32).  Against the generic path above this moves 8 B per
// entry instead of 12, uses 4096-entry tiles (a quarter of the per-tile histograms to scan)
// and -- the main point -- stages every tile through LDS in sorted order, so each digit's
// run leaves the tile as ONE coalesced burst: with 256 digits a 2048-entry tile wrote
// 8-entry (32 / 64 B) runs straight from registers and the key passes ran at a third of
// the speed of the (nearly sequential) count passes.
// The first pass reads (keys, counts) and packs; the last unpacks into (keys, counts).
constexpr int kS2BS = 256, kS2Rows = 16, kS2Tile = kS2BS * kS2Rows;  // 4096 entries

__device__ __forceinline__ uint64_t comp_make(int32_t key, int64_t cnt) {
  return ((uint64_t)(~(uint32_t)cnt) << 32) | (uint64_t)((uint32_t)key ^ 0x80000000u);
}
__device__ __forceinline__ int32_t comp_key(uint64_t c) { return (int32_t)((uint32_t)c ^ 0x80000000u); }
__device__ __forceinline__ int64_t comp_cnt(uint64_t c) { return (int64_t)(uint32_t)~(uint32_t)(c >> 32); }

// element (wave w, row r, lane l) of a tile: waves own contiguous 1024-element runs (stability)
__device__ __forceinline__ uint64_t s2_elem(uint64_t tile, unsigned w, unsigned r, unsigned l) {
  return tile * kS2Tile + (uint64_t)w * (kS2Rows * kWave) + (uint64_t)r * kWave + l;
}

template <bool FIRST>
__global__ __launch_bounds__(kS2BS) void sort2_hist_kernel(const uint64_t *__restrict__ comp,
                                                           const int32_t *__restrict__ keys,
                                                           const int64_t *__restrict__ cnts,
                                                           uint64_t n, int shift,
                                                           unsigned *tile_hist, uint64_t ntiles) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kS2Tile;
#pragma unroll 4
  for (int r = 0; r < kS2Rows; ++r) {
    const uint64_t i = base + (uint64_t)r * kS2BS + threadIdx.x;  // any order: only counts matter
    const bool act = i < n;
    uint64_t c = 0;
    if (act) c = FIRST ? comp_make(keys[i], cnts[i]) : comp[i];
    const unsigned d = (unsigned)(c >> shift) & 0xFF;
    if (shift < 32) {
      if (act) atomicAdd(&h[d], 1u);  // key bytes: digits spread over 256 bins
    } else {
      // count passes see one or two digit values: aggregate equal digits per wave first
      const unsigned long long peers = match_digit(d, act);
      if (act && (peers & ((1ull << lane_id()) - 1ull)) == 0)
        atomicAdd(&h[d], (unsigned)__popcll(peers));
    }
  }
  __syncthreads();
  tile_hist[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(kS2BS) void sort2_scatter_kernel(
    const uint64_t *__restrict__ comp, const int32_t *__restrict__ keys,
    const int64_t *__restrict__ cnts, uint64_t n, int shift, const unsigned *__restrict__ tile_off,
    const unsigned long long *__restrict__ chunk_base, uint64_t ntiles, uint64_t *out_comp,
    int32_t *out_keys, int64_t *out_cnts) {
  constexpr int NW = kS2BS / kWave;
  __shared__ unsigned wcnt[NW][256];
  __shared__ unsigned goff[256];
  __shared__ unsigned wtot[NW];
  __shared__ uint64_t stage[kS2Tile];
  const unsigned w = threadIdx.x / kWave, l = lane_id();
#pragma unroll
  for (int q = 0; q < NW; ++q) wcnt[q][threadIdx.x] = 0;
  __syncthreads();
  uint64_t c[kS2Rows];
  unsigned short local[kS2Rows];
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    const uint64_t i = s2_elem(blockIdx.x, w, r, l);
    c[r] = ~0ull;
    if (i < n) c[r] = FIRST ? comp_make(keys[i], cnts[i]) : comp[i];
  }
  const uint64_t tile_base = (uint64_t)blockIdx.x * kS2Tile;
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    const bool act = s2_elem(blockIdx.x, w, r, l) < n;
    const unsigned d = (unsigned)(c[r] >> shift) & 0xFF;
    const unsigned long long peers = match_digit(d, act);
    const unsigned rank = __popcll(peers & ((1ull << l) - 1ull));
    const unsigned before = act ? wcnt[w][d] : 0;  // equal digits in earlier rows of this wave
    __builtin_amdgcn_wave_barrier();
    if (act && rank == 0) wcnt[w][d] = before + (unsigned)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    local[r] = (unsigned short)(before + rank);
  }
  __syncthreads();
  {  // thread d: tile-local start of digit d (block exclusive scan) + per-wave bases
    const unsigned d = threadIdx.x;
    unsigned t[NW], tot = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      t[q] = wcnt[q][d];
      tot += t[q];
    }
    unsigned inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o = __shfl_up(inc, off, 64);
      if (l >= (unsigned)off) inc += o;
    }
    if (l == 63) wtot[w] = inc;
    __syncthreads();
    unsigned wbase = 0;
    for (unsigned q = 0; q < w; ++q) wbase += wtot[q];
    const unsigned dstart = wbase + inc - tot;
    unsigned run = dstart;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      wcnt[q][d] = run;
      run += t[q];
    }
    goff[d] = scan_lookup(tile_off, chunk_base, (uint64_t)d * ntiles + blockIdx.x) - dstart;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    if (s2_elem(blockIdx.x, w, r, l) < n) {
      const unsigned d = (unsigned)(c[r] >> shift) & 0xFF;
      stage[wcnt[w][d] + local[r]] = c[r];
    }
  }
  __syncthreads();
  const unsigned tile_n = (unsigned)(n - tile_base < (uint64_t)kS2Tile ? n - tile_base : kS2Tile);
#pragma unroll 4
  for (int j = 0; j < kS2Rows; ++j) {
    const unsigned idx = j * kS2BS + threadIdx.x;
    if (idx < tile_n) {
      const uint64_t v = stage[idx];
      const unsigned d = (unsigned)(v >> shift) & 0xFF;
      const unsigned dst = goff[d] + idx;
      if (LAST) {
        out_keys[dst] = comp_key(v);
        out_cnts[dst] = comp_cnt(v);
      } else {
        out_comp[dst] = v;
      }
    }
  }
}

inline uint64_t pad16(uint64_t x) { return (x + 15) & ~15ull; }

// ---- int32 keys, packed words, ONESWEEP: one histogram read + one scatter launch per pass ---
// The three-launch passes above (tile histogram -> device scan -> scatter) cost ~21-28 launches
// and two extra reads of the array per sort; Criteo's 26 vocabularies spent more time in the
// 7-17 us hist / scan kernels than in the scatters (profiles/r01_final_kernel_stats.csv).  Here:
//   os_hist_kernel    ONE read of (keys, counts): the digit histograms of EVERY pass
//   os_base_kernel    per pass: bucket totals -> exclusive scan = global digit bases
//   os_scatter_kernel per pass: tile ranks as in sort2_scatter_kernel; the tile's offset inside
//                     each digit bucket comes from a decoupled look-back over the preceding
//                     tiles' published (aggregate | inclusive prefix) words instead of a scanned
//                     per-tile histogram.  Tile ids are handed out by an atomic ticket, so a tile
//                     only ever waits for tiles that are already running (no deadlock whatever
//                     the dispatch order).  A status word is flag (2 bits) + count (30 bits):
//                     the data IS the flag (one relaxed agent-scope 4-byte store / load, the
//                     "R2 granule" form of cdna_hip_programming.md G16), so no fences are needed.
constexpr int kOsMaxPass = 8;
constexpr int kOsHistBlocks = 1024;  // 4 per CU: 256 left 4 waves per CU waiting for their loads
constexpr unsigned kOsAgg = 1u << 30, kOsPrefix = 2u << 30, kOsMask = (1u << 30) - 1u;

__global__ __launch_bounds__(kS2BS) void os_hist_kernel(const int32_t *__restrict__ keys,
                                                        const int64_t *__restrict__ cnts,
                                                        uint64_t n, int npass,
                                                        unsigned *__restrict__ block_hist) {
  __shared__ unsigned h[kOsMaxPass * 256];
  for (int i = threadIdx.x; i < npass * 256; i += kS2BS) h[i] = 0;
  __syncthreads();
  const unsigned l = lane_id();
  unsigned c255[kOsMaxPass];  // digit 0xFF of the upper count bytes (nearly every entry) in registers
#pragma unroll
  for (int p = 0; p < kOsMaxPass; ++p) c255[p] = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kS2BS;
  const uint64_t iters = (n + stride - 1) / stride;
  for (uint64_t it = 0; it < iters; ++it) {
    const uint64_t i = it * stride + (uint64_t)blockIdx.x * kS2BS + threadIdx.x;
    const bool act = i < n;
    const uint64_t c = act ? comp_make(keys[i], cnts[i]) : 0ull;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (act) atomicAdd(&h[p * 256 + ((unsigned)(c >> (8 * p)) & 0xFF)], 1u);
    if (npass > 4) {  // lowest count byte: a few hot digits -> aggregate equal digits per wave
      const unsigned d = (unsigned)(c >> 32) & 0xFF;
      const unsigned long long peers = match_digit(d, act);
      if (act && (peers & ((1ull << l) - 1ull)) == 0)
        atomicAdd(&h[4 * 256 + d], (unsigned)__popcll(peers));
    }
#pragma unroll
    for (int p = 5; p < kOsMaxPass; ++p) {
      if (p < npass && act) {
        const unsigned d = (unsigned)(c >> (8 * p)) & 0xFF;
        if (d == 0xFF)
          ++c255[p];
        else
          atomicAdd(&h[p * 256 + d], 1u);
      }
    }
  }
#pragma unroll
  for (int p = 5; p < kOsMaxPass; ++p) {
    if (p < npass) {
      unsigned v = c255[p];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (l == 0 && v) atomicAdd(&h[p * 256 + 0xFF], v);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < npass * 256; i += kS2BS)
    block_hist[(uint64_t)blockIdx.x * (kOsMaxPass * 256) + i] = h[i];
}

// one workgroup per pass: base[p][d] = number of entries whose digit (pass p) is < d
__global__ __launch_bounds__(256) void os_base_kernel(const unsigned *__restrict__ block_hist,
                                                      int nblocks, unsigned *__restrict__ base) {
  __shared__ unsigned wtot[4];
  const int p = blockIdx.x, d = threadIdx.x;
  unsigned tot = 0;
#pragma unroll 8
  for (int b = 0; b < nblocks; ++b) tot += block_hist[(uint64_t)b * (kOsMaxPass * 256) + p * 256 + d];
  unsigned inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned o = __shfl_up(inc, off, 64);
    if (lane_id() >= (unsigned)off) inc += o;
  }
  const unsigned w = threadIdx.x / kWave;
  if (lane_id() == 63) wtot[w] = inc;
  __syncthreads();
  unsigned wb = 0;
  for (unsigned q = 0; q < w; ++q) wb += wtot[q];
  base[p * 256 + d] = wb + inc - tot;
}

// (LOAD: element index -> packed word; the kernels below differ only in where a word comes from)
template <bool LAST, typename LOAD>
__device__ __forceinline__ void os_scatter_body(
    LOAD load, uint64_t n, int shift, const unsigned *__restrict__ base,
    unsigned *status, unsigned *ticket, uint64_t *out_comp, int32_t *out_keys, int64_t *out_cnts) {
  constexpr int NW = kS2BS / kWave;
  __shared__ unsigned wcnt[NW][256];
  __shared__ unsigned goff[256];
  __shared__ unsigned wtot[NW];
  __shared__ unsigned s_tile;
  __shared__ uint64_t stage[kS2Tile];
  const unsigned w = threadIdx.x / kWave, l = lane_id();
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
  for (int q = 0; q < NW; ++q) wcnt[q][threadIdx.x] = 0;
  __syncthreads();
  const unsigned tile = s_tile;
  uint64_t c[kS2Rows];
  unsigned short local[kS2Rows];
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    const uint64_t i = s2_elem(tile, w, r, l);
    c[r] = ~0ull;
    if (i < n) c[r] = load(i);
  }
  const uint64_t tile_base = (uint64_t)tile * kS2Tile;
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    const bool act = s2_elem(tile, w, r, l) < n;
    const unsigned d = (unsigned)(c[r] >> shift) & 0xFF;
    const unsigned long long peers = match_digit(d, act);
    const unsigned rank = __popcll(peers & ((1ull << l) - 1ull));
    const unsigned before = act ? wcnt[w][d] : 0;  // equal digits in earlier rows of this wave
    __builtin_amdgcn_wave_barrier();
    if (act && rank == 0) wcnt[w][d] = before + (unsigned)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    local[r] = (unsigned short)(before + rank);
  }
  __syncthreads();
  {  // thread d: tile count of digit d, tile-local start, look-back for the global offset
    const unsigned d = threadIdx.x;
    unsigned t[NW], tot = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      t[q] = wcnt[q][d];
      tot += t[q];
    }
    // publish this tile's aggregate first: the tiles behind us only need this word
    unsigned *my = status + (uint64_t)tile * 256 + d;
    __hip_atomic_store(my, (tile == 0 ? kOsPrefix : kOsAgg) | tot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    unsigned inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o = __shfl_up(inc, off, 64);
      if (l >= (unsigned)off) inc += o;
    }
    if (l == 63) wtot[w] = inc;
    __syncthreads();
    unsigned wbase = 0;
    for (unsigned q = 0; q < w; ++q) wbase += wtot[q];
    const unsigned dstart = wbase + inc - tot;
    unsigned run = dstart;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      wcnt[q][d] = run;
      run += t[q];
    }
    unsigned excl = 0;
    if (tile > 0) {
      unsigned tb = tile - 1;
      while (true) {
        const unsigned v = __hip_atomic_load(status + (uint64_t)tb * 256 + d, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        const unsigned f = v >> 30;
        if (f == 0) {
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        excl += v & kOsMask;
        if (f == 2) break;
        --tb;  // tile 0 always publishes a prefix: never runs below 0
      }
      __hip_atomic_store(my, kOsPrefix | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    goff[d] = base[d] + excl - dstart;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    if (s2_elem(tile, w, r, l) < n) {
      const unsigned d = (unsigned)(c[r] >> shift) & 0xFF;
      stage[wcnt[w][d] + local[r]] = c[r];
    }
  }
  __syncthreads();
  const unsigned tile_n = (unsigned)(n - tile_base < (uint64_t)kS2Tile ? n - tile_base : kS2Tile);
#pragma unroll 4
  for (int j = 0; j < kS2Rows; ++j) {
    const unsigned idx = j * kS2BS + threadIdx.x;
    if (idx < tile_n) {
      const uint64_t v = stage[idx];
      const unsigned d = (unsigned)(v >> shift) & 0xFF;
      const unsigned dst = goff[d] + idx;
      if (LAST) {
        out_keys[dst] = comp_key(v);
        out_cnts[dst] = comp_cnt(v);
      } else {
        out_comp[dst] = v;
      }
    }
  }
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(kS2BS) void os_scatter_kernel(
    const uint64_t *__restrict__ comp, const int32_t *__restrict__ keys,
    const int64_t *__restrict__ cnts, uint64_t n, int shift, const unsigned *__restrict__ base,
    unsigned *status, unsigned *ticket, uint64_t *out_comp, int32_t *out_keys, int64_t *out_cnts) {
  os_scatter_body<LAST>([=](uint64_t i) -> uint64_t { return FIRST ? comp_make(keys[i], cnts[i]) : comp[i]; },
                        n, shift, base, status, ticket, out_comp, out_keys, out_cnts);
}

// the word of row i of a key column as nvt_sgb_sort sorts it: (order-preserving 32-bit key image
// << 32) | fold << rb | row -- the first pass and the histogram read the COLUMN (4 / 8 + 1 bytes per
// row) instead of a packed copy of it (a pass that wrote 8 bytes per row and two that read them)
template <typename K>
struct PackedKeyWord {
  const K *keys;
  const uint8_t *fold;
  int64_t bias;
  int rb;
  __device__ __forceinline__ uint64_t operator()(uint64_t i) const {
    const uint64_t img = (uint32_t)((uint64_t)(int64_t)keys[i] - (uint64_t)bias);
    const uint64_t f = fold ? (uint64_t)fold[i] : 0ull;
    return (img << 32) | (f << rb) | i;
  }
};
template <typename K>
__global__ __launch_bounds__(kS2BS) void os_scatter_pack_kernel(
    PackedKeyWord<K> src, uint64_t n, int shift, const unsigned *__restrict__ base, unsigned *status,
    unsigned *ticket, uint64_t *out_comp) {
  os_scatter_body<false>(src, n, shift, base, status, ticket, out_comp, nullptr, nullptr);
}

// tmp layout: compA[n] | compB[n] | block_hist | base | status[npass][ntiles][256] | tickets
inline uint64_t os_tmp_bytes(uint64_t n) {
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile;
  return 2 * n * 8 + (uint64_t)kOsHistBlocks * kOsMaxPass * 256 * 4 + kOsMaxPass * 256 * 4 +
         (uint64_t)kOsMaxPass * ntiles * 256 * 4 + kOsMaxPass * 4 + 256;
}

inline int vocab_sort_onesweep(int32_t *keys, int64_t *counts, uint64_t n, int64_t max_count,
                               void *tmp, hipStream_t stream) {
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile;
  char *p = reinterpret_cast<char *>(tmp);
  uint64_t *bufs[2];
  bufs[0] = reinterpret_cast<uint64_t *>(p);
  p += n * 8;
  bufs[1] = reinterpret_cast<uint64_t *>(p);
  p += n * 8;
  unsigned *block_hist = reinterpret_cast<unsigned *>(p);
  p += (uint64_t)kOsHistBlocks * kOsMaxPass * 256 * 4;
  unsigned *base = reinterpret_cast<unsigned *>(p);
  p += kOsMaxPass * 256 * 4;
  unsigned *status = reinterpret_cast<unsigned *>(p);
  int count_bytes = 0;
  for (uint64_t m = (uint64_t)max_count; m; m >>= 8) ++count_bytes;
  const int npass = 4 + count_bytes;  // 4 key bytes, then the live bytes of ~count
  const uint64_t status_words = (uint64_t)npass * ntiles * 256;
  unsigned *tickets = status + status_words;
  NVT_CHECK_HIP(hipMemsetAsync(status, 0, (status_words + kOsMaxPass) * 4, stream));
  const unsigned hb = (unsigned)(ntiles < (uint64_t)kOsHistBlocks ? ntiles : kOsHistBlocks);
  os_hist_kernel<<<hb, kS2BS, 0, stream>>>(keys, counts, n, npass, block_hist);
  NVT_CHECK_LAUNCH();
  os_base_kernel<<<npass, 256, 0, stream>>>(block_hist, (int)hb, base);
  NVT_CHECK_LAUNCH();
  const uint64_t *src = nullptr;
  int flip = 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int shift = 8 * pass;
    const bool first = pass == 0, last = pass == npass - 1;
    unsigned *st = status + (uint64_t)pass * ntiles * 256;
    uint64_t *dst = bufs[flip];
    if (first)
      os_scatter_kernel<true, false><<<(unsigned)ntiles, kS2BS, 0, stream>>>(
          nullptr, keys, counts, n, shift, base + pass * 256, st, tickets + pass, dst, nullptr,
          nullptr);
    else if (last)
      os_scatter_kernel<false, true><<<(unsigned)ntiles, kS2BS, 0, stream>>>(
          src, nullptr, nullptr, n, shift, base + pass * 256, st, tickets + pass, nullptr, keys,
          counts);
    else
      os_scatter_kernel<false, false><<<(unsigned)ntiles, kS2BS, 0, stream>>>(
          src, nullptr, nullptr, n, shift, base + pass * 256, st, tickets + pass, dst, nullptr,
          nullptr);
    NVT_CHECK_LAUNCH();
    src = dst;
    flip ^= 1;
  }
  return NVT_OK;
}

// ---- generic: stable LSD radix sort of packed 64-bit words on the bit range [bit_lo, bit_hi) ----
// (the onesweep scatter above with FIRST = LAST = false).  Used by the multi-key groupby update:
// words are (slot << 32 | row), sorted by slot, so that every group's rows become one run in
// row order.
template <typename LOAD>
__device__ __forceinline__ void os_hist_words_body(LOAD load, uint64_t n, int bit_lo, int npass,
                                                   unsigned *__restrict__ block_hist) {
  __shared__ unsigned h[kOsMaxPass * 256];
  for (int i = threadIdx.x; i < npass * 256; i += kS2BS) h[i] = 0;
  __syncthreads();
  const unsigned l = lane_id();
  const uint64_t stride = (uint64_t)gridDim.x * kS2BS;
  const uint64_t iters = (n + stride - 1) / stride;
  constexpr int U = 4;  // loads in flight per thread
  for (uint64_t it = 0; it < iters; it += U) {
    uint64_t cw[U];
    bool av[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t i = (it + u) * stride + (uint64_t)blockIdx.x * kS2BS + threadIdx.x;
      av[u] = (it + u) < iters && i < n;
      cw[u] = av[u] ? load(i) >> bit_lo : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool act = av[u];
      const uint64_t c = cw[u];
#pragma unroll
      for (int p = 0; p < kOsMaxPass; ++p) {
        if (p < npass) {
          // sorted-by-group inputs are heavily skewed (one hot group = one digit value): a wave
          // whose 64 digits are all equal adds once; otherwise plain LDS atomics (a full
          // match-any aggregation per digit cost more than the conflicts it saved)
          const unsigned d = (unsigned)(c >> (8 * p)) & 0xFF;
          const unsigned long long am = __ballot(act);
          const unsigned d0 = __builtin_amdgcn_readfirstlane(d);
          if (am == ~0ull && __ballot(d == d0) == ~0ull) {
            if (l == 0) atomicAdd(&h[p * 256 + d0], 64u);
          } else if (act) {
            atomicAdd(&h[p * 256 + d], 1u);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < npass * 256; i += kS2BS)
    block_hist[(uint64_t)blockIdx.x * (kOsMaxPass * 256) + i] = h[i];
}

__global__ __launch_bounds__(kS2BS) void os_hist_words_kernel(const uint64_t *__restrict__ w,
                                                              uint64_t n, int bit_lo, int npass,
                                                              unsigned *__restrict__ block_hist) {
  os_hist_words_body([=](uint64_t i) -> uint64_t { return w[i]; }, n, bit_lo, npass, block_hist);
}
template <typename K>
__global__ __launch_bounds__(kS2BS) void os_hist_pack_kernel(PackedKeyWord<K> src, uint64_t n, int bit_lo,
                                                             int npass, unsigned *__restrict__ block_hist) {
  os_hist_words_body(src, n, bit_lo, npass, block_hist);
}

uint64_t sort_words_tmp_bytes(uint64_t n) { return os_tmp_bytes(n); }

// Sorts data[0..n) by bits [bit_lo, bit_hi) (stable).  *result = data or a buffer inside tmp.
// keys != nullptr: the words are the packed rows of a key column (PackedKeyWord: key image, fold,
// row) that nobody has written out -- the histogram and the first pass read the column.
static int sort_words_impl(uint64_t *data, const void *keys, int key_dtype, int64_t key_bias,
                           const uint8_t *fold, int rb, uint64_t n, int bit_lo, int bit_hi, void *tmp,
                           uint64_t **result, hipStream_t stream) {
  NVT_CHECK_ARG(n < (1ull << 30), "at most 2^30-1 words");
  const int npass = (bit_hi - bit_lo + 7) / 8;
  NVT_CHECK_ARG(npass >= 1 && npass <= kOsMaxPass, "bit range too wide");
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile;
  char *p = reinterpret_cast<char *>(tmp);
  uint64_t *bufs[2];
  bufs[0] = reinterpret_cast<uint64_t *>(p);
  p += n * 8;
  bufs[1] = reinterpret_cast<uint64_t *>(p);
  p += n * 8;
  unsigned *block_hist = reinterpret_cast<unsigned *>(p);
  p += (uint64_t)kOsHistBlocks * kOsMaxPass * 256 * 4;
  unsigned *base = reinterpret_cast<unsigned *>(p);
  p += kOsMaxPass * 256 * 4;
  unsigned *status = reinterpret_cast<unsigned *>(p);
  const uint64_t status_words = (uint64_t)npass * ntiles * 256;
  unsigned *tickets = status + status_words;
  NVT_CHECK_HIP(hipMemsetAsync(status, 0, (status_words + kOsMaxPass) * 4, stream));
  const unsigned hb = (unsigned)(ntiles < (uint64_t)kOsHistBlocks ? ntiles : kOsHistBlocks);
  const PackedKeyWord<int32_t> s32{reinterpret_cast<const int32_t *>(keys), fold, key_bias, rb};
  const PackedKeyWord<int64_t> s64{reinterpret_cast<const int64_t *>(keys), fold, key_bias, rb};
  if (!keys)
    os_hist_words_kernel<<<hb, kS2BS, 0, stream>>>(data, n, bit_lo, npass, block_hist);
  else if (key_dtype == NVT_I32)
    os_hist_pack_kernel<int32_t><<<hb, kS2BS, 0, stream>>>(s32, n, bit_lo, npass, block_hist);
  else
    os_hist_pack_kernel<int64_t><<<hb, kS2BS, 0, stream>>>(s64, n, bit_lo, npass, block_hist);
  NVT_CHECK_LAUNCH();
  os_base_kernel<<<npass, 256, 0, stream>>>(block_hist, (int)hb, base);
  NVT_CHECK_LAUNCH();
  const uint64_t *src = data;
  int flip = 0;
  for (int pass = 0; pass < npass; ++pass) {
    uint64_t *dst = bufs[flip];
    unsigned *st = status + (uint64_t)pass * ntiles * 256;
    if (pass == 0 && keys && key_dtype == NVT_I32)
      os_scatter_pack_kernel<int32_t><<<(unsigned)ntiles, kS2BS, 0, stream>>>(
          s32, n, bit_lo, base, st, tickets, dst);
    else if (pass == 0 && keys)
      os_scatter_pack_kernel<int64_t><<<(unsigned)ntiles, kS2BS, 0, stream>>>(
          s64, n, bit_lo, base, st, tickets, dst);
    else
      os_scatter_kernel<false, false><<<(unsigned)ntiles, kS2BS, 0, stream>>>(
          src, nullptr, nullptr, n, bit_lo + 8 * pass, base + pass * 256, st, tickets + pass, dst,
          nullptr, nullptr);
    NVT_CHECK_LAUNCH();
    src = dst;
    flip ^= 1;
  }
  *result = const_cast<uint64_t *>(src);
  return NVT_OK;
}

int sort_words_bits(uint64_t *data, uint64_t n, int bit_lo, int bit_hi, void *tmp, uint64_t **result,
                    hipStream_t stream) {
  *result = data;
  if (n <= 1 || bit_hi <= bit_lo) return NVT_OK;
  return sort_words_impl(data, nullptr, 0, 0, nullptr, 0, n, bit_lo, bit_hi, tmp, result, stream);
}

// The packed rows of a key column ((key - bias) image << 32 | fold << rb | row), sorted by
// bits [rb, 64) (key, then fold; stable): *result = a buffer inside tmp (sort_words_tmp_bytes(n)).
int sort_packed_keys(const void *keys, int key_dtype, int64_t key_bias, const uint8_t *fold, int rb,
                     uint64_t n, void *tmp, uint64_t **result, hipStream_t stream) {
  NVT_CHECK_ARG(keys && n >= 1 && rb >= 24 && rb <= 32, "keys / rows / row bits");
  return sort_words_impl(nullptr, keys, key_dtype, key_bias, fold, rb, n, rb, 64, tmp, result, stream);
}

// ---- all small vocabularies of a fit in ONE launch: workgroup b sorts vocabulary b --------
// packed words in LDS (128 KiB for up to 16384 entries), bitonic network, ascending.
constexpr int kSmallPackedMax = 16384;
struct SmallCol {
  int32_t *keys;
  int64_t *counts;
  unsigned n;
};
constexpr int kSmallBatch = 64;
struct SmallBatch {
  SmallCol c[kSmallBatch];
};
__global__ __launch_bounds__(kSmallBS) void sort_small_packed_many_kernel(SmallBatch b) {
  __shared__ uint64_t sv[kSmallPackedMax];
  const SmallCol col = b.c[blockIdx.x];
  const unsigned n = col.n;
  unsigned m = 1;
  while (m < n) m <<= 1;
  for (unsigned i = threadIdx.x; i < m; i += kSmallBS)
    sv[i] = i < n ? comp_make(col.keys[i], col.counts[i]) : ~0ull;  // padding sorts last
  __syncthreads();
  for (unsigned size = 2; size <= m; size <<= 1) {
    for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
      for (unsigned t = threadIdx.x; t < m / 2; t += kSmallBS) {
        const unsigned lo = 2 * t - (t & (stride - 1));
        const unsigned hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint64_t a = sv[lo], c = sv[hi];
        if (up ? (c < a) : (a < c)) {
          sv[lo] = c;
          sv[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (unsigned i = threadIdx.x; i < n; i += kSmallBS) {
    const uint64_t v = sv[i];
    col.keys[i] = comp_key(v);
    col.counts[i] = comp_cnt(v);
  }
}

// tmp layout of the packed path: compA[n] | compB[n] | tile_hist | chunk_tot
inline uint64_t sort2_tmp_bytes(uint64_t n) {
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile, hist_len = 256 * ntiles;
  return 2 * n * 8 + pad16(hist_len * 4) + scan_chunks(hist_len) * 8 + 64;
}

inline int vocab_sort_packed(int32_t *keys, int64_t *counts, uint64_t n, int64_t max_count,
                             void *tmp, hipStream_t stream) {
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile, hist_len = 256 * ntiles;
  char *p = reinterpret_cast<char *>(tmp);
  uint64_t *bufs[2];
  bufs[0] = reinterpret_cast<uint64_t *>(p);
  p += n * 8;
  bufs[1] = reinterpret_cast<uint64_t *>(p);
  p += n * 8;
  unsigned *tile_hist = reinterpret_cast<unsigned *>(p);
  p += pad16(hist_len * 4);
  unsigned long long *chunk_tot = reinterpret_cast<unsigned long long *>(p);
  int count_bytes = 0;
  for (uint64_t m = (uint64_t)max_count; m; m >>= 8) ++count_bytes;
  const int npass = 4 + count_bytes;  // 4 key bytes, then the live bytes of ~count
  const uint64_t *src = nullptr;
  int flip = 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int shift = 8 * pass;
    const bool first = pass == 0, last = pass == npass - 1;
    if (first)
      sort2_hist_kernel<true><<<(unsigned)ntiles, kS2BS, 0, stream>>>(nullptr, keys, counts, n,
                                                                       shift, tile_hist, ntiles);
    else
      sort2_hist_kernel<false><<<(unsigned)ntiles, kS2BS, 0, stream>>>(src, nullptr, nullptr, n,
                                                                        shift, tile_hist, ntiles);
    NVT_CHECK_LAUNCH();
    const unsigned long long *cbase = nullptr;
    {
      int rc = exclusive_scan_u32_deferred(tile_hist, hist_len, chunk_tot, &cbase, stream);
      if (rc) return rc;
    }
    uint64_t *dst = bufs[flip];
    if (first)
      sort2_scatter_kernel<true, false><<<(unsigned)ntiles, kS2BS, 0, stream>>>(
          nullptr, keys, counts, n, shift, tile_hist, cbase, ntiles, dst, nullptr, nullptr);
    else if (last)
      sort2_scatter_kernel<false, true><<<(unsigned)ntiles, kS2BS, 0, stream>>>(
          src, nullptr, nullptr, n, shift, tile_hist, cbase, ntiles, nullptr, keys, counts);
    else
      sort2_scatter_kernel<false, false><<<(unsigned)ntiles, kS2BS, 0, stream>>>(
          src, nullptr, nullptr, n, shift, tile_hist, cbase, ntiles, dst, nullptr, nullptr);
    NVT_CHECK_LAUNCH();
    src = dst;
    flip ^= 1;
  }
  return NVT_OK;
}

template <typename K>
int vocab_sort(K *keys, int64_t *counts, uint64_t n, int64_t max_count, void *tmp,
               hipStream_t stream) {
  constexpr int NP = (int)sizeof(K) + 8;
  if (n <= 1) return NVT_OK;
  NVT_CHECK_ARG(n < (1ull << 32), "at most 2^32-1 vocabulary entries");
  if (n <= kSmallMax) {
    sort_small_kernel<K><<<1, kSmallBS, 0, stream>>>(keys, counts, (unsigned)n);
    NVT_CHECK_LAUNCH();
    return NVT_OK;
  }
  if constexpr (sizeof(K) == 4) {
    if (max_count > 0 && max_count < (1ll << 32) && n < (1ull << 30) && !ab_env("NVT_SORT_LEGACY"))
      return vocab_sort_onesweep(keys, counts, n, max_count, tmp, stream);
    if (max_count > 0 && max_count < (1ll << 32) && n < (1ull << 31))
      return vocab_sort_packed(keys, counts, n, max_count, tmp, stream);
  }
  const uint64_t ntiles = (n + kTileSort - 1) / kTileSort;
  // tmp layout: counts2 | keys2 | tile_hist | chunk_tot | pass_hist
  char *p = reinterpret_cast<char *>(tmp);
  int64_t *counts2 = reinterpret_cast<int64_t *>(p);
  p += n * sizeof(int64_t);
  K *keys2 = reinterpret_cast<K *>(p);
  p += pad16(n * sizeof(K));
  unsigned *tile_hist = reinterpret_cast<unsigned *>(p);
  const uint64_t hist_len = 256 * ntiles;
  p += pad16(hist_len * sizeof(unsigned));
  const uint64_t nchunks = (hist_len + kScanChunk - 1) / kScanChunk;
  unsigned long long *chunk_tot = reinterpret_cast<unsigned long long *>(p);
  p += nchunks * sizeof(unsigned long long);
  unsigned long long *pass_hist = reinterpret_cast<unsigned long long *>(p);

  bool run_pass[NP];
  if (max_count > 0) {
    // counts lie in [0, max_count]: the inverted-count bytes above its top byte are all 0xFF
    constexpr int KB = (int)sizeof(K);
    int count_bytes = 0;
    for (uint64_t m = (uint64_t)max_count; m; m >>= 8) ++count_bytes;
    for (int p = 0; p < NP; ++p) run_pass[p] = p < KB || (p - KB) < count_bytes;
  } else {
    NVT_CHECK_HIP(hipMemsetAsync(pass_hist, 0, NP * 256 * sizeof(unsigned long long), stream));
    sort_pass_hist_kernel<K><<<stream_grid(n, kBlock * 8, 4), kBlock, 0, stream>>>(keys, counts, n,
                                                                                     pass_hist);
    NVT_CHECK_LAUNCH();
    unsigned long long host_hist[NP * 256];
    NVT_CHECK_HIP(hipMemcpyAsync(host_hist, pass_hist, sizeof(host_hist), hipMemcpyDeviceToHost,
                                 stream));
    NVT_CHECK_HIP(hipStreamSynchronize(stream));
    for (int p = 0; p < NP; ++p) {
      run_pass[p] = true;
      for (int d = 0; d < 256; ++d)
        if (host_hist[p * 256 + d] == n) run_pass[p] = false;
    }
  }

  K *src_k = keys, *dst_k = keys2;
  int64_t *src_c = counts, *dst_c = counts2;
  for (int pass = 0; pass < NP; ++pass) {
    if (!run_pass[pass]) continue;
    sort_tile_hist_kernel<K><<<(unsigned)ntiles, kBlock, 0, stream>>>(src_k, src_c, n, pass,
                                                                      tile_hist, ntiles);
    NVT_CHECK_LAUNCH();
    {
      int rc = exclusive_scan_u32(tile_hist, hist_len, chunk_tot, stream);
      if (rc) return rc;
    }
    sort_scatter_kernel<K><<<(unsigned)ntiles, kBlock, 0, stream>>>(src_k, src_c, n, pass,
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

int vocab_sort_any(int key_bytes, void *keys, int64_t *counts, uint64_t n, int64_t max_count,
                   void *tmp, hipStream_t s) {
  NVT_PROF("vocab_sort", 0, s);
  if (key_bytes == 4) return vocab_sort<int32_t>((int32_t *)keys, counts, n, max_count, tmp, s);
  return vocab_sort<int64_t>((int64_t *)keys, counts, n, max_count, tmp, s);
}

bool vocab_sort_small_eligible(int key_bytes, uint64_t n, int64_t max_count) {
  return key_bytes == 4 && n >= 2 && n <= (uint64_t)kSmallPackedMax && max_count > 0 &&
         max_count < (1ll << 32);
}

int vocab_sort_small_batch(const SmallSortDesc *cols, int ncols, hipStream_t s) {
  for (int c0 = 0; c0 < ncols; c0 += kSmallBatch) {
    const int nc = ncols - c0 < kSmallBatch ? ncols - c0 : kSmallBatch;
    SmallBatch b;
    memset(&b, 0, sizeof(b));
    for (int i = 0; i < nc; ++i) {
      b.c[i].keys = cols[c0 + i].keys;
      b.c[i].counts = cols[c0 + i].counts;
      b.c[i].n = cols[c0 + i].n;
    }
    NVT_PROF("vocab_sort_small", 0, s);
    sort_small_packed_many_kernel<<<nc, kSmallBS, 0, s>>>(b);
    NVT_CHECK_LAUNCH();
  }
  return NVT_OK;
}

// ---- vocabulary order from a KEY-SORTED (key, count) list: ONE stable counting pass ---------
// The range path of the counting stage (nvt_range_count.hip) emits its list in key order and a
// histogram of cls = min(count, 255).  "count descending, key ascending" (categorify.py:1300,
// 1316) is then: class 255 (count >= 255; a few thousand entries of a 45 M-row power-law
// column, never more than rows / 255) in front, then classes 254 .. 1, every class in the key
// order it already has.  One stable scatter by class does that for all but the first class,
// whose entries are sorted afterwards by the (small) generic sort; the encode table is filled
// by the same scatter, where every entry learns its label.  Against the 7-pass radix sort +
// separate table build: 12 B read + 12 B written per entry instead of ~120, 4 launches
// instead of 11.
// Same tile geometry, ballot ranking and decoupled look-back as os_scatter_kernel above.
__device__ __forceinline__ unsigned cls_digit(uint64_t comp) {
  const uint32_t cnt = ~(uint32_t)(comp >> 32);
  return 255u - (cnt < 255u ? cnt : 255u);
}

__device__ __forceinline__ void cls_scatter_body(
    const int32_t *__restrict__ keys, const int64_t *__restrict__ cnts, uint64_t n,
    const unsigned *__restrict__ cls_hist, unsigned *status, unsigned *ticket, int32_t *out_keys,
    int64_t *out_cnts, unsigned long long *table, uint64_t mask, int64_t first_label,
    int64_t *sentinel_label, int32_t *label_of, int32_t *big_src = nullptr) {
  // big_src set (a SHARD of a list that several ranks own, nvt_vocab_label_shard): only the
  // entries of class 255 are written out (compacted in key order at the front of out_*, with
  // their positions in the shard), every other entry only gets its label
  constexpr int NW = kS2BS / kWave;
  __shared__ unsigned wcnt[NW][256];
  __shared__ unsigned goff[256];
  __shared__ unsigned wtot[NW], btot[NW];
  __shared__ unsigned s_tile;
  __shared__ uint64_t stage[kS2Tile];
  const unsigned w = threadIdx.x / kWave, l = lane_id();
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
  for (int q = 0; q < NW; ++q) wcnt[q][threadIdx.x] = 0;
  // class bases: digit d = 255 - cls, base[d] = entries of the classes in front of it
  unsigned cbase;
  {
    const unsigned v = cls_hist[255 - threadIdx.x];
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o = __shfl_up(inc, off, 64);
      if (l >= (unsigned)off) inc += o;
    }
    if (l == 63) btot[w] = inc;
    __syncthreads();
    unsigned wb = 0;
    for (unsigned q = 0; q < w; ++q) wb += btot[q];
    cbase = wb + inc - v;
  }
  const unsigned tile = s_tile;
  uint64_t c[kS2Rows];
  unsigned short local[kS2Rows];
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    const uint64_t i = s2_elem(tile, w, r, l);
    c[r] = ~0ull;
    if (i < n) c[r] = comp_make(keys[i], cnts[i]);
  }
  const uint64_t tile_base = (uint64_t)tile * kS2Tile;
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    const bool act = s2_elem(tile, w, r, l) < n;
    const unsigned d = cls_digit(c[r]);
    const unsigned long long peers = match_digit(d, act);
    const unsigned rank = __popcll(peers & ((1ull << l) - 1ull));
    const unsigned before = act ? wcnt[w][d] : 0;
    __builtin_amdgcn_wave_barrier();
    if (act && rank == 0) wcnt[w][d] = before + (unsigned)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    local[r] = (unsigned short)(before + rank);
  }
  __syncthreads();
  {
    const unsigned d = threadIdx.x;
    unsigned t[NW], tot = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      t[q] = wcnt[q][d];
      tot += t[q];
    }
    unsigned *my = status + (uint64_t)tile * 256 + d;
    __hip_atomic_store(my, (tile == 0 ? kOsPrefix : kOsAgg) | tot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    unsigned inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o = __shfl_up(inc, off, 64);
      if (l >= (unsigned)off) inc += o;
    }
    if (l == 63) wtot[w] = inc;
    __syncthreads();
    unsigned wbase = 0;
    for (unsigned q = 0; q < w; ++q) wbase += wtot[q];
    const unsigned dstart = wbase + inc - tot;
    unsigned run = dstart;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      wcnt[q][d] = run;
      run += t[q];
    }
    unsigned excl = 0;
    if (tile > 0) {
      unsigned tb = tile - 1;
      while (true) {
        const unsigned v = __hip_atomic_load(status + (uint64_t)tb * 256 + d, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        const unsigned f = v >> 30;
        if (f == 0) {
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        excl += v & kOsMask;
        if (f == 2) break;
        --tb;
      }
      __hip_atomic_store(my, kOsPrefix | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    goff[d] = cbase + excl - dstart;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    const uint64_t i = s2_elem(tile, w, r, l);
    if (i < n) {
      const unsigned d = cls_digit(c[r]);
      const unsigned sidx = wcnt[w][d] + local[r];
      stage[sidx] = c[r];
      // range table: label of the entry at position i of the key-ordered list (class 255 is
      // labelled after its own sort: -1 here)
      if (label_of != nullptr) label_of[i] = d != 0 ? (int32_t)(first_label + goff[d] + sidx) : -1;
      if (big_src != nullptr && d == 0) big_src[goff[0] + sidx] = (int32_t)i;
    }
  }
  __syncthreads();
  const unsigned tile_n = (unsigned)(n - tile_base < (uint64_t)kS2Tile ? n - tile_base : kS2Tile);
#pragma unroll 4
  for (int j = 0; j < kS2Rows; ++j) {
    const unsigned idx = j * kS2BS + threadIdx.x;
    if (idx < tile_n) {
      const uint64_t v = stage[idx];
      const unsigned d = cls_digit(v);
      if (big_src != nullptr && d != 0) continue;
      const unsigned dst = goff[d] + idx;
      const int32_t key = comp_key(v);
      out_keys[dst] = key;
      out_cnts[dst] = comp_cnt(v);
      if (key == INT32_MIN && d != 0 && sentinel_label != nullptr) {
        *sentinel_label = first_label + (int64_t)dst;
      } else if (table != nullptr && d != 0) {  // class 255 gets its labels after its own sort
        const int64_t label = first_label + (int64_t)dst;
        {
          const unsigned long long want = ((unsigned long long)(uint32_t)label << 32) | (uint32_t)key;
          uint64_t slot = (uint64_t)slot_hash(key) & mask;
          while (atomicCAS(&table[slot], kEncEmptySlot, want) != kEncEmptySlot) slot = (slot + 1) & mask;
        }
      }
    }
  }
}

__global__ __launch_bounds__(kS2BS) void cls_scatter_kernel(
    const int32_t *__restrict__ keys, const int64_t *__restrict__ cnts, uint64_t n,
    const unsigned *__restrict__ cls_hist, unsigned *status, unsigned *ticket, int32_t *out_keys,
    int64_t *out_cnts, unsigned long long *table, uint64_t mask, int64_t first_label,
    int64_t *sentinel_label, int32_t *label_of) {
  cls_scatter_body(keys, cnts, n, cls_hist, status, ticket, out_keys, out_cnts, table, mask,
                   first_label, sentinel_label, label_of);
}

// ---- the same ordering for SEVERAL vocabularies per launch --------------------------------
// A Criteo fit orders 13 key-sorted vocabularies; one launch chain per vocabulary (memsets,
// scatter, patch / build: ~10 launches each) kept the HOST busy for as long as the kernels ran
// (~130 launches, 1.0 ms of a 12 ms step).  Here every stage is ONE launch for all vocabularies
// of the call: the tiles of all lists form one grid (a block finds its vocabulary in a prefix
// table of 16 entries), streaming stages use blockIdx.y = vocabulary.
constexpr int kOrdBatch = 16;
struct OrdJob {
  const int32_t *keys;
  const int64_t *cnts;
  const unsigned *cls_hist;
  unsigned *status, *ticket;
  int32_t *out_keys;
  int64_t *out_cnts;
  int32_t *label_of;
  unsigned long long *table;
  int64_t *sentinel_label;
  int32_t *aux;
  unsigned long long *fb_status;
  unsigned long long n, capacity, nslots, flat_slots, first_label, status_words, fb_words;
};
struct OrdBatch {
  OrdJob j[kOrdBatch];
  unsigned tile_start[kOrdBatch + 1];
  unsigned flat_start[kOrdBatch + 1];
  int njobs;
};
__device__ __forceinline__ int ord_job_of(const unsigned *start, int n, unsigned b) {
  int c = 0;
  while (c + 1 < n && b >= start[c + 1]) ++c;
  return c;
}

__global__ __launch_bounds__(kBlock) void ord_prep_kernel(OrdBatch b) {
  const OrdJob &j = b.j[blockIdx.y];
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  const uint64_t t0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  for (uint64_t i = t0; i < j.status_words; i += stride) j.status[i] = 0;   // + the ticket word
  for (uint64_t i = t0; i < j.fb_words; i += stride) j.fb_status[i] = 0;
  if (j.flat_slots) {  // flat table: every slot empty before the build
    for (uint64_t i = t0; i < j.capacity; i += stride) j.table[i] = kEncEmptySlot;
  }
  if (t0 == 0) *j.sentinel_label = -1;
}

__global__ __launch_bounds__(kS2BS) void cls_scatter_many_kernel(OrdBatch b) {
  const int ji = ord_job_of(b.tile_start, b.njobs, blockIdx.x);
  const OrdJob &j = b.j[ji];
  cls_scatter_body(j.keys, j.cnts, j.n, j.cls_hist, j.status, j.ticket, j.out_keys, j.out_cnts,
                   nullptr, 0, (int64_t)j.first_label, j.sentinel_label, j.label_of);
}

// Range table (dumped by the counting pass: slot = {key, position in the key-ordered list}):
// positions -> labels.  One streaming pass: the slots are in key order, so label_of[] is read
// front to back as well.
__global__ __launch_bounds__(kBlock) void range_patch_kernel(unsigned long long *table,
                                                             uint64_t nslots,
                                                             const int32_t *__restrict__ label_of) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock * 2;
  for (uint64_t s0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 2; s0 < nslots; s0 += stride) {
    ulonglong2 e = *reinterpret_cast<ulonglong2 *>(table + s0);  // nslots is even, 16-byte aligned
    bool dirty = false;
    if ((int32_t)(uint32_t)e.x != INT32_MIN) {
      e.x = ((unsigned long long)(uint32_t)label_of[(uint32_t)(e.x >> 32)] << 32) | (uint32_t)e.x;
      dirty = true;
    }
    if ((int32_t)(uint32_t)e.y != INT32_MIN) {
      e.y = ((unsigned long long)(uint32_t)label_of[(uint32_t)(e.y >> 32)] << 32) | (uint32_t)e.y;
      dirty = true;
    }
    if (dirty) *reinterpret_cast<ulonglong2 *>(table + s0) = e;
  }
}

// labels of the (few) entries of class 255 after their own sort: vocab[j] -> first_label + j
__global__ __launch_bounds__(kBlock) void range_fix_prefix_kernel(
    unsigned long long *table, const int32_t *__restrict__ aux, const int32_t *__restrict__ vocab,
    uint64_t n_big, int64_t first_label, int64_t *sentinel_label, uint64_t table_slots) {
  const RangeMap map = load_map(aux);
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n_big; j += stride) {
    const int32_t key = vocab[j];
    const int64_t label = first_label + (int64_t)j;
    if (key == INT32_MIN) {
      *sentinel_label = label;
      continue;
    }
    uint64_t s = map.table_slot(key);
    if (map.flat) {  // runs in key order: bounded search (keys that cluster in their range)
      s = flat_find_from(table, table_slots, s, key, table[s]);
      if (s != ~0ull) table[s] = ((unsigned long long)(uint32_t)label << 32) | (uint32_t)key;
      continue;
    }
    while (true) {
      const unsigned long long e = table[s];
      if ((int32_t)(uint32_t)e == key) {
        table[s] = ((unsigned long long)(uint32_t)label << 32) | (uint32_t)key;
        break;
      }
      if ((int32_t)(uint32_t)e == INT32_MIN) break;  // cannot happen for a key of the vocabulary
      ++s;
    }
  }
}

__global__ __launch_bounds__(kBlock) void range_patch_many_kernel(OrdBatch b) {
  const OrdJob &j = b.j[blockIdx.y];
  if (j.flat_slots || j.nslots == 0) return;
  unsigned long long *table = j.table;
  const int32_t *__restrict__ label_of = j.label_of;
  const uint64_t nslots = j.nslots;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock * 2;
  for (uint64_t s0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 2; s0 < nslots; s0 += stride) {
    ulonglong2 e = *reinterpret_cast<ulonglong2 *>(table + s0);
    bool dirty = false;
    if ((int32_t)(uint32_t)e.x != INT32_MIN) {
      e.x = ((unsigned long long)(uint32_t)label_of[(uint32_t)(e.x >> 32)] << 32) | (uint32_t)e.x;
      dirty = true;
    }
    if ((int32_t)(uint32_t)e.y != INT32_MIN) {
      e.y = ((unsigned long long)(uint32_t)label_of[(uint32_t)(e.y >> 32)] << 32) | (uint32_t)e.y;
      dirty = true;
    }
    if (dirty) *reinterpret_cast<ulonglong2 *>(table + s0) = e;
  }
}

struct FixJob {
  unsigned long long *table;
  const int32_t *aux, *vocab;
  int64_t *sentinel_label;
  unsigned long long n_big, first_label, table_slots;
};
struct FixBatch {
  FixJob j[kOrdBatch];
};
__global__ __launch_bounds__(kBlock) void range_fix_prefix_many_kernel(FixBatch b) {
  const FixJob &f = b.j[blockIdx.y];
  const RangeMap map = load_map(f.aux);
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < f.n_big; j += stride) {
    const int32_t key = f.vocab[j];
    const int64_t label = (int64_t)f.first_label + (int64_t)j;
    if (key == INT32_MIN) {
      *f.sentinel_label = label;
      continue;
    }
    uint64_t s = map.table_slot(key);
    if (map.flat) {
      s = flat_find_from(f.table, f.table_slots, s, key, f.table[s]);
      if (s != ~0ull) f.table[s] = ((unsigned long long)(uint32_t)label << 32) | (uint32_t)key;
      continue;
    }
    while (true) {
      const unsigned long long e = f.table[s];
      if ((int32_t)(uint32_t)e == key) {
        f.table[s] = ((unsigned long long)(uint32_t)label << 32) | (uint32_t)key;
        break;
      }
      if ((int32_t)(uint32_t)e == INT32_MIN) break;
      ++s;
    }
  }
}

// histogram of min(count, 255) of a (key, count) list that did not come from the range path (the
// multi-GPU merge gathers key-sorted owner shards): what cls_scatter_kernel needs
__global__ __launch_bounds__(kBlock) void class_hist_kernel(const int64_t *__restrict__ cnts,
                                                            uint64_t n, unsigned *hist) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  unsigned ones = 0;  // class 1 is most of a power-law vocabulary: counted in a register
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t c = cnts[i];
    if (c == 1)
      ++ones;
    else
      atomicAdd(&h[c < 255 ? (c < 0 ? 0 : (unsigned)c) : 255u], 1u);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ones += __shfl_down(ones, off, 64);
  if (lane_id() == 0 && ones) atomicAdd(&h[1], ones);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// ---- flat range table from a KEY-SORTED list: no atomics, no random inserts ---------------------
// A vocabulary table with linear probing and a MONOTONE slot function can be laid out directly
// from the sorted keys: home slots h_i = f(K_i) are non-decreasing in i, so the position of entry
// i is p_i = max(h_i, p_{i-1} + 1) = i + max_{j <= i}(h_j - j) -- a prefix MAXIMUM over the list
// (decoupled look-back over the tiles), after which every entry is written once, positions
// increasing.  Against 36 M random 8-byte CAS inserts (the sort path's vocabularies, or the
// union a multi-GPU merge gathers): one streaming pass.  Lookups probe forward from f(key) to
// the key or an empty slot, exactly like the dumped tables of the range path (RangeMap.flat).
// The largest displacement p_i - h_i goes to aux[NVT_FLAT_AUX_MAXDISP]: keys that cluster in
// their range make long runs, the caller then builds an ordinary hashed table instead.
__device__ __forceinline__ void flat_params_body(const int32_t *__restrict__ keys, uint64_t n,
                                                 uint64_t slots, int32_t *aux) {
  // span of the (sorted) keys, the sentinel key (smallest int32, not in the table) left out
  const uint64_t first = (n > 1 && keys[0] == INT32_MIN) ? 1 : 0;
  const uint64_t lo = ukey(keys[first]), hi = ukey(keys[n - 1]);
  const uint64_t span = hi - lo, F = slots;  // any slot count < 2^32 (no power of two needed)
  uint32_t mul;
  int sh;
  range_map_params(span, F, &mul, &sh);
  aux[NVT_RANGE_AUX_LO] = (int32_t)(uint32_t)lo;
  aux[NVT_RANGE_AUX_LO + 1] = (int32_t)(uint32_t)span;
  aux[NVT_RANGE_AUX_LO + 2] = (int32_t)mul;
  aux[NVT_RANGE_AUX_LO + 3] = 0;
  aux[NVT_RANGE_AUX_LO + 4] = sh;
  aux[NVT_RANGE_AUX_LO + 5] = 1;  // flat layout
  aux[NVT_RANGE_AUX_LO + 6] = keys[0] == INT32_MIN ? 1 : 0;  // position 0 holds the smallest int32 (not in the table)
  aux[NVT_FLAT_AUX_MAXDISP] = 0;
}
__global__ void flat_params_kernel(const int32_t *__restrict__ keys, uint64_t n, uint64_t slots,
                                   int32_t *aux) {
  flat_params_body(keys, n, slots, aux);
}

constexpr unsigned long long kFbAgg = 1ull << 62, kFbPrefix = 2ull << 62, kFbMask = (1ull << 62) - 1ull;
constexpr long long kFbBias = 1ll << 40;  // h - i is > -2^30: biased to an unsigned value

__device__ __forceinline__ void flat_build_body(
    const int32_t *__restrict__ keys, const int32_t *__restrict__ label_of, uint64_t n,
    int32_t *aux, unsigned long long *status, unsigned *ticket, unsigned long long *table,
    uint64_t table_slots) {
  constexpr int NW = kS2BS / kWave;
  __shared__ unsigned long long wmax[NW];
  __shared__ unsigned long long s_carry;
  __shared__ unsigned s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const unsigned tile = s_tile, w = threadIdx.x / kWave, l = lane_id();
  const RangeMap map = load_map(aux);
  // element (wave w, row r, lane l): waves own contiguous 1024-entry runs (s2_elem)
  int32_t k[kS2Rows];
  unsigned long long d[kS2Rows];  // biased h - i, 0 = no entry
  unsigned long long run = 0;     // running maximum over this wave's rows so far
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    const uint64_t i = s2_elem(tile, w, r, l);
    k[r] = i < n ? keys[i] : INT32_MIN;
    d[r] = 0;
    if (i < n && k[r] != INT32_MIN) d[r] = (unsigned long long)((long long)map.fine(k[r]) - (long long)i + kFbBias);
  }
  // inclusive prefix maximum inside the wave's run: lanes of a row, then the rows in order
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    unsigned long long v = d[r];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long o = __shfl_up(v, off, 64);
      if (l >= (unsigned)off) v = o > v ? o : v;
    }
    v = run > v ? run : v;
    d[r] = v;
    run = __shfl(v, 63, 64);
  }
  if (l == 63) wmax[w] = run;
  __syncthreads();
  unsigned long long wprev = 0, tmax = 0;
  for (int q = 0; q < NW; ++q) {
    if (q < (int)w) wprev = wmax[q] > wprev ? wmax[q] : wprev;
    tmax = wmax[q] > tmax ? wmax[q] : tmax;
  }
  if (threadIdx.x == 0) {
    unsigned long long *my = status + tile;
    __hip_atomic_store(my, (tile == 0 ? kFbPrefix : kFbAgg) | tmax, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long carry = 0;
    if (tile > 0) {
      unsigned tb = tile - 1;
      while (true) {
        const unsigned long long v = __hip_atomic_load(status + tb, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        const unsigned f = (unsigned)(v >> 62);
        if (f == 0) {
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        const unsigned long long val = v & kFbMask;
        carry = val > carry ? val : carry;
        if (f == 2) break;
        --tb;
      }
      const unsigned long long incl = carry > tmax ? carry : tmax;
      __hip_atomic_store(my, kFbPrefix | incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_carry = carry;
  }
  __syncthreads();
  const unsigned long long before = s_carry > wprev ? s_carry : wprev;
  unsigned maxdisp = 0;
#pragma unroll
  for (int r = 0; r < kS2Rows; ++r) {
    const uint64_t i = s2_elem(tile, w, r, l);
    if (i >= n || k[r] == INT32_MIN) continue;
    const unsigned long long m = d[r] > before ? d[r] : before;
    const uint64_t p = (uint64_t)((long long)i + ((long long)m - kFbBias));
    const uint64_t h = map.fine(k[r]);
    const unsigned disp = (unsigned)(p - h < 0xFFFFFFFFull ? p - h : 0xFFFFFFFFull);
    maxdisp = disp > maxdisp ? disp : maxdisp;
    if (p < table_slots)
      table[p] = ((unsigned long long)(uint32_t)(label_of ? label_of[i] : (int32_t)i) << 32) | (uint32_t)k[r];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = __shfl_down(maxdisp, off, 64);
    maxdisp = o > maxdisp ? o : maxdisp;
  }
  if (l == 0 && maxdisp > 0) atomicMax(reinterpret_cast<unsigned *>(aux + NVT_FLAT_AUX_MAXDISP), maxdisp);
}
__global__ __launch_bounds__(kS2BS) void flat_build_kernel(
    const int32_t *__restrict__ keys, const int32_t *__restrict__ label_of, uint64_t n,
    int32_t *aux, unsigned long long *status, unsigned *ticket, unsigned long long *table,
    uint64_t table_slots) {
  flat_build_body(keys, label_of, n, aux, status, ticket, table, table_slots);
}
__global__ void flat_params_many_kernel(OrdBatch b) {
  const OrdJob &j = b.j[blockIdx.x];
  if (j.flat_slots) flat_params_body(j.keys, j.n, j.flat_slots, j.aux);
}
__global__ __launch_bounds__(kS2BS) void flat_build_many_kernel(OrdBatch b) {
  const int ji = ord_job_of(b.flat_start, b.njobs, blockIdx.x);
  const OrdJob &j = b.j[ji];
  const uint64_t ntiles = (j.n + kS2Tile - 1) / kS2Tile;
  flat_build_body(j.keys, j.label_of, j.n, j.aux, j.fb_status,
                  reinterpret_cast<unsigned *>(j.fb_status + ntiles), j.table, j.capacity);
}

// key -> position in the sorted list through a flat range table whose labels are the positions
// (groupby group ids, join_groupby.py:198-203 / target_encoding.py:350-371: the reference's left
// merge on the key column).  Probing runs forward from the key's home slot; the entries along a
// run are in key order, so a larger key ends an unsuccessful probe as an empty slot does.
struct FlatIndexView {
  RangeMap map;
  int64_t offset;  // table key = column key - offset (0 for int32 columns)
  bool has_min;
  int64_t null_group;  // group of the rows whose key is null (-1: none; aux word LO + 10 holds it + 1)
  const unsigned long long *table;
  uint64_t slots;
};

__device__ __forceinline__ FlatIndexView flat_view(const int32_t *__restrict__ aux,
                                                   const unsigned long long *table, uint64_t slots,
                                                   int64_t offset) {
  FlatIndexView v;
  v.map = load_map(aux);
  v.offset = offset;
  v.has_min = aux[NVT_RANGE_AUX_LO + 6] != 0;
  v.null_group = (int64_t)aux[NVT_RANGE_AUX_LO + 10] - 1;
  v.table = table;
  v.slots = slots;
  return v;
}

template <typename K>
__device__ __forceinline__ int64_t flat_probe(const FlatIndexView &v, const K *__restrict__ keys,
                                              const uint8_t *__restrict__ valid, uint64_t i) {
  int64_t kv;
  if (!bit_valid(valid, i)) return v.null_group;   // null keys are one group (groupby dropna=False)
  if (__builtin_sub_overflow((int64_t)keys[i], v.offset, &kv)) return -1;
  if (kv < (int64_t)INT32_MIN || kv > (int64_t)INT32_MAX) return -1;
  const int32_t k = (int32_t)kv;
  if (k == INT32_MIN) return v.has_min ? 0 : -1;
  const uint64_t home = v.map.fine(k);
  if (home >= v.slots) return -1;
  unsigned long long w = 0;
  const uint64_t sl = flat_find_from(v.table, v.slots, home, k, v.table[home], &w);
  return sl == ~0ull ? -1 : (int64_t)(uint32_t)(w >> 32);
}

template <typename K>
__global__ __launch_bounds__(kBlock) void flat_lookup_kernel(
    const K *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n,
    const int32_t *__restrict__ aux, const unsigned long long *__restrict__ table, uint64_t slots,
    int64_t offset, int64_t *__restrict__ out) {
  const FlatIndexView v = flat_view(aux, table, slots, offset);
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    out[i] = flat_probe(v, keys, valid, i);
}

// JoinGroupby.transform in one pass (join_groupby.py:198-217): probe, then the group's record
// of `ncols` float64 statistics (one 32-byte sector for count / sum / mean / std) instead of a
// group-id column in HBM and one random gather per statistic.
constexpr int kGatherMaxCols = 16;
struct GatherOuts {
  void *out[kGatherMaxCols];
  int dtype[kGatherMaxCols];
  double miss[kGatherMaxCols];
};

template <typename OUT>
__device__ __forceinline__ void gather_store(void *out, uint64_t i, double x) {
  reinterpret_cast<OUT *>(out)[i] = (OUT)x;
}

template <typename K, int NC>
__global__ __launch_bounds__(kBlock) void flat_lookup_gather_kernel(
    const K *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n,
    const int32_t *__restrict__ aux, const unsigned long long *__restrict__ table, uint64_t slots,
    int64_t offset, const double *__restrict__ records, GatherOuts o, unsigned long long *unseen) {
  const FlatIndexView v = flat_view(aux, table, slots, offset);
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  bool any_unseen = false;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t g = flat_probe(v, keys, valid, i);
    any_unseen |= g < 0;
    double x[NC];
    const double *rec = records + (uint64_t)(g < 0 ? 0 : g) * NC;
    if constexpr (NC % 2 == 0) {  // records are 16-byte aligned: two statistics per load
#pragma unroll
      for (int c = 0; c < NC; c += 2) {
        const double2 p = *reinterpret_cast<const double2 *>(rec + c);
        x[c] = p.x;
        x[c + 1] = p.y;
      }
    } else {
#pragma unroll
      for (int c = 0; c < NC; ++c) x[c] = rec[c];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {  // compile-time c: descriptors stay in scalar registers
      const double y = g < 0 ? o.miss[c] : x[c];
      switch (o.dtype[c]) {
        case NVT_F32: gather_store<float>(o.out[c], i, y); break;
        case NVT_F64: gather_store<double>(o.out[c], i, y); break;
        case NVT_I32: gather_store<int32_t>(o.out[c], i, y); break;
        default: gather_store<int64_t>(o.out[c], i, y); break;
      }
    }
  }
  if (unseen && __ballot(any_unseen) != 0ull && lane_id() == 0) atomicOr(unseen, 1ull);
}

template <typename K>
static int launch_gather(int ncols, unsigned grid, hipStream_t s, const K *keys, const uint8_t *valid,
                         uint64_t n, const int32_t *aux, const unsigned long long *tab,
                         uint64_t capacity, int64_t offset, const double *records,
                         const GatherOuts &o, unsigned long long *flag) {
#define NVT_G(NC)                                                                                 \
  case NC:                                                                                        \
    flat_lookup_gather_kernel<K, NC><<<grid, kBlock, 0, s>>>(keys, valid, n, aux, tab, capacity, \
                                                             offset, records, o, flag);          \
    break;
  switch (ncols) {
    NVT_G(1) NVT_G(2) NVT_G(3) NVT_G(4) NVT_G(5) NVT_G(6) NVT_G(7) NVT_G(8)
    NVT_G(9) NVT_G(10) NVT_G(11) NVT_G(12) NVT_G(13) NVT_G(14) NVT_G(15) NVT_G(16)
    default: return NVT_EINVAL;
  }
#undef NVT_G
  return NVT_OK;
}

// TargetEncoding.transform in one pass (target_encoding.py:341-371): probe, then the group's
// record {sum, count, (sum_f, count_f) for every fold} -- 16 * (kfold + 1) contiguous bytes.
// A (group, fold) pair without rows is the reference's unmatched [fold, key] merge: y_mean.
template <typename K, typename OUT>
__global__ __launch_bounds__(kBlock) void flat_lookup_te_kernel(
    const K *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n,
    const int32_t *__restrict__ aux, const unsigned long long *__restrict__ table, uint64_t slots,
    int64_t offset, const uint8_t *__restrict__ fold, unsigned kfold,
    const double *__restrict__ records, double p, double y_mean, OUT *__restrict__ out) {
  const FlatIndexView v = flat_view(aux, table, slots, offset);
  const unsigned stride_rec = 2 * (kfold + 1);
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t g = flat_probe(v, keys, valid, i);
    double r = y_mean;
    if (g >= 0) {
      const double *rec = records + (uint64_t)g * stride_rec;
      const double2 tot = *reinterpret_cast<const double2 *>(rec);
      if (fold) {
        const double2 f = *reinterpret_cast<const double2 *>(rec + 2 + 2 * (unsigned)fold[i]);
        if (f.y > 0.0) r = (tot.x - f.x + p * y_mean) / (tot.y - f.y + p);
      } else {
        r = (tot.x + p * y_mean) / (tot.y + p);
      }
    }
    out[i] = (OUT)r;
  }
}

// ---- lookup images: ONE probe and ONE record per row for every operator on a key column ----
// JoinGroupby.transform and TargetEncoding.transform on the same key column are two left merges
// on the same key in the reference (join_groupby.py:198-217, target_encoding.py:341-371).  Here
// every such operator ("consumer") owns a byte range of ONE packed per-group record whose values
// are already what a row receives, in the OUTPUT dtype: JoinGroupby's statistics cast to
// float32 / int32, TargetEncoding's smoothed value for every fold ((kfold + 1) values: slot 0 =
// no fold, slot 1 + f = rows of fold f) -- the formula depends on (group, fold) only, so
// evaluating it per group at the end of the fit gives the row's value bit for bit.  A row then
// costs one random sector for the probe and one for its record (<= 64 bytes), whatever the
// number of operators and statistics; the kernel moves 4- or 8-byte words, it does not convert.
template <typename K, int MAXC>
__global__ __launch_bounds__(kBlock) void flat_lookup_image_kernel(
    const K *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n,
    const int32_t *__restrict__ aux, const unsigned long long *__restrict__ table, uint64_t slots,
    int64_t offset, const int32_t *__restrict__ gid_in, int32_t *__restrict__ gid_out,
    const uint8_t *__restrict__ image, uint32_t stride_bytes, int ncols, ImageOuts o,
    unsigned long long *unseen) {
  const FlatIndexView v = flat_view(aux, table, slots, offset);
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  bool any_unseen = false;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t g = gid_in ? (int64_t)gid_in[i] : flat_probe(v, keys, valid, i);
    if (gid_out) gid_out[i] = (int32_t)g;
    any_unseen |= g < 0;
    const uint8_t *rec = image + (uint64_t)(g < 0 ? 0 : g) * stride_bytes;
    uint64_t x[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {  // all loads first: they hit the one sector of the record
      if (c < ncols) {
        uint32_t at = o.off[c];
        if (o.fold[c]) at += (1u + (uint32_t)o.fold[c][i]) * o.fstride[c];
        x[c] = o.size[c] == 8 ? *reinterpret_cast<const uint64_t *>(rec + at)
                              : (uint64_t)*reinterpret_cast<const uint32_t *>(rec + at);
      }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < ncols) {
        const uint64_t y = g < 0 ? o.miss[c] : x[c];
        if (o.size[c] == 8) __builtin_nontemporal_store(y, reinterpret_cast<uint64_t *>(o.out[c]) + i);
        else __builtin_nontemporal_store((uint32_t)y, reinterpret_cast<uint32_t *>(o.out[c]) + i);
      }
    }
  }
  if (unseen && __ballot(any_unseen) != 0ull && lane_id() == 0) atomicOr(unseen, 1ull);
}

// image[g * stride + off + 4|8 * c] = (dst dtype) src[c][g]: a consumer's statistics (float64 /
// int64 arrays of one value per group) written into its byte range of the records
struct ImagePackArgs {
  const void *src[kImageMaxCols];
  int src_dtype[kImageMaxCols];  // NVT_F64 / NVT_I64
  int dst_dtype[kImageMaxCols];  // NVT_F32 / NVT_F64 / NVT_I32 / NVT_I64
  uint32_t off[kImageMaxCols];
};

__global__ __launch_bounds__(kBlock) void image_pack_kernel(ImagePackArgs a, int ncols, uint64_t groups,
                                                            uint8_t *__restrict__ image,
                                                            uint32_t stride_bytes) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += stride) {
    uint8_t *rec = image + g * stride_bytes;
    for (int c = 0; c < ncols; ++c) {
      const bool is_int = a.src_dtype[c] == NVT_I64;
      const double x = is_int ? 0.0 : reinterpret_cast<const double *>(a.src[c])[g];
      const int64_t xi = is_int ? reinterpret_cast<const int64_t *>(a.src[c])[g] : 0;
      image_store(rec + a.off[c], a.dst_dtype[c], x, xi, is_int);
    }
  }
}

// JoinGroupby's byte range straight from the fit's accumulators (join_groupby.py:175-217 over
// categorify.py:1087-1131 _bottom_level_groupby): count, sum, mean = sum / n, var = (sumsq -
// sum * sum / n) / max(n - 1, 1) (NaN for n = 1), std = sqrt(var), min, max -- evaluated per group
// in float64 like the column-wise path (ops/_groupby.py derive_stats), stored in the output dtype.
__global__ __launch_bounds__(kBlock) void jg_image_kernel(JgImageArgs a, int ncols, uint64_t groups,
                                                          uint8_t *__restrict__ image, uint32_t stride_bytes) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += stride) {
    uint8_t *rec = image + g * stride_bytes;
    const int64_t ni = a.count[g];
    for (int c = 0; c < ncols; ++c) {
      bool is_int;
      const double x = jg_stat(a, c, g, ni, &is_int);
      image_store(rec + a.off[c], a.dst_dtype[c], x, ni, is_int);
    }
  }
}

// TargetEncoding's byte range: (kfold + 1) values per group from the fit's statistics -- totals
// {count, sum}[g] and the dense per-(group, fold) {count, sum}[g * kfold + f] of the sort path
// (nvt_sgb_reduce) -- exactly the expression nvt_te_apply_folds evaluates per row
// (target_encoding.py:350-371), once per (group, fold).  A thread per value: the fold arrays are
// read in memory order, a record's values leave as one contiguous run.
template <typename OUT>
__global__ __launch_bounds__(kBlock) void te_image_kernel(
    const int64_t *__restrict__ tot_count, const double *__restrict__ tot_sum,
    const int64_t *__restrict__ fold_count, const double *__restrict__ fold_sum, unsigned kfold,
    uint64_t groups, double p, double y_mean, uint8_t *__restrict__ image, uint32_t stride_bytes,
    uint32_t off) {
  const unsigned per = kfold + 1;
  const uint64_t total = groups * per, stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t e = (uint64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
    const uint64_t g = e / per;
    const unsigned slot = (unsigned)(e - g * per);
    const double r = te_value(tot_count, tot_sum, fold_count, fold_sum, kfold, g, slot, p, y_mean);
    *reinterpret_cast<OUT *>(image + g * stride_bytes + off + slot * sizeof(OUT)) = (OUT)r;
  }
}

// ---- vocabulary order of a list that is SHARDED over the ranks of a multi-GPU fit -------------
// Every rank owns a key range of the merged (key, count) list.  The order "count descending, key
// ascending" of the union is: class 255 (count >= 255, sorted exactly once all ranks' few such
// entries are gathered), then classes 254 .. 1, each in key order = owner by owner, every owner's
// entries in the order they have.  The label of an entry of class c < 255 is therefore
//   (entries of the classes in front of c, all owners) + (entries of class c on the owners in
//   front of this one) + (its rank among this shard's entries of class c)
// -- the last term is what the class scatter computes; the first two come in as class bases
// (`cls_hist` here is the caller's difference array of those bases: the kernel's exclusive prefix
// over the digits reproduces them modulo 2^32).  Every rank orders 1 / G of the union instead of
// all of it.
__global__ __launch_bounds__(kS2BS) void label_shard_kernel(
    const int32_t *__restrict__ keys, const int64_t *__restrict__ cnts, uint64_t n,
    const unsigned *__restrict__ cls_hist, unsigned *status, unsigned *ticket, int32_t *big_keys,
    int64_t *big_cnts, int32_t *label_of, int32_t *big_src) {
  cls_scatter_body(keys, cnts, n, cls_hist, status, ticket, big_keys, big_cnts, nullptr, 0, 0, nullptr,
                   label_of, big_src);
}

// vocabulary + table from a key-sorted list whose entries carry their position in the vocabulary
// order (labels[i], 0-based): ordered arrays by ONE scatter, absolute labels for the table build
__global__ __launch_bounds__(kBlock) void label_scatter_kernel(
    const int32_t *__restrict__ keys, const int64_t *__restrict__ cnts, const int32_t *__restrict__ labels,
    uint64_t n, int64_t first_label, int32_t *__restrict__ out_keys, int64_t *__restrict__ out_cnts,
    int32_t *__restrict__ abs_label, int64_t *sentinel_label) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const int32_t k = keys[i];
    const uint32_t l = (uint32_t)labels[i];
    out_keys[l] = k;
    out_cnts[l] = cnts[i];
    abs_label[i] = (int32_t)(first_label + (int64_t)l);
    if (k == INT32_MIN && sentinel_label != nullptr) *sentinel_label = first_label + (int64_t)l;
  }
}

int vocab_from_labels(const int32_t *src_keys, const int64_t *src_cnts, const int32_t *labels, uint64_t n,
                      int32_t *out_keys, int64_t *out_cnts, void *tmp, int64_t first_label, void *table,
                      uint64_t capacity, int64_t *sentinel_label, const int32_t *range_aux,
                      uint64_t flat_slots, hipStream_t s) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(n < (1ull << 30), "at most 2^30-1 vocabulary entries");
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile;
  // same layout as vocab_order_from_sorted with n_big = 0: status | label_of[n] | flat-build status
  int32_t *abs_label = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(tmp) + pad16(ntiles * 256 * 4 + 64));
  unsigned long long *fb_status =
      reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(abs_label) + pad16(n * 4));
  const bool flat = table != nullptr && range_aux != nullptr && flat_slots > 0;
  if (table != nullptr) {
    if (flat) {
      NVT_CHECK_ARG(flat_slots >= 64 && flat_slots < (1ull << 32), "flat table: 64 .. 2^32-1 slots");
      NVT_CHECK_ARG(capacity >= flat_slots + n + 64, "flat table: slots + n + 64");
    }
    int rc = encode_clear_any(4, table, capacity, sentinel_label, s);  // (also: no sentinel key yet)
    if (rc) return rc;
  }
  NVT_PROF("vocab_order", 0, s);
  label_scatter_kernel<<<stream_grid(n, kBlock, 8), kBlock, 0, s>>>(src_keys, src_cnts, labels, n, first_label,
                                                                    out_keys, out_cnts, abs_label,
                                                                    table != nullptr ? sentinel_label : nullptr);
  NVT_CHECK_LAUNCH();
  if (flat) {
    int32_t *aux = const_cast<int32_t *>(range_aux);
    flat_params_kernel<<<1, 1, 0, s>>>(src_keys, n, flat_slots, aux);
    NVT_CHECK_LAUNCH();
    NVT_CHECK_HIP(hipMemsetAsync(fb_status, 0, ntiles * 8 + 64, s));
    flat_build_kernel<<<(unsigned)ntiles, kS2BS, 0, s>>>(src_keys, abs_label, n, aux, fb_status,
                                                        reinterpret_cast<unsigned *>(fb_status + ntiles),
                                                        (unsigned long long *)table, capacity);
    NVT_CHECK_LAUNCH();
  } else if (table != nullptr) {
    // an ordinary hashed table: the ordered keys carry the labels first_label + position
    int rc = encode_insert_any(4, out_keys, n, first_label, table, capacity, sentinel_label, s);
    if (rc) return rc;
  }
  return NVT_OK;
}

uint64_t vocab_order_tmp_bytes(uint64_t n, uint64_t n_big) {
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile;
  uint64_t sort_bytes = 0;
  if (n_big > 1) (void)nvt_vocab_sort_tmp_bytes(4, n_big, &sort_bytes);
  // status words of the class scatter | sort scratch | label_of[n] | status words of the flat build
  return pad16(ntiles * 256 * 4 + 64) + pad16(sort_bytes) + pad16(n * 4) + pad16(ntiles * 8 + 64) + 64;
}

int vocab_order_from_sorted(const int32_t *src_keys, const int64_t *src_cnts, uint64_t n,
                            const unsigned *cls_hist, uint64_t n_big, int64_t max_count,
                            int32_t *out_keys, int64_t *out_cnts, void *tmp, int64_t first_label,
                            void *table, uint64_t capacity, int64_t *sentinel_label,
                            const int32_t *range_aux, int range_nb_log2, hipStream_t s,
                            bool *tail_deferred, uint64_t flat_slots) {
  if (tail_deferred) *tail_deferred = false;
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(n < (1ull << 30), "at most 2^30-1 vocabulary entries");
  NVT_CHECK_ARG(n_big <= n, "n_big > n");
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile;
  unsigned *status = reinterpret_cast<unsigned *>(tmp);
  unsigned *ticket = status + ntiles * 256;
  uint64_t sort_bytes = 0;
  if (n_big > 1) (void)nvt_vocab_sort_tmp_bytes(4, n_big, &sort_bytes);
  char *sort_tmp = reinterpret_cast<char *>(tmp) + pad16(ntiles * 256 * 4 + 64);
  int32_t *label_of = reinterpret_cast<int32_t *>(sort_tmp + pad16(sort_bytes));
  unsigned long long *fb_status =
      reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(label_of) + pad16(n * 4));
  // range table (range_aux set): `table` holds {key, position} slots already (dumped by the
  // counting pass) and only needs its positions replaced by labels -- no clear, no inserts.
  // flat (flat_slots > 0, range_aux = the block that RECEIVES the map): the table is laid
  // out from the sorted keys by a prefix maximum, see flat_build_kernel.
  const bool flat = table != nullptr && range_aux != nullptr && flat_slots > 0;
  const bool ranged = table != nullptr && range_aux != nullptr;
  if (flat) {
    NVT_CHECK_ARG(flat_slots >= 64 && flat_slots < (1ull << 32), "flat table: 64 .. 2^32-1 slots");
    NVT_CHECK_ARG(capacity >= flat_slots + n + 64, "flat table: slots + n + 64");
    int rc = encode_clear_any(4, table, capacity, sentinel_label, s);
    if (rc) return rc;
  } else if (table && !ranged) {
    int rc = encode_clear_any(4, table, capacity, sentinel_label, s);
    if (rc) return rc;
  } else if (ranged) {
    NVT_CHECK_HIP(hipMemsetAsync(sentinel_label, 0xFF, 8, s));  // -1: no sentinel key
  }
  {
    NVT_PROF("vocab_order", 0, s);
    NVT_CHECK_HIP(hipMemsetAsync(status, 0, ntiles * 256 * 4 + 64, s));
    cls_scatter_kernel<<<(unsigned)ntiles, kS2BS, 0, s>>>(
        src_keys, src_cnts, n, cls_hist, status, ticket, out_keys, out_cnts,
        ranged ? nullptr : (unsigned long long *)table, capacity - 1, first_label, sentinel_label,
        ranged ? label_of : nullptr);
    NVT_CHECK_LAUNCH();
    if (flat) {
      int32_t *aux = const_cast<int32_t *>(range_aux);
      flat_params_kernel<<<1, 1, 0, s>>>(src_keys, n, flat_slots, aux);
      NVT_CHECK_LAUNCH();
      NVT_CHECK_HIP(hipMemsetAsync(fb_status, 0, ntiles * 8 + 64, s));
      flat_build_kernel<<<(unsigned)ntiles, kS2BS, 0, s>>>(
          src_keys, label_of, n, aux, fb_status, reinterpret_cast<unsigned *>(fb_status + ntiles),
          (unsigned long long *)table, capacity);
      NVT_CHECK_LAUNCH();
    } else if (ranged) {
      const uint64_t nslots = ((uint64_t)1 << range_nb_log2) * kRpRegion + kRpGuard;
      range_patch_kernel<<<stream_grid(nslots / 2, kBlock, 8), kBlock, 0, s>>>(
          (unsigned long long *)table, nslots, label_of);
      NVT_CHECK_LAUNCH();
    }
  }
  // the sort of class 255 and the labels of its entries: left to ONE batched launch for all the
  // vocabularies of the call when the caller asks for it (vocab_order_tail_batch)
  if (tail_deferred && vocab_sort_small_eligible(4, n_big, max_count)) {
    *tail_deferred = true;
    return NVT_OK;
  }
  if (n_big > 1) {
    int rc = vocab_sort_any(4, out_keys, out_cnts, n_big, max_count, sort_tmp, s);
    if (rc) return rc;
  }
  if (table && n_big > 0) {
    if (ranged) {
      NVT_PROF("encode_build", 0, s);
      range_fix_prefix_kernel<<<stream_grid(n_big, kBlock), kBlock, 0, s>>>(
          (unsigned long long *)table, range_aux, out_keys, n_big, first_label, sentinel_label,
          capacity);
      NVT_CHECK_LAUNCH();
    } else {
      int rc = encode_insert_any(4, out_keys, n_big, first_label, table, capacity, sentinel_label, s);
      if (rc) return rc;
    }
  }
  return NVT_OK;
}

// vocab_order_from_sorted for several vocabularies that own a range table (dumped or flat): every
// stage one launch (ord_prep / cls_scatter_many / flat_params_many + flat_build_many /
// range_patch_many), then the class-255 tails (one batched small sort + one label launch; a tail
// too long for the small sort is sorted on its own).
int vocab_order_sorted_batch(const OrderSortedJob *jobs, int njobs, hipStream_t s) {
  for (int j0 = 0; j0 < njobs; j0 += kOrdBatch) {
    const int nj = njobs - j0 < kOrdBatch ? njobs - j0 : kOrdBatch;
    OrdBatch b;
    memset(&b, 0, sizeof(b));
    std::vector<OrderTail> tails;
    bool any_flat = false, any_ranged = false;
    for (int i = 0; i < nj; ++i) {
      const OrderSortedJob &q = jobs[j0 + i];
      NVT_CHECK_ARG(q.n > 0 && q.n < (1ull << 30), "1 .. 2^30-1 vocabulary entries");
      NVT_CHECK_ARG(q.n_big <= q.n, "n_big > n");
      NVT_CHECK_ARG(q.table && q.range_aux && q.tmp && q.sentinel_label, "range table jobs only");
      const uint64_t ntiles = (q.n + kS2Tile - 1) / kS2Tile;
      unsigned *status = reinterpret_cast<unsigned *>(q.tmp);
      uint64_t sort_bytes = 0;
      if (q.n_big > 1) (void)nvt_vocab_sort_tmp_bytes(4, q.n_big, &sort_bytes);
      char *sort_tmp = reinterpret_cast<char *>(q.tmp) + pad16(ntiles * 256 * 4 + 64);
      int32_t *label_of = reinterpret_cast<int32_t *>(sort_tmp + pad16(sort_bytes));
      OrdJob &o = b.j[i];
      o.keys = q.src_keys;
      o.cnts = q.src_cnts;
      o.cls_hist = q.cls_hist;
      o.status = status;
      o.ticket = status + ntiles * 256;
      o.out_keys = q.out_keys;
      o.out_cnts = q.out_cnts;
      o.label_of = label_of;
      o.table = reinterpret_cast<unsigned long long *>(q.table);
      o.sentinel_label = q.sentinel_label;
      o.aux = const_cast<int32_t *>(q.range_aux);
      o.fb_status = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(label_of) + pad16(q.n * 4));
      o.n = q.n;
      o.capacity = q.capacity;
      o.flat_slots = q.flat_slots;
      o.nslots = q.flat_slots ? 0 : ((uint64_t)1 << q.range_nb_log2) * kRpRegion + kRpGuard;
      o.first_label = (unsigned long long)q.first_label;
      o.status_words = ntiles * 256 + 16;
      o.fb_words = q.flat_slots ? ntiles + 8 : 0;
      if (q.flat_slots) {
        NVT_CHECK_ARG(q.flat_slots >= 64 && q.flat_slots < (1ull << 32), "flat table: 64 .. 2^32-1 slots");
        NVT_CHECK_ARG(q.capacity >= q.flat_slots + q.n + 64, "flat table: slots + n + 64");
        any_flat = true;
      } else {
        any_ranged = true;
      }
      b.tile_start[i + 1] = b.tile_start[i] + (unsigned)ntiles;
      b.flat_start[i + 1] = b.flat_start[i] + (q.flat_slots ? (unsigned)ntiles : 0u);
    }
    b.njobs = nj;
    {
      NVT_PROF("vocab_order", 0, s);
      ord_prep_kernel<<<dim3(any_flat ? 2048 : 64, nj), kBlock, 0, s>>>(b);
      NVT_CHECK_LAUNCH();
      cls_scatter_many_kernel<<<b.tile_start[nj], kS2BS, 0, s>>>(b);
      NVT_CHECK_LAUNCH();
      if (any_flat) {
        flat_params_many_kernel<<<nj, 1, 0, s>>>(b);
        NVT_CHECK_LAUNCH();
        flat_build_many_kernel<<<b.flat_start[nj], kS2BS, 0, s>>>(b);
        NVT_CHECK_LAUNCH();
      }
      if (any_ranged) {
        range_patch_many_kernel<<<dim3(1024, nj), kBlock, 0, s>>>(b);
        NVT_CHECK_LAUNCH();
      }
    }
    for (int i = 0; i < nj; ++i) {
      const OrderSortedJob &q = jobs[j0 + i];
      if (q.n_big == 0) continue;
      if (vocab_sort_small_eligible(4, q.n_big, q.max_count)) {
        tails.push_back({q.out_keys, q.out_cnts, q.n_big, q.first_label, q.table, q.capacity,
                         q.sentinel_label, q.range_aux});
        continue;
      }
      // a class 255 beyond the one-workgroup sort (merged multi-partition vocabularies), or of
      // one entry (nothing to sort)
      if (q.n_big > 1) {
        const uint64_t ntiles = (q.n + kS2Tile - 1) / kS2Tile;
        char *sort_tmp = reinterpret_cast<char *>(q.tmp) + pad16(ntiles * 256 * 4 + 64);
        int rc = vocab_sort_any(4, q.out_keys, q.out_cnts, q.n_big, q.max_count, sort_tmp, s);
        if (rc) return rc;
      }
      NVT_PROF("encode_build", 0, s);
      range_fix_prefix_kernel<<<stream_grid(q.n_big, kBlock), kBlock, 0, s>>>(
          (unsigned long long *)q.table, q.range_aux, q.out_keys, q.n_big, q.first_label,
          q.sentinel_label, q.capacity);
      NVT_CHECK_LAUNCH();
    }
    if (!tails.empty()) {
      int rc = vocab_order_tail_batch(tails.data(), (int)tails.size(), s);
      if (rc) return rc;
    }
  }
  return NVT_OK;
}

// class 255 of several vocabularies (vocab_order_from_sorted with tail_deferred): ONE batched
// sort launch (a workgroup per vocabulary) instead of a one-workgroup launch per vocabulary,
// then the labels of the sorted entries
int vocab_order_tail_batch(const OrderTail *t, int nt, hipStream_t s) {
  if (nt == 0) return NVT_OK;
  std::vector<SmallSortDesc> d(nt);
  for (int i = 0; i < nt; ++i) {
    d[i].keys = t[i].keys;
    d[i].counts = t[i].counts;
    d[i].n = (unsigned)t[i].n_big;
  }
  int rc = vocab_sort_small_batch(d.data(), nt, s);
  if (rc) return rc;
  NVT_PROF("encode_build", 0, s);
  // range tables (dumped or flat): the labels of all vocabularies in one launch
  for (int i0 = 0; i0 < nt;) {
    FixBatch fb;
    memset(&fb, 0, sizeof(fb));
    int nf = 0;
    uint64_t longest = 0;
    for (; i0 < nt && nf < kOrdBatch; ++i0) {
      if (!t[i0].table || !t[i0].range_aux) continue;
      fb.j[nf++] = {(unsigned long long *)t[i0].table, t[i0].range_aux, t[i0].keys,
                    t[i0].sentinel_label, t[i0].n_big, (unsigned long long)t[i0].first_label,
                    t[i0].capacity};
      longest = t[i0].n_big > longest ? t[i0].n_big : longest;
    }
    if (nf) {
      range_fix_prefix_many_kernel<<<dim3(stream_grid(longest, kBlock), nf), kBlock, 0, s>>>(fb);
      NVT_CHECK_LAUNCH();
    }
  }
  for (int i = 0; i < nt; ++i) {
    if (!t[i].table) continue;
    if (t[i].range_aux) {
      continue;  // (labelled above)
    } else {
      rc = encode_insert_any(4, t[i].keys, t[i].n_big, t[i].first_label, t[i].table, t[i].capacity,
                             t[i].sentinel_label, s);
      if (rc) return rc;
    }
  }
  return NVT_OK;
}

}  // namespace nvt

using namespace nvt;

extern "C" {

int nvt_vocab_sort_tmp_bytes(int key_bytes, uint64_t n, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes && (key_bytes == 4 || key_bytes == 8), "key_bytes must be 4 or 8");
  const uint64_t ntiles = (n + kTileSort - 1) / kTileSort;
  const uint64_t hist_len = 256 * ntiles;
  const uint64_t nchunks = (hist_len + kScanChunk - 1) / kScanChunk;
  *bytes = n * 8 + pad16(n * key_bytes) + pad16(hist_len * 4) + nchunks * 8 +
           (uint64_t)(key_bytes + 8) * 256 * 8 + 64;
  if (key_bytes == 4 && sort2_tmp_bytes(n) > *bytes) *bytes = sort2_tmp_bytes(n);
  if (key_bytes == 4 && os_tmp_bytes(n) > *bytes) *bytes = os_tmp_bytes(n);
  return NVT_OK;
}
int nvt_class_hist(const int64_t *counts, uint64_t n, uint32_t *hist, void *stream) {
  NVT_CHECK_ARG(hist && (n == 0 || counts), "null pointer");
  hipStream_t s = (hipStream_t)stream;
  NVT_CHECK_HIP(hipMemsetAsync(hist, 0, 256 * 4, s));
  if (n == 0) return NVT_OK;
  class_hist_kernel<<<stream_grid(n, kBlock * 8, 4), kBlock, 0, s>>>(counts, n, hist);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}
int nvt_vocab_label_shard(const int32_t *keys, const int64_t *counts, uint64_t n, const uint32_t *class_base_diff,
                          void *tmp, int32_t *label_of, int32_t *big_keys, int64_t *big_counts,
                          int32_t *big_src, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(keys && counts && class_base_diff && tmp && label_of && big_keys && big_counts && big_src,
                "null pointer");
  NVT_CHECK_ARG(n < (1ull << 30), "at most 2^30-1 entries");
  hipStream_t s = (hipStream_t)stream;
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile;
  unsigned *status = reinterpret_cast<unsigned *>(tmp);
  NVT_PROF("vocab_order", 0, s);
  NVT_CHECK_HIP(hipMemsetAsync(status, 0, ntiles * 256 * 4 + 64, s));
  label_shard_kernel<<<(unsigned)ntiles, kS2BS, 0, s>>>(keys, counts, n, class_base_diff, status,
                                                       status + ntiles * 256, big_keys, big_counts, label_of,
                                                       big_src);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_vocab_order_tmp_bytes(uint64_t n, uint64_t n_big, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes, "null out pointer");
  *bytes = vocab_order_tmp_bytes(n, n_big);
  return NVT_OK;
}
int nvt_vocab_sort_i32(int32_t *keys, int64_t *counts, uint64_t n, int64_t max_count, void *tmp,
                       void *stream) {
  NVT_CHECK_ARG(n <= 1 || (keys && counts && tmp), "null pointer");
  return vocab_sort_any(4, keys, counts, n, max_count, tmp, (hipStream_t)stream);
}
int nvt_vocab_sort_i64(int64_t *keys, int64_t *counts, uint64_t n, int64_t max_count, void *tmp,
                       void *stream) {
  NVT_CHECK_ARG(n <= 1 || (keys && counts && tmp), "null pointer");
  return vocab_sort_any(8, keys, counts, n, max_count, tmp, (hipStream_t)stream);
}

int nvt_flat_index_tmp_bytes(uint64_t n, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes, "null out");
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile;
  *bytes = pad16(ntiles * 8 + 64) + 64;
  return NVT_OK;
}

int nvt_flat_index_build(const int32_t *keys, uint64_t n, uint64_t slots, int32_t *aux, void *table,
                         uint64_t capacity, void *tmp, void *stream) {
  NVT_CHECK_ARG(keys && aux && table && tmp, "null pointer");
  NVT_CHECK_ARG(n >= 1 && n < (1ull << 30), "1 .. 2^30-1 keys");
  NVT_CHECK_ARG(slots >= 64 && slots < (1ull << 32), "slots must be 64 .. 2^32-1");
  NVT_CHECK_ARG(capacity >= slots + n + 64, "flat table: slots + n + 64");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_index", 0, s);
  const uint64_t ntiles = (n + kS2Tile - 1) / kS2Tile;
  unsigned long long *status = reinterpret_cast<unsigned long long *>(tmp);
  // the sentinel label of an encode table has no meaning here: it lands in the status block and
  // is wiped with it
  int rc = encode_clear_any(4, table, capacity, reinterpret_cast<int64_t *>(status), s);
  if (rc) return rc;
  NVT_CHECK_HIP(hipMemsetAsync(status, 0, ntiles * 8 + 64, s));
  flat_params_kernel<<<1, 1, 0, s>>>(keys, n, slots, aux);
  NVT_CHECK_LAUNCH();
  flat_build_kernel<<<(unsigned)ntiles, kS2BS, 0, s>>>(
      keys, nullptr, n, aux, status, reinterpret_cast<unsigned *>(status + ntiles),
      (unsigned long long *)table, capacity);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_flat_lookup(const void *keys, int dtype, const uint8_t *valid, uint64_t n, const int32_t *aux,
                    const void *table, uint64_t capacity, int64_t key_offset, int64_t *out,
                    void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(keys && aux && table && out, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_lookup", n * (dtype == NVT_I64 ? 8ull : 4ull), s);
  const unsigned grid = stream_grid(n, kBlock * 2);
  const unsigned long long *tab = reinterpret_cast<const unsigned long long *>(table);
  switch (dtype) {
    case NVT_I32:
      flat_lookup_kernel<int32_t><<<grid, kBlock, 0, s>>>((const int32_t *)keys, valid, n, aux, tab, capacity,
                                                          key_offset, out);
      break;
    case NVT_I64:
      flat_lookup_kernel<int64_t><<<grid, kBlock, 0, s>>>((const int64_t *)keys, valid, n, aux, tab, capacity,
                                                          key_offset, out);
      break;
    default:
      set_error("nvt_flat_lookup: key dtype must be int32 / int64 (got %d)", dtype);
      return NVT_EINVAL;
  }
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}


int nvt_flat_lookup_gather(const void *keys, int dtype, const uint8_t *valid, uint64_t n,
                           const int32_t *aux, const void *table, uint64_t capacity,
                           int64_t key_offset, const double *records, int ncols, void *const *outs,
                           const int *out_dtypes, const double *miss, uint64_t *unseen, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(keys && aux && table && records && outs && out_dtypes && miss, "null pointer");
  NVT_CHECK_ARG(ncols >= 1 && ncols <= kGatherMaxCols, "1..16 statistics per call");
  GatherOuts o;
  memset(&o, 0, sizeof(o));
  for (int c = 0; c < ncols; ++c) {
    NVT_CHECK_ARG(outs[c], "null output column");
    NVT_CHECK_ARG(out_dtypes[c] == NVT_F32 || out_dtypes[c] == NVT_F64 || out_dtypes[c] == NVT_I32 ||
                      out_dtypes[c] == NVT_I64, "output dtype must be f32 / f64 / i32 / i64");
    o.out[c] = outs[c];
    o.dtype[c] = out_dtypes[c];
    o.miss[c] = miss[c];
  }
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_lookup", n * (dtype == NVT_I64 ? 8ull : 4ull), s);
  const unsigned grid = stream_grid(n, kBlock * 2);
  const unsigned long long *tab = reinterpret_cast<const unsigned long long *>(table);
  unsigned long long *flag = reinterpret_cast<unsigned long long *>(unseen);
  int rc;
  if (dtype == NVT_I32)
    rc = launch_gather<int32_t>(ncols, grid, s, (const int32_t *)keys, valid, n, aux, tab, capacity,
                                key_offset, records, o, flag);
  else if (dtype == NVT_I64)
    rc = launch_gather<int64_t>(ncols, grid, s, (const int64_t *)keys, valid, n, aux, tab, capacity,
                                key_offset, records, o, flag);
  else {
    set_error("nvt_flat_lookup_gather: key dtype must be int32 / int64 (got %d)", dtype);
    return NVT_EINVAL;
  }
  if (rc) return rc;
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_flat_lookup_te(const void *keys, int dtype, const uint8_t *valid, uint64_t n, const int32_t *aux,
                       const void *table, uint64_t capacity, int64_t key_offset, const uint8_t *fold,
                       int kfold, const double *records, double p_smooth, double y_mean, void *out,
                       int out_dtype, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(keys && aux && table && records && out, "null pointer");
  NVT_CHECK_ARG(kfold >= 1 && kfold <= 256 && ((kfold > 1) == (fold != nullptr)), "fold ids come with kfold > 1");
  NVT_CHECK_ARG(dtype == NVT_I32 || dtype == NVT_I64, "key dtype must be int32 / int64");
  NVT_CHECK_ARG(out_dtype == NVT_F32 || out_dtype == NVT_F64, "out dtype must be f32 / f64");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("te_apply", n * (dtype == NVT_I64 ? 8ull : 4ull), s);
  const unsigned grid = stream_grid(n, kBlock * 2);
  const unsigned long long *tab = reinterpret_cast<const unsigned long long *>(table);
  const unsigned kf = fold ? (unsigned)kfold : 0u;  // record stride 2 * (kf + 1)
#define NVT_TE_LAUNCH(K, OUT)                                                                   \
  flat_lookup_te_kernel<K, OUT><<<grid, kBlock, 0, s>>>((const K *)keys, valid, n, aux, tab,    \
                                                        capacity, key_offset, fold, kf, records, \
                                                        p_smooth, y_mean, (OUT *)out)
  if (dtype == NVT_I32 && out_dtype == NVT_F32) NVT_TE_LAUNCH(int32_t, float);
  else if (dtype == NVT_I32) NVT_TE_LAUNCH(int32_t, double);
  else if (out_dtype == NVT_F32) NVT_TE_LAUNCH(int64_t, float);
  else NVT_TE_LAUNCH(int64_t, double);
#undef NVT_TE_LAUNCH
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_flat_lookup_image(const void *keys, int dtype, const uint8_t *valid, uint64_t n,
                          const int32_t *aux, const void *table, uint64_t capacity, int64_t key_offset,
                          const int32_t *gid_in, int32_t *gid_out, const void *image,
                          uint32_t stride_bytes, int ncols, void *const *outs,
                          const uint8_t *const *folds, const uint32_t *offs, const uint32_t *sizes,
                          const uint64_t *miss_bits, uint64_t *unseen, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(aux && table && image && outs && offs && sizes && miss_bits, "null pointer");
  NVT_CHECK_ARG(keys || gid_in, "keys or group ids");
  NVT_CHECK_ARG(ncols >= 1 && ncols <= kImageMaxCols, "1..24 outputs");
  NVT_CHECK_ARG(stride_bytes >= 8 && stride_bytes % 8 == 0, "record stride: a multiple of 8 bytes");
  NVT_CHECK_ARG(dtype == NVT_I32 || dtype == NVT_I64, "key dtype must be int32 / int64");
  ImageOuts o;
  memset(&o, 0, sizeof(o));
  for (int c = 0; c < ncols; ++c) {
    NVT_CHECK_ARG(outs[c], "null output");
    NVT_CHECK_ARG(sizes[c] == 4 || sizes[c] == 8, "values are 4 or 8 bytes");
    NVT_CHECK_ARG(offs[c] % sizes[c] == 0, "value offsets are aligned to the value size");
    o.out[c] = outs[c];
    o.fold[c] = folds ? folds[c] : nullptr;
    o.miss[c] = miss_bits[c];
    o.off[c] = offs[c];
    o.fstride[c] = sizes[c];
    o.size[c] = sizes[c];
    // (with a fold column the caller guarantees off + (kfold + 1) * size <= stride)
    NVT_CHECK_ARG((uint64_t)offs[c] + sizes[c] <= stride_bytes, "value outside the record");
  }
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_lookup", n * (dtype == NVT_I64 ? 8ull : 4ull), s);
  const unsigned grid = stream_grid(n, kBlock * 2);
  const unsigned long long *tab = reinterpret_cast<const unsigned long long *>(table);
  unsigned long long *flag = reinterpret_cast<unsigned long long *>(unseen);
  const uint8_t *img = reinterpret_cast<const uint8_t *>(image);
#define NVT_IMG(K, MAXC)                                                                          \
  flat_lookup_image_kernel<K, MAXC><<<grid, kBlock, 0, s>>>((const K *)keys, valid, n, aux, tab,  \
                                                            capacity, key_offset, gid_in, gid_out, \
                                                            img, stride_bytes, ncols, o, flag)
#define NVT_IMG_K(K)                    \
  do {                                  \
    if (ncols <= 2) NVT_IMG(K, 2);      \
    else if (ncols <= 4) NVT_IMG(K, 4); \
    else if (ncols <= 8) NVT_IMG(K, 8); \
    else if (ncols <= 16) NVT_IMG(K, 16); \
    else NVT_IMG(K, 24);                \
  } while (0)
  if (dtype == NVT_I32) NVT_IMG_K(int32_t);
  else NVT_IMG_K(int64_t);
#undef NVT_IMG_K
#undef NVT_IMG
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_image_pack(const void *const *src, const int *src_dtypes, const int *dst_dtypes,
                   const uint32_t *offs, int ncols, uint64_t groups, void *image,
                   uint32_t stride_bytes, void *stream) {
  if (groups == 0 || ncols == 0) return NVT_OK;
  NVT_CHECK_ARG(src && src_dtypes && dst_dtypes && offs && image, "null pointer");
  NVT_CHECK_ARG(ncols >= 1 && ncols <= kImageMaxCols, "1..24 columns");
  ImagePackArgs a;
  memset(&a, 0, sizeof(a));
  for (int c = 0; c < ncols; ++c) {
    NVT_CHECK_ARG(src[c], "null source column");
    NVT_CHECK_ARG(src_dtypes[c] == NVT_F64 || src_dtypes[c] == NVT_I64, "sources are float64 / int64");
    const int d = dst_dtypes[c];
    NVT_CHECK_ARG(d == NVT_F32 || d == NVT_F64 || d == NVT_I32 || d == NVT_I64, "values are f32 / f64 / i32 / i64");
    const uint32_t sz = (d == NVT_F32 || d == NVT_I32) ? 4u : 8u;
    NVT_CHECK_ARG(offs[c] % sz == 0 && (uint64_t)offs[c] + sz <= stride_bytes, "value outside the record");
    a.src[c] = src[c];
    a.src_dtype[c] = src_dtypes[c];
    a.dst_dtype[c] = d;
    a.off[c] = offs[c];
  }
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_index", groups * 8ull * ncols, s);
  image_pack_kernel<<<stream_grid(groups, kBlock, 8), kBlock, 0, s>>>(
      a, ncols, groups, reinterpret_cast<uint8_t *>(image), stride_bytes);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_jg_image(const int64_t *count, const double *const *sum, const double *const *sumsq,
                 const double *const *mn, const double *const *mx, int nvals, const int *kinds,
                 const int *vals, const int *dst_dtypes, const uint32_t *offs, int ncols, uint64_t groups,
                 void *image, uint32_t stride_bytes, void *stream) {
  if (groups == 0 || ncols == 0) return NVT_OK;
  NVT_CHECK_ARG(count && kinds && vals && dst_dtypes && offs && image, "null pointer");
  NVT_CHECK_ARG(ncols >= 1 && ncols <= kImageMaxCols, "1..24 columns");
  NVT_CHECK_ARG(nvals >= 0 && nvals <= kJgMaxVals, "0..8 value columns");
  JgImageArgs a;
  memset(&a, 0, sizeof(a));
  a.count = count;
  for (int j = 0; j < nvals; ++j) {
    a.sum[j] = sum ? sum[j] : nullptr;
    a.sumsq[j] = sumsq ? sumsq[j] : nullptr;
    a.mn[j] = mn ? mn[j] : nullptr;
    a.mx[j] = mx ? mx[j] : nullptr;
  }
  for (int c = 0; c < ncols; ++c) {
    const int k = kinds[c], j = vals[c], d = dst_dtypes[c];
    NVT_CHECK_ARG(k >= 0 && k <= 6, "statistic kind 0..6");
    NVT_CHECK_ARG(k == 0 || (j >= 0 && j < nvals), "value column out of range");
    NVT_CHECK_ARG(k == 0 || a.sum[j] || k == 3 || k == 4, "null sum array");
    NVT_CHECK_ARG((k != 3 || a.mn[j]) && (k != 4 || a.mx[j]) && (k < 5 || (a.sum[j] && a.sumsq[j])),
                  "null accumulator array for a requested statistic");
    NVT_CHECK_ARG(d == NVT_F32 || d == NVT_F64 || d == NVT_I32 || d == NVT_I64, "values are f32 / f64 / i32 / i64");
    const uint32_t sz = (d == NVT_F32 || d == NVT_I32) ? 4u : 8u;
    NVT_CHECK_ARG(offs[c] % sz == 0 && (uint64_t)offs[c] + sz <= stride_bytes, "value outside the record");
    a.kind[c] = k;
    a.val[c] = k == 0 ? 0 : j;
    a.dst_dtype[c] = d;
    a.off[c] = offs[c];
  }
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_index", groups * 8ull * ncols, s);
  jg_image_kernel<<<stream_grid(groups, kBlock, 8), kBlock, 0, s>>>(a, ncols, groups,
                                                                    reinterpret_cast<uint8_t *>(image), stride_bytes);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_te_image(const int64_t *tot_count, const double *tot_sum, const int64_t *fold_count,
                 const double *fold_sum, int kfold, uint64_t groups, double p_smooth, double y_mean,
                 int out_dtype, void *image, uint32_t stride_bytes, uint32_t off, void *stream) {
  if (groups == 0) return NVT_OK;
  NVT_CHECK_ARG(tot_count && tot_sum && image, "null pointer");
  NVT_CHECK_ARG(kfold >= 0 && kfold <= 256, "kfold must be 0 (no folds) .. 256");
  NVT_CHECK_ARG(kfold == 0 || (fold_count && fold_sum), "fold statistics come with kfold > 0");
  NVT_CHECK_ARG(out_dtype == NVT_F32 || out_dtype == NVT_F64, "out dtype must be f32 / f64");
  const uint32_t sz = out_dtype == NVT_F32 ? 4u : 8u;
  NVT_CHECK_ARG(off % sz == 0 && (uint64_t)off + (uint64_t)(kfold + 1) * sz <= stride_bytes,
                "values outside the record");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_index", groups * 16ull * (kfold + 1), s);
  const unsigned grid = stream_grid(groups * (uint64_t)(kfold + 1), kBlock * 2, 8);
  uint8_t *img = reinterpret_cast<uint8_t *>(image);
  if (out_dtype == NVT_F32)
    te_image_kernel<float><<<grid, kBlock, 0, s>>>(tot_count, tot_sum, fold_count, fold_sum, (unsigned)kfold,
                                                   groups, p_smooth, y_mean, img, stride_bytes, off);
  else
    te_image_kernel<double><<<grid, kBlock, 0, s>>>(tot_count, tot_sum, fold_count, fold_sum, (unsigned)kfold,
                                                    groups, p_smooth, y_mean, img, stride_bytes, off);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // extern "C"
