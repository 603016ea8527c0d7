// Categorify.fit groupby-size, path 10 ("sort path"): int32 keys, unweighted, more distinct keys
// than the range path's 1024 LDS buckets hold (> ~6.5 M: Criteo-1TB's 38-40 M-unique columns,
// bench/examples/dask-nvtabular-criteo-benchmark.py:360-366).  Replaces categorify.py:955-1051
// (_top_level_groupby, size only) for columns in which nearly every row carries its own key:
// counting such rows in hash tables means two partition passes plus one LDS table per 3 k keys
// (path 3) and leaves an unordered list of tens of millions of entries to a 7-pass sort.  Here:
//
//   sc_pack_kernel    row -> 32-bit order-preserving key image; null rows -> 0xFFFFFFFF
//   s32_* kernels     onesweep LSD radix sort of the 32-bit words (4 passes of 8 bits, one
//                     histogram read; the scheme of nvt_sort.hip on half the bytes)
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
constexpr uint32_t kNullWord = 0xFFFFFFFFu;  // also the image of INT32_MAX: null rows sort last,
                                             // the first rows - nulls sorted words are the valid ones

__global__ __launch_bounds__(kBlock) void sc_pack_kernel(const int32_t *__restrict__ keys,
                                                         const uint8_t *__restrict__ valid,
                                                         uint64_t n, uint32_t *__restrict__ words,
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
    uint4 w;
    w.x = (vb & 1) ? (uint32_t)p.x ^ 0x80000000u : kNullWord;
    w.y = (vb & 2) ? (uint32_t)p.y ^ 0x80000000u : kNullWord;
    w.z = (vb & 4) ? (uint32_t)p.z ^ 0x80000000u : kNullWord;
    w.w = (vb & 8) ? (uint32_t)p.w ^ 0x80000000u : kNullWord;
    nulls += 4 - __popc(vb);
    reinterpret_cast<uint4 *>(words)[v] = w;
  }
  for (uint64_t i = nvec * 4 + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const bool ok = bit_valid(valid, i);
    words[i] = ok ? (uint32_t)keys[i] ^ 0x80000000u : kNullWord;
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

// ---- onesweep LSD radix sort of 32-bit words (the scheme of nvt_sort.hip's os_* kernels on half
// the bytes: 8192-word tiles so that a digit's run leaves a tile as >= 128 bytes) ------------------
constexpr int kS32BS = 256, kS32Rows = 32, kS32Tile = kS32BS * kS32Rows;  // 8192 words per tile
constexpr int kS32HistBlocks = 256;

__device__ __forceinline__ unsigned long long match8(unsigned digit, bool active) {
  unsigned long long peers = __ballot(active);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const unsigned long long m = __ballot((digit >> b) & 1);
    peers &= ((digit >> b) & 1) ? m : ~m;
  }
  return peers;
}
// element (wave w, row r, lane l) of a tile: waves own contiguous 2048-word runs (stability)
__device__ __forceinline__ uint64_t s32_elem(uint64_t tile, unsigned w, unsigned r, unsigned l) {
  return tile * kS32Tile + (uint64_t)w * (kS32Rows * kWave) + (uint64_t)r * kWave + l;
}

// ONE read of the words: the digit histograms of all four passes
__global__ __launch_bounds__(kS32BS) void s32_hist_kernel(const uint32_t *__restrict__ w, uint64_t n,
                                                          unsigned *__restrict__ block_hist) {
  __shared__ unsigned h[4 * 256];
  for (int i = threadIdx.x; i < 4 * 256; i += kS32BS) h[i] = 0;
  __syncthreads();
  const uint64_t nvec = n / 4;
  const uint64_t stride = (uint64_t)gridDim.x * kS32BS;
  for (uint64_t v = (uint64_t)blockIdx.x * kS32BS + threadIdx.x; v < nvec; v += stride) {
    const uint4 x = reinterpret_cast<const uint4 *>(w)[v];
    const uint32_t e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int p = 0; p < 4; ++p) atomicAdd(&h[p * 256 + ((e[j] >> (8 * p)) & 0xFF)], 1u);
  }
  for (uint64_t i = nvec * 4 + (uint64_t)blockIdx.x * kS32BS + threadIdx.x; i < n; i += stride) {
    const uint32_t e = w[i];
#pragma unroll
    for (int p = 0; p < 4; ++p) atomicAdd(&h[p * 256 + ((e >> (8 * p)) & 0xFF)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * 256; i += kS32BS) block_hist[(uint64_t)blockIdx.x * 1024 + i] = h[i];
}

// one workgroup per pass: base[p][d] = number of words whose digit (pass p) is < d
__global__ __launch_bounds__(256) void s32_base_kernel(const unsigned *__restrict__ block_hist,
                                                       int nblocks, unsigned *__restrict__ base) {
  __shared__ unsigned wtot[4];
  const int p = blockIdx.x, d = threadIdx.x;
  unsigned tot = 0;
#pragma unroll 8
  for (int b = 0; b < nblocks; ++b) tot += block_hist[(uint64_t)b * 1024 + p * 256 + d];
  unsigned inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(inc, off, 64);
    if (lane_id() >= (unsigned)off) inc += o;
  }
  const unsigned w = threadIdx.x / kWave;
  if (lane_id() == 63) wtot[w] = inc;
  __syncthreads();
  unsigned wb = 0;
  for (unsigned q = 0; q < w; ++q) wb += wtot[q];
  base[p * 256 + d] = wb + inc - tot;
}

__global__ __launch_bounds__(kS32BS) void s32_scatter_kernel(
    const uint32_t *__restrict__ src, uint64_t n, int shift, const unsigned *__restrict__ base,
    unsigned *status, unsigned *ticket, uint32_t *__restrict__ dst) {
  constexpr int NW = kS32BS / kWave;
  __shared__ unsigned wcnt[NW][256];
  __shared__ unsigned goff[256];
  __shared__ unsigned wtot[NW];
  __shared__ unsigned s_tile;
  __shared__ uint32_t stage[kS32Tile];
  const unsigned w = threadIdx.x / kWave, l = lane_id();
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
  for (int q = 0; q < NW; ++q) wcnt[q][threadIdx.x] = 0;
  __syncthreads();
  const unsigned tile = s_tile;
  uint32_t c[kS32Rows];
  unsigned short local[kS32Rows];
#pragma unroll
  for (int r = 0; r < kS32Rows; ++r) {
    const uint64_t i = s32_elem(tile, w, r, l);
    c[r] = i < n ? src[i] : 0xFFFFFFFFu;
  }
  const uint64_t tile_base = (uint64_t)tile * kS32Tile;
#pragma unroll
  for (int r = 0; r < kS32Rows; ++r) {
    const bool act = s32_elem(tile, w, r, l) < n;
    const unsigned d = (c[r] >> shift) & 0xFF;
    const unsigned long long peers = match8(d, act);
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
    __hip_atomic_store(my, (tile == 0 ? kScPrefix : kScAgg) | tot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    unsigned inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(inc, off, 64);
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
        excl += v & kScMask;
        if (f == 2) break;
        --tb;
      }
      __hip_atomic_store(my, kScPrefix | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    goff[d] = base[d] + excl - dstart;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kS32Rows; ++r) {
    if (s32_elem(tile, w, r, l) < n) {
      const unsigned d = (c[r] >> shift) & 0xFF;
      stage[wcnt[w][d] + local[r]] = c[r];
    }
  }
  __syncthreads();
  const unsigned tile_n = (unsigned)(n - tile_base < (uint64_t)kS32Tile ? n - tile_base : kS32Tile);
#pragma unroll 4
  for (int j = 0; j < kS32Rows; ++j) {
    const unsigned idx = j * kS32BS + threadIdx.x;
    if (idx < tile_n) {
      const uint32_t v = stage[idx];
      dst[goff[(v >> shift) & 0xFF] + idx] = v;
    }
  }
}

// run heads of the sorted words.  Tile = 4096 consecutive words staged in LDS; thread t owns 16
// consecutive words; tile offsets by a decoupled look-back (ticketed tile ids)
__global__ __launch_bounds__(kScBS) void sc_rle_kernel(const uint32_t *__restrict__ sorted,
                                                       uint64_t n, unsigned *status,
                                                       unsigned *ticket, int32_t *__restrict__ out_keys,
                                                       unsigned *__restrict__ head_pos,
                                                       uint64_t out_cap, uint64_t *state) {
  // key images of the tile, slot 0 = the word in front of it; one pad word per 32 so that the
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
    hi[at(1 + r * kScBS + threadIdx.x)] = i < n ? sorted[i] : kNullWord;
  }
  if (threadIdx.x == 0) hi[at(0)] = base_i > 0 ? sorted[base_i - 1] : kNullWord;
  __syncthreads();
  // validity cannot be told from the word (0xFFFFFFFF is also the image of INT32_MAX): nulls
  // sort last, so "valid" = index < n_valid, n_valid = rows - nulls (the pack kernel is done)
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

// ws: words A | words B (ping-pong; the run starts reuse whichever is free) | block histograms |
//     bases | status words of the four passes + tickets | status words of the run-length pass
uint64_t sort_count_ws_bytes(uint64_t n) {
  const uint64_t t32 = (n + kS32Tile - 1) / kS32Tile, trle = (n + kScTile - 1) / kScTile;
  return 3 * al256(n * 4) + al256((uint64_t)kS32HistBlocks * 1024 * 4) + al256(1024 * 4) +
         al256(4 * t32 * 256 * 4 + 64) + al256(trle * 4 + 64) + 256;
}

// hist = device uint32[256] (cleared here)
int sort_count_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, void *wsp, unsigned *hist,
                   int32_t *out_keys, int64_t *out_cnt, uint64_t out_cap, uint64_t *state,
                   hipStream_t s) {
  NVT_CHECK_ARG(hist != nullptr, "sort path: the column needs its histogram block (hot_image)");
  NVT_CHECK_ARG(n < (1ull << 30), "sort path: at most 2^30-1 rows per call");
  NVT_PROF("dense_count_s10", n * 4, s);
  const uint64_t t32 = (n + kS32Tile - 1) / kS32Tile, trle = (n + kScTile - 1) / kScTile;
  char *p = reinterpret_cast<char *>(wsp);
  uint32_t *bufA = reinterpret_cast<uint32_t *>(p);
  p += al256(n * 4);
  uint32_t *bufB = reinterpret_cast<uint32_t *>(p);
  p += al256(n * 4);
  unsigned *head_pos = reinterpret_cast<unsigned *>(p);
  p += al256(n * 4);
  unsigned *block_hist = reinterpret_cast<unsigned *>(p);
  p += al256((uint64_t)kS32HistBlocks * 1024 * 4);
  unsigned *base = reinterpret_cast<unsigned *>(p);
  p += al256(1024 * 4);
  unsigned *status = reinterpret_cast<unsigned *>(p);
  unsigned *tickets = status + 4 * t32 * 256;
  p += al256(4 * t32 * 256 * 4 + 64);
  unsigned *rle_status = reinterpret_cast<unsigned *>(p);
  unsigned *rle_ticket = rle_status + trle;
  NVT_CHECK_HIP(hipMemsetAsync(status, 0, 4 * t32 * 256 * 4 + 64, s));
  NVT_CHECK_HIP(hipMemsetAsync(rle_status, 0, trle * 4 + 64, s));
  NVT_CHECK_HIP(hipMemsetAsync(hist, 0, 256 * 4, s));
  sc_pack_kernel<<<stream_grid(n / 4 + 1, kBlock, 8), kBlock, 0, s>>>(keys, valid, n, bufA, state);
  NVT_CHECK_LAUNCH();
  const unsigned hb = (unsigned)(t32 < (uint64_t)kS32HistBlocks ? t32 : kS32HistBlocks);
  s32_hist_kernel<<<hb, kS32BS, 0, s>>>(bufA, n, block_hist);
  NVT_CHECK_LAUNCH();
  s32_base_kernel<<<4, 256, 0, s>>>(block_hist, (int)hb, base);
  NVT_CHECK_LAUNCH();
  uint32_t *src = bufA, *dst = bufB;
  for (int pass = 0; pass < 4; ++pass) {
    s32_scatter_kernel<<<(unsigned)t32, kS32BS, 0, s>>>(src, n, 8 * pass, base + pass * 256,
                                                        status + (uint64_t)pass * t32 * 256,
                                                        tickets + pass, dst);
    NVT_CHECK_LAUNCH();
    uint32_t *t = src;
    src = dst;
    dst = t;
  }
  sc_rle_kernel<<<(unsigned)trle, kScBS, 0, s>>>(src, n, rle_status, rle_ticket, out_keys, head_pos,
                                                 out_cap, state);
  NVT_CHECK_LAUNCH();
  sc_counts_kernel<<<stream_grid(n / 2 + 1, kBlock, 8), kBlock, 0, s>>>(head_pos, out_cnt, hist, state);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // namespace nvt
