// Device side of the multi-GPU vocabulary exchange (the reference's tree reduce of
// per-partition groupby frames, categorify.py:1423-1529): everything around the collectives of
// nvtabular_amd/dist.py::merge_counts_many that is O(#distinct keys) work, for ALL columns of a
// fit in one launch each.
//
//   nvt_exchange_ranges   per column {-min key, max key, sum of counts}      (one MAX all-reduce)
//   nvt_exchange_hist     rows per (owner rank, column), owner = key range    (the count matrix)
//   nvt_exchange_scatter  (count << 32 | key) words grouped by (owner, column): the send buffer
//   nvt_exchange_unpack   the all-gathered words -> per column contiguous keys / counts
//
// The torch formulation of the same steps (26 columns x ~8 elementwise kernels, a radix sort of
// destination ids, a gather, a bincount; per-column concatenations on the way back) cost 3.2 +
// 1.8 ms per fit at the bench's size (31 M entries, tools/dist_ops_probe.py).
#include "nvt_common.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"

namespace nvt {

constexpr int kXMaxCols = 64;
constexpr int kXTile = kBlock * 16;
constexpr int kXMaxCells = 4096;  // owner x column counters of one workgroup (LDS)

struct XBatch {
  const int32_t *keys[kXMaxCols];
  const int64_t *cnts[kXMaxCols];
  uint64_t pre[kXMaxCols + 1];  // entries in front of column j in the virtual concatenation
  int64_t lo[kXMaxCols];        // owner(key) = min((key - lo) / width, G - 1)
  uint64_t width[kXMaxCols];
  int ncol, G;
};

__device__ __forceinline__ int x_col_of(const XBatch &b, uint64_t i) {
  int lo = 0, hi = b.ncol;  // the column with pre[c] <= i < pre[c + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (b.pre[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ unsigned x_owner(const XBatch &b, int c, int32_t key) {
  const uint64_t d = (uint64_t)((int64_t)key - b.lo[c]);  // key >= lo: the global minimum
  const uint64_t o = d / b.width[c];
  return (unsigned)(o < (uint64_t)(b.G - 1) ? o : (uint64_t)(b.G - 1));
}

__global__ void x_ranges_init_kernel(long long *__restrict__ rng, int ncol) {
  for (int j = threadIdx.x; j < ncol; j += blockDim.x) {
    rng[3 * j + 0] = -INT64_MAX;  // -min key (no entry on any rank: stays)
    rng[3 * j + 1] = -INT64_MAX;  // max key
    rng[3 * j + 2] = 0;           // rows counted on this rank
  }
}

__global__ __launch_bounds__(kBlock) void x_ranges_kernel(XBatch b, long long *__restrict__ rng) {
  __shared__ int smin[kXMaxCols], smax[kXMaxCols];
  __shared__ unsigned long long ssum[kXMaxCols];
  if (threadIdx.x < kXMaxCols) {
    smin[threadIdx.x] = INT32_MAX;
    smax[threadIdx.x] = INT32_MIN;
    ssum[threadIdx.x] = 0;
  }
  __syncthreads();
  const uint64_t n = b.pre[b.ncol];
  const unsigned lane = lane_id();
  for (uint64_t t0 = (uint64_t)blockIdx.x * kXTile; t0 < n; t0 += (uint64_t)gridDim.x * kXTile) {
    for (int u = 0; u < kXTile / kBlock; ++u) {
      const uint64_t i = t0 + (uint64_t)u * kBlock + threadIdx.x;
      const bool act = i < n;
      int c = -1;
      int32_t k = 0;
      unsigned long long w = 0;
      if (act) {
        c = x_col_of(b, i);
        const uint64_t r = i - b.pre[c];
        k = b.keys[c][r];
        w = (unsigned long long)b.cnts[c][r];
      }
      // a wave inside one column (all but the few that straddle a boundary): one LDS atomic per
      // statistic instead of 64 on the same word
      const int c0 = __builtin_amdgcn_readfirstlane(c);
      if (__ballot(c == c0) == __ballot(true) && c0 >= 0) {
        int mn = k, mx = k;
        unsigned long long sm = w;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const int a = __shfl_down(mn, off, 64), d = __shfl_down(mx, off, 64);
          const unsigned long long e = __shfl_down(sm, off, 64);
          mn = a < mn ? a : mn;
          mx = d > mx ? d : mx;
          sm += e;
        }
        if (lane == 0) {
          atomicMin(&smin[c0], mn);
          atomicMax(&smax[c0], mx);
          atomicAdd(&ssum[c0], sm);
        }
      } else if (act) {
        atomicMin(&smin[c], k);
        atomicMax(&smax[c], k);
        atomicAdd(&ssum[c], w);
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < b.ncol && ssum[threadIdx.x] > 0) {
    const int c = threadIdx.x;
    atomicMax(&rng[3 * c + 0], -(long long)smin[c]);
    atomicMax(&rng[3 * c + 1], (long long)smax[c]);
    atomicAdd(reinterpret_cast<unsigned long long *>(&rng[3 * c + 2]), ssum[c]);
  }
}

// key-sorted lists: the range is the first and the last key (rows: the caller's bound)
__global__ void x_ranges_sorted_kernel(XBatch b, long long *__restrict__ rng) {
  for (int j = threadIdx.x; j < b.ncol; j += blockDim.x) {
    const uint64_t n = b.pre[j + 1] - b.pre[j];
    rng[3 * j + 0] = n ? -(long long)b.keys[j][0] : -INT64_MAX;
    rng[3 * j + 1] = n ? (long long)b.keys[j][n - 1] : -INT64_MAX;
    rng[3 * j + 2] = 0;
  }
}

// SCATTER = false: send_mat[g * ncol + c] += rows of column c owned by rank g.
// SCATTER = true: `cursor` holds the start of every (owner, column) group in `rows_out` (and is
// advanced); the order of the rows inside a group is whatever the workgroups make it.
template <bool SCATTER>
__global__ __launch_bounds__(kBlock) void x_group_kernel(XBatch b, unsigned long long *__restrict__ cells,
                                                        int64_t *__restrict__ rows_out) {
  __shared__ unsigned cnt[kXMaxCells];
  __shared__ unsigned long long base[SCATTER ? kXMaxCells : 1];
  const int ncell = b.G * b.ncol;
  const uint64_t n = b.pre[b.ncol];
  for (uint64_t t0 = (uint64_t)blockIdx.x * kXTile; t0 < n; t0 += (uint64_t)gridDim.x * kXTile) {
    for (int q = threadIdx.x; q < ncell; q += kBlock) cnt[q] = 0;
    __syncthreads();
    unsigned cell[kXTile / kBlock], rank[kXTile / kBlock];
    int64_t word[kXTile / kBlock];
#pragma unroll
    for (int u = 0; u < kXTile / kBlock; ++u) {
      const uint64_t i = t0 + (uint64_t)u * kBlock + threadIdx.x;
      cell[u] = 0xFFFFFFFFu;
      if (i < n) {
        const int c = x_col_of(b, i);
        const uint64_t r = i - b.pre[c];
        const int32_t k = b.keys[c][r];
        cell[u] = x_owner(b, c, k) * (unsigned)b.ncol + (unsigned)c;
        if (SCATTER) word[u] = (int64_t)(((uint64_t)b.cnts[c][r] << 32) | (uint64_t)(uint32_t)k);
      }
      // key-ordered lists put whole waves into one (owner, column) cell: one LDS atomic for the
      // wave instead of 64 on the same word
      const unsigned c0 = __builtin_amdgcn_readfirstlane(cell[u]);
      if (c0 != 0xFFFFFFFFu && __ballot(cell[u] == c0) == ~0ull) {
        unsigned first = 0;
        if (lane_id() == 0) first = atomicAdd(&cnt[c0], (unsigned)kWave);
        rank[u] = __shfl(first, 0, 64) + lane_id();
      } else if (cell[u] != 0xFFFFFFFFu) {
        rank[u] = atomicAdd(&cnt[cell[u]], 1u);
      }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < ncell; q += kBlock) {
      const unsigned m = cnt[q];
      if (m) {
        const unsigned long long at = atomicAdd(&cells[q], (unsigned long long)m);
        if (SCATTER) base[q] = at;
      }
    }
    __syncthreads();
    if (SCATTER) {
#pragma unroll
      for (int u = 0; u < kXTile / kBlock; ++u)
        if (cell[u] != 0xFFFFFFFFu) rows_out[base[cell[u]] + rank[u]] = word[u];
    }
    __syncthreads();
  }
}

// KEY-SORTED lists: the rows of an (owner, column) group are a contiguous slice of the column
// (the owner is monotone in the key), so the position of a row in the send buffer follows from
// its index alone -- no cursors, no atomics, and the slice arrives IN KEY ORDER: the owner then
// merges G sorted runs per column (nvt_merge_sorted_many) instead of sorting what it received.
// first_row[cell] = rows of the column in front of the slice, start[cell] = first position of
// the group in rows_out.
__global__ __launch_bounds__(kBlock) void x_pack_ordered_kernel(XBatch b,
                                                                const uint64_t *__restrict__ first_row,
                                                                const uint64_t *__restrict__ start,
                                                                int64_t *__restrict__ rows_out) {
  const uint64_t n = b.pre[b.ncol];
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const int c = x_col_of(b, i);
    const uint64_t r = i - b.pre[c];
    const int32_t k = b.keys[c][r];
    const unsigned cell = x_owner(b, c, k) * (unsigned)b.ncol + (unsigned)c;
    rows_out[start[cell] + (r - first_row[cell])] =
        (int64_t)(((uint64_t)b.cnts[c][r] << 32) | (uint64_t)(uint32_t)k);
  }
}

// the all-gathered words (rank-major, column-minor segments) -> column-major: segment s goes to
// [dst_off[s], ...) of keys_out / cnts_out, so every column is one contiguous, key-ordered list
__global__ __launch_bounds__(kBlock) void x_unpack_kernel(const int64_t *__restrict__ words, uint64_t n,
                                                          const uint64_t *__restrict__ seg_off,
                                                          const uint64_t *__restrict__ dst_off,
                                                          int nseg, int32_t *__restrict__ keys_out,
                                                          int64_t *__restrict__ cnts_out) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int lo = 0, hi = nseg;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (seg_off[mid] <= i) lo = mid; else hi = mid;
    }
    const uint64_t d = dst_off[lo] + (i - seg_off[lo]);
    const int64_t w = words[i];
    keys_out[d] = (int32_t)(uint32_t)(uint64_t)w;
    cnts_out[d] = w >> 32;
  }
}

// the same with a second payload: one int32 per word (labels of the distributed ordering)
__global__ __launch_bounds__(kBlock) void x_unpack2_kernel(const int64_t *__restrict__ words,
                                                           const int32_t *__restrict__ extra, uint64_t n,
                                                           const uint64_t *__restrict__ seg_off,
                                                           const uint64_t *__restrict__ dst_off, int nseg,
                                                           int32_t *__restrict__ keys_out,
                                                           int64_t *__restrict__ cnts_out,
                                                           int32_t *__restrict__ extra_out) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int lo = 0, hi = nseg;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (seg_off[mid] <= i) lo = mid; else hi = mid;
    }
    const uint64_t d = dst_off[lo] + (i - seg_off[lo]);
    const int64_t w = words[i];
    keys_out[d] = (int32_t)(uint32_t)(uint64_t)w;
    cnts_out[d] = w >> 32;
    extra_out[d] = extra[i];
  }
}

static int fill_batch(XBatch &b, const nvt_xcol *cols, int ncol, const int64_t *lo, const uint64_t *width,
                      int G) {
  NVT_CHECK_ARG(cols && ncol >= 1 && ncol <= kXMaxCols, "1..64 columns");
  memset(&b, 0, sizeof(b));
  b.ncol = ncol;
  b.G = G;
  for (int j = 0; j < ncol; ++j) {
    NVT_CHECK_ARG(cols[j].n == 0 || (cols[j].keys && cols[j].counts), "null column");
    b.keys[j] = cols[j].keys;
    b.cnts[j] = cols[j].counts;
    b.pre[j + 1] = b.pre[j] + cols[j].n;
    b.lo[j] = lo ? lo[j] : 0;
    b.width[j] = width ? (width[j] ? width[j] : 1) : 1;
  }
  NVT_CHECK_ARG(b.pre[ncol] < (1ull << 40), "too many entries");
  return NVT_OK;
}

}  // namespace nvt

using namespace nvt;

extern "C" {

int nvt_exchange_ranges(const nvt_xcol *cols, int ncol, int64_t *rng, void *stream) {
  NVT_CHECK_ARG(rng, "null out");
  XBatch b;
  int rc = fill_batch(b, cols, ncol, nullptr, nullptr, 1);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("exchange_prep", b.pre[ncol] * 12ull, s);
  x_ranges_init_kernel<<<1, 64, 0, s>>>((long long *)rng, ncol);
  NVT_CHECK_LAUNCH();
  if (b.pre[ncol] == 0) return NVT_OK;
  x_ranges_kernel<<<stream_grid(b.pre[ncol], kXTile, 4), kBlock, 0, s>>>(b, (long long *)rng);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_exchange_ranges_sorted(const nvt_xcol *cols, int ncol, int64_t *rng, void *stream) {
  NVT_CHECK_ARG(rng, "null out");
  XBatch b;
  int rc = fill_batch(b, cols, ncol, nullptr, nullptr, 1);
  if (rc) return rc;
  x_ranges_sorted_kernel<<<1, 64, 0, (hipStream_t)stream>>>(b, (long long *)rng);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_exchange_hist(const nvt_xcol *cols, int ncol, const int64_t *lo, const uint64_t *width, int G,
                      uint64_t *send_mat, void *stream) {
  NVT_CHECK_ARG(send_mat && lo && width, "null pointer");
  NVT_CHECK_ARG(G >= 1 && (int64_t)G * ncol <= kXMaxCells, "ranks x columns must be <= 4096");
  XBatch b;
  int rc = fill_batch(b, cols, ncol, lo, width, G);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("exchange_prep", b.pre[ncol] * 4ull, s);
  NVT_CHECK_HIP(hipMemsetAsync(send_mat, 0, (size_t)G * ncol * 8, s));
  if (b.pre[ncol] == 0) return NVT_OK;
  x_group_kernel<false><<<stream_grid(b.pre[ncol], kXTile, 4), kBlock, 0, s>>>(
      b, (unsigned long long *)send_mat, nullptr);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_exchange_scatter(const nvt_xcol *cols, int ncol, const int64_t *lo, const uint64_t *width, int G,
                         uint64_t *cursors, int64_t *rows_out, void *stream) {
  NVT_CHECK_ARG(cursors && rows_out && lo && width, "null pointer");
  NVT_CHECK_ARG(G >= 1 && (int64_t)G * ncol <= kXMaxCells, "ranks x columns must be <= 4096");
  XBatch b;
  int rc = fill_batch(b, cols, ncol, lo, width, G);
  if (rc) return rc;
  if (b.pre[ncol] == 0) return NVT_OK;
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("exchange_prep", b.pre[ncol] * 20ull, s);
  x_group_kernel<true><<<stream_grid(b.pre[ncol], kXTile, 4), kBlock, 0, s>>>(
      b, (unsigned long long *)cursors, rows_out);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_exchange_pack_ordered(const nvt_xcol *cols, int ncol, const int64_t *lo, const uint64_t *width,
                              int G, const uint64_t *first_row, const uint64_t *start, int64_t *rows_out,
                              void *stream) {
  NVT_CHECK_ARG(first_row && start && rows_out && lo && width, "null pointer");
  NVT_CHECK_ARG(G >= 1 && (int64_t)G * ncol <= kXMaxCells, "ranks x columns must be <= 4096");
  XBatch b;
  int rc = fill_batch(b, cols, ncol, lo, width, G);
  if (rc) return rc;
  if (b.pre[ncol] == 0) return NVT_OK;
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("exchange_prep", b.pre[ncol] * 20ull, s);
  x_pack_ordered_kernel<<<stream_grid(b.pre[ncol], kBlock * 4), kBlock, 0, s>>>(b, first_row, start, rows_out);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_exchange_unpack(const int64_t *words, uint64_t n, const uint64_t *seg_off, const uint64_t *dst_off,
                        int nseg, int32_t *keys_out, int64_t *counts_out, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(words && seg_off && dst_off && keys_out && counts_out && nseg >= 1, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("exchange_unpack", n * 20ull, s);
  x_unpack_kernel<<<stream_grid(n, kBlock * 4), kBlock, 0, s>>>(words, n, seg_off, dst_off, nseg, keys_out,
                                                             counts_out);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_exchange_unpack2(const int64_t *words, const int32_t *extra, uint64_t n, const uint64_t *seg_off,
                         const uint64_t *dst_off, int nseg, int32_t *keys_out, int64_t *counts_out,
                         int32_t *extra_out, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(words && extra && seg_off && dst_off && keys_out && counts_out && extra_out && nseg >= 1,
                "null pointer");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("exchange_unpack", n * 28ull, s);
  x_unpack2_kernel<<<stream_grid(n, kBlock * 4), kBlock, 0, s>>>(words, extra, n, seg_off, dst_off, nseg,
                                                              keys_out, counts_out, extra_out);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // extern "C"
