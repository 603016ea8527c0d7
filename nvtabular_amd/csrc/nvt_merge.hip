// Merge of KEY-SORTED lists -- the tree-merge step of a multi-partition fit.
//
// Reference: nvtabular/ops/categorify.py:1054-1070 (_mid_level_groupby: concat the partial
// groupby frames of `split_every` partitions and group them again) inside the tree of
// categorify.py:1423-1478; join_groupby.py:140-173 / target_encoding.py:171-214 reach the same
// tree through _groupby_to_disk with their sum / count aggregates.
//
// Here every partition's partial result is already ORDERED BY KEY (range path, sort path, the
// sort-path groupby), so "concat + group again" is a 2-way merge: the union of two ascending,
// duplicate-free int32 key lists with the payloads of equal keys combined.  One pass, no hash
// table, no sort, and the output is key-ordered again -- the vocabulary of a multi-partition fit
// keeps the one-pass ordering (cls_scatter) and the flat range table of the single-partition
// fit instead of falling back to radix sort + random inserts.
//
// Kernels (HBM-bound streaming, 12 B read + <= 12 B written per entry):
//   merge_split_kernel   one thread per tile boundary: merge-path search on the two lists
//                        (ties: A first), so tile t covers merged ranks [t * TILE, (t+1) * TILE)
//   merge_tile_kernel    one workgroup per tile (ticketed): the tile's runs of A and B are staged
//                        in LDS, every thread merges VT consecutive ranks serially, an entry of B
//                        that equals the entry of A in front of it is dropped and its count added
//                        there (a key occurs at most once per list, so duplicates come in pairs:
//                        neighbours across thread and tile borders are read directly), the kept
//                        entries are compacted through LDS and written at the tile's offset, which
//                        comes from a decoupled look-back over the preceding tiles' status words.
//                        Optionally the positions (in A, in B) every output entry came from are
//                        written too (src maps): nvt_merge_payload then combines any number of
//                        payload arrays (sums, per-fold counts, minima ...) of a groupby result.
#include <vector>

#include "nvt_common.hpp"
#include "nvt_prof.hpp"

namespace nvt {

#ifndef NVT_MG_BS
#define NVT_MG_BS 1024
#endif
#ifndef NVT_MG_VT
#define NVT_MG_VT 5
#endif
constexpr int kMgBS = NVT_MG_BS;
constexpr int kMgVT = NVT_MG_VT;  // odd: the threads' serial LDS walks start on different banks; 1024 x 5 measured
                                  // best of {128..1024} x {3..15} (tools/var_merge.sh: short serial walks, 32 waves per CU)
constexpr int kMgTile = kMgBS * kMgVT;
constexpr int kMgMaxCols = 30;

constexpr unsigned long long kMgAgg = 1ull << 62, kMgPrefix = 2ull << 62, kMgMask = (1ull << 62) - 1ull;

struct MergeCol {
  const int32_t *a_keys, *b_keys;
  const int64_t *a_cnt, *b_cnt;
  int32_t *out_keys;
  int64_t *out_cnt;
  int32_t *src_a, *src_b;
  unsigned long long *out_n;
  unsigned *splits;             // [ntiles + 1] entries of A in front of every tile boundary
  unsigned long long *status;   // [ntiles] look-back words, zeroed
  unsigned *ticket;             // zeroed
  unsigned na, nb;
};
struct MergeBatch {
  MergeCol c[kMgMaxCols];
  unsigned tile_start[kMgMaxCols + 1];  // first block of every column
  unsigned bnd_start[kMgMaxCols + 1];   // first boundary thread of every column
  int ncols;
};

__device__ __forceinline__ int batch_col(const unsigned *start, int ncols, unsigned i) {
  int c = 0;
  while (c + 1 < ncols && i >= start[c + 1]) ++c;
  return c;
}

// entries of A among the first d merged entries (ties: A first)
__device__ __forceinline__ unsigned merge_path(const int32_t *__restrict__ a, unsigned na,
                                               const int32_t *__restrict__ b, unsigned nb, unsigned d) {
  unsigned lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  while (lo < hi) {
    const unsigned mid = (lo + hi) >> 1;
    if (a[mid] <= b[d - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(kMgBS) void merge_split_kernel(MergeBatch b) {
  const unsigned i = blockIdx.x * kMgBS + threadIdx.x;
  if (i >= b.bnd_start[b.ncols]) return;
  const int ci = batch_col(b.bnd_start, b.ncols, i);
  const MergeCol &c = b.c[ci];
  const unsigned t = i - b.bnd_start[ci];
  const uint64_t total = (uint64_t)c.na + c.nb;
  const uint64_t d64 = (uint64_t)t * kMgTile;
  const unsigned d = (unsigned)(d64 < total ? d64 : total);
  c.splits[t] = merge_path(c.a_keys, c.na, c.b_keys, c.nb, d);
  if (t == 0 && total == 0) *c.out_n = 0;
}

template <bool SRC>
__global__ __launch_bounds__(kMgBS) void merge_tile_kernel(MergeBatch b) {
  __shared__ int32_t s_key[kMgTile];
  __shared__ long long s_cnt[kMgTile];
  __shared__ unsigned s_wsum[kMgBS / kWave];
  __shared__ unsigned s_tile;
  __shared__ unsigned long long s_base;
  const int ci = batch_col(b.tile_start, b.ncols, blockIdx.x);
  const MergeCol &c = b.c[ci];
  if (threadIdx.x == 0) s_tile = atomicAdd(c.ticket, 1u);
  __syncthreads();
  const unsigned tile = s_tile, t = threadIdx.x, w = t / kWave, l = lane_id();
  const unsigned na = c.na, nb = c.nb;
  const uint64_t total = (uint64_t)na + nb;
  const uint64_t d0 = (uint64_t)tile * kMgTile;
  if (d0 >= total) return;
  const uint64_t d1 = d0 + kMgTile < total ? d0 + kMgTile : total;
  const unsigned a0 = c.splits[tile], a1 = c.splits[tile + 1];
  const unsigned b0 = (unsigned)(d0 - a0), b1 = (unsigned)(d1 - a1);
  const unsigned na_t = a1 - a0, nb_t = b1 - b0, nt = na_t + nb_t;
  const bool has_cnt = c.a_cnt != nullptr;
  // stage the tile's runs: A at [0, na_t), B behind it
  for (unsigned i = t; i < nt; i += kMgBS) {
    const bool fa = i < na_t;
    const unsigned j = fa ? a0 + i : b0 + (i - na_t);
    s_key[i] = fa ? c.a_keys[j] : c.b_keys[j];
    if (has_cnt) s_cnt[i] = fa ? c.a_cnt[j] : c.b_cnt[j];
  }
  // neighbours outside the tile: the entry of A in front of it (a leading B entry may repeat
  // it) and the entry of B behind it (the last A entry may be repeated there)
  const bool has_pa = a0 > 0, has_nb = b1 < nb;
  const int32_t pa_key = has_pa ? c.a_keys[a0 - 1] : 0;
  const int32_t nb_key = has_nb ? c.b_keys[b1] : 0;
  const long long nb_cnt = (has_nb && has_cnt) ? c.b_cnt[b1] : 0;
  __syncthreads();
  const int32_t *sA = s_key, *sB = s_key + na_t;
  const long long *cA = s_cnt, *cB = s_cnt + na_t;
  const unsigned diag = t * kMgVT < nt ? t * kMgVT : nt;
  unsigned ai = merge_path(sA, na_t, sB, nb_t, diag);
  unsigned bi = diag - ai;
  int32_t ok[kMgVT];
  long long oc[kMgVT];
  int32_t oa[SRC ? kMgVT : 1], ob[SRC ? kMgVT : 1];
  unsigned keep = 0;
#pragma unroll
  for (int j = 0; j < kMgVT; ++j) {
    ok[j] = 0;
    oc[j] = 0;
    if (SRC) oa[j] = ob[j] = -1;
    if (diag + j >= nt) continue;
    const bool hasA = ai < na_t, hasB = bi < nb_t;
    const int32_t ka = hasA ? sA[ai] : 0, kb = hasB ? sB[bi] : 0;
    if (hasA && (!hasB || ka <= kb)) {
      // the equal entry of B, if there is one, is the next entry of B
      const bool dupB = hasB ? kb == ka : (has_nb && bi == nb_t && nb_key == ka);
      ok[j] = ka;
      if (has_cnt) oc[j] = cA[ai] + (dupB ? (hasB ? cB[bi] : nb_cnt) : 0);
      if (SRC) {
        oa[j] = (int32_t)(a0 + ai);
        ob[j] = dupB ? (int32_t)(b0 + bi) : -1;
      }
      keep |= 1u << j;
      ++ai;
    } else {
      // dropped when the entry of A in front of it carries the same key
      const bool dup = ai > 0 ? sA[ai - 1] == kb : (has_pa && pa_key == kb);
      ok[j] = kb;
      if (has_cnt) oc[j] = cB[bi];
      if (SRC) ob[j] = (int32_t)(b0 + bi);
      if (!dup) keep |= 1u << j;
      ++bi;
    }
  }
  // offsets of the kept entries: thread -> wave -> workgroup -> tile (look-back)
  const unsigned mine = __popc(keep);
  unsigned inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(inc, off, 64);
    if (l >= (unsigned)off) inc += o;
  }
  if (l == 63) s_wsum[w] = inc;
  __syncthreads();  // (also: every thread is done reading s_key / s_cnt)
  unsigned wbase = 0, ttot = 0;
  for (unsigned q = 0; q < kMgBS / kWave; ++q) {
    if (q < w) wbase += s_wsum[q];
    ttot += s_wsum[q];
  }
  if (t == 0) {
    unsigned long long *my = c.status + tile;
    __hip_atomic_store(my, (tile == 0 ? kMgPrefix : kMgAgg) | ttot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long carry = 0;
    if (tile > 0) {
      unsigned tb = tile - 1;
      while (true) {
        const unsigned long long v = __hip_atomic_load(c.status + tb, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        const unsigned f = (unsigned)(v >> 62);
        if (f == 0) {
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        carry += v & kMgMask;
        if (f == 2) break;
        --tb;
      }
      __hip_atomic_store(my, kMgPrefix | (carry + ttot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_base = carry;
    if (d1 == total) *c.out_n = carry + ttot;
  }
  unsigned pos = wbase + inc - mine;
#pragma unroll
  for (int j = 0; j < kMgVT; ++j) {
    if (keep & (1u << j)) {
      s_key[pos] = ok[j];
      if (has_cnt) s_cnt[pos] = oc[j];
      ++pos;
    }
  }
  __syncthreads();
  const unsigned long long base = s_base;
  for (unsigned i = t; i < ttot; i += kMgBS) {
    c.out_keys[base + i] = s_key[i];
    if (has_cnt) c.out_cnt[base + i] = s_cnt[i];
  }
  if (SRC) {
    __syncthreads();
    pos = wbase + inc - mine;
#pragma unroll
    for (int j = 0; j < kMgVT; ++j) {
      if (keep & (1u << j)) {
        s_cnt[pos] = (long long)(((unsigned long long)(uint32_t)ob[j] << 32) | (uint32_t)oa[j]);
        ++pos;
      }
    }
    __syncthreads();
    for (unsigned i = t; i < ttot; i += kMgBS) {
      const unsigned long long v = (unsigned long long)s_cnt[i];
      c.src_a[base + i] = (int32_t)(uint32_t)v;
      c.src_b[base + i] = (int32_t)(uint32_t)(v >> 32);
    }
  }
}

// out[i, :] = op(A[src_a[i], :], B[src_b[i], :]) over rows of `width` values; a missing side
// (index -1) contributes the identity.  op: 0 add, 1 min, 2 max; NaN = "no value yet" for min / max
template <typename T, int OP>
__global__ __launch_bounds__(kBlock) void merge_payload_kernel(
    const int32_t *__restrict__ src_a, const int32_t *__restrict__ src_b, uint64_t n, unsigned width,
    const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out) {
  const uint64_t total = n * width;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
    const uint64_t r = i / width;
    const unsigned j = (unsigned)(i - r * width);
    const int32_t ia = src_a[r], ib = src_b[r];
    T v;
    if (ia >= 0 && ib >= 0) {
      const T x = a[(uint64_t)ia * width + j], y = b[(uint64_t)ib * width + j];
      if (OP == 0) v = x + y;
      else if (OP == 1) v = (y < x || x != x) ? y : x;
      else v = (y > x || x != x) ? y : x;
    } else {
      v = ia >= 0 ? a[(uint64_t)ia * width + j] : b[(uint64_t)ib * width + j];
    }
    out[i] = v;
  }
}

static inline uint64_t mg_pad16(uint64_t x) { return (x + 15) & ~15ull; }
static inline uint64_t mg_tiles(uint64_t total) { return (total + kMgTile - 1) / kMgTile; }
static inline uint64_t mg_col_ws(uint64_t total) {
  const uint64_t nt = mg_tiles(total);
  return mg_pad16((nt + 1) * 4) + nt * 8 + 16;
}

}  // namespace nvt

using namespace nvt;

extern "C" {

int nvt_merge_sorted_ws_bytes(const nvt_merge_col *cols, int ncols, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes && (ncols == 0 || cols), "null pointer");
  uint64_t tot = 64;
  for (int i = 0; i < ncols; ++i) tot += mg_col_ws(cols[i].na + cols[i].nb);
  *bytes = tot;
  return NVT_OK;
}

int nvt_merge_sorted_many(const nvt_merge_col *cols, int ncols, void *ws, uint64_t ws_bytes,
                          void *stream) {
  NVT_CHECK_ARG(ncols == 0 || (cols && ws), "null descriptors / workspace");
  hipStream_t s = (hipStream_t)stream;
  uint64_t need = 0;
  nvt_merge_sorted_ws_bytes(cols, ncols, &need);
  NVT_CHECK_ARG(ws_bytes >= need, "workspace smaller than nvt_merge_sorted_ws_bytes");
  NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "workspace must be 16-byte aligned");
  if (ncols == 0) return NVT_OK;
  NVT_CHECK_HIP(hipMemsetAsync(ws, 0, need, s));
  char *p = reinterpret_cast<char *>(ws);
  for (int c0 = 0; c0 < ncols; c0 += kMgMaxCols) {
    const int nc = ncols - c0 < kMgMaxCols ? ncols - c0 : kMgMaxCols;
    MergeBatch b;
    memset(&b, 0, sizeof(b));
    bool src = false;
    uint64_t bytes = 0;
    for (int i = 0; i < nc; ++i) {
      const nvt_merge_col &d = cols[c0 + i];
      const uint64_t total = d.na + d.nb;
      NVT_CHECK_ARG(total < (1ull << 31), "na + nb must be below 2^31");
      NVT_CHECK_ARG(d.out_n && (total == 0 || d.out_keys), "null output");
      NVT_CHECK_ARG((d.na == 0 || d.a_keys) && (d.nb == 0 || d.b_keys), "null key list");
      NVT_CHECK_ARG((d.a_counts != nullptr) == (d.b_counts != nullptr) || d.na == 0 || d.nb == 0,
                    "counts on both lists or on neither");
      NVT_CHECK_ARG((d.src_a != nullptr) == (d.src_b != nullptr), "src_a and src_b go together");
      const bool has_cnt = d.a_counts != nullptr || d.b_counts != nullptr;
      NVT_CHECK_ARG(!has_cnt || total == 0 || d.out_counts, "null out_counts");
      MergeCol &m = b.c[i];
      m.a_keys = d.a_keys;
      m.b_keys = d.b_keys;
      // (an empty list needs no counts of its own; the kernel only tests a_cnt)
      m.a_cnt = has_cnt ? (d.a_counts ? d.a_counts : d.b_counts) : nullptr;
      m.b_cnt = has_cnt ? (d.b_counts ? d.b_counts : d.a_counts) : nullptr;
      m.out_keys = d.out_keys;
      m.out_cnt = d.out_counts;
      m.src_a = d.src_a;
      m.src_b = d.src_b;
      m.out_n = reinterpret_cast<unsigned long long *>(d.out_n);
      m.na = (unsigned)d.na;
      m.nb = (unsigned)d.nb;
      const uint64_t nt = mg_tiles(total);
      m.splits = reinterpret_cast<unsigned *>(p);
      m.status = reinterpret_cast<unsigned long long *>(p + mg_pad16((nt + 1) * 4));
      m.ticket = reinterpret_cast<unsigned *>(m.status + nt);
      p += mg_col_ws(total);
      b.tile_start[i + 1] = b.tile_start[i] + (unsigned)nt;
      b.bnd_start[i + 1] = b.bnd_start[i] + (unsigned)nt + 1;
      src = src || d.src_a != nullptr;
      bytes += total * (has_cnt ? 12 : 4);
    }
    for (int i = 0; i < nc; ++i)
      NVT_CHECK_ARG(!src || cols[c0 + i].src_a, "src maps on every column of a call or on none");
    b.ncols = nc;
    NVT_PROF("merge_sorted", bytes, s);
    merge_split_kernel<<<(b.bnd_start[nc] + kMgBS - 1) / kMgBS, kMgBS, 0, s>>>(b);
    NVT_CHECK_LAUNCH();
    if (b.tile_start[nc] == 0) continue;
    if (src) merge_tile_kernel<true><<<b.tile_start[nc], kMgBS, 0, s>>>(b);
    else merge_tile_kernel<false><<<b.tile_start[nc], kMgBS, 0, s>>>(b);
    NVT_CHECK_LAUNCH();
  }
  return NVT_OK;
}

int nvt_merge_payload(const int32_t *src_a, const int32_t *src_b, uint64_t n, int width, int dtype,
                      int op, const void *a, const void *b, void *out, void *stream) {
  if (n == 0 || width == 0) return NVT_OK;
  NVT_CHECK_ARG(src_a && src_b && out && width > 0, "null pointer / width");
  NVT_CHECK_ARG(dtype == NVT_I64 || dtype == NVT_F64, "payload dtype must be int64 or float64");
  NVT_CHECK_ARG(op >= 0 && op <= 2, "op: 0 add, 1 min, 2 max");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("merge_sorted", n * (uint64_t)width * 8, s);
  const unsigned grid = stream_grid(n * (uint64_t)width, kBlock * 4);
#define NVT_MP(T, OP)                                                                      \
  merge_payload_kernel<T, OP><<<grid, kBlock, 0, s>>>(src_a, src_b, n, (unsigned)width,     \
                                                      (const T *)a, (const T *)b, (T *)out)
  if (dtype == NVT_I64) {
    if (op == 0) NVT_MP(long long, 0); else if (op == 1) NVT_MP(long long, 1); else NVT_MP(long long, 2);
  } else {
    if (op == 0) NVT_MP(double, 0); else if (op == 1) NVT_MP(double, 1); else NVT_MP(double, 2);
  }
#undef NVT_MP
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // extern "C"
