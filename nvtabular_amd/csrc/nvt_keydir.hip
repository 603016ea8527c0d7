// Key directory: the transform side of JoinGroupby / TargetEncoding on the sort path's groups
// (join_groupby.py:198-217, target_encoding.py:341-371: the reference's left merges on the key
// column) with ONE random line of HBM per row.
//
// Why: nvt_flat_lookup_image reads a {key, group} slot of the flat range table (8 bytes out of a
// 120 MB table at 5 M groups) and then the group's record (<= 64 bytes out of a 320 MB image) --
// two random lines per row, and the PMC passes show each costing a full 128-byte line of HBM
// traffic (profiles/r06_cfg4_pmc_traffic.json: 6.7 GB per launch for 0.5 GB of rows and outputs;
// the kernel runs AT the rate the fabric takes L2 misses -- 39 G/s next to the streams, 44-55 G/s
// in tools/micro/probe_rate.hip whatever the table size beyond the L2 -- with ~2.4 misses per row:
// the probe walks across 64-byte lines).  Here the key -> group step is ONE 16-byte read:
//   * the groups' keys are ascending (the sort path's group order): group g = position g;
//   * bucket b of the directory = {first, k0, k1, k2}: `first` = index of the first key that maps
//     to a bucket >= b under the monotone range map of nvt_range.hpp (range_map_params over
//     [first key, last key], B ~ groups buckets), k0..k2 = the bucket's first three keys (a
//     shorter bucket repeats its last key; an empty one holds the next key of the list, which
//     maps to another bucket and therefore never equals a key that asks here).
// A row whose key is k0 / k1 / k2 has its group from that read; a key above k2 walks the key list
// from first + 3 (2 % of the buckets hold more than three keys at one key per bucket; bisection
// when the keys cluster).  The map's parameters are recomputed by every workgroup from the first
// and the last key: no parameter block, no read-back of a displacement (the flat table's failure
// mode for clustered keys does not exist here), and laying the directory out is two passes over
// n + B words instead of a prefix-maximum scan, a table clear and a scatter.
//
// nvt_image_build writes the records whole: every operator's byte range of a tile of 256 records
// is assembled in LDS and leaves as full lines (the per-operator kernels nvt_jg_image /
// nvt_te_image each wrote 16-24 bytes of every 64-byte record: read-modify-write of the image per
// operator, 0.67 + 0.53 GB of traffic for a 0.32 GB image).
#include <cstring>

#include "nvt_common.hpp"
#include "nvt_image.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"
#include "nvt_range.hpp"

namespace nvt {

struct KeyMap {
  uint32_t ulo, span, mul;
  int sh;
  __device__ __forceinline__ uint32_t bucket(int32_t key) const {
    const uint32_t u = ukey(key);
    uint32_t d = u > ulo ? u - ulo : 0u;
    d = d < span ? d : span;
    return __umulhi(d << sh, mul);
  }
};

// every thread of the workgroup calls this (one barrier inside): the map over [klo, khi] onto B buckets
__device__ __forceinline__ KeyMap keymap_of(int32_t klo, int32_t khi, uint64_t B, uint32_t *lds4) {
  if (threadIdx.x == 0) {
    const uint64_t lo = ukey(klo), hi = ukey(khi);
    uint32_t mul;
    int sh;
    range_map_params(hi - lo, B, &mul, &sh);
    lds4[0] = (uint32_t)lo;
    lds4[1] = (uint32_t)(hi - lo);
    lds4[2] = mul;
    lds4[3] = (uint32_t)sh;
  }
  __syncthreads();
  KeyMap m;
  m.ulo = lds4[0];
  m.span = lds4[1];
  m.mul = lds4[2];
  m.sh = (int)lds4[3];
  return m;
}

// dir[b].first = first i with bucket(keys[i]) >= b, b = 0 .. B (dir[B].first = n).  Thread i writes
// the buckets in (bucket(keys[i - 1]), bucket(keys[i])]; thread n the ones behind the last key.
// Long gaps (clustered keys leave most buckets empty) are written by the whole wave.
__global__ __launch_bounds__(kBlock) void keydir_first_kernel(const int32_t *__restrict__ keys, uint64_t n,
                                                              uint64_t B, uint32_t *__restrict__ dir) {
  __shared__ uint32_t prm[4];
  const KeyMap m = keymap_of(keys[0], keys[n - 1], B, prm);
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  const uint64_t iters = (n + 1 + stride - 1) / stride;
  const unsigned l = lane_id();
  for (uint64_t it = 0; it < iters; ++it) {
    const uint64_t i = it * stride + (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    uint64_t from = 0, to = 0;  // buckets [from, to) receive i
    if (i <= n) {
      from = i == 0 ? 0 : (uint64_t)m.bucket(keys[i - 1]) + 1;
      to = i == n ? B + 1 : (uint64_t)m.bucket(keys[i]) + 1;
    }
    uint64_t len = to > from ? to - from : 0;
    if (len <= 4) {
      for (uint64_t b = from; b < to; ++b) dir[4 * b] = (uint32_t)i;
      len = 0;
    }
    unsigned long long todo = __ballot(len > 0);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      const uint64_t f = __shfl(from, src, 64), t = __shfl(to, src, 64), v = __shfl(i, src, 64);
      for (uint64_t b = f + l; b < t; b += kWave) dir[4 * b] = (uint32_t)v;
      todo &= todo - 1;
    }
  }
}

// the first three keys of every bucket beside its `first`
__global__ __launch_bounds__(kBlock) void keydir_keys_kernel(const int32_t *__restrict__ keys, uint64_t n,
                                                             uint64_t B, uint32_t *__restrict__ dir) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b <= B; b += stride) {
    const uint64_t first = dir[4 * b];
    const uint64_t next = b < B ? dir[4 * (b + 1)] : n;
    const uint64_t cnt = next - first;
    // (an empty bucket: the next key of the list -- or the last one -- maps to another bucket)
    const int32_t k0 = keys[first < n ? first : n - 1];
    const int32_t k1 = cnt > 1 ? keys[first + 1] : k0;
    const int32_t k2 = cnt > 2 ? keys[first + 2] : k1;
    dir[4 * b + 1] = (uint32_t)k0;
    dir[4 * b + 2] = (uint32_t)k1;
    dir[4 * b + 3] = (uint32_t)k2;
  }
}

// position of `k` in the ascending list keys[0 .. nkeys) through its bucket's {first, k0, k1, k2};
// -1: no such key.
constexpr int kDirSteps = 4;
__device__ __forceinline__ int64_t keydir_find(const int32_t *__restrict__ keys, uint64_t nkeys,
                                               const uint32_t *__restrict__ dir, uint32_t b, int32_t k) {
  const uint4 e = *reinterpret_cast<const uint4 *>(dir + 4ull * b);
  if ((int32_t)e.y == k) return (int64_t)e.x;
  if ((int32_t)e.z == k) return (int64_t)e.x + 1;
  if ((int32_t)e.w == k) return (int64_t)e.x + 2;
  const uint32_t uk = ukey(k);
  if (ukey((int32_t)e.w) > uk) return -1;  // (ascending: the bucket's other keys are above k2)
  // a bucket with more than three keys, or an unseen key above the bucket's keys: the list
  uint64_t j = (uint64_t)e.x + 3;
#pragma unroll 1
  for (int step = 0; step < kDirSteps; ++step, ++j) {
    if (j >= nkeys) return -1;
    const int32_t kj = keys[j];
    if (ukey(kj) >= uk) return kj == k ? (int64_t)j : -1;
  }
  // a crowded bucket (keys that cluster in their range): bisection over its keys
  uint64_t lo = j, hi = dir[4ull * (b + 1)];  // the first key >= k is in [lo, hi]
  hi = hi < nkeys ? hi : nkeys;
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if (ukey(keys[mid]) < uk) lo = mid + 1; else hi = mid;
  }
  return (lo < nkeys && keys[lo] == k) ? (int64_t)lo : -1;
}

__device__ __forceinline__ uint32_t word_of(const uint4 &v, unsigned w) {  // (w is wave-uniform)
  return w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w;
}

template <typename K, int MAXC>
__global__ __launch_bounds__(kBlock) void keydir_lookup_image_kernel(
    const K *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n,
    const uint32_t *__restrict__ dir, uint64_t B, const int32_t *__restrict__ keys32, uint64_t nkeys,
    int64_t offset, int64_t null_group, const uint8_t *__restrict__ image, uint32_t stride_bytes,
    int ncols, ImageOuts o, unsigned long long *unseen) {
  constexpr int MAXG = MAXC / 2;
  __shared__ uint32_t prm[4];
  const KeyMap m = keymap_of(keys32[0], keys32[nkeys - 1], B, prm);
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  bool any_unseen = false;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int64_t g = -1, kv;
    if (!bit_valid(valid, i)) {
      g = null_group;  // null keys are one group (groupby dropna=False)
    } else if (!__builtin_sub_overflow((int64_t)keys[i], offset, &kv) && kv >= (int64_t)INT32_MIN &&
               kv <= (int64_t)INT32_MAX) {
      g = keydir_find(keys32, nkeys, dir, m.bucket((int32_t)kv), (int32_t)kv);
    }
    any_unseen |= g < 0;
    const uint8_t *rec = image + (uint64_t)(g < 0 ? 0 : g) * stride_bytes;
    // all loads first: the shared windows, then the outputs that have a load of their own
    uint4 G[MAXG];
#pragma unroll
    for (int q = 0; q < MAXG; ++q)
      if (q < o.ngroups) G[q] = *reinterpret_cast<const uint4 *>(rec + o.goff[q]);
    uint64_t x[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < ncols && o.grp[c] == 0xFF) {
        uint32_t at = o.off[c];
        if (o.fold[c]) at += (1u + (uint32_t)o.fold[c][i]) * o.fstride[c];
        x[c] = o.size[c] == 8 ? *reinterpret_cast<const uint64_t *>(rec + at)
                              : (uint64_t)*reinterpret_cast<const uint32_t *>(rec + at);
      }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < ncols && o.grp[c] != 0xFF) {
        uint4 v = G[0];
#pragma unroll
        for (int q = 1; q < MAXG; ++q)
          if (o.grp[c] == q) v = G[q];
        const unsigned w = o.word[c];
        x[c] = word_of(v, w);
        if (o.size[c] == 8) x[c] |= (uint64_t)word_of(v, w + 1) << 32;
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

// ---- whole records in one pass ------------------------------------------------------------------
constexpr int kBuildMaxParts = 4;
constexpr int kBuildTile = 256;         // records per tile (a thread per record)
constexpr int kBuildMaxStride = 192;    // bytes: the tile (padded to an odd number of words per record) fits 64 KiB
struct TePart {
  const int64_t *tot_count, *fold_count;
  const double *tot_sum, *fold_sum;
  unsigned kfold;
  int out_dtype;  // NVT_F32 / NVT_F64
  uint32_t off;
  double p, y_mean;
  const double *moments;  // {count, sum} of the target on the device (nullptr: y_mean is the number)
};
struct BuildParts {
  int nparts;
  int kind[kBuildMaxParts];          // 0: JoinGroupby statistics, 1: TargetEncoding values
  uint64_t groups[kBuildMaxParts];   // records the part fills (the others keep zero bytes there)
  int jg_ncols[kBuildMaxParts];
  JgImageArgs jg[kBuildMaxParts];
  TePart te[kBuildMaxParts];
};

__global__ __launch_bounds__(kBuildTile) void image_build_kernel(
    BuildParts P, uint64_t records, uint32_t *__restrict__ image, uint32_t stride_words) {
  extern __shared__ uint32_t tile[];  // [kBuildTile][stride_words + 1]: a thread per record hits 64 banks
  const uint32_t sw = stride_words + 1;
  const uint64_t ntiles = (records + kBuildTile - 1) / kBuildTile;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t g0 = t * kBuildTile;
    const unsigned cnt = (unsigned)(records - g0 < (uint64_t)kBuildTile ? records - g0 : kBuildTile);
    for (unsigned e = threadIdx.x; e < kBuildTile * sw; e += kBuildTile) tile[e] = 0u;
    __syncthreads();
    for (int q = 0; q < P.nparts; ++q) {
      if (P.kind[q] == 0) {
        const uint64_t g = g0 + threadIdx.x;
        if (threadIdx.x < cnt && g < P.groups[q]) {
          const JgImageArgs &a = P.jg[q];
          const int64_t ni = a.count[g];
          for (int c = 0; c < P.jg_ncols[q]; ++c) {
            bool is_int;
            const double x = jg_stat(a, c, g, ni, &is_int);
            uint32_t *at = tile + threadIdx.x * sw + a.off[c] / 4;
            switch (a.dst_dtype[c]) {
              case NVT_F32: at[0] = __float_as_uint(is_int ? (float)ni : (float)x); break;
              case NVT_I32: at[0] = (uint32_t)(is_int ? (int32_t)ni : (int32_t)x); break;
              case NVT_F64: {
                const unsigned long long b = (unsigned long long)__double_as_longlong(is_int ? (double)ni : x);
                at[0] = (uint32_t)b;
                at[1] = (uint32_t)(b >> 32);
                break;
              }
              default: {
                const unsigned long long b = (unsigned long long)(is_int ? ni : (int64_t)x);
                at[0] = (uint32_t)b;
                at[1] = (uint32_t)(b >> 32);
              }
            }
          }
        }
      } else {
        // a thread per value: the fold arrays are read in memory order
        const TePart &a = P.te[q];
        const unsigned per = a.kfold + 1;
        const double y_mean = a.moments ? __ddiv_rn(a.moments[1], a.moments[0]) : a.y_mean;
        const uint64_t gend = P.groups[q] < g0 + cnt ? P.groups[q] : g0 + cnt;
        const unsigned vals = gend > g0 ? (unsigned)(gend - g0) * per : 0u;
        for (unsigned e = threadIdx.x; e < vals; e += kBuildTile) {
          const unsigned r = e / per, slot = e - r * per;
          const double v = te_value(a.tot_count, a.tot_sum, a.fold_count, a.fold_sum, a.kfold, g0 + r, slot,
                                    a.p, y_mean);
          uint32_t *rec = tile + r * sw + a.off / 4;
          if (a.out_dtype == NVT_F32) {
            rec[slot] = __float_as_uint((float)v);
          } else {
            const unsigned long long b = (unsigned long long)__double_as_longlong(v);
            rec[2 * slot] = (uint32_t)b;
            rec[2 * slot + 1] = (uint32_t)(b >> 32);
          }
        }
      }
    }
    __syncthreads();
    // the tile leaves as one contiguous run of cnt * stride_words words
    uint32_t *dst = image + g0 * stride_words;
    const unsigned total = cnt * stride_words;
    for (unsigned e = threadIdx.x; e < total; e += kBuildTile) {
      const unsigned r = e / stride_words, w = e - r * stride_words;
      __builtin_nontemporal_store(tile[r * sw + w], dst + e);
    }
    __syncthreads();
  }
}

}  // namespace nvt

using namespace nvt;

extern "C" {

int nvt_keydir_build(const int32_t *keys32, uint64_t n, uint64_t dir_slots, uint32_t *dir, void *stream) {
  NVT_CHECK_ARG(keys32 && dir, "null pointer");
  NVT_CHECK_ARG(n >= 1 && n < (1ull << 32) - 1, "1 .. 2^32-2 keys");
  NVT_CHECK_ARG(dir_slots >= 1 && dir_slots < (1ull << 32) - 1, "1 .. 2^32-2 buckets");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_index", n * 4ull + dir_slots * 16ull, s);
  NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(dir) & 15) == 0, "dir must be 16-byte aligned");
  keydir_first_kernel<<<stream_grid(n + 1, kBlock, 8), kBlock, 0, s>>>(keys32, n, dir_slots, dir);
  keydir_keys_kernel<<<stream_grid(dir_slots + 1, kBlock, 8), kBlock, 0, s>>>(keys32, n, dir_slots, dir);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_keydir_lookup_image(const void *keys, int dtype, const uint8_t *valid, uint64_t n,
                            const uint32_t *dir, uint64_t dir_slots, const int32_t *keys32, uint64_t nkeys,
                            int64_t key_offset, int64_t null_group, const void *image,
                            uint32_t stride_bytes, int ncols,
                            void *const *outs, const uint8_t *const *folds, const uint32_t *offs,
                            const uint32_t *sizes, const uint64_t *miss_bits, uint64_t *unseen,
                            void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(keys && dir && keys32 && image && outs && offs && sizes && miss_bits, "null pointer");
  NVT_CHECK_ARG(nkeys >= 1 && nkeys < (1ull << 32) - 1 && dir_slots >= 1, "1 .. 2^32-2 keys, >= 1 bucket");
  NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(dir) & 15) == 0, "dir must be 16-byte aligned");
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
    NVT_CHECK_ARG((uint64_t)offs[c] + sizes[c] <= stride_bytes, "value outside the record");
    o.grp[c] = 0xFF;
  }
  // fixed-offset outputs that share an aligned 16-byte window of the record: one load per window
  // (windows are 16-byte aligned in memory when the stride and the image are)
  if (stride_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(image) & 15) == 0) {
    const int maxg = (ncols <= 2 ? 2 : ncols <= 4 ? 4 : ncols <= 8 ? 8 : ncols <= 16 ? 16 : 24) / 2;
    for (int c = 0; c < ncols; ++c) {
      if (o.fold[c] || o.grp[c] != 0xFF) continue;
      const uint32_t win = offs[c] / 16;
      int members = 0;
      for (int d = c; d < ncols; ++d) members += (!o.fold[d] && offs[d] / 16 == win) ? 1 : 0;
      if (members < 2 || o.ngroups >= maxg) continue;
      const int q = o.ngroups++;
      o.goff[q] = win * 16;
      for (int d = c; d < ncols; ++d) {
        if (!o.fold[d] && offs[d] / 16 == win) {
          o.grp[d] = (uint8_t)q;
          o.word[d] = (uint8_t)((offs[d] % 16) / 4);
        }
      }
    }
  }
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_lookup", n * (dtype == NVT_I64 ? 8ull : 4ull), s);
  const unsigned grid = stream_grid(n, kBlock * 2);
  unsigned long long *flag = reinterpret_cast<unsigned long long *>(unseen);
  const uint8_t *img = reinterpret_cast<const uint8_t *>(image);
#define NVT_IMG(K, MAXC)                                                                              \
  keydir_lookup_image_kernel<K, MAXC><<<grid, kBlock, 0, s>>>((const K *)keys, valid, n, dir, dir_slots, \
                                                              keys32, nkeys, key_offset, null_group,  \
                                                              img, stride_bytes, ncols, o, flag)
#define NVT_IMG_K(K)                      \
  do {                                    \
    if (ncols <= 2) NVT_IMG(K, 2);        \
    else if (ncols <= 4) NVT_IMG(K, 4);   \
    else if (ncols <= 8) NVT_IMG(K, 8);   \
    else if (ncols <= 16) NVT_IMG(K, 16); \
    else NVT_IMG(K, 24);                  \
  } while (0)
  if (dtype == NVT_I32) NVT_IMG_K(int32_t);
  else NVT_IMG_K(int64_t);
#undef NVT_IMG_K
#undef NVT_IMG
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_image_build(const nvt_image_part *parts, int nparts, uint64_t records, void *image,
                    uint32_t stride_bytes, void *stream) {
  if (records == 0) return NVT_OK;
  NVT_CHECK_ARG(image && (parts || nparts == 0), "null pointer");
  NVT_CHECK_ARG(nparts >= 0 && nparts <= kBuildMaxParts, "0..4 parts");
  NVT_CHECK_ARG(stride_bytes >= 8 && stride_bytes % 8 == 0 && stride_bytes <= (uint32_t)kBuildMaxStride,
                "record stride: a multiple of 8 bytes, <= 192");
  BuildParts P;
  memset(&P, 0, sizeof(P));
  P.nparts = nparts;
  uint64_t in_bytes = 0;
  for (int q = 0; q < nparts; ++q) {
    const nvt_image_part &p = parts[q];
    NVT_CHECK_ARG(p.kind == NVT_IMAGE_PART_JG || p.kind == NVT_IMAGE_PART_TE, "part kind 0 / 1");
    NVT_CHECK_ARG(p.groups <= records, "a part with more groups than records");
    P.kind[q] = p.kind;
    P.groups[q] = p.groups;
    if (p.kind == NVT_IMAGE_PART_JG) {
      NVT_CHECK_ARG(p.count && p.kinds && p.vals && p.dst_dtypes && p.offs, "null pointer");
      NVT_CHECK_ARG(p.ncols >= 1 && p.ncols <= kImageMaxCols, "1..24 columns");
      NVT_CHECK_ARG(p.nvals >= 0 && p.nvals <= kJgMaxVals, "0..8 value columns");
      JgImageArgs &a = P.jg[q];
      a.count = p.count;
      for (int j = 0; j < p.nvals; ++j) {
        a.sum[j] = p.sum ? p.sum[j] : nullptr;
        a.sumsq[j] = p.sumsq ? p.sumsq[j] : nullptr;
        a.mn[j] = p.mn ? p.mn[j] : nullptr;
        a.mx[j] = p.mx ? p.mx[j] : nullptr;
      }
      for (int c = 0; c < p.ncols; ++c) {
        const int k = p.kinds[c], j = p.vals[c], d = p.dst_dtypes[c];
        NVT_CHECK_ARG(k >= 0 && k <= 6, "statistic kind 0..6");
        NVT_CHECK_ARG(k == 0 || (j >= 0 && j < p.nvals), "value column out of range");
        NVT_CHECK_ARG(k == 0 || a.sum[j] || k == 3 || k == 4, "null sum array");
        NVT_CHECK_ARG((k != 3 || a.mn[j]) && (k != 4 || a.mx[j]) && (k < 5 || (a.sum[j] && a.sumsq[j])),
                      "null accumulator array for a requested statistic");
        NVT_CHECK_ARG(d == NVT_F32 || d == NVT_F64 || d == NVT_I32 || d == NVT_I64, "values are f32 / f64 / i32 / i64");
        const uint32_t sz = (d == NVT_F32 || d == NVT_I32) ? 4u : 8u;
        NVT_CHECK_ARG(p.offs[c] % sz == 0 && (uint64_t)p.offs[c] + sz <= stride_bytes,
                      "value outside the record");
        a.kind[c] = k;
        a.val[c] = k == 0 ? 0 : j;
        a.dst_dtype[c] = d;
        a.off[c] = p.offs[c];
      }
      P.jg_ncols[q] = p.ncols;
      in_bytes += p.groups * 8ull * (1 + p.nvals);
    } else {
      NVT_CHECK_ARG(p.tot_count && p.tot_sum, "null pointer");
      NVT_CHECK_ARG(p.kfold >= 0 && p.kfold <= 256, "kfold must be 0 (no folds) .. 256");
      NVT_CHECK_ARG(p.kfold == 0 || (p.fold_count && p.fold_sum), "fold statistics come with kfold > 0");
      NVT_CHECK_ARG(p.out_dtype == NVT_F32 || p.out_dtype == NVT_F64, "out dtype must be f32 / f64");
      const uint32_t sz = p.out_dtype == NVT_F32 ? 4u : 8u;
      NVT_CHECK_ARG(p.offset % sz == 0 && (uint64_t)p.offset + (uint64_t)(p.kfold + 1) * sz <= stride_bytes,
                    "values outside the record");
      TePart &a = P.te[q];
      a.tot_count = p.tot_count;
      a.tot_sum = p.tot_sum;
      a.fold_count = p.fold_count;
      a.fold_sum = p.fold_sum;
      a.kfold = (unsigned)p.kfold;
      a.out_dtype = p.out_dtype;
      a.off = p.offset;
      a.p = p.p_smooth;
      a.y_mean = p.y_mean;
      a.moments = p.moments;
      in_bytes += p.groups * 16ull * (p.kfold + 1);
    }
  }
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("groupby_index", in_bytes + records * stride_bytes, s);
  const uint32_t sw = stride_bytes / 4;
  const size_t lds = (size_t)kBuildTile * (sw + 1) * 4;
  const uint64_t ntiles = (records + kBuildTile - 1) / kBuildTile;
  const unsigned grid = (unsigned)(ntiles < 256ull * 8 ? ntiles : 256ull * 8);
  image_build_kernel<<<grid, kBuildTile, lds, s>>>(P, records, reinterpret_cast<uint32_t *>(image), sw);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // extern "C"
