// Continuous-column kernels: Normalize.fit moments, NormalizeMinMax.fit,
// fused FillMissing + Normalize.transform, and the row-wise joins of
// JoinGroupby / TargetEncoding.  All are pure streaming passes (HBM-bound):
// 16-byte loads per lane, fp64 arithmetic, 16/32-byte stores.
//
// Reference: moments.py:64-116, normalize.py:71-90,150-186, fill.py:49-57,
// join_groupby.py:175-217, target_encoding.py:340-374.
#include <limits>
#include <type_traits>
#include <vector>

#include "nvt_common.hpp"
#include "nvt_prof.hpp"

namespace nvt {

constexpr unsigned kReduceGrid = 1024;  // fixed so partial sums combine in a fixed order

template <typename T>
struct VecOf {
  static constexpr int n = 16 / sizeof(T);
};

template <typename T>
__device__ __forceinline__ void load_vec(const T *p, T (&v)[VecOf<T>::n]) {
  // (non-temporal: a column is streamed once per kernel; moments 0.40 -> 0.377 ms, fill +
  // normalize 1.49 -> 1.44 ms for the 13 Criteo columns.  Non-temporal STORES of the float64
  // output made fill + normalize slower: 1.47 -> 1.74 ms)
#ifdef NVT_CONT_PLAIN_LOAD
  int4 raw = *reinterpret_cast<const int4 *>(p);
#else
  typedef int v4i_ntl __attribute__((ext_vector_type(4)));
  v4i_ntl raw = __builtin_nontemporal_load(reinterpret_cast<const v4i_ntl *>(p));
#endif
  memcpy(v, &raw, 16);
}

// ---------------------------------------------------------------------------
// moments: per-block partial {count, sum, sumsq} -> deterministic final reduce
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void moments_body(const T *__restrict__ x,
                                             const uint8_t *__restrict__ valid, uint64_t n,
                                             int has_fill, double fill_val,
                                             double *__restrict__ partials) {
  constexpr int VEC = VecOf<T>::n;
  double cnt = 0, sum = 0, sq = 0;
  auto acc = [&](T raw, bool ok) {
    double v = (double)raw;
    if (!ok || is_nan(raw)) {
      if (!has_fill) return;
      v = fill_val;
    }
    cnt += 1.0;
    sum += v;
    sq += v * v;
  };
  const uint64_t nvec = n / VEC;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  // 4 independent 16-byte loads (+ bitmap bytes) in flight per lane: with one load per
  // iteration the kernel ran at 2.9 TB/s, latency-bound
  constexpr int U = 4;
  for (uint64_t i0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i0 < nvec; i0 += stride * U) {
    T v[U][VEC];
    unsigned vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t i = i0 + (uint64_t)u * stride;
      vb[u] = 0;
      if (i < nvec) {
        load_vec<T>(x + i * VEC, v[u]);
        vb[u] = 0x100u | (valid != nullptr ? (unsigned)valid[(i * VEC) >> 3] : 0xFFu);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!vb[u]) continue;
      const uint64_t row = (i0 + (uint64_t)u * stride) * VEC;
      const unsigned vbits = (vb[u] & 0xFFu) >> (row & 7);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc(v[u][j], (vbits >> j) & 1);
    }
  }
  for (uint64_t i = nvec * VEC + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    acc(x[i], bit_valid(valid, i));

  __shared__ double red[3][kBlock / kWave];
  cnt = wave_sum(cnt);
  sum = wave_sum(sum);
  sq = wave_sum(sq);
  const unsigned w = threadIdx.x / kWave;
  if (lane_id() == 0) {
    red[0][w] = cnt;
    red[1][w] = sum;
    red[2][w] = sq;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0;
    for (int k = 0; k < kBlock / kWave; ++k) t += red[threadIdx.x][k];
    partials[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void moments_kernel(const T *__restrict__ x,
                                                         const uint8_t *__restrict__ valid,
                                                         uint64_t n, int has_fill, double fill_val,
                                                         double *__restrict__ partials) {
  moments_body<T>(x, valid, n, has_fill, fill_val, partials);
}

// Batched form (nvt_moments_many): blockIdx.y = column; every column of a Normalize.fit
// partition in ONE launch (the per-column version was 26 launches per partition for Criteo's
// 13 continuous columns).  Same grid.x and block-to-row mapping as the single-column kernel,
// so the partial sums -- and therefore the results -- are bit-identical to it.
constexpr int kBatchCols = 32;
struct MomCol {
  const void *x;
  const uint8_t *valid;
  uint64_t n;
  double fill_val;
  double *out3;
  int dtype, has_fill;
};
struct MomBatch {
  MomCol c[kBatchCols];
};
__global__ __launch_bounds__(kBlock) void moments_many_kernel(MomBatch b, double *partials) {
  const MomCol &c = b.c[blockIdx.y];
  double *p = partials + (uint64_t)blockIdx.y * 3 * gridDim.x;
  switch (c.dtype) {
    case NVT_F32: moments_body<float>((const float *)c.x, c.valid, c.n, c.has_fill, c.fill_val, p); break;
    case NVT_F64: moments_body<double>((const double *)c.x, c.valid, c.n, c.has_fill, c.fill_val, p); break;
    case NVT_I32: moments_body<int32_t>((const int32_t *)c.x, c.valid, c.n, c.has_fill, c.fill_val, p); break;
    default: moments_body<int64_t>((const int64_t *)c.x, c.valid, c.n, c.has_fill, c.fill_val, p); break;
  }
}

__device__ __forceinline__ void moments_final_body(const double *__restrict__ partials,
                                                   unsigned nblocks, double *out3) {
  __shared__ double red[kBlock / kWave];
  for (int q = 0; q < 3; ++q) {
    double t = 0;
    for (unsigned i = threadIdx.x; i < nblocks; i += kBlock) t += partials[(uint64_t)q * nblocks + i];
    t = wave_sum(t);
    if (lane_id() == 0) red[threadIdx.x / kWave] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = 0;
      for (int k = 0; k < kBlock / kWave; ++k) s += red[k];
      out3[q] += s;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(kBlock) void moments_final_kernel(const double *__restrict__ partials,
                                                               unsigned nblocks, double *out3) {
  moments_final_body(partials, nblocks, out3);
}
__global__ __launch_bounds__(kBlock) void moments_final_many_kernel(MomBatch b,
                                                                    const double *__restrict__ partials,
                                                                    unsigned nblocks) {
  moments_final_body(partials + (uint64_t)blockIdx.x * 3 * nblocks, nblocks, b.c[blockIdx.x].out3);
}

// ---------------------------------------------------------------------------
// min / max
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void minmax_kernel(const T *__restrict__ x,
                                                        const uint8_t *__restrict__ valid,
                                                        uint64_t n, double *__restrict__ partials) {
  constexpr int VEC = VecOf<T>::n;
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  double mn = qnan, mx = qnan;
  auto acc = [&](T raw, bool ok) {
    if (!ok || is_nan(raw)) return;
    double v = (double)raw;
    mn = (v < mn || mn != mn) ? v : mn;
    mx = (v > mx || mx != mx) ? v : mx;
  };
  const uint64_t nvec = n / VEC;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    T v[VEC];
    load_vec<T>(x + i * VEC, v);
    unsigned vbits = 0xF;
    if (valid != nullptr) {
      uint64_t row = i * VEC;
      vbits = valid[row >> 3] >> (row & 7);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc(v[j], (vbits >> j) & 1);
  }
  for (uint64_t i = nvec * VEC + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    acc(x[i], bit_valid(valid, i));
  __shared__ double red[2][kBlock / kWave];
  mn = wave_min(mn);
  mx = wave_max(mx);
  const unsigned w = threadIdx.x / kWave;
  if (lane_id() == 0) {
    red[0][w] = mn;
    red[1][w] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = qnan, b = qnan;
    for (int k = 0; k < kBlock / kWave; ++k) {
      double p = red[0][k], q = red[1][k];
      a = (p < a || a != a) ? p : a;
      b = (q > b || b != b) ? q : b;
    }
    partials[blockIdx.x] = a;
    partials[gridDim.x + blockIdx.x] = b;
  }
}
__global__ void minmax_final_kernel(const double *__restrict__ partials, unsigned nblocks,
                                    int accumulate, double *out2) {
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  double mn = qnan, mx = qnan;
  for (unsigned i = threadIdx.x; i < nblocks; i += kWave) {
    double p = partials[i], q = partials[nblocks + i];
    mn = (p < mn || mn != mn) ? p : mn;
    mx = (q > mx || mx != mx) ? q : mx;
  }
  mn = wave_min(mn);
  mx = wave_max(mx);
  if (threadIdx.x == 0) {
    if (accumulate) {
      double p = out2[0], q = out2[1];
      mn = (p < mn || mn != mn) ? p : mn;
      mx = (q > mx || mx != mx) ? q : mx;
    }
    out2[0] = mn;
    out2[1] = mx;
  }
}

// ---------------------------------------------------------------------------
// 4-byte values in, 8-byte values out (int32 / float32 -> float64 / int64: the Criteo continuous
// columns through FillMissing >> Normalize, Clip, LogOp): with a 16-byte load per lane a lane
// owns 32 output bytes and each of its two 16-byte stores writes every other 16 bytes of the
// wave's 2 KiB -- half-written lines per store instruction, 4.7-4.9 TB/s for this 4-in / 8-out
// stream however the grid is cut.  Here a lane takes TWO elements per run of 128 (an 8-byte load,
// one 16-byte store): every store instruction of a wave covers 1024 contiguous bytes, 5.6-6.0
// TB/s for the same stream (tools/micro/stream_ratio.hip: the store shape, not the grid, the
// unroll or the load width, is what separates the two).  f(raw, valid bit, was_null&) -> OUT.
// ---------------------------------------------------------------------------
template <typename T, typename OUT, typename F>
__device__ __forceinline__ void widen_stream(const T *__restrict__ x, const uint8_t *__restrict__ valid,
                                             uint64_t n, OUT *__restrict__ out,
                                             uint8_t *__restrict__ filled, F &&f) {
  static_assert(sizeof(T) == 4 && sizeof(OUT) == 8, "4-byte values in, 8-byte values out");
  constexpr int RUN = 2 * kWave, U = 4;
  typedef int v2i_nt __attribute__((ext_vector_type(2)));
  typedef int v4i_nt __attribute__((ext_vector_type(4)));
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  const unsigned lane = threadIdx.x & (kWave - 1);
  const uint64_t nruns = n / RUN;
  // (the wave index through readfirstlane: run numbers, bounds and base addresses are scalar)
  const uint64_t wave = (uint64_t)blockIdx.x * (kBlock / kWave) +
                        (uint64_t)__builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const uint64_t nwaves = stride / kWave;
  for (uint64_t r0 = wave * U; r0 < nruns; r0 += nwaves * U) {
    v2i_nt raw[U];
    unsigned vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      raw[u] = v2i_nt{0, 0};
      vb[u] = 3u;
      if (r0 + u < nruns) {
        const uint64_t e = (r0 + u) * RUN + 2 * lane;
        raw[u] = __builtin_nontemporal_load(reinterpret_cast<const v2i_nt *>(x + e));
        if (valid != nullptr) vb[u] = (unsigned)valid[e >> 3] >> (e & 7);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r0 + u >= nruns) break;
      const uint64_t e = (r0 + u) * RUN + 2 * lane;
      T v[2];
      memcpy(v, &raw[u], 8);
      OUT r[2];
      uint8_t m[2];
      r[0] = f(v[0], (bool)(vb[u] & 1), m[0]);
      r[1] = f(v[1], (bool)((vb[u] >> 1) & 1), m[1]);
      v4i_nt o;
      memcpy(&o, r, 16);
      __builtin_nontemporal_store(o, reinterpret_cast<v4i_nt *>(out + e));
      if (filled != nullptr) {
        uint16_t pk;
        memcpy(&pk, m, 2);
        *reinterpret_cast<uint16_t *>(filled + e) = pk;
      }
    }
  }
  for (uint64_t i = nruns * RUN + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    uint8_t m;
    out[i] = f(x[i], bit_valid(valid, i), m);
    if (filled != nullptr) filled[i] = m;
  }
}

// ---------------------------------------------------------------------------
// fused FillMissing + Normalize
// ---------------------------------------------------------------------------
template <typename T, typename OUT>
__device__ __forceinline__ void fill_norm_body(
    const T *__restrict__ x, const uint8_t *__restrict__ valid, uint64_t n, int has_fill,
    double fill_val, int do_norm, double shift, double scale, OUT *__restrict__ out,
    uint8_t *__restrict__ filled) {
  constexpr int VEC = VecOf<T>::n;
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  const double inv_is_div = scale > 0 ? 1.0 : 0.0;
  auto f = [&](T raw, bool ok, uint8_t &was_null) -> OUT {
    bool isnull = !ok || is_nan(raw);
    was_null = isnull ? 1 : 0;
    if (!do_norm) {
      // pure fill: stay in the output type (exact for integers)
      if (isnull) {
        if (has_fill) return (OUT)fill_val;
        if constexpr (std::is_floating_point<OUT>::value) return (OUT)qnan;
        return (OUT)0;
      }
      return (OUT)raw;
    }
    if constexpr (std::is_same<T, float>::value) {
      // a float32 column stays float32 through pandas' `(values - mean) / std` (the Python
      // float operands are weak scalars, normalize.py:79-84) and is widened afterwards: the
      // same two correctly rounded fp32 operations here give the reference's bits
      float v = isnull ? (has_fill ? (float)fill_val : std::numeric_limits<float>::quiet_NaN()) : raw;
      v = v - (float)shift;
      if (inv_is_div != 0.0) v = v / (float)scale;
      return (OUT)v;
    }
    double v = isnull ? (has_fill ? fill_val : qnan) : (double)raw;
    v -= shift;
    if (inv_is_div != 0.0) v /= scale;
    return (OUT)v;
  };
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  if constexpr (sizeof(T) == 4 && sizeof(OUT) == 8) {
    widen_stream<T, OUT>(x, valid, n, out, filled, f);
    return;
  }
  const uint64_t nvec = n / VEC;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    T v[VEC];
    load_vec<T>(x + i * VEC, v);
    unsigned vbits = 0xF;
    if (valid != nullptr) {
      uint64_t row = i * VEC;
      vbits = valid[row >> 3] >> (row & 7);
    }
    OUT r[VEC];
    uint8_t m[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) r[j] = f(v[j], (vbits >> j) & 1, m[j]);
    OUT *dst = out + i * VEC;
    constexpr int OB = VEC * (int)sizeof(OUT);
    if constexpr (OB == 32) {
      int4 a, b;
      memcpy(&a, &r[0], 16);
      memcpy(&b, &r[VEC / 2], 16);
      reinterpret_cast<int4 *>(dst)[0] = a;
      reinterpret_cast<int4 *>(dst)[1] = b;
    } else if constexpr (OB == 16) {
      int4 a;
      memcpy(&a, &r[0], 16);
      reinterpret_cast<int4 *>(dst)[0] = a;
    } else {
      int2 a;
      memcpy(&a, &r[0], 8);
      reinterpret_cast<int2 *>(dst)[0] = a;
    }
    if (filled != nullptr) {
      if constexpr (VEC == 4) {
        uint32_t pk;
        memcpy(&pk, m, 4);
        reinterpret_cast<uint32_t *>(filled)[i] = pk;
      } else {
        uint16_t pk;
        memcpy(&pk, m, 2);
        reinterpret_cast<uint16_t *>(filled)[i] = pk;
      }
    }
  }
  for (uint64_t i = nvec * VEC + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    uint8_t m;
    out[i] = f(x[i], bit_valid(valid, i), m);
    if (filled != nullptr) filled[i] = m;
  }
}

template <typename T, typename OUT>
__global__ __launch_bounds__(kBlock) void fill_norm_kernel(
    const T *__restrict__ x, const uint8_t *__restrict__ valid, uint64_t n, int has_fill,
    double fill_val, int do_norm, double shift, double scale, OUT *__restrict__ out,
    uint8_t *__restrict__ filled) {
  fill_norm_body<T, OUT>(x, valid, n, has_fill, fill_val, do_norm, shift, scale, out, filled);
}

// Batched form (nvt_fill_normalize_many): blockIdx.y = column, one launch per (input dtype,
// output dtype) combination of a FillMissing >> Normalize.transform partition.
struct FnCol {
  const void *x;
  const uint8_t *valid;
  uint64_t n;
  double fill_val, shift, scale;
  void *out;
  uint8_t *filled;
  int has_fill, do_norm;
  const double *moments;  // {count, sum, sum of squares} on the device (nullptr: shift / scale are the numbers)
};

// (mean, std) from {count, sum, sum of squares} as the host finishes them (moments.py:89-116 /
// ops/normalize.py finalize_moments): every operation rounded once, in the same order
__device__ __forceinline__ void finish_moments(const double *__restrict__ m, double *mean, double *std) {
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  const double n = m[0], s = m[1], s2 = m[2];
  if (n == 0.0) {
    *mean = qnan;
    *std = qnan;
    return;
  }
  double var = __dsub_rn(s2, __ddiv_rn(__dmul_rn(s, s), n));
  const double dn = __dsub_rn(n, 1.0);
  var = __ddiv_rn(var, dn < 1.0 ? 1.0 : dn);
  if (dn == 0.0) var = qnan;
  *mean = __ddiv_rn(s, n);
  *std = (var == var && var >= 0.0) ? __dsqrt_rn(var) : qnan;
}
struct FnBatch {
  FnCol c[kBatchCols];
};
template <typename T, typename OUT>
__global__ __launch_bounds__(kBlock) void fill_norm_many_kernel(FnBatch b) {
  const FnCol &c = b.c[blockIdx.y];
  double shift = c.shift, scale = c.scale;
  if (c.moments) {
    double mean, std;
    finish_moments(c.moments, &mean, &std);
    shift = mean;
    scale = std > 0.0 ? std : 0.0;  // normalize.py:79-82: std == 0 -> x - mean
  }
  fill_norm_body<T, OUT>((const T *)c.x, c.valid, c.n, c.has_fill, c.fill_val, c.do_norm, shift, scale,
                         (OUT *)c.out, c.filled);
}

// Clip (clip.py:49-55) and LogOp (logop.py:43-53), fused with a pending FillMissing constant:
//   v = isnull ? (has_fill ? fill : null) : x;  v = clamp(v, lo, hi);  out = do_log ? log(f32(v) + 1) : v
// Nulls stay null: NaN for float outputs, untouched validity bit for integer outputs.
template <typename T, typename OUT>
__global__ __launch_bounds__(kBlock) void clip_log_kernel(
    const T *__restrict__ x, const uint8_t *__restrict__ valid, uint64_t n, int has_fill,
    double fill_val, int has_min, double vmin, int has_max, double vmax, int do_log,
    OUT *__restrict__ out) {
  constexpr int VEC = VecOf<T>::n;
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  auto f = [&](T raw, bool ok) -> OUT {
    bool isnull = !ok || is_nan(raw);
    double v = (double)raw;
    if (isnull) {
      if (!has_fill) {
        if constexpr (std::is_floating_point<OUT>::value) return (OUT)qnan;
        return (OUT)0;
      }
      v = fill_val;
    }
    if (has_min && v < vmin) v = vmin;
    if (has_max && v > vmax) v = vmax;
    if (do_log) return (OUT)logf((float)v + 1.0f);  // the reference computes in float32
    return (OUT)v;
  };
  if constexpr (sizeof(T) == 4 && sizeof(OUT) == 8) {
    widen_stream<T, OUT>(x, valid, n, out, nullptr,
                         [&](T raw, bool ok, uint8_t &m) -> OUT { m = 0; return f(raw, ok); });
    return;
  }
  const uint64_t nvec = n / VEC;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    T v[VEC];
    load_vec<T>(x + i * VEC, v);
    unsigned vbits = 0xF;
    if (valid != nullptr) {
      uint64_t row = i * VEC;
      vbits = valid[row >> 3] >> (row & 7);
    }
    OUT r[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) r[j] = f(v[j], (vbits >> j) & 1);
    OUT *dst = out + i * VEC;
    constexpr int OB = VEC * (int)sizeof(OUT);
    if constexpr (OB == 32) {
      int4 a, b;
      memcpy(&a, &r[0], 16);
      memcpy(&b, &r[VEC / 2], 16);
      reinterpret_cast<int4 *>(dst)[0] = a;
      reinterpret_cast<int4 *>(dst)[1] = b;
    } else if constexpr (OB == 16) {
      int4 a;
      memcpy(&a, &r[0], 16);
      reinterpret_cast<int4 *>(dst)[0] = a;
    } else {
      int2 a;
      memcpy(&a, &r[0], 8);
      reinterpret_cast<int2 *>(dst)[0] = a;
    }
  }
  for (uint64_t i = nvec * VEC + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    out[i] = f(x[i], bit_valid(valid, i));
}

// Bucketize (bucketize.py:76-94): out = np.digitize(x, boundaries, right=False) = number of
// boundaries <= x, boundaries ascending; NaN / null rows get len(boundaries) like numpy's NaN.
// Boundaries live in LDS; one branch-free binary search per element.
constexpr int kMaxBoundaries = 8192;
template <typename T>
__global__ __launch_bounds__(kBlock) void bucketize_kernel(const T *__restrict__ x,
                                                           const uint8_t *__restrict__ valid,
                                                           uint64_t n,
                                                           const double *__restrict__ bounds,
                                                           int nb, int32_t *__restrict__ out) {
  __shared__ double b[kMaxBoundaries];
  for (int i = threadIdx.x; i < nb; i += kBlock) b[i] = bounds[i];
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const T raw = x[i];
    int r = nb;
    if (bit_valid(valid, i) && !is_nan(raw)) {
      const double v = (double)raw;
      int lo = 0, hi = nb;  // first index with b[idx] > v
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (b[mid] <= v) lo = mid + 1; else hi = mid;
      }
      r = lo;
    }
    out[i] = r;
  }
}

template <typename OUT>
__global__ __launch_bounds__(kBlock) void gather_kernel(const double *__restrict__ src,
                                                        const int64_t *__restrict__ group,
                                                        uint64_t n, double miss,
                                                        OUT *__restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int64_t g = group[i];
    double v = g >= 0 ? src[g] : miss;
    out[i] = (OUT)v;
  }
}

template <typename OUT>
__global__ __launch_bounds__(kBlock) void te_kernel(
    const int64_t *__restrict__ group_all, const int64_t *__restrict__ group_fold,
    const double *__restrict__ sum_all, const int64_t *__restrict__ cnt_all,
    const double *__restrict__ sum_fold, const int64_t *__restrict__ cnt_fold, uint64_t n,
    double p, double y_mean, OUT *__restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int64_t g = group_all[i];
    double v = y_mean;
    if (group_fold != nullptr) {
      // with folds the reference merges on [fold, key]: an unseen (fold,key) pair -> y_mean
      int64_t f = group_fold[i];
      if (g >= 0 && f >= 0) {
        double s = sum_all[g] - sum_fold[f];
        double c = (double)(cnt_all[g] - cnt_fold[f]);
        v = (s + p * y_mean) / (c + p);
      }
    } else if (g >= 0) {
      v = (sum_all[g] + p * y_mean) / ((double)cnt_all[g] + p);
    }
    out[i] = (OUT)v;
  }
}

// the same with DENSE fold statistics (nvt_sgb_reduce): entry g * kfold + fold[i] of
// sum_fold / cnt_fold belongs to (group g, fold of row i); a pair without rows has count 0 and
// is the reference's unmatched [fold, key] merge: y_mean
template <typename OUT>
__global__ __launch_bounds__(kBlock) void te_dense_kernel(
    const int64_t *__restrict__ group_all, const uint8_t *__restrict__ fold, unsigned kfold,
    const double *__restrict__ sum_all, const int64_t *__restrict__ cnt_all,
    const double *__restrict__ sum_fold, const int64_t *__restrict__ cnt_fold, uint64_t n,
    double p, double y_mean, OUT *__restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t g = group_all[i];
    double v = y_mean;
    if (g >= 0) {
      const uint64_t f = (uint64_t)g * kfold + fold[i];
      const int64_t cf = cnt_fold[f];
      if (cf > 0) {
        const double s = sum_all[g] - sum_fold[f];
        const double c = (double)(cnt_all[g] - cf);
        v = (s + p * y_mean) / (c + p);
      }
    }
    out[i] = (OUT)v;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void widen_kernel(const T *__restrict__ src, uint64_t n,
                                                       int64_t *__restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    out[i] = (int64_t)src[i];
}

__global__ __launch_bounds__(kBlock) void popcount_kernel(const uint8_t *__restrict__ valid,
                                                          uint64_t n, uint64_t *out) {
  const uint64_t nbytes = n >> 3;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  unsigned long long c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nbytes; i += stride)
    c += __popc((unsigned)valid[i]);
  if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 7))
    c += __popc((unsigned)valid[nbytes] & ((1u << (n & 7)) - 1u));
  double d = wave_sum((double)c);
  if (lane_id() == 0 && d > 0) atomicAdd((unsigned long long *)out, (unsigned long long)d);
}

template <typename T>
int moments_launch(const T *x, const uint8_t *valid, uint64_t n, int has_fill, double fill_val,
                   double *out3, double *partials, hipStream_t s) {
  unsigned grid = stream_grid(n / VecOf<T>::n + 1, kBlock * 4, 4);
  NVT_PROF("moments", n * sizeof(T), s);
  moments_kernel<T><<<grid, kBlock, 0, s>>>(x, valid, n, has_fill, fill_val, partials);
  NVT_CHECK_LAUNCH();
  moments_final_kernel<<<1, kBlock, 0, s>>>(partials, grid, out3);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}
template <typename T>
int minmax_launch(const T *x, const uint8_t *valid, uint64_t n, int accumulate, double *out2,
                  double *partials, hipStream_t s) {
  unsigned grid = stream_grid(n / VecOf<T>::n + 1, kBlock * 4, 4);
  minmax_kernel<T><<<grid, kBlock, 0, s>>>(x, valid, n, partials);
  NVT_CHECK_LAUNCH();
  minmax_final_kernel<<<1, kWave, 0, s>>>(partials, grid, accumulate, out2);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}
template <typename T, typename OUT>
int fill_norm_launch(const void *x, const uint8_t *valid, uint64_t n, int has_fill, double fill_val,
                     int do_norm, double shift, double scale, void *out, uint8_t *filled,
                     hipStream_t s) {
  unsigned grid = stream_grid(n / VecOf<T>::n + 1, kBlock * 2, 8);
  NVT_PROF("fill_normalize", n * (sizeof(T) + sizeof(OUT)), s);
  fill_norm_kernel<T, OUT><<<grid, kBlock, 0, s>>>(reinterpret_cast<const T *>(x), valid, n,
                                                   has_fill, fill_val, do_norm, shift, scale,
                                                   reinterpret_cast<OUT *>(out), filled);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // namespace nvt

using namespace nvt;

extern "C" {

uint64_t nvt_moments_scratch_bytes(void) { return (uint64_t)kReduceGrid * 3 * sizeof(double); }

int nvt_moments(const void *x, int dtype, const uint8_t *valid, uint64_t n, int has_fill,
                double fill_val, double *out3, void *partials, void *stream) {
  NVT_CHECK_ARG(out3 && partials, "null out/partials");
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(x && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "x must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  double *p = reinterpret_cast<double *>(partials);
  switch (dtype) {
    case NVT_F32:
      return moments_launch<float>((const float *)x, valid, n, has_fill, fill_val, out3, p, s);
    case NVT_F64:
      return moments_launch<double>((const double *)x, valid, n, has_fill, fill_val, out3, p, s);
    case NVT_I32:
      return moments_launch<int32_t>((const int32_t *)x, valid, n, has_fill, fill_val, out3, p, s);
    case NVT_I64:
      return moments_launch<int64_t>((const int64_t *)x, valid, n, has_fill, fill_val, out3, p, s);
  }
  set_error("nvt_moments: unsupported dtype %d", dtype);
  return NVT_EINVAL;
}

static inline int dtype_bytes(int dtype) {
  return dtype == NVT_F32 || dtype == NVT_I32 ? 4 : dtype == NVT_U8 ? 1 : 8;
}

int nvt_moments_many(const nvt_moments_col *cols, int ncols, void *partials, void *stream) {
  NVT_CHECK_ARG(ncols == 0 || (cols && partials), "null descriptors/partials");
  hipStream_t s = (hipStream_t)stream;
  double *pbase = reinterpret_cast<double *>(partials);
  for (int c0 = 0; c0 < ncols; c0 += kBatchCols) {
    const int nc = ncols - c0 < kBatchCols ? ncols - c0 : kBatchCols;
    MomBatch b;
    memset(&b, 0, sizeof(b));
    unsigned grid = 1;
    uint64_t bytes = 0;
    int live = 0;
    for (int i = 0; i < nc; ++i) {
      const nvt_moments_col &c = cols[c0 + i];
      if (c.n == 0) continue;
      NVT_CHECK_ARG(c.x && c.out3 && (reinterpret_cast<uintptr_t>(c.x) & 15) == 0,
                    "x must be non-null and 16-byte aligned");
      NVT_CHECK_ARG(c.dtype == NVT_F32 || c.dtype == NVT_F64 || c.dtype == NVT_I32 ||
                        c.dtype == NVT_I64, "unsupported dtype");
      MomCol &m = b.c[live++];
      m.x = c.x;
      m.valid = c.valid;
      m.n = c.n;
      m.fill_val = c.fill_val;
      m.out3 = c.out3;
      m.dtype = c.dtype;
      m.has_fill = c.has_fill;
      const unsigned g = stream_grid(c.n / (16 / dtype_bytes(c.dtype)) + 1, kBlock * 4, 4);
      grid = g > grid ? g : grid;
      bytes += c.n * dtype_bytes(c.dtype);
    }
    if (!live) continue;
    double *p = pbase + (uint64_t)c0 * 3 * kReduceGrid;
    NVT_PROF("moments", bytes, s);
    moments_many_kernel<<<dim3(grid, live), kBlock, 0, s>>>(b, p);
    NVT_CHECK_LAUNCH();
    moments_final_many_kernel<<<live, kBlock, 0, s>>>(b, p, grid);
    NVT_CHECK_LAUNCH();
  }
  return NVT_OK;
}

int nvt_fill_normalize_many(const nvt_fillnorm_col *cols, int ncols, void *stream) {
  NVT_CHECK_ARG(ncols == 0 || cols, "null descriptors");
  hipStream_t s = (hipStream_t)stream;
  // group the columns by (input dtype, output dtype): one launch per combination and
  // kBatchCols columns
  std::vector<char> done(ncols > 0 ? ncols : 0, 0);
  for (int i0 = 0; i0 < ncols; ++i0) {
    if (done[i0]) continue;
    const int dt = cols[i0].dtype, odt = cols[i0].out_dtype;
    FnBatch b;
    memset(&b, 0, sizeof(b));
    int live = 0;
    unsigned grid = 1;
    uint64_t bytes = 0;
    auto flush = [&]() -> int {
      if (!live) return NVT_OK;
      NVT_PROF("fill_normalize", bytes, s);
#define NVT_FNM(T, O)                                                    \
  do {                                                                   \
    fill_norm_many_kernel<T, O><<<dim3(grid, live), kBlock, 0, s>>>(b);  \
    NVT_CHECK_LAUNCH();                                                  \
    live = 0;                                                            \
    grid = 1;                                                            \
    bytes = 0;                                                           \
    return NVT_OK;                                                       \
  } while (0)
      if (odt == NVT_F64) {
        switch (dt) {
          case NVT_F32: NVT_FNM(float, double);
          case NVT_F64: NVT_FNM(double, double);
          case NVT_I32: NVT_FNM(int32_t, double);
          case NVT_I64: NVT_FNM(int64_t, double);
        }
      } else if (odt == NVT_F32) {
        switch (dt) {
          case NVT_F32: NVT_FNM(float, float);
          case NVT_F64: NVT_FNM(double, float);
          case NVT_I32: NVT_FNM(int32_t, float);
          case NVT_I64: NVT_FNM(int64_t, float);
        }
      } else if (odt == dt) {
        switch (dt) {
          case NVT_I32: NVT_FNM(int32_t, int32_t);
          case NVT_I64: NVT_FNM(int64_t, int64_t);
        }
      }
#undef NVT_FNM
      set_error("nvt_fill_normalize_many: unsupported dtype combination in=%d out=%d", dt, odt);
      return NVT_EINVAL;
    };
    for (int i = i0; i < ncols; ++i) {
      const nvt_fillnorm_col &c = cols[i];
      if (done[i] || c.dtype != dt || c.out_dtype != odt) continue;
      done[i] = 1;
      if (c.n == 0) continue;
      NVT_CHECK_ARG(c.x && c.out && (reinterpret_cast<uintptr_t>(c.x) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(c.out) & 15) == 0,
                    "x/out must be non-null and 16-byte aligned");
      NVT_CHECK_ARG(!c.filled || (reinterpret_cast<uintptr_t>(c.filled) & 3) == 0,
                    "filled must be 4-byte aligned");
      NVT_CHECK_ARG(odt == NVT_F32 || odt == NVT_F64 || !c.do_norm,
                    "normalised output must be f32/f64");
      FnCol &f = b.c[live++];
      f.x = c.x;
      f.valid = c.valid;
      f.n = c.n;
      f.fill_val = c.fill_val;
      f.shift = c.shift;
      f.scale = c.scale;
      f.out = c.out;
      f.filled = c.filled;
      f.has_fill = c.has_fill;
      f.do_norm = c.do_norm;
      f.moments = c.do_norm ? c.moments : nullptr;
      const unsigned g = stream_grid(c.n / (16 / dtype_bytes(dt)) + 1, kBlock * 2, 8);
      grid = g > grid ? g : grid;
      bytes += c.n * (uint64_t)(dtype_bytes(dt) + dtype_bytes(odt));
      if (live == kBatchCols) {
        int rc = flush();
        if (rc) return rc;
      }
    }
    int rc = flush();
    if (rc) return rc;
  }
  return NVT_OK;
}

int nvt_minmax(const void *x, int dtype, const uint8_t *valid, uint64_t n, int accumulate,
               double *out2, void *partials, void *stream) {
  NVT_CHECK_ARG(out2 && partials, "null out/partials");
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(x && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "x must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  double *p = reinterpret_cast<double *>(partials);
  switch (dtype) {
    case NVT_F32:
      return minmax_launch<float>((const float *)x, valid, n, accumulate, out2, p, s);
    case NVT_F64:
      return minmax_launch<double>((const double *)x, valid, n, accumulate, out2, p, s);
    case NVT_I32:
      return minmax_launch<int32_t>((const int32_t *)x, valid, n, accumulate, out2, p, s);
    case NVT_I64:
      return minmax_launch<int64_t>((const int64_t *)x, valid, n, accumulate, out2, p, s);
  }
  set_error("nvt_minmax: unsupported dtype %d", dtype);
  return NVT_EINVAL;
}

int nvt_fill_normalize(const void *x, int dtype, const uint8_t *valid, uint64_t n, int has_fill,
                       double fill_val, int do_norm, double shift, double scale, void *out,
                       int out_dtype, uint8_t *filled, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(x && out, "null x/out");
  NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                "x/out must be 16-byte aligned");
  NVT_CHECK_ARG(!filled || (reinterpret_cast<uintptr_t>(filled) & 3) == 0,
                "filled must be 4-byte aligned");
  hipStream_t s = (hipStream_t)stream;
#define NVT_FN(T, O) \
  return fill_norm_launch<T, O>(x, valid, n, has_fill, fill_val, do_norm, shift, scale, out, filled, s)
  if (out_dtype == NVT_F64) {
    switch (dtype) {
      case NVT_F32: NVT_FN(float, double);
      case NVT_F64: NVT_FN(double, double);
      case NVT_I32: NVT_FN(int32_t, double);
      case NVT_I64: NVT_FN(int64_t, double);
    }
  } else if (out_dtype == NVT_F32) {
    switch (dtype) {
      case NVT_F32: NVT_FN(float, float);
      case NVT_F64: NVT_FN(double, float);
      case NVT_I32: NVT_FN(int32_t, float);
      case NVT_I64: NVT_FN(int64_t, float);
    }
  } else if (!do_norm && out_dtype == dtype) {
    switch (dtype) {
      case NVT_I32: NVT_FN(int32_t, int32_t);
      case NVT_I64: NVT_FN(int64_t, int64_t);
    }
  }
#undef NVT_FN
  set_error("nvt_fill_normalize: unsupported dtype combination in=%d out=%d norm=%d", dtype,
            out_dtype, do_norm);
  return NVT_EINVAL;
}

int nvt_clip_log(const void *x, int dtype, const uint8_t *valid, uint64_t n, int has_fill,
                 double fill_val, int has_min, double vmin, int has_max, double vmax, int do_log,
                 void *out, int out_dtype, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(x && out, "null x/out");
  NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                "x/out must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
#define NVT_CL(T, O)                                                                           \
  do {                                                                                         \
    unsigned grid = stream_grid(n / VecOf<T>::n + 1, kBlock * 2, 8);                           \
    NVT_PROF("clip_log", n * (sizeof(T) + sizeof(O)), s);                                      \
    clip_log_kernel<T, O><<<grid, kBlock, 0, s>>>((const T *)x, valid, n, has_fill, fill_val,  \
                                                  has_min, vmin, has_max, vmax, do_log,        \
                                                  (O *)out);                                   \
    NVT_CHECK_LAUNCH();                                                                        \
    return NVT_OK;                                                                             \
  } while (0)
  if (out_dtype == NVT_F32) {
    switch (dtype) {
      case NVT_F32: NVT_CL(float, float);
      case NVT_F64: NVT_CL(double, float);
      case NVT_I32: NVT_CL(int32_t, float);
      case NVT_I64: NVT_CL(int64_t, float);
    }
  } else if (out_dtype == NVT_F64) {
    switch (dtype) {
      case NVT_F32: NVT_CL(float, double);
      case NVT_F64: NVT_CL(double, double);
      case NVT_I32: NVT_CL(int32_t, double);
      case NVT_I64: NVT_CL(int64_t, double);
    }
  } else if (!do_log && out_dtype == dtype) {
    switch (dtype) {
      case NVT_I32: NVT_CL(int32_t, int32_t);
      case NVT_I64: NVT_CL(int64_t, int64_t);
    }
  }
#undef NVT_CL
  set_error("nvt_clip_log: unsupported dtype combination in=%d out=%d log=%d", dtype, out_dtype,
            do_log);
  return NVT_EINVAL;
}

int nvt_bucketize(const void *x, int dtype, const uint8_t *valid, uint64_t n,
                  const double *boundaries, int n_boundaries, int32_t *out, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(x && out, "null x/out");
  NVT_CHECK_ARG(n_boundaries >= 0 && n_boundaries <= kMaxBoundaries, "at most 8192 boundaries");
  NVT_CHECK_ARG(n_boundaries == 0 || boundaries, "null boundaries");
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = stream_grid(n, kBlock * 4, 8);
  switch (dtype) {
    case NVT_F32:
      bucketize_kernel<float><<<grid, kBlock, 0, s>>>((const float *)x, valid, n, boundaries, n_boundaries, out);
      break;
    case NVT_F64:
      bucketize_kernel<double><<<grid, kBlock, 0, s>>>((const double *)x, valid, n, boundaries, n_boundaries, out);
      break;
    case NVT_I32:
      bucketize_kernel<int32_t><<<grid, kBlock, 0, s>>>((const int32_t *)x, valid, n, boundaries, n_boundaries, out);
      break;
    case NVT_I64:
      bucketize_kernel<int64_t><<<grid, kBlock, 0, s>>>((const int64_t *)x, valid, n, boundaries, n_boundaries, out);
      break;
    default:
      set_error("nvt_bucketize: unsupported dtype %d", dtype);
      return NVT_EINVAL;
  }
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_gather_f64(const double *src, const int64_t *group, uint64_t n, double miss, void *out,
                   int out_dtype, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(src && group && out, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("gather", 0, s);
  unsigned grid = stream_grid(n, kBlock * 4);
  switch (out_dtype) {
    case NVT_F64:
      gather_kernel<double><<<grid, kBlock, 0, s>>>(src, group, n, miss, (double *)out);
      break;
    case NVT_F32:
      gather_kernel<float><<<grid, kBlock, 0, s>>>(src, group, n, miss, (float *)out);
      break;
    case NVT_I32:
      gather_kernel<int32_t><<<grid, kBlock, 0, s>>>(src, group, n, miss, (int32_t *)out);
      break;
    case NVT_I64:
      gather_kernel<int64_t><<<grid, kBlock, 0, s>>>(src, group, n, miss, (int64_t *)out);
      break;
    default:
      set_error("nvt_gather_f64: unsupported out dtype %d", out_dtype);
      return NVT_EINVAL;
  }
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_te_apply(const int64_t *group_all, const int64_t *group_fold, const double *sum_all,
                 const int64_t *cnt_all, const double *sum_fold, const int64_t *cnt_fold,
                 uint64_t n, double p_smooth, double y_mean, void *out, int out_dtype,
                 void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(group_all && sum_all && cnt_all && out, "null pointer");
  NVT_CHECK_ARG(!group_fold || (sum_fold && cnt_fold), "fold stats missing");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("te_apply", 0, s);
  unsigned grid = stream_grid(n, kBlock * 4);
  if (out_dtype == NVT_F32)
    te_kernel<float><<<grid, kBlock, 0, s>>>(group_all, group_fold, sum_all, cnt_all, sum_fold,
                                             cnt_fold, n, p_smooth, y_mean, (float *)out);
  else if (out_dtype == NVT_F64)
    te_kernel<double><<<grid, kBlock, 0, s>>>(group_all, group_fold, sum_all, cnt_all, sum_fold,
                                              cnt_fold, n, p_smooth, y_mean, (double *)out);
  else {
    set_error("nvt_te_apply: out dtype must be f32/f64");
    return NVT_EINVAL;
  }
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_te_apply_folds(const int64_t *group_all, const uint8_t *fold, int kfold,
                       const double *sum_all, const int64_t *cnt_all, const double *sum_fold,
                       const int64_t *cnt_fold, uint64_t n, double p_smooth, double y_mean,
                       void *out, int out_dtype, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(group_all && fold && sum_all && cnt_all && sum_fold && cnt_fold && out, "null pointer");
  NVT_CHECK_ARG(kfold >= 2 && kfold <= 256, "kfold must be 2..256");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("te_apply", 0, s);
  unsigned grid = stream_grid(n, kBlock * 4);
  if (out_dtype == NVT_F32)
    te_dense_kernel<float><<<grid, kBlock, 0, s>>>(group_all, fold, (unsigned)kfold, sum_all, cnt_all,
                                                   sum_fold, cnt_fold, n, p_smooth, y_mean, (float *)out);
  else if (out_dtype == NVT_F64)
    te_dense_kernel<double><<<grid, kBlock, 0, s>>>(group_all, fold, (unsigned)kfold, sum_all, cnt_all,
                                                    sum_fold, cnt_fold, n, p_smooth, y_mean, (double *)out);
  else {
    set_error("nvt_te_apply_folds: out dtype must be f32/f64");
    return NVT_EINVAL;
  }
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_widen_i64(const void *src, int dtype, uint64_t n, int64_t *out, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(src && out, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  unsigned grid = stream_grid(n, kBlock * 4);
  switch (dtype) {
    case NVT_I32:
      widen_kernel<int32_t><<<grid, kBlock, 0, s>>>((const int32_t *)src, n, out);
      break;
    case NVT_U8:
      widen_kernel<uint8_t><<<grid, kBlock, 0, s>>>((const uint8_t *)src, n, out);
      break;
    case NVT_I64:
      NVT_CHECK_HIP(hipMemcpyAsync(out, src, n * 8, hipMemcpyDeviceToDevice, s));
      return NVT_OK;
    default:
      set_error("nvt_widen_i64: unsupported dtype %d", dtype);
      return NVT_EINVAL;
  }
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int nvt_popcount(const uint8_t *valid, uint64_t n, uint64_t *out_device, void *stream) {
  NVT_CHECK_ARG(out_device, "null out");
  hipStream_t s = (hipStream_t)stream;
  NVT_CHECK_HIP(hipMemsetAsync(out_device, 0, sizeof(uint64_t), s));
  if (n == 0 || valid == nullptr) return NVT_OK;
  popcount_kernel<<<stream_grid(n / 8 + 1, kBlock * 16), kBlock, 0, s>>>(valid, n, out_device);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // extern "C"
