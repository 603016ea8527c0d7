// Device-wide exclusive scan of a uint32 array (chunk scan -> chunk-total scan -> add), shared
// by the vocabulary radix sort and the partitioned count (per-tile bucket offsets).
#pragma once
#include "nvt_common.hpp"

namespace nvt {

// Exclusive scan of `len` uint32 in three steps (chunk scan, chunk-total scan, add).
constexpr int kScanChunk = 2048;  // 256 threads x 8
static __global__ __launch_bounds__(kBlock) void scan_chunk_kernel(unsigned *data, uint64_t len,
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
static __global__ __launch_bounds__(kBlock) void scan_totals_kernel(unsigned long long *chunk_tot,
                                                             uint64_t nchunks) {
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
static __global__ __launch_bounds__(kBlock) void scan_add_kernel(unsigned *data, uint64_t len,
                                                          const unsigned long long *chunk_tot) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * 8;
  const unsigned add = (unsigned)chunk_tot[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (base + j < len) data[base + j] += add;
}


// Short arrays (the radix passes of a <= ~1 M-entry vocabulary, small partition histograms):
// ONE workgroup walks the array in 8192-element blocks with a running carry -- one launch
// instead of three, which is what those passes were bound by.
constexpr int kScanSmallBS = 1024;
constexpr uint64_t kScanSmallMax = 1 << 17;
static __global__ __launch_bounds__(kScanSmallBS) void scan_small_kernel(unsigned *data, uint64_t len) {
  __shared__ unsigned wsum[2][kScanSmallBS / kWave];
  const unsigned w = threadIdx.x / kWave;
  auto load8 = [&](uint64_t base, unsigned (&v)[8]) {
    if (base + 8 <= len) {
      const uint4 a = *reinterpret_cast<const uint4 *>(data + base);
      const uint4 c = *reinterpret_cast<const uint4 *>(data + base + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
      v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (base + j < len) ? data[base + j] : 0;
    }
  };
  unsigned nxt[8];
  load8((uint64_t)threadIdx.x * 8, nxt);
  unsigned carry = 0;  // identical in every thread
  int par = 0;
  for (uint64_t b = 0; b < len; b += (uint64_t)kScanSmallBS * 8, par ^= 1) {
    const uint64_t base = b + (uint64_t)threadIdx.x * 8;
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = nxt[j];
    if (b + (uint64_t)kScanSmallBS * 8 < len) load8(base + (uint64_t)kScanSmallBS * 8, nxt);
    unsigned tot = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) tot += v[j];
    unsigned inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o = __shfl_up(inc, off, 64);
      if (lane_id() >= (unsigned)off) inc += o;
    }
    if (lane_id() == 63) wsum[par][w] = inc;
    __syncthreads();  // double-buffered wsum: one barrier per block
    unsigned wbase = carry, all = 0;
    for (unsigned i = 0; i < kScanSmallBS / kWave; ++i) {
      const unsigned x = wsum[par][i];
      if (i < w) wbase += x;
      all += x;
    }
    unsigned run = wbase + inc - tot;
    unsigned o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = run;
      run += v[j];
    }
    if (base + 8 <= len) {
      *reinterpret_cast<uint4 *>(data + base) = make_uint4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<uint4 *>(data + base + 4) = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (base + j < len) data[base + j] = o[j];
    }
    carry += all;
  }
}

inline uint64_t scan_chunks(uint64_t len) { return (len + kScanChunk - 1) / kScanChunk; }

// data[0..len) -> exclusive prefix sums in place; chunk_tot: scan_chunks(len) uint64 scratch
inline int exclusive_scan_u32(unsigned *data, uint64_t len, unsigned long long *chunk_tot,
                              hipStream_t stream) {
  if (len == 0) return NVT_OK;
  if (len <= kScanSmallMax && (reinterpret_cast<uintptr_t>(data) & 15) == 0) {
    scan_small_kernel<<<1, kScanSmallBS, 0, stream>>>(data, len);
    NVT_CHECK_LAUNCH();
    return NVT_OK;
  }
  const uint64_t nchunks = scan_chunks(len);
  scan_chunk_kernel<<<(unsigned)nchunks, kBlock, 0, stream>>>(data, len, chunk_tot);
  NVT_CHECK_LAUNCH();
  scan_totals_kernel<<<1, kBlock, 0, stream>>>(chunk_tot, nchunks);
  NVT_CHECK_LAUNCH();
  scan_add_kernel<<<(unsigned)nchunks, kBlock, 0, stream>>>(data, len, chunk_tot);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

// Same scan, but the last step is left to the consumer: on return
//   prefix(i) = data[i] + (unsigned)chunk_base[i / kScanChunk]      (scan_lookup below)
// with chunk_base == nullptr meaning data[] already holds the full prefix (short arrays).
// Saves one launch + one pass over the array per radix pass of a large vocabulary.
inline int exclusive_scan_u32_deferred(unsigned *data, uint64_t len, unsigned long long *chunk_tot,
                                       const unsigned long long **chunk_base, hipStream_t stream) {
  *chunk_base = nullptr;
  if (len == 0) return NVT_OK;
  if (len <= kScanSmallMax && (reinterpret_cast<uintptr_t>(data) & 15) == 0) {
    scan_small_kernel<<<1, kScanSmallBS, 0, stream>>>(data, len);
    NVT_CHECK_LAUNCH();
    return NVT_OK;
  }
  const uint64_t nchunks = scan_chunks(len);
  scan_chunk_kernel<<<(unsigned)nchunks, kBlock, 0, stream>>>(data, len, chunk_tot);
  NVT_CHECK_LAUNCH();
  scan_totals_kernel<<<1, kBlock, 0, stream>>>(chunk_tot, nchunks);
  NVT_CHECK_LAUNCH();
  *chunk_base = chunk_tot;
  return NVT_OK;
}
__device__ __forceinline__ unsigned scan_lookup(const unsigned *data,
                                                const unsigned long long *chunk_base, uint64_t i) {
  unsigned v = data[i];
  if (chunk_base != nullptr) v += (unsigned)chunk_base[i / kScanChunk];
  return v;
}

}  // namespace nvt
