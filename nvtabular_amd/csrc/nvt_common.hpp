// Shared device helpers for the gfx950 kernels (wave64, 256 CUs / 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/nvt_hip.h"

namespace nvt {

constexpr int kBlock = 256;        // 4 waves, one per SIMD
constexpr int kCUs = 256;          // MI355X
constexpr int kWave = 64;

void set_error(const char *fmt, ...);

#define NVT_CHECK_ARG(cond, msg)                 \
  do {                                           \
    if (!(cond)) {                               \
      nvt::set_error("%s: %s", __func__, msg);   \
      return NVT_EINVAL;                         \
    }                                            \
  } while (0)

#define NVT_CHECK_HIP(expr)                                                       \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      nvt::set_error("%s: %s -> %s", __func__, #expr, hipGetErrorString(_e));     \
      return NVT_EHIP;                                                            \
    }                                                                             \
  } while (0)

#define NVT_CHECK_LAUNCH() NVT_CHECK_HIP(hipGetLastError())

// Run-time A / B switches between kernel variants (NVT_ENC_PIPE, NVT_SORT_LEGACY ...) exist only in
// variant libraries built with -DNVT_AB_SWITCHES (tools/build_variant.sh, tools/var_libs.sh): the
// default library takes the default side of every one of them at compile time.  (Configuration that
// a deployment may set -- NVT_ENCODE_STREAMS, NVT_FINALIZE_SERIAL, NVT_ROCTX, NVT_EVENT_TIMING,
// NVT_ENC_STATS -- stays on getenv.)
inline const char *ab_env(const char *name) {
#ifdef NVT_AB_SWITCHES
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Grid for a streaming (HBM-bound) kernel: enough workgroups to fill 256 CUs
// several times over, capped so every block still gets a long grid-stride run.
inline unsigned stream_grid(uint64_t work_items, unsigned per_block, unsigned blocks_per_cu = 8) {
  uint64_t need = (work_items + per_block - 1) / per_block;
  uint64_t cap = (uint64_t)kCUs * blocks_per_cu;
  if (need < 1) need = 1;
  return (unsigned)(need < cap ? need : cap);
}

// ---- hashing ---------------------------------------------------------------
// Public hash: murmur3 fmix64 of the sign-extended key.
__host__ __device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xFF51AFD7ED558CCDull;
  k ^= k >> 33;
  k *= 0xC4CEB9FE1A85EC53ull;
  k ^= k >> 33;
  return k;
}
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__host__ __device__ __forceinline__ uint64_t key_hash64(int64_t key) { return fmix64((uint64_t)key); }
__host__ __device__ __forceinline__ uint32_t key_hash32(int64_t key) {
  return (uint32_t)(key_hash64(key) >> 32);
}
// Slot hash (internal, never visible in results): cheaper 32-bit mix for int32 keys.
__device__ __forceinline__ uint32_t slot_hash(int32_t key) { return fmix32((uint32_t)key); }
__device__ __forceinline__ uint64_t slot_hash(int64_t key) { return fmix64((uint64_t)key); }
// Cheaper still, for the two loops that hash EVERY row of a column and are bound by instruction
// issue (hot-image lookup of the partition kernels, home slot of the LDS-resident counting):
// two 24-bit multiplies (full rate; v_mul_lo_u32 is a quarter-rate instruction and fmix32 holds
// two of them).  Only the TOP bits of the sum are mixed well enough to address a table: every
// input bit reaches bits >= 20 through one of the two products.  Measured on the first 8192 ids
// of a scrambled / dense / strided / random column: as many keys find a free slot of a 4096 x 2
// table as with fmix32 (tools/hash_quality.py).
__device__ __forceinline__ uint32_t mul24_hash(int32_t key) {
  const uint32_t k = (uint32_t)key;
  return __umul24(k >> 8, 0x5BD1E9u) + __umul24(k ^ (k >> 7), 0x9E3779u);
}
// bucket of the hot-key image (sample kernel and every kernel that looks keys up in it)
__device__ __forceinline__ uint32_t hot_image_bucket(int32_t key, uint32_t mask) {
#ifndef NVT_HOT_FMIX
  return (mul24_hash(key) >> 20) & mask;  // (mask <= 4095)
#else
  return (slot_hash(key) >> 13) & mask;
#endif
}

// ---- validity bitmaps --------------------------------------------------------
__device__ __forceinline__ bool bit_valid(const uint8_t *valid, uint64_t i) {
  return valid == nullptr || ((valid[i >> 3] >> (i & 7)) & 1);
}

template <typename T>
__device__ __forceinline__ bool is_nan(T) { return false; }
template <>
__device__ __forceinline__ bool is_nan<float>(float v) { return v != v; }
template <>
__device__ __forceinline__ bool is_nan<double>(double v) { return v != v; }

// ---- wave helpers ------------------------------------------------------------
__device__ __forceinline__ unsigned lane_id() { return __lane_id(); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o = __shfl_down(v, off, 64);
    v = (o < v || v != v) ? o : v;
  }
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o = __shfl_down(v, off, 64);
    v = (o > v || v != v) ? o : v;
  }
  return v;
}

}  // namespace nvt
