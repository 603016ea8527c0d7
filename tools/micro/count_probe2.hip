// Micro-benchmark (tools only): what bounds the LDS-table counting loop?  Variants from a pure
// stream read up to the production-shaped loop, 45 M int32 keys, Zipf over `card` ids.
//   hipcc --offload-arch=gfx950 -O3 -o count_probe2 count_probe2.hip && ./count_probe2
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e = (x);                                                    \
    if (e != hipSuccess) {                                                 \
      printf("err %s line %d\n", hipGetErrorString(e), __LINE__);          \
      exit(1);                                                             \
    }                                                                      \
  } while (0)
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__global__ void gen(int32_t *k, uint64_t n, double card, double s, uint32_t seed) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t st = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    uint32_t r = fmix32((uint32_t)i * 2654435761u + seed);
    double u = (r + 0.5) / 4294967296.0;
    double x = pow((pow(card, 1.0 - s) - 1.0) * u + 1.0, 1.0 / (1.0 - s));
    int64_t v = (int64_t)floor(x);
    if (v < 1) v = 1;
    if (v > card) v = (int64_t)card;
    k[i] = (int32_t)((v * 2654435761ull) % 2147483648ull);
  }
}
constexpr int EMPTY = INT32_MIN;

// ---- V0: stream floor ------------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(BS) void v_stream(const int32_t *__restrict__ keys, uint64_t n,
                                               unsigned *out) {
  const int4 *vk = (const int4 *)keys;
  uint64_t nv = n / 4, st = (uint64_t)gridDim.x * BS;
  unsigned acc = 0;
  constexpr int U = 4;
  for (uint64_t v0 = blockIdx.x * (uint64_t)BS + threadIdx.x; v0 < nv; v0 += st * U) {
    int4 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (v0 + u * st < nv) p[u] = vk[v0 + u * st]; else p[u] = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += p[u].x ^ p[u].y ^ p[u].z ^ p[u].w;
  }
  if (acc == 0x12345678) out[0] = acc;
}

// hash variants
template <int HASH, int BITS>
__device__ __forceinline__ uint32_t slot_of(int key) {
  if (HASH == 0) return fmix32((uint32_t)key) >> (32 - BITS);
  if (HASH == 1) return ((uint32_t)key * 0x9E3779B1u) >> (32 - BITS);
  // HASH 2: xor-fold then one multiply (better for keys that differ in high bits only)
  uint32_t k = (uint32_t)key;
  k ^= k >> 15;
  return (k * 0x2C1B3C6Du) >> (32 - BITS);
}

// ---- V1: pure LDS atomic histogram on hash slot (no key verification): LDS atomic ceiling --
template <int BITS, int BS, int HASH>
__global__ __launch_bounds__(BS) void v_hist(const int32_t *__restrict__ keys, uint64_t n,
                                             unsigned *out) {
  constexpr int SLOTS = 1 << BITS;
  __shared__ unsigned lcnt[SLOTS];
  for (int i = threadIdx.x; i < SLOTS; i += BS) lcnt[i] = 0;
  __syncthreads();
  const int4 *vk = (const int4 *)keys;
  uint64_t nv = n / 4, st = (uint64_t)gridDim.x * BS;
  constexpr int U = 4;
  for (uint64_t v0 = blockIdx.x * (uint64_t)BS + threadIdx.x; v0 < nv; v0 += st * U) {
    int4 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (v0 + u * st < nv) p[u] = vk[v0 + u * st]; else p[u] = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      atomicAdd(&lcnt[slot_of<HASH, BITS>(p[u].x)], 1u);
      atomicAdd(&lcnt[slot_of<HASH, BITS>(p[u].y)], 1u);
      atomicAdd(&lcnt[slot_of<HASH, BITS>(p[u].z)], 1u);
      atomicAdd(&lcnt[slot_of<HASH, BITS>(p[u].w)], 1u);
    }
  }
  __syncthreads();
  unsigned acc = 0;
  for (int i = threadIdx.x; i < SLOTS; i += BS) acc += lcnt[i];
  if (acc == 0x12345678) out[0] = acc;
}

// ---- V2: read key, compare, add (hit path branch-free, misses to a slow path) ------------
// MODE 0: read + add;  MODE 1: 64-bit {key|count} word, ONE returning atomic add, verify after
template <int BITS, int BS, int HASH, int MODE, int U, bool VALID>
__global__ __launch_bounds__(BS) void v_count(const int32_t *__restrict__ keys,
                                              const uint8_t *__restrict__ valid, uint64_t n,
                                              unsigned *out) {
  constexpr int SLOTS = 1 << BITS;
  __shared__ int lkeys[MODE == 0 ? SLOTS : 1];
  __shared__ unsigned lcnt[MODE == 0 ? SLOTS + 64 : 1];
  __shared__ unsigned long long lw[MODE == 1 ? SLOTS + 64 : 1];
  if (MODE == 0) {
    for (int i = threadIdx.x; i < SLOTS; i += BS) {
      lkeys[i] = EMPTY;
      lcnt[i] = 0;
    }
    if (threadIdx.x < 64) lcnt[SLOTS + threadIdx.x] = 0;
  } else {
    for (int i = threadIdx.x; i < SLOTS + 64; i += BS) lw[i] = ((unsigned long long)(uint32_t)EMPTY) << 32;
  }
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  const int4 *vk = (const int4 *)keys;
  const uint64_t nv = n / 4;
  // contiguous slab per workgroup
  const uint64_t per = (nv + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = blockIdx.x * per, hi = lo + per < nv ? lo + per : nv;
  unsigned nulls = 0;
  int4 np[U];
  unsigned nb[U];
  auto issue = [&](uint64_t v0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t v = v0 + (uint64_t)u * BS;
      nb[u] = 0x10000;
      if (v < hi) {
        np[u] = vk[v];
        nb[u] = VALID ? valid[(v * 4) >> 3] : 0xFF;
      }
    }
  };
  issue(lo + threadIdx.x);
  for (uint64_t v0 = lo + threadIdx.x; v0 < hi; v0 += (uint64_t)BS * U) {
    int4 p[U];
    unsigned b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      p[u] = np[u];
      b[u] = nb[u];
    }
    issue(v0 + (uint64_t)BS * U);
    int k[U * 4];
    uint32_t s[U * 4];
    unsigned live = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool inr = !(b[u] & 0x10000);
      const unsigned vb = inr ? (b[u] >> (((v0 + (uint64_t)u * BS) * 4) & 7)) & 15u : 0u;
      k[u * 4 + 0] = p[u].x;
      k[u * 4 + 1] = p[u].y;
      k[u * 4 + 2] = p[u].z;
      k[u * 4 + 3] = p[u].w;
      nulls += inr ? 4 - __popc(vb) : 0;
      live |= vb << (u * 4);
    }
#pragma unroll
    for (int q = 0; q < U * 4; ++q) {
      const bool lv = (live >> q) & 1;
      s[q] = lv ? slot_of<HASH, BITS>(k[q]) : (uint32_t)SLOTS + lane;
    }
    unsigned miss = 0;
    if (MODE == 0) {
      int cur[U * 4];
#pragma unroll
      for (int q = 0; q < U * 4; ++q) cur[q] = lkeys[s[q] & (SLOTS - 1)];
#pragma unroll
      for (int q = 0; q < U * 4; ++q) {
        const bool lv = (live >> q) & 1;
        const bool hit = lv & (cur[q] == k[q]);
        atomicAdd(&lcnt[hit ? s[q] : (uint32_t)SLOTS + lane], 1u);
        miss |= ((lv & !hit) ? 1u : 0u) << q;
      }
      if (miss) {
#pragma unroll
        for (int q = 0; q < U * 4; ++q) {
          if (!((miss >> q) & 1)) continue;
          uint32_t h = s[q];
          for (int pr = 0; pr < 256; ++pr) {
            uint32_t sl = (h + pr) & (SLOTS - 1);
            int c = lkeys[sl];
            if (c == EMPTY) {
              c = atomicCAS(&lkeys[sl], EMPTY, k[q]);
              if (c == EMPTY) c = k[q];
            }
            if (c == k[q]) {
              atomicAdd(&lcnt[sl], 1u);
              break;
            }
          }
        }
      }
    } else {
      unsigned long long old[U * 4];
#pragma unroll
      for (int q = 0; q < U * 4; ++q) old[q] = atomicAdd(&lw[s[q]], 1ull);
#pragma unroll
      for (int q = 0; q < U * 4; ++q) {
        const bool lv = (live >> q) & 1;
        const bool hit = (int)(old[q] >> 32) == k[q];
        miss |= ((lv & !hit) ? 1u : 0u) << q;
      }
      if (miss) {
#pragma unroll
        for (int q = 0; q < U * 4; ++q) {
          if (!((miss >> q) & 1)) continue;
          atomicAdd(&lw[s[q]], ~0ull);  // undo
          uint32_t h = s[q];
          for (int pr = 0; pr < 256; ++pr) {
            uint32_t sl = (h + pr) & (SLOTS - 1);
            unsigned long long c = lw[sl];
            if ((int)(c >> 32) == EMPTY) {
              unsigned long long want = ((unsigned long long)(uint32_t)k[q] << 32) | (c & 0xFFFFFFFFull);
              unsigned long long prev = atomicCAS(&lw[sl], c, want);
              c = prev == c ? want : prev;
            }
            if ((int)(c >> 32) == k[q]) {
              atomicAdd(&lw[sl], 1ull);
              break;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  unsigned acc = nulls;
  if (MODE == 0)
    for (int i = threadIdx.x; i < SLOTS; i += BS) acc += lcnt[i];
  else
    for (int i = threadIdx.x; i < SLOTS; i += BS) acc += (unsigned)lw[i];
  atomicAdd(&out[1], acc);
}

// ---- X: throughput vs latency.  XM 1: two blind atomics per key; XM 2: blind read + blind atomic
// (independent); XM 3: read one batch AHEAD (software-pipelined verify): reads of batch i+1 are
// in the LDS queue while batch i is compared and added.
template <int BITS, int BS, int XM>
__global__ __launch_bounds__(BS) void v_x(const int32_t *__restrict__ keys, uint64_t n, unsigned *out) {
  constexpr int SLOTS = 1 << BITS;
  constexpr int U = 4;
  __shared__ int lkeys[SLOTS];
  __shared__ unsigned lcnt[SLOTS + 64];
  for (int i = threadIdx.x; i < SLOTS; i += BS) {
    lkeys[i] = EMPTY;
    lcnt[i] = 0;
  }
  if (threadIdx.x < 64) lcnt[SLOTS + threadIdx.x] = 0;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  const int4 *vk = (const int4 *)keys;
  const uint64_t nv = n / 4;
  const uint64_t per = (nv + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = blockIdx.x * per, hi = lo + per < nv ? lo + per : nv;
  unsigned acc = 0;
  int4 np[U];
  auto issue = [&](uint64_t v0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t v = v0 + (uint64_t)u * BS;
      np[u] = v < hi ? vk[v] : make_int4(0, 0, 0, 0);
    }
  };
  issue(lo + threadIdx.x);
  if (XM == 3) {
    // pipelined: state of the batch whose reads are in flight
    int kc[U * 4];
    uint32_t sc[U * 4];
    int cur[U * 4];
    bool have = false;
    for (uint64_t v0 = lo + threadIdx.x; v0 < hi + (uint64_t)BS * U; v0 += (uint64_t)BS * U) {
      int kn[U * 4];
      uint32_t sn[U * 4];
      int curn[U * 4];
      const bool more = v0 < hi;
      if (more) {
        int4 p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = np[u];
        issue(v0 + (uint64_t)BS * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          kn[u * 4 + 0] = p[u].x;
          kn[u * 4 + 1] = p[u].y;
          kn[u * 4 + 2] = p[u].z;
          kn[u * 4 + 3] = p[u].w;
        }
#pragma unroll
        for (int q = 0; q < U * 4; ++q) sn[q] = slot_of<1, BITS>(kn[q]);
#pragma unroll
        for (int q = 0; q < U * 4; ++q) curn[q] = lkeys[sn[q]];  // reads of the NEXT batch
      }
      if (have) {
        unsigned miss = 0;
#pragma unroll
        for (int q = 0; q < U * 4; ++q) {
          const bool hit = cur[q] == kc[q];
          atomicAdd(&lcnt[hit ? sc[q] : (uint32_t)SLOTS + lane], 1u);
          miss |= (hit ? 0u : 1u) << q;
        }
        if (miss) {
#pragma unroll
          for (int q = 0; q < U * 4; ++q) {
            if (!((miss >> q) & 1)) continue;
            uint32_t h = sc[q];
            for (int pr = 0; pr < 256; ++pr) {
              uint32_t sl = (h + pr) & (SLOTS - 1);
              int c = lkeys[sl];
              if (c == EMPTY) {
                c = atomicCAS(&lkeys[sl], EMPTY, kc[q]);
                if (c == EMPTY) c = kc[q];
              }
              if (c == kc[q]) {
                atomicAdd(&lcnt[sl], 1u);
                break;
              }
            }
          }
        }
      }
      if (more) {
#pragma unroll
        for (int q = 0; q < U * 4; ++q) {
          kc[q] = kn[q];
          sc[q] = sn[q];
          cur[q] = curn[q];
        }
      }
      have = more;
    }
  } else {
    for (uint64_t v0 = lo + threadIdx.x; v0 < hi; v0 += (uint64_t)BS * U) {
      int4 p[U];
#pragma unroll
      for (int u = 0; u < U; ++u) p[u] = np[u];
      issue(v0 + (uint64_t)BS * U);
      int k[U * 4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        k[u * 4 + 0] = p[u].x;
        k[u * 4 + 1] = p[u].y;
        k[u * 4 + 2] = p[u].z;
        k[u * 4 + 3] = p[u].w;
      }
#pragma unroll
      for (int q = 0; q < U * 4; ++q) {
        const uint32_t s = slot_of<1, BITS>(k[q]);
        if (XM == 1) {
          atomicAdd(&lcnt[s], 1u);
          atomicAdd((unsigned *)&lkeys[s ^ 1], 1u);
        } else {
          acc += lkeys[s];
          atomicAdd(&lcnt[s], 1u);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SLOTS; i += BS) acc += lcnt[i];
  atomicAdd(&out[1], acc);
}

// ---- Y: branch-free loop: unconditional (clamped) loads, fixed trip count, verify + add ------
// YM 0: loads at the top of the iteration (no software prefetch)   YM 1: prefetch one iteration
template <int BITS, int BS, int YM, int U>
__global__ __launch_bounds__(BS) void v_y(const int32_t *__restrict__ keys, uint64_t n, unsigned *out) {
  constexpr int SLOTS = 1 << BITS;
  __shared__ int lkeys[SLOTS];
  __shared__ unsigned lcnt[SLOTS + 64];
  for (int i = threadIdx.x; i < SLOTS; i += BS) {
    lkeys[i] = EMPTY;
    lcnt[i] = 0;
  }
  if (threadIdx.x < 64) lcnt[SLOTS + threadIdx.x] = 0;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  const int4 *vk = (const int4 *)keys;
  const uint64_t nv = n / 4;
  const uint64_t per = (nv + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = blockIdx.x * per, hi = lo + per < nv ? lo + per : nv;
  const int iters = (int)((per + (uint64_t)BS * U - 1) / ((uint64_t)BS * U));
  const uint64_t last = hi - 1;
  int4 np[U];
  if (YM == 1) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t v = lo + threadIdx.x + (uint64_t)u * BS;
      np[u] = vk[v < hi ? v : last];
    }
  }
  for (int it = 0; it < iters; ++it) {
    const uint64_t v0 = lo + threadIdx.x + (uint64_t)it * BS * U;
    int4 p[U];
    if (YM == 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) p[u] = np[u];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint64_t v = v0 + (uint64_t)BS * U + (uint64_t)u * BS;
        np[u] = vk[v < hi ? v : last];
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint64_t v = v0 + (uint64_t)u * BS;
        p[u] = vk[v < hi ? v : last];
      }
    }
    int k[U * 4];
    uint32_t s[U * 4];
    unsigned live = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      k[u * 4 + 0] = p[u].x;
      k[u * 4 + 1] = p[u].y;
      k[u * 4 + 2] = p[u].z;
      k[u * 4 + 3] = p[u].w;
      live |= ((v0 + (uint64_t)u * BS < hi) ? 15u : 0u) << (u * 4);
    }
#pragma unroll
    for (int q = 0; q < U * 4; ++q) s[q] = slot_of<1, BITS>(k[q]);
    int cur[U * 4];
#pragma unroll
    for (int q = 0; q < U * 4; ++q) cur[q] = lkeys[s[q]];
    unsigned miss = 0;
#pragma unroll
    for (int q = 0; q < U * 4; ++q) {
      const bool lv = (live >> q) & 1;
      const bool hit = lv & (cur[q] == k[q]);
      atomicAdd(&lcnt[hit ? s[q] : (uint32_t)SLOTS + lane], 1u);
      miss |= ((lv & !hit) ? 1u : 0u) << q;
    }
    if (miss) {
#pragma unroll 1
      for (int q = 0; q < U * 4; ++q) {
        if (!((miss >> q) & 1)) continue;
        uint32_t h = s[q];
        for (int pr = 0; pr < 256; ++pr) {
          uint32_t sl = (h + pr) & (SLOTS - 1);
          int c = lkeys[sl];
          if (c == EMPTY) {
            c = atomicCAS(&lkeys[sl], EMPTY, k[q]);
            if (c == EMPTY) c = k[q];
          }
          if (c == k[q]) {
            atomicAdd(&lcnt[sl], 1u);
            break;
          }
        }
      }
    }
  }
  __syncthreads();
  unsigned acc = 0;
  for (int i = threadIdx.x; i < SLOTS; i += BS) acc += lcnt[i];
  atomicAdd(&out[1], acc);
}

template <typename F>
float timeit(F f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 3; r++) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / 3 * 1000;
}

int main() {
  uint64_t n = 45000000;
  int32_t *k;
  unsigned *out;
  uint8_t *valid;
  CK(hipMalloc(&k, n * 4));
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&valid, n / 8 + 16));
  CK(hipMemset(valid, 0xF7, n / 8 + 16));
  CK(hipMemset(out, 0, 64));
  double cards[] = {3, 36, 976, 7420};
  printf("%6s | %7s | %7s %7s %7s %7s | %7s %7s %7s %7s | %7s %7s %7s | %7s %7s %7s\n", "card", "stream",
         "h14f", "h14m", "h13mx2", "h12mx4", "c14f", "c14m", "c14x", "c14mU8", "c13mx2", "c13mx2v", "c12mx4",
         "w13m", "w13mx2", "w12mx4");
  for (double c : cards) {
    gen<<<2048, 256>>>(k, n, c, 1.1, (uint32_t)c);
    CK(hipDeviceSynchronize());
    float s0 = timeit([&] { v_stream<256><<<2048, 256>>>(k, n, out); });
    float h0 = timeit([&] { v_hist<14, 1024, 0><<<256, 1024>>>(k, n, out); });
    float h1 = timeit([&] { v_hist<14, 1024, 1><<<256, 1024>>>(k, n, out); });
    float h2 = timeit([&] { v_hist<13, 1024, 1><<<512, 1024>>>(k, n, out); });
    float h3 = timeit([&] { v_hist<12, 512, 1><<<1024, 512>>>(k, n, out); });
    float c0 = timeit([&] { v_count<14, 1024, 0, 0, 4, false><<<256, 1024>>>(k, valid, n, out); });
    float c1 = timeit([&] { v_count<14, 1024, 1, 0, 4, false><<<256, 1024>>>(k, valid, n, out); });
    float c2 = timeit([&] { v_count<14, 1024, 2, 0, 4, false><<<256, 1024>>>(k, valid, n, out); });
    float c3 = timeit([&] { v_count<14, 1024, 1, 0, 8, false><<<256, 1024>>>(k, valid, n, out); });
    float c4 = timeit([&] { v_count<13, 1024, 1, 0, 4, false><<<512, 1024>>>(k, valid, n, out); });
    float c5 = timeit([&] { v_count<13, 1024, 1, 0, 4, true><<<512, 1024>>>(k, valid, n, out); });
    float c6 = timeit([&] { v_count<12, 512, 1, 0, 4, false><<<1024, 512>>>(k, valid, n, out); });
    float w0 = timeit([&] { v_count<13, 1024, 1, 1, 4, false><<<256, 1024>>>(k, valid, n, out); });
    float w1 = timeit([&] { v_count<13, 1024, 1, 1, 4, false><<<512, 1024>>>(k, valid, n, out); });
    float w2 = timeit([&] { v_count<12, 512, 1, 1, 4, false><<<1024, 512>>>(k, valid, n, out); });
    float x1 = timeit([&] { v_x<13, 1024, 1><<<256, 1024>>>(k, n, out); });
    float x2 = timeit([&] { v_x<13, 1024, 2><<<256, 1024>>>(k, n, out); });
    float x3 = timeit([&] { v_x<14, 1024, 3><<<256, 1024>>>(k, n, out); });
    float x4 = timeit([&] { v_x<13, 1024, 3><<<512, 1024>>>(k, n, out); });
    float y0 = timeit([&] { v_y<14, 1024, 0, 4><<<256, 1024>>>(k, n, out); });
    float y1 = timeit([&] { v_y<14, 1024, 1, 4><<<256, 1024>>>(k, n, out); });
    float y2 = timeit([&] { v_y<14, 1024, 0, 2><<<256, 1024>>>(k, n, out); });
    float y3 = timeit([&] { v_y<13, 1024, 0, 4><<<512, 1024>>>(k, n, out); });
    float y4 = timeit([&] { v_y<13, 512, 0, 4><<<512, 512>>>(k, n, out); });
    float y5 = timeit([&] { v_y<14, 512, 0, 4><<<256, 512>>>(k, n, out); });
    float y6 = timeit([&] { v_y<14, 1024, 0, 1><<<256, 1024>>>(k, n, out); });
    printf("   Y: noPF %.1f  PF %.1f  U2 %.1f  13x2 %.1f  13/512x2 %.1f  14/512 %.1f  U1 %.1f\n", y0, y1, y2, y3, y4, y5, y6);
    printf("%6.0f | %7.1f | %7.1f %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f || 2atom %.1f rd+atom %.1f pipe14 %.1f pipe13x2 %.1f\n",
           c, s0, h0, h1, h2, h3, c0, c1, c2, c3, c4, c5, c6, w0, w1, w2, x1, x2, x3, x4);
  }
  return 0;
}
