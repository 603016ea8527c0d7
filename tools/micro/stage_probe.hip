// Micro-benchmark (tools only): build the path-S stage-1 loop up feature by feature from the
// simplest stream-fed verify+insert loop, to see which feature costs what.
//   F bit0: validity bitmap   bit1: sentinel key   bit2: fill check   bit3: tiny replication
//   bit4: nulls+sentinel counted in registers instead of dummy slots
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
__global__ void gen(int32_t *k, uint64_t n, double card, double s, uint32_t seed) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, st = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    uint32_t r = fmix32((uint32_t)i * 2654435761u + seed);
    double u = (r + 0.5) / 4294967296.0;
    double x = pow((pow(card, 1.0 - s) - 1.0) * u + 1.0, 1.0 / (1.0 - s));
    int64_t v = (int64_t)floor(x); if (v < 1) v = 1; if (v > card) v = (int64_t)card;
    k[i] = (int32_t)((v * 2654435761ull) % 2147483648ull);
  }
}
constexpr int EMPTY = INT32_MIN;
template <int BITS, int BS, int F, int HASH, int REP = 8>
__global__ __launch_bounds__(BS) void stage(const int32_t *__restrict__ keys, const uint8_t *__restrict__ valid,
                                            uint64_t n, unsigned long long *out) {
  constexpr int SLOTS = 1 << BITS;
  __shared__ int lkeys[SLOTS];
  __shared__ unsigned lcnt[SLOTS + 128];  // + per-lane dummy words (nulls / sentinels)
  __shared__ unsigned lfill;
  for (int i = threadIdx.x; i < SLOTS; i += BS) { lkeys[i] = EMPTY; lcnt[i] = 0; }
  if (threadIdx.x < 128) lcnt[SLOTS + threadIdx.x] = 0;
  if (threadIdx.x == 0) lfill = 0;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  const int4 *vk = (const int4 *)keys;
  const uint64_t nv = n / 4;
  const uint64_t per = (nv + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = blockIdx.x * per, hi = lo + per < nv ? lo + per : nv;
  const int iters = hi > lo ? (int)((hi - lo + (uint64_t)BS * 4 - 1) / ((uint64_t)BS * 4)) : 0;
  const uint64_t last = hi ? hi - 1 : 0;
  const uint32_t rep = (F & 8) ? (lane & (unsigned)(REP - 1)) * 2053u : 0u;
  unsigned my_nulls = 0, my_sent = 0;
  bool failed = false;
  for (int it = 0; it < iters; ++it) {
    const uint64_t v0 = lo + (uint64_t)it * BS * 4 + threadIdx.x;
    int4 p[4];
    unsigned vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t v = v0 + (uint64_t)u * BS;
      const uint64_t vc = v < hi ? v : last;
      p[u] = vk[vc];
      vb[u] = (F & 1) ? (unsigned)valid[(vc * 4) >> 3] : 0xFFu;
    }
    int kk[16];
    uint32_t sl[16];
    unsigned live = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t v = v0 + (uint64_t)u * BS;
      const unsigned inr = v < hi ? 15u : 0u;
      const unsigned bits = (vb[u] >> ((v * 4) & 7)) & inr;
      kk[u*4] = p[u].x; kk[u*4+1] = p[u].y; kk[u*4+2] = p[u].z; kk[u*4+3] = p[u].w;
      if (F & 16) my_nulls += __popc(inr & ~bits);
      live |= bits << (u * 4);
    }
    unsigned cls = 0;  // bit q: goes to the table (valid, not the sentinel)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool v = (live >> q) & 1;
      const bool sent = (F & 2) ? (kk[q] == EMPTY) : false;
      const bool t = v & !sent;
      uint32_t h = HASH == 0 ? (fmix32((uint32_t)kk[q]) >> (32 - BITS)) : (((uint32_t)kk[q] * 0x9E3779B1u) >> (32 - BITS));
      h = (h + rep) & (SLOTS - 1);

      // arithmetic select: with `t ? h : dummy` the compiler sinks the multiply into an
      // exec-masked block per key (16 s_and_saveexec / s_cbranch_execz ladders per batch)
      const uint32_t tm = 0u - (uint32_t)t;
      if (F & 16) {
        my_sent += (v & sent) ? 1u : 0u;
        const uint32_t dm = (uint32_t)SLOTS + lane;
        sl[q] = dm ^ ((h ^ dm) & tm);
      } else {
        const uint32_t dm = (uint32_t)SLOTS + lane + (v ? 64u : 0u);  // nulls / sentinels counted in dummies
        sl[q] = dm ^ ((h ^ dm) & tm);
      }
      cls |= (t ? 1u : 0u) << q;
    }
    int cur[16];
    // every read of the batch is issued before the first add: an LDS read queued behind this
    // wave's own (conflict-serialised) atomics waits for all of them -- with the reads split
    // around the adds the same loop ran at 107 instead of 59 us
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) cur[q] = lkeys[sl[q] & (SLOTS - 1)];
    unsigned fill_now = (F & 4) ? lfill : 0u;
    __builtin_amdgcn_sched_barrier(0);
    unsigned miss = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool t = (cls >> q) & 1;
      const bool hit = t & (cur[q] == kk[q]);
      // table rows that miss go through the slow path; everything else lands on its target
      const bool direct = hit | !t;
      if (F & 16) {
        atomicAdd(&lcnt[hit ? sl[q] : (uint32_t)SLOTS + lane], 1u);
      } else {
        atomicAdd(&lcnt[direct ? sl[q] : (uint32_t)SLOTS + 127u], direct ? 1u : 0u);
      }
      miss |= ((t & !hit) ? 1u : 0u) << q;
    }
    if ((F & 4) && fill_now > (unsigned)(SLOTS / 4 * 3)) break;
    if (miss) {
#pragma unroll 1
      for (int q = 0; q < 16; ++q) {
        if (!((miss >> q) & 1)) continue;
        bool done = false;
        for (int pr = 0; pr < 512; ++pr) {
          const uint32_t a = (sl[q] + pr) & (SLOTS - 1);
          int c = lkeys[a];
          if (c == EMPTY) {
            c = atomicCAS(&lkeys[a], EMPTY, kk[q]);
            if (c == EMPTY) { c = kk[q]; if (F & 4) atomicAdd(&lfill, 1u); }
          }
          if (c == kk[q]) { atomicAdd(&lcnt[a], 1u); done = true; break; }
        }
        if (!done) failed = true;
      }
    }
  }
  __syncthreads();
  unsigned long long acc = my_nulls + my_sent + (failed ? 1u << 30 : 0u);
  for (int i = threadIdx.x; i < SLOTS + 128; i += BS) acc += lcnt[i];
  atomicAdd(&out[0], acc);
}

// kl2: the fast loop of lds_rate.hip (59 us) morphed step by step towards `stage` (96 us)
//  V>=1 proper power-of-two probe mask   V>=2 separate key / count arrays
//  V>=3 clamped loads + live mask        V>=4 `direct` form of the add
template <int V, int EXTRA = 128>
__global__ __launch_bounds__(1024) void kl2(const int *__restrict__ keys, uint64_t n, unsigned long long *out) {
  constexpr int BS = 1024, H = 16384;
  __shared__ unsigned l[2 * H + EXTRA];
  unsigned *lk = l, *lc = l + H;
  constexpr unsigned DUMMY = EXTRA >= 128 ? (unsigned)H : (unsigned)H - 64u;  // inside the table when no room
  for (int i = threadIdx.x; i < 2 * H + EXTRA; i += BS) l[i] = 0;
  __syncthreads();
  unsigned acc = 0;
  const unsigned lane = threadIdx.x & 63;
  const int4 *vk = (const int4 *)keys;
  const uint64_t nv = n / 4;
  const uint64_t per = V >= 3 ? (nv + gridDim.x - 1) / gridDim.x : nv / gridDim.x;
  const uint64_t lo = blockIdx.x * per, hi = lo + per < nv ? lo + per : nv;
  const int iters = V >= 3 ? (int)((hi - lo + (uint64_t)BS * 4 - 1) / ((uint64_t)BS * 4)) : (int)(per / ((uint64_t)BS * 4));
  const uint64_t last = hi - 1;
  for (int it = 0; it < iters; ++it) {
    int4 p[4];
    unsigned live = 0xFFFF;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t v = lo + (uint64_t)it * BS * 4 + (uint64_t)u * BS + threadIdx.x;
      if (V >= 3) {
        p[u] = vk[v < hi ? v : last];
        if (!(v < hi)) live &= ~(15u << (u * 4));
      } else {
        p[u] = vk[v];
      }
    }
    int kk[16];
    unsigned sl[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) { kk[u*4] = p[u].x; kk[u*4+1] = p[u].y; kk[u*4+2] = p[u].z; kk[u*4+3] = p[u].w; }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      unsigned s = ((unsigned)kk[q] * 0x9E3779B1u) >> 18;
      if (V >= 3) {
        const unsigned tm = 0u - ((live >> q) & 1u), dm = DUMMY + lane;
        sl[q] = dm ^ ((s ^ dm) & tm);
      } else {
        sl[q] = s;
      }
    }
    unsigned cur[16], miss = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) cur[q] = lk[sl[q] & (H - 1)];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool t = V >= 3 ? (live >> q) & 1 : true;
      const bool hit = t & (cur[q] == (unsigned)kk[q]);
      if (V >= 4) {
        const bool direct = hit | !t;
        atomicAdd(&lc[direct ? sl[q] : DUMMY + 63u], direct ? 1u : 0u);
      } else {
        atomicAdd(&lc[hit ? sl[q] : DUMMY + lane], 1u);
      }
      miss |= ((t & !hit) ? 1u : 0u) << q;
    }
    if (miss) {
#pragma unroll 1
      for (int q = 0; q < 16; ++q) {
        if (!((miss >> q) & 1)) continue;
        for (int pr = 0; pr < 64; ++pr) {
          unsigned a = (sl[q] + pr) & (V >= 1 ? (unsigned)H - 1 : (unsigned)H - 1 - 64);
          unsigned c = lk[a];
          if (c == 0) { c = atomicCAS(&lk[a], 0u, (unsigned)kk[q]); if (c == 0) c = (unsigned)kk[q]; }
          if (c == (unsigned)kk[q]) { atomicAdd(&lc[a], 1u); break; }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * H + EXTRA; i += BS) acc += l[i];
  atomicAdd(&out[0], (unsigned long long)acc);
}
// LDS ops fed by a global stream: LD 0 = keys from an LCG (no loads), 1 = int4 loads per 4 ops
template <int OP, int LD, int BS, int SLOTS>
__global__ __launch_bounds__(BS) void kl_orig(const int *__restrict__ keys, uint64_t n, unsigned *out) {
  __shared__ unsigned l[SLOTS];
  for (int i = threadIdx.x; i < SLOTS; i += BS) l[i] = 0;
  __syncthreads();
  unsigned acc = 0;
  const int4 *vk = (const int4 *)keys;
  const uint64_t nv = n / 4;
  const uint64_t per = nv / gridDim.x;
  const uint64_t lo = blockIdx.x * per;
  const int iters = (int)(per / ((uint64_t)BS * 4));
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int it = 0; it < iters; ++it) {
    int4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (LD == 1) {
        p[u] = vk[lo + (uint64_t)it * BS * 4 + (uint64_t)u * BS + threadIdx.x];
      } else {
        x = x * 1664525u + 1013904223u; p[u].x = x >> 3;
        x = x * 1664525u + 1013904223u; p[u].y = x >> 3;
        x = x * 1664525u + 1013904223u; p[u].z = x >> 3;
        x = x * 1664525u + 1013904223u; p[u].w = x >> 3;
      }
    }
    int kk[16];
    unsigned sl[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) { kk[u*4] = p[u].x; kk[u*4+1] = p[u].y; kk[u*4+2] = p[u].z; kk[u*4+3] = p[u].w; }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      unsigned s = ((unsigned)kk[q] * 0x9E3779B1u) >> 18;  // 14 bits
      s &= (SLOTS / 2 - 1);
      if (OP == 0) atomicAdd(&l[s + SLOTS / 2], 1u);
      else if (OP == 1) acc += l[s];
      else if (OP == 2) { acc += l[s]; atomicAdd(&l[s + SLOTS / 2], 1u); }
      else sl[q] = s;
    }
    if (OP >= 3) {  // verify: read all, compare, add to the slot on a hit else to a per-lane dummy
      unsigned cur[16], miss = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) cur[q] = l[sl[q]];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const bool hit = cur[q] == (unsigned)kk[q];
        atomicAdd(&l[hit ? sl[q] + SLOTS / 2 : (unsigned)SLOTS / 2 + 16320u + (threadIdx.x & 63)], 1u);
        miss |= (hit ? 0u : 1u) << q;
      }
      if (OP >= 4 && miss) {
#pragma unroll 1
        for (int q = 0; q < 16; ++q) {
          if (!((miss >> q) & 1)) continue;
          for (int pr = 0; pr < 64; ++pr) {
            unsigned a = (sl[q] + pr) & (SLOTS / 2 - 1 - 64);
            unsigned c = l[a];
            if (c == 0) { c = atomicCAS(&l[a], 0u, (unsigned)kk[q]); if (c == 0) c = (unsigned)kk[q]; }
            if (c == (unsigned)kk[q]) { atomicAdd(&l[a + SLOTS / 2], 1u); break; }
          }
        }
      }
      if (OP == 3) acc += miss;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SLOTS; i += BS) acc += l[i];
  if (acc == 0x12345678) out[0] = acc;
}

// W: ONE returning 64-bit LDS atomic per key on a {key:32 | count:32} word, verification of the
// returned key deferred by one half-batch (8 keys) so that the wave never drains its LDS queue.
//   WM 0: immediate verify (wait for all 8)   WM 1: deferred (pipelined by 8)   WM 2: no verify at all
template <int WM, int BS>
__global__ __launch_bounds__(BS) void wk(const int *__restrict__ keys, uint64_t n, unsigned long long *out) {
  constexpr int H = 16384;
  __shared__ unsigned long long lw[H];
  const unsigned long long EW = ((unsigned long long)(uint32_t)EMPTY) << 32;
  for (int i = threadIdx.x; i < H; i += BS) lw[i] = EW;
  __syncthreads();
  const int4 *vk = (const int4 *)keys;
  const uint64_t nv = n / 4;
  const uint64_t per = nv / gridDim.x;
  const uint64_t lo = blockIdx.x * per;
  const int iters = (int)(per / ((uint64_t)BS * 2));
  unsigned bad = 0;
  auto fix = [&](int key, unsigned slot) {
    // undo the blind add, then insert / find the key by linear probing with 64-bit CAS
    atomicAdd(&lw[slot], ~0ull);
    for (int pr = 0; pr < 512; ++pr) {
      const unsigned a = (slot + pr) & (H - 1);
      unsigned long long c = lw[a];
      bool done = false;
      while ((int)(c >> 32) == EMPTY) {  // claim: keep whatever transient count the word carries
        const unsigned long long want = ((unsigned long long)(uint32_t)key << 32) | (uint32_t)((uint32_t)c + 1u);
        const unsigned long long prev = atomicCAS(&lw[a], c, want);
        if (prev == c) { done = true; break; }
        c = prev;
      }
      if (done) return;
      if ((int)(c >> 32) == key) { atomicAdd(&lw[a], 1ull); return; }
    }
    bad = 1;
  };
  int kp[8];
  unsigned sp[8];
  unsigned long long op[8];
  bool have = false;
  for (int it = 0; it < iters; ++it) {
    int4 p[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) p[u] = vk[lo + (uint64_t)it * BS * 2 + (uint64_t)u * BS + threadIdx.x];
    int kk[8] = {p[0].x, p[0].y, p[0].z, p[0].w, p[1].x, p[1].y, p[1].z, p[1].w};
    unsigned sl[8];
    unsigned long long old[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) sl[q] = ((unsigned)kk[q] * 0x9E3779B1u) >> 18;
#pragma unroll
    for (int q = 0; q < 8; ++q) old[q] = atomicAdd(&lw[sl[q]], 1ull);
    if (WM == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if ((int)(old[q] >> 32) != kk[q]) fix(kk[q], sl[q]);
    } else if (WM == 1) {
      if (have) {
        unsigned miss = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) miss |= (((int)(op[q] >> 32) != kp[q]) ? 1u : 0u) << q;
        if (miss) {
#pragma unroll 1
          for (int q = 0; q < 8; ++q)
            if ((miss >> q) & 1) fix(kp[q], sp[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) { kp[q] = kk[q]; sp[q] = sl[q]; op[q] = old[q]; }
      have = true;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) bad += (unsigned)(old[q] >> 40);
    }
  }
  if (WM == 1 && have) {
#pragma unroll 1
    for (int q = 0; q < 8; ++q)
      if ((int)(op[q] >> 32) != kp[q]) fix(kp[q], sp[q]);
  }
  __syncthreads();
  unsigned long long acc = bad;
  for (int i = threadIdx.x; i < H; i += BS) acc += (uint32_t)lw[i];
  atomicAdd(&out[1], acc);
}

// HK: the NH hottest keys are counted in per-lane registers (compare against wave-uniform values),
// only the rest goes through the LDS table: removes the same-address atomic storms of Zipf data.
template <int NH, int BS>
__global__ __launch_bounds__(BS) void hk(const int *__restrict__ keys, uint64_t n, unsigned long long *out) {
  constexpr int H = 16384;
  __shared__ int lk[H];
  __shared__ unsigned lc[H + 64];
  for (int i = threadIdx.x; i < H; i += BS) { lk[i] = EMPTY; lc[i] = 0; }
  if (threadIdx.x < 64) lc[H + threadIdx.x] = 0;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  int hot[NH > 0 ? NH : 1];
#pragma unroll
  for (int j = 0; j < NH; ++j) hot[j] = (int)((((unsigned long long)(j + 1)) * 2654435761ull) % 2147483648ull);
  unsigned hc[NH > 0 ? NH : 1];
#pragma unroll
  for (int j = 0; j < NH; ++j) hc[j] = 0;
  const int4 *vk = (const int4 *)keys;
  const uint64_t nv = n / 4;
  const uint64_t per = nv / gridDim.x;
  const uint64_t lo = blockIdx.x * per;
  const int iters = (int)(per / ((uint64_t)BS * 4));
  for (int it = 0; it < iters; ++it) {
    int4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = vk[lo + (uint64_t)it * BS * 4 + (uint64_t)u * BS + threadIdx.x];
    int kk[16];
    unsigned sl[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) { kk[u*4] = p[u].x; kk[u*4+1] = p[u].y; kk[u*4+2] = p[u].z; kk[u*4+3] = p[u].w; }
    unsigned cold = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      unsigned is_hot = 0;
#pragma unroll
      for (int j = 0; j < NH; ++j) {
        const unsigned e = kk[q] == hot[j] ? 1u : 0u;
        hc[j] += e;
        is_hot |= e;
      }
      const unsigned s = ((unsigned)kk[q] * 0x9E3779B1u) >> 18;
      const unsigned tm = is_hot - 1u, dm = (unsigned)H + lane;  // tm = all ones when not hot
      sl[q] = dm ^ ((s ^ dm) & tm);
      cold |= (is_hot ^ 1u) << q;
    }
    int cur[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) cur[q] = lk[sl[q] & (H - 1)];
    unsigned miss = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool t = (cold >> q) & 1;
      const bool hit = t & (cur[q] == kk[q]);
      atomicAdd(&lc[hit ? sl[q] : (unsigned)H + lane], 1u);
      miss |= ((t & !hit) ? 1u : 0u) << q;
    }
    if (miss) {
#pragma unroll 1
      for (int q = 0; q < 16; ++q) {
        if (!((miss >> q) & 1)) continue;
        for (int pr = 0; pr < 512; ++pr) {
          const unsigned a = (sl[q] + pr) & (H - 1);
          int c = lk[a];
          if (c == EMPTY) { c = atomicCAS(&lk[a], EMPTY, kk[q]); if (c == EMPTY) c = kk[q]; }
          if (c == kk[q]) { atomicAdd(&lc[a], 1u); break; }
        }
      }
    }
  }
  __syncthreads();
  unsigned long long acc = 0;
#pragma unroll
  for (int j = 0; j < NH; ++j) acc += hc[j];
  for (int i = threadIdx.x; i < H; i += BS) acc += lc[i];
  atomicAdd(&out[1], acc);
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < 3; r++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 3 * 1000;
}
int main() {
  uint64_t n = 45000000; int32_t *k; unsigned long long *out; uint8_t *valid;
  CK(hipMalloc(&k, n * 4)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&valid, n / 8 + 64));
  CK(hipMemset(valid, 0xB7, n / 8 + 64));
  double cards[] = {3, 36, 976, 7420};
  printf("%6s | %7s %7s %7s %7s %7s %7s %7s %7s | %7s %7s | check\n", "card", "base", "+valid", "+sent", "+fill", "v+s+f", "vsf+reg", "vsf+rep", "fmix", "512thr", "x2/13");
  for (double c : cards) {
    gen<<<2048, 256>>>(k, n, c, 1.1, (uint32_t)c); CK(hipDeviceSynchronize());
    auto run = [&](auto kern, int grid, int bs) { return timeit([&] { kern<<<grid, bs>>>(k, valid, n, out); }); };
    float a0 = run(stage<14, 1024, 0, 1>, 256, 1024);
    float a1 = run(stage<14, 1024, 1, 1>, 256, 1024);
    float a2 = run(stage<14, 1024, 2, 1>, 256, 1024);
    float a3 = run(stage<14, 1024, 4, 1>, 256, 1024);
    float a4 = run(stage<14, 1024, 7, 1>, 256, 1024);
    float a5 = run(stage<14, 1024, 23, 1>, 256, 1024);
    float a6 = run(stage<14, 1024, 15, 1>, 256, 1024);
    float a7 = run(stage<14, 1024, 7, 0>, 256, 1024);
    float b0 = run(stage<14, 512, 7, 1>, 256, 512);
    float b1 = run(stage<13, 1024, 7, 1>, 512, 1024);
    CK(hipMemset(out, 0, 64));
    wk<1, 1024><<<256, 1024>>>(k, n, out);
    unsigned long long wtot[2]; CK(hipMemcpy(wtot, out, 16, hipMemcpyDeviceToHost));
    printf("        W (64-bit rtn atomic): immediate %.1f  deferred %.1f  no-verify %.1f | 512thr deferred %.1f | counted %llu (expect %llu)\n",
           timeit([&] { wk<0, 1024><<<256, 1024>>>(k, n, out); }), timeit([&] { wk<1, 1024><<<256, 1024>>>(k, n, out); }),
           timeit([&] { wk<2, 1024><<<256, 1024>>>(k, n, out); }), timeit([&] { wk<1, 512><<<256, 512>>>(k, n, out); }),
           wtot[1], (unsigned long long)(n / 4 / 256 / 2048 * 2048 * 256 * 4));
    printf("        hot keys in registers: 0: %.1f  1: %.1f  2: %.1f  4: %.1f  8: %.1f  16: %.1f | 512thr 8: %.1f\n",
           timeit([&] { hk<0, 1024><<<256, 1024>>>(k, n, out); }), timeit([&] { hk<1, 1024><<<256, 1024>>>(k, n, out); }),
           timeit([&] { hk<2, 1024><<<256, 1024>>>(k, n, out); }), timeit([&] { hk<4, 1024><<<256, 1024>>>(k, n, out); }),
           timeit([&] { hk<8, 1024><<<256, 1024>>>(k, n, out); }), timeit([&] { hk<16, 1024><<<256, 1024>>>(k, n, out); }),
           timeit([&] { hk<8, 512><<<256, 512>>>(k, n, out); }));
    { unsigned *o2 = (unsigned *)out;
      printf("        kl_orig OP4: %.1f   kl_orig OP2 (blind rd+atom): %.1f\n",
             timeit([&] { kl_orig<4, 1, 1024, 32768><<<256, 1024>>>(k, n, o2); }),
             timeit([&] { kl_orig<2, 1, 1024, 32768><<<256, 1024>>>(k, n, o2); })); }
    printf("        kl2 128 KiB exactly: V0 %.1f V1 %.1f V3 %.1f V4 %.1f | +64 B: V1 %.1f | +4 KiB: V1 %.1f\n",
           timeit([&] { kl2<0, 0><<<256, 1024>>>(k, n, out); }), timeit([&] { kl2<1, 0><<<256, 1024>>>(k, n, out); }),
           timeit([&] { kl2<3, 0><<<256, 1024>>>(k, n, out); }), timeit([&] { kl2<4, 0><<<256, 1024>>>(k, n, out); }),
           timeit([&] { kl2<1, 16><<<256, 1024>>>(k, n, out); }), timeit([&] { kl2<1, 1024><<<256, 1024>>>(k, n, out); }));
    printf("        kl2 morph: V0 %.1f  V1 %.1f  V2(=V1) -  V3 %.1f  V4 %.1f\n",
           timeit([&] { kl2<0><<<256, 1024>>>(k, n, out); }), timeit([&] { kl2<1><<<256, 1024>>>(k, n, out); }),
           timeit([&] { kl2<3><<<256, 1024>>>(k, n, out); }), timeit([&] { kl2<4><<<256, 1024>>>(k, n, out); }));
    printf("        replication (v+s+f): x1 %.1f  x2 %.1f  x4 %.1f  x8 %.1f | 512thr x2 %.1f x4 %.1f\n",
           run(stage<14, 1024, 7, 1>, 256, 1024), run(stage<14, 1024, 15, 1, 2>, 256, 1024),
           run(stage<14, 1024, 15, 1, 4>, 256, 1024), run(stage<14, 1024, 15, 1, 8>, 256, 1024),
           run(stage<14, 512, 15, 1, 2>, 256, 512), run(stage<14, 512, 15, 1, 4>, 256, 512));
    CK(hipMemset(out, 0, 8));
    stage<14, 1024, 7, 1><<<256, 1024>>>(k, valid, n, out);
    unsigned long long tot; CK(hipMemcpy(&tot, out, 8, hipMemcpyDeviceToHost));
    printf("%6.0f | %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f | %7.1f %7.1f | rows counted %llu of %llu\n", c, a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, tot, (unsigned long long)(n / 4 * 4));
  }
  return 0;
}
