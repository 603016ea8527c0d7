// Micro-benchmark (tools only): raw LDS op rates on gfx950 -- ds_add_u32 / ds_read_b32 /
// ds_write_b32 with conflict-free, random and skewed (hot address) patterns.  No global loads.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// OP 0: atomic add (no return), 1: read, 2: write, 3: read then atomic (independent), 4: atomic rtn
// PAT 0: lane-linear (conflict-free), 1: random uniform, 2: 15 % of lanes on ONE hot address, rest random
template <int OP, int PAT, int BS, int SLOTS>
__global__ __launch_bounds__(BS) void k(unsigned *out, int iters) {
  __shared__ unsigned l[SLOTS];
  for (int i = threadIdx.x; i < SLOTS; i += BS) l[i] = 0;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
  const unsigned lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x = x * 1664525u + 1013904223u;
      unsigned r = x >> 8;
      unsigned s;
      if (PAT == 0) s = (lane + 64 * (r & 127)) & (SLOTS - 1);
      else if (PAT == 1) s = r & (SLOTS - 1);
      else s = ((r & 1023) < 154) ? 777u : ((r >> 10) & (SLOTS - 1));
      if (OP == 0) atomicAdd(&l[s], 1u);
      else if (OP == 1) acc += l[s];
      else if (OP == 2) l[s] = r;
      else if (OP == 3) { acc += l[s]; atomicAdd(&l[s ^ 64], 1u); }
      else if (OP == 5) { acc += l[s & (SLOTS / 2 - 1)]; atomicAdd(&l[(s & (SLOTS / 2 - 1)) + SLOTS / 2], 1u); }
      else acc += atomicAdd(&l[s], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SLOTS; i += BS) acc += l[i];
  if (acc == 0x12345678) out[0] = acc;
}

// LDS ops fed by a global stream: LD 0 = keys from an LCG (no loads), 1 = int4 loads per 4 ops
template <int OP, int LD, int BS, int SLOTS>
__global__ __launch_bounds__(BS) void kl(const int *__restrict__ keys, uint64_t n, unsigned *out) {
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
template <typename F> float timeit(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < 3; r++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 3 * 1000;
}
int main() {
  unsigned *out; CK(hipMalloc(&out, 64));
  const int iters = 11;  // 256 WG x 1024 thr x 11 x 16 = 46 M ops, like one 45 M-row column
  const char *ops[] = {"atomic", "read", "write", "rd+atom", "atom_rtn", "rdLo+atHi"};
  printf("46 M ops per launch, 256 WG x 1024 thr (16 waves/CU), 16384 slots; us per launch (=> clk per wave-op at 2.4 GHz)\n");
  printf("%9s %18s %18s %18s\n", "op", "linear", "random", "15% hot");
#define ROW(OP) { float a = timeit([&]{ k<OP,0,1024,16384><<<256,1024>>>(out, iters); }); \
                  float b = timeit([&]{ k<OP,1,1024,16384><<<256,1024>>>(out, iters); }); \
                  float c = timeit([&]{ k<OP,2,1024,16384><<<256,1024>>>(out, iters); }); \
                  const double w = 256.0 * 1024 * 11 * 16 / 64 / 256; /* wave-ops per CU */ \
                  printf("%9s %8.1f (%5.1f) %8.1f (%5.1f) %8.1f (%5.1f)\n", ops[OP], a, a*2400/w, b, b*2400/w, c, c*2400/w); }
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4)
  printf("32 waves/CU (512 WG x 1024 thr, 8192 slots, half the iterations per WG):\n");
#define ROW2(OP) { float a = timeit([&]{ k<OP,0,1024,8192><<<512,1024>>>(out, 6); }); \
                  float b = timeit([&]{ k<OP,1,1024,8192><<<512,1024>>>(out, 6); }); \
                  float c = timeit([&]{ k<OP,2,1024,8192><<<512,1024>>>(out, 6); }); \
                  printf("%9s %8.1f %8.1f %8.1f  (x 11/12 to compare)\n", ops[OP], a, b, c); }
  ROW2(0) ROW2(1) ROW2(3)
  printf("4 waves/CU (256 WG x 256 thr, 4x iterations):\n");
#define ROW3(OP) { float a = timeit([&]{ k<OP,0,256,16384><<<256,256>>>(out, 44); }); \
                  float b = timeit([&]{ k<OP,1,256,16384><<<256,256>>>(out, 44); }); \
                  float c = timeit([&]{ k<OP,2,256,16384><<<256,256>>>(out, 44); }); \
                  printf("%9s %8.1f %8.1f %8.1f\n", ops[OP], a, b, c); }
  ROW3(0) ROW3(1) ROW3(3)
  printf("128 KiB table (32768 slots), 16 waves/CU; OP3 = read low half idx, atomic at idx^16384 (other half):\n");
#define ROW4(OP) { float a = timeit([&]{ k<OP,0,1024,32768><<<256,1024>>>(out, iters); }); \
                  float b = timeit([&]{ k<OP,1,1024,32768><<<256,1024>>>(out, iters); }); \
                  float c = timeit([&]{ k<OP,2,1024,32768><<<256,1024>>>(out, iters); }); \
                  printf("%9s %8.1f %8.1f %8.1f\n", ops[OP], a, b, c); }
  ROW4(0) ROW4(1) ROW4(3) ROW4(5)
  {
    int *keys; uint64_t n = 45000000; CK(hipMalloc(&keys, n * 4));
    std::vector<int> h(1 << 20); unsigned x = 1; for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (int)(x >> 3); }
    for (uint64_t off = 0; off < n; off += h.size()) CK(hipMemcpy(keys + off, h.data(), std::min<uint64_t>(h.size(), n - off) * 4, hipMemcpyHostToDevice));
    printf("stream-fed (45 M keys, uniform random, 32768 slots = 2 x 64 KiB): us\n");
#define ROW5(OP, name) { float a = timeit([&]{ kl<OP,0,1024,32768><<<256,1024>>>(keys, n, out); }); \
                   float b = timeit([&]{ kl<OP,1,1024,32768><<<256,1024>>>(keys, n, out); }); \
                   float c = timeit([&]{ kl<OP,1,512,32768><<<256,512>>>(keys, n, out); }); \
                   float d = timeit([&]{ kl<OP,1,256,32768><<<256,256>>>(keys, n, out); }); \
                   printf("%9s  lcg %8.1f  loads/1024thr %8.1f  loads/512thr %8.1f loads/256thr %8.1f\n", name, a, b, c, d); }
    ROW5(0, "atomic") ROW5(1, "read") ROW5(2, "rd+atom")
    double cards[] = {36, 976, 7420, 1e6};
    for (double c : cards) {
      gen<<<2048, 256>>>(keys, n, c, 1.1, (uint32_t)c); CK(hipDeviceSynchronize());
      printf("zipf(1.1) over %.0f ids:\n", c);
      ROW5(0, "atomic") ROW5(1, "read") ROW5(2, "rd+atom") ROW5(3, "verify") ROW5(4, "verify+ins")
    }
  }
  return 0;
}
