// What limits the random probes of the cache-mode encode (one 8-byte slot per missing row)?
//   hipcc --offload-arch=gfx950 -O3 -o probe_rate probe_rate.hip
// (1) random 8-byte loads per second over tables of 2 MB .. 2 GB: 256 workgroups x 1024 threads
//     (the encode's shape: 16 waves per CU) and 2048 x 256 (32 waves per CU), 8 independent loads in
//     flight per lane, `frac` of the lanes active per load instruction (a miss rate).
// (2) the same with a 4-in / 8-out stream running beside it on a second stream (what the encode's own
//     key / label streams do to the table's residency in L2 / Infinity Cache).
// (3) integer VALU issue: a chain-free block of v_add / v_xor / v_mul_u32_u24 per wave -> cycles per
//     wave instruction and SIMD (is a wave64 VALU instruction 2 or 4 cycles on gfx950?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// every lane: `iters` batches of 8 independent loads at hashed slots; lanes with (hash & 255) >= act skip
// FLAVOUR of the probe load: 0 plain, 1 non-temporal, 2 relaxed agent-scope atomic load (sc1),
// 3 relaxed system-scope atomic load (sc0 sc1), 4 plain 4-byte load
template <int FLAVOUR>
__device__ __forceinline__ unsigned long long probe_load(const unsigned long long *p) {
  if constexpr (FLAVOUR == 1) return __builtin_nontemporal_load(p);
  if constexpr (FLAVOUR == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if constexpr (FLAVOUR == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if constexpr (FLAVOUR == 4) return (unsigned long long)*reinterpret_cast<const unsigned *>(p);
  return *p;
}
template <int BS, int FLAVOUR = 0>
__global__ __launch_bounds__(BS) void probe_kernel(const unsigned long long *__restrict__ table, uint64_t slots,
                                                   int iters, unsigned act, unsigned long long *sink) {
  const uint32_t gid = blockIdx.x * BS + threadIdx.x;
  unsigned long long acc = 0;
  uint32_t s = gid * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    unsigned long long e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      s = mix(s + q + 1);
      const bool on = (s & 255u) < act;
      const uint64_t sl = ((uint64_t)mix(s ^ 0x9E3779B9u) * slots) >> 32;
      e[q] = probe_load<FLAVOUR>(table + (on ? sl : 0));
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc += e[q];
  }
  if (acc == 0x1234567ull) sink[0] = acc;
}

typedef int v2i __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void stream_kernel(const int *__restrict__ x, double *__restrict__ y, size_t n, int reps) {
  const size_t nr = n / 128;
  const unsigned lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = (size_t)gridDim.x * 4;
  for (int rep = 0; rep < reps; ++rep)
    for (size_t r = wave; r < nr; r += nw) {
      v2i a = __builtin_nontemporal_load((const v2i *)(x + r * 128) + lane);
      v2d o = {(double)a.x, (double)a.y};
      __builtin_nontemporal_store(o, (v2d *)(y + r * 128) + lane);
    }
}

template <int KIND>
__global__ __launch_bounds__(1024) void valu_kernel(uint32_t *out, int iters) {
  uint32_t a[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q] = threadIdx.x * 77u + q;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (KIND == 0) a[q] = a[q] + (a[(q + 1) & 15] ^ 0x55u);          // v_xor + v_add (or one v_xad)
        if (KIND == 1) a[q] = __umul24(a[q], 0x5bd1e9u) + a[(q + 1) & 15];  // v_mad_u32_u24
        if (KIND == 2) a[q] = a[q] * 0x9E3779B1u + a[(q + 1) & 15];       // v_mul_lo_u32 + add
        if (KIND == 3) a[q] = (a[q] < a[(q + 1) & 15]) ? a[q] + 3u : a[(q + 1) & 15];  // v_cmp + v_cndmask + add
        if (KIND == 4) a[q] = (a[q] * 0x9E3779B1u) ^ a[(q + 1) & 15];      // v_mul_lo_u32 + v_xor
        if (KIND == 5) a[q] = __umulhi(a[q], 0x9E3779B1u) ^ a[(q + 1) & 15];  // v_mul_hi_u32 + v_xor
        if (KIND == 6) a[q] = a[q] ^ (a[(q + 1) & 15] >> 7);               // v_lshrrev + v_xor
      }
  }
  uint32_t s = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) s ^= a[q];
  if (s == 0x12345u) out[0] = s;
}

int main(int argc, char **argv) {
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1));
  CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned long long *sink;
  CK(hipMalloc(&sink, 64));
  const size_t maxb = 2048ull << 20;
  unsigned long long *table;
  CK(hipMalloc(&table, maxb));
  CK(hipMemset(table, 1, maxb));
  const size_t n = 45000000;
  int *x;
  double *y;
  CK(hipMalloc(&x, n * 4));
  CK(hipMalloc(&y, n * 8));
  CK(hipMemset(x, 1, n * 4));
  auto time_probe = [&](int shape, size_t mb, unsigned act, bool with_stream) {
    const uint64_t slots = (mb << 20) / 8;
    const int iters = 24;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      if (with_stream) stream_kernel<<<512, 256, 0, s2>>>(x, y, n, 6);
      CK(hipEventRecord(e0, s1));
      if (shape == 0) probe_kernel<1024><<<256, 1024, 0, s1>>>(table, slots, iters, act, sink);
      else probe_kernel<256><<<2048, 256, 0, s1>>>(table, slots, iters * 1024 * 256 / (2048 * 256), act, sink);
      CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    const double threads = shape == 0 ? 256.0 * 1024 : 2048.0 * 256;
    const double it = shape == 0 ? iters : iters * 1024 * 256 / (2048 * 256);
    const double probes = threads * it * 8 * (act / 256.0);
    return probes / (best * 1e-3) / 1e9;
  };
  const bool only_valu = argc > 1 && argv[1][0] == 'v';
  if (!only_valu) {
  printf("random 8-byte loads, G/s   (shape 0: 256 x 1024 threads, shape 1: 2048 x 256)\n");
  printf("%8s %6s | %10s %10s | %10s %10s\n", "table MB", "active", "s0 alone", "s0+stream", "s1 alone", "s1+stream");
  for (size_t mb : {2, 128})
    for (unsigned act : {64u})
      printf("%8zu %6.2f | %10.1f %10.1f | %10.1f %10.1f\n", mb, act / 256.0, time_probe(0, mb, act, false),
             time_probe(0, mb, act, true), time_probe(1, mb, act, false), time_probe(1, mb, act, true));
  auto time_flavour = [&](int fl, size_t mb, unsigned act) {
    const uint64_t slots = (mb << 20) / 8;
    const int iters = 24;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s1));
      if (fl == 0) probe_kernel<1024, 0><<<256, 1024, 0, s1>>>(table, slots, iters, act, sink);
      if (fl == 1) probe_kernel<1024, 1><<<256, 1024, 0, s1>>>(table, slots, iters, act, sink);
      if (fl == 2) probe_kernel<1024, 2><<<256, 1024, 0, s1>>>(table, slots, iters, act, sink);
      if (fl == 3) probe_kernel<1024, 3><<<256, 1024, 0, s1>>>(table, slots, iters, act, sink);
      if (fl == 4) probe_kernel<1024, 4><<<256, 1024, 0, s1>>>(table, slots, iters, act, sink);
      CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    return 256.0 * 1024 * iters * 8 * (act / 256.0) / (best * 1e-3) / 1e9;
  };
  printf("load flavours (256 x 1024 threads), G/s:  plain / nt / agent sc1 / system sc0 sc1 / 4-byte\n");
  for (size_t mb : {2, 32, 128, 1024})
    for (unsigned act : {64u, 256u})
      printf("%8zu %6.2f | %8.1f %8.1f %8.1f %8.1f %8.1f\n", mb, act / 256.0, time_flavour(0, mb, act),
             time_flavour(1, mb, act), time_flavour(2, mb, act), time_flavour(3, mb, act), time_flavour(4, mb, act));
  // memory kinds: the same random loads from uncached / fine-grained device allocations
  for (unsigned flag : {(unsigned)hipDeviceMallocUncached, (unsigned)hipDeviceMallocFinegrained}) {
    unsigned long long *t2 = nullptr;
    if (hipExtMallocWithFlags((void **)&t2, 1024ull << 20, flag) != hipSuccess) {
      printf("hipExtMallocWithFlags(%u) failed\n", flag);
      (void)hipGetLastError();
      continue;
    }
    CK(hipMemset(t2, 1, 1024ull << 20));
    unsigned long long *keep = table;
    table = t2;
    printf("allocation flag %u (1 uncached / 3?): plain / nt / sc1 / sc0sc1 / 4-byte at 128 MB, 1 GB:\n", flag);
    for (size_t mb : {128, 1024})
      for (unsigned act : {64u, 256u})
        printf("%8zu %6.2f | %8.1f %8.1f %8.1f %8.1f %8.1f\n", mb, act / 256.0, time_flavour(0, mb, act),
               time_flavour(1, mb, act), time_flavour(2, mb, act), time_flavour(3, mb, act), time_flavour(4, mb, act));
    table = keep;
    CK(hipFree(t2));
  }
  }
  // (3) VALU
  uint32_t *o;
  CK(hipMalloc(&o, 64));
  auto time_valu = [&](int kind) {
    const int iters = 2000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s1));
      if (kind == 0) valu_kernel<0><<<256, 1024, 0, s1>>>(o, iters);
      if (kind == 1) valu_kernel<1><<<256, 1024, 0, s1>>>(o, iters);
      if (kind == 2) valu_kernel<2><<<256, 1024, 0, s1>>>(o, iters);
      if (kind == 3) valu_kernel<3><<<256, 1024, 0, s1>>>(o, iters);
      if (kind == 4) valu_kernel<4><<<256, 1024, 0, s1>>>(o, iters);
      if (kind == 5) valu_kernel<5><<<256, 1024, 0, s1>>>(o, iters);
      if (kind == 6) valu_kernel<6><<<256, 1024, 0, s1>>>(o, iters);
      CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    // per SIMD: 4 waves x iters x 128 statement instances
    return best * 1e-3 * 2.4e9 / (4.0 * iters * 128);
  };
  printf("VALU: cycles (at 2.4 GHz) per wave and statement, 4 waves per SIMD:\n");
  printf("  xor+add %.2f   mad_u24 %.2f   mul_lo+add %.2f   cmp+cndmask+add %.2f   mul_lo,xor %.2f   mul_hi,xor %.2f   shr,xor %.2f\n",
         time_valu(0), time_valu(1), time_valu(2), time_valu(3), time_valu(4), time_valu(5), time_valu(6));
  return 0;
}
