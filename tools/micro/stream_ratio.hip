// What does a 4-byte-in / 8-byte-out stream (fill + normalize: int32 -> double) reach on this GPU,
// and with which access shape?  hipcc --offload-arch=gfx950 -O3 -o stream_ratio stream_ratio.hip
//   A  lane loads int4 (16 B), stores two 16-B halves of its 32 B (the shape of fill_norm_body)
//   B  lane loads two int2 512 B apart, every store instruction covers 1024 contiguous bytes
//   C  A with two vectors in flight per thread
//   D  B with four int2 per thread
// each with plain / non-temporal loads and stores, over a few grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool NT, typename V> __device__ __forceinline__ V ld(const V *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT, typename V> __device__ __forceinline__ void st(V *p, V v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ double cv(int x, double sh, double sc) { return ((double)x - sh) / sc; }

template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void kA(const int *__restrict__ x, double *__restrict__ y, size_t n, double sh, double sc) {
  const size_t nv = n / 4, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride * U) {
    v4i a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < nv) a[u] = ld<NTL>((const v4i *)x + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < nv) {
      v2d lo = {cv(a[u].x, sh, sc), cv(a[u].y, sh, sc)}, hi = {cv(a[u].z, sh, sc), cv(a[u].w, sh, sc)};
      v2d *d = (v2d *)y + 2 * (i + u * stride);
      st<NTS>(d, lo);
      st<NTS>(d + 1, hi);
    }
  }
}
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void kB(const int *__restrict__ x, double *__restrict__ y, size_t n, double sh, double sc) {
  // a wave handles U runs of 128 ints; lane l: ints [2l, 2l + 1] of each run
  const size_t nr = n / 128;                       // runs
  const unsigned lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = (size_t)gridDim.x * 4;
  for (size_t r = wave * U; r < nr; r += nw * U) {
    v2i a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (r + u < nr) a[u] = ld<NTL>((const v2i *)(x + (r + u) * 128) + lane);
#pragma unroll
    for (int u = 0; u < U; ++u) if (r + u < nr) {
      v2d o = {cv(a[u].x, sh, sc), cv(a[u].y, sh, sc)};
      st<NTS>((v2d *)(y + (r + u) * 128) + lane, o);
    }
  }
}

int main() {
  const size_t n = 45000000ull / 128 * 128, cols = 13;
  int *x; double *y;
  CK(hipMalloc(&x, n * cols * 4)); CK(hipMalloc(&y, n * cols * 8));
  CK(hipMemset(x, 1, n * cols * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char *name, auto launch) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      for (size_t c = 0; c < cols; ++c) launch(x + c * n, y + c * n);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    printf("%-28s %7.3f ms  %6.2f TB/s\n", name, best, (double)n * cols * 12 / best / 1e9);
  };
  const double sh = 3.5, sc = 7.25;
  for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
    printf("-- grid %d x 256\n", g);
#define RUN(K, NTL, NTS, U) time(#K " ntl=" #NTL " nts=" #NTS " U=" #U, [&](int *xi, double *yo) { K<NTL, NTS, U><<<g, 256>>>(xi, yo, n, sh, sc); })
    RUN(kA, true, false, 1); RUN(kA, true, true, 1); RUN(kA, false, false, 1);
    RUN(kA, true, false, 2); RUN(kA, true, true, 2);
    RUN(kB, true, false, 2); RUN(kB, true, true, 2); RUN(kB, false, false, 2);
    RUN(kB, true, false, 4); RUN(kB, true, true, 4);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
