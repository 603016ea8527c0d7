// Error reporting, version, and launch-side instrumentation (HIP-event profiler + roctx
// ranges) of the C ABI (include/nvt_hip.h).
#include <dlfcn.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "nvt_common.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"

namespace nvt {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- internal stream pools ------------------------------------------------------
namespace {
std::mutex g_pool_mu;
std::vector<SidePool *> g_pools;
}  // namespace
int side_pool(int which, SidePool **out) {
  int dev = 0;
  NVT_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (SidePool *p : g_pools)
    if (p->dev == dev && p->which == which) {
      *out = p;
      return NVT_OK;
    }
  SidePool *p = new SidePool();
  p->dev = dev;
  p->which = which;
  for (int i = 0; i < kSideStreams; ++i) {
    NVT_CHECK_HIP(hipStreamCreateWithFlags(&p->s[i], hipStreamNonBlocking));
    NVT_CHECK_HIP(hipEventCreateWithFlags(&p->join[i], hipEventDisableTiming));
  }
  NVT_CHECK_HIP(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
  NVT_CHECK_HIP(hipEventCreateWithFlags(&p->aux, hipEventDisableTiming));
  g_pools.push_back(p);
  *out = p;
  return NVT_OK;
}

// ---- HIP-event profiler ---------------------------------------------------------
namespace {
struct Rec {
  const char *name;  // string literals only
  uint64_t bytes;
  hipEvent_t a, b;
};
std::atomic<int> g_on{0};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_base = nullptr;

hipEvent_t take_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
void recycle_all() {
  for (auto &r : g_recs) {
    if (r.a) g_pool.push_back(r.a);
    if (r.b) g_pool.push_back(r.b);
  }
  g_recs.clear();
}
}  // namespace

bool prof_enabled() { return g_on.load(std::memory_order_relaxed) != 0; }

int prof_open(const char *name, uint64_t alg_bytes, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r{name, alg_bytes, take_event(), take_event()};
  if (!r.a || !r.b) return -1;
  (void)hipEventRecord(r.a, stream);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}
void prof_close(int id, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (id >= 0 && id < (int)g_recs.size()) (void)hipEventRecord(g_recs[id].b, stream);
}

// ---- roctx (optional, dlopen'ed) ----------------------------------------------------
namespace {
typedef int (*roctx_push_t)(const char *);
typedef int (*roctx_pop_t)(void);
std::once_flag g_roctx_once;
roctx_push_t g_push = nullptr;
roctx_pop_t g_pop = nullptr;
void roctx_init() {
  const char *off = getenv("NVT_ROCTX");
  if (off && off[0] == '0') return;
  const char *libs[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1",
                        "libroctx64.so", "libroctx64.so.4"};
  for (const char *l : libs) {
    void *h = dlopen(l, RTLD_LAZY | RTLD_GLOBAL);
    if (!h) continue;
    g_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
    g_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
    if (g_push && g_pop) return;
    g_push = nullptr;
    g_pop = nullptr;
  }
}
}  // namespace
void roctx_push(const char *name) {
  std::call_once(g_roctx_once, roctx_init);
  if (g_push) g_push(name);
}
void roctx_pop() {
  if (g_pop) g_pop();
}
}  // namespace nvt

// ---- mailbox: device -> host read-back through coherent pinned memory + host spin ---------
struct nvt_mailbox {
  char *host;        // [64-byte header: seq word | payload]
  uint64_t bytes;    // payload capacity
  uint64_t next_seq;
};

namespace nvt {
__global__ __launch_bounds__(256) void mailbox_post_kernel(const uint64_t *__restrict__ src,
                                                           uint64_t *dst, uint64_t nwords,
                                                           uint64_t *flag, uint64_t seq) {
  for (uint64_t i = threadIdx.x; i < nwords; i += 256) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace nvt

using namespace nvt;

extern "C" {
int nvt_mailbox_create(uint64_t bytes, nvt_mailbox **out) {
  NVT_CHECK_ARG(out, "null out pointer");
  bytes = (bytes + 7) & ~7ull;
  void *p = nullptr;
  NVT_CHECK_HIP(hipHostMalloc(&p, bytes + 64, hipHostMallocCoherent | hipHostMallocMapped));
  memset(p, 0, bytes + 64);
  nvt_mailbox *mb = new nvt_mailbox{(char *)p, bytes, 1};
  *out = mb;
  return NVT_OK;
}
void nvt_mailbox_destroy(nvt_mailbox *mb) {
  if (!mb) return;
  (void)hipHostFree(mb->host);
  delete mb;
}
void *nvt_mailbox_data(nvt_mailbox *mb) { return mb ? mb->host + 64 : nullptr; }
uint64_t nvt_mailbox_capacity(nvt_mailbox *mb) { return mb ? mb->bytes : 0; }
int nvt_mailbox_post(nvt_mailbox *mb, const void *src_device, uint64_t bytes, void *stream,
                     uint64_t *seq_out) {
  NVT_CHECK_ARG(mb && src_device && seq_out, "null pointer");
  NVT_CHECK_ARG(bytes % 8 == 0 && bytes <= mb->bytes, "bytes must be a multiple of 8 within capacity");
  NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(src_device) & 7) == 0, "source must be 8-byte aligned");
  const uint64_t seq = mb->next_seq++;
  void *dev = nullptr;  // device-side address of the mapped host buffer
  NVT_CHECK_HIP(hipHostGetDevicePointer(&dev, mb->host, 0));
  mailbox_post_kernel<<<1, 256, 0, (hipStream_t)stream>>>(
      (const uint64_t *)src_device, (uint64_t *)((char *)dev + 64), bytes / 8, (uint64_t *)dev, seq);
  NVT_CHECK_LAUNCH();
  *seq_out = seq;
  return NVT_OK;
}
int nvt_mailbox_wait(nvt_mailbox *mb, uint64_t seq, double timeout_s) {
  NVT_CHECK_ARG(mb, "null mailbox");
  const volatile uint64_t *flag = reinterpret_cast<const volatile uint64_t *>(mb->host);
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (uint64_t spins = 0;; ++spins) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) >= seq) return NVT_OK;
    if ((spins & 0xFFF) == 0xFFF) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
      if (dt > timeout_s) {
        // a failed launch never writes the word: report it instead of spinning forever
        hipError_t e = hipPeekAtLastError();
        set_error("nvt_mailbox_wait: sequence %llu not reached after %.1f s (%s)",
                  (unsigned long long)seq, dt, hipGetErrorString(e));
        return NVT_EHIP;
      }
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
}

int nvt_version(void) { return 200; }  // 0.2.0
const char *nvt_last_error(void) { return nvt::g_err; }

int nvt_prof_begin(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  recycle_all();
  if (!g_base) NVT_CHECK_HIP(hipEventCreate(&g_base));
  NVT_CHECK_HIP(hipEventRecord(g_base, nullptr));
  g_on.store(1);
  return NVT_OK;
}

// Synchronises the device, then writes a JSON object
//   {"kernels": {"<name>": [total_ms, launches, algorithmic_bytes], ...},
//    "busy_ms": <union of all scope intervals>, "span_ms": <first start .. last stop>}
// into buf (NUL-terminated, truncated to cap); *needed = bytes required incl. NUL.
int nvt_prof_report(char *buf, uint64_t cap, uint64_t *needed) {
  g_on.store(0);
  NVT_CHECK_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g_mu);
  struct Agg {
    double ms = 0;
    uint64_t n = 0, bytes = 0;
  };
  std::map<std::string, Agg> agg;
  std::vector<std::pair<float, float>> iv;
  for (auto &r : g_recs) {
    float ms = 0, t0 = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    Agg &a = agg[r.name];
    a.ms += ms;
    a.n += 1;
    a.bytes += r.bytes;
    if (g_base && hipEventElapsedTime(&t0, g_base, r.a) == hipSuccess) iv.push_back({t0, t0 + ms});
  }
  double busy = 0, span = 0;
  if (!iv.empty()) {
    std::sort(iv.begin(), iv.end());
    float lo = iv[0].first, hi = iv[0].second, last = hi;
    for (size_t i = 1; i < iv.size(); ++i) {
      if (iv[i].first > hi) {
        busy += hi - lo;
        lo = iv[i].first;
        hi = iv[i].second;
      } else if (iv[i].second > hi) {
        hi = iv[i].second;
      }
      last = std::max(last, iv[i].second);
    }
    busy += hi - lo;
    span = last - iv[0].first;
  }
  std::string s = "{\"kernels\": {";
  bool first = true;
  char tmp[256];
  for (auto &kv : agg) {
    snprintf(tmp, sizeof(tmp), "%s\"%s\": [%.6f, %llu, %llu]", first ? "" : ", ", kv.first.c_str(),
             kv.second.ms, (unsigned long long)kv.second.n, (unsigned long long)kv.second.bytes);
    s += tmp;
    first = false;
  }
  snprintf(tmp, sizeof(tmp), "}, \"busy_ms\": %.6f, \"span_ms\": %.6f}", busy, span);
  s += tmp;
  if (needed) *needed = s.size() + 1;
  if (buf && cap) {
    const size_t m = std::min<size_t>(s.size(), cap - 1);
    memcpy(buf, s.data(), m);
    buf[m] = 0;
  }
  recycle_all();
  return NVT_OK;
}

// roctx range around host-side phases (operator fit / transform), named after the reference's
// @annotate strings; no-ops when the roctx library is absent
void nvt_range_push(const char *name) { roctx_push(name ? name : ""); }
void nvt_range_pop(void) { roctx_pop(); }
}

// ---------------------------------------------------------------------------------------------
// Seeded TargetEncoding folds on the device (target_encoding.py:427-439):
//     state = numpy.random.RandomState(fold_seed); fold = state.choice(arange(kfold), n)
// is, for integer seeds, MT19937 seeded by init_genrand(seed), 32-bit outputs masked to the
// smallest 2^k - 1 >= kfold - 1 and REJECTED while above kfold - 1 (numpy's legacy bounded
// integers); the i-th accepted value is fold[i].  One workgroup walks the generator: the
// twist of a 624-word state is three dependent phases of <= 227 independent words, the
// tempered words are filtered and compacted through a block scan.  ~1500 cycles per 624 draws:
// 45 M folds of kfold = 5 (72 M draws) in ~70 ms, once per (kfold, seed) -- the reference
// re-seeds per partition, so every partition uses a prefix of the same sequence and the
// engine keeps the longest one.  (Before: numpy on the host + a copy, 0.3 s for 45 M rows.)
// ---------------------------------------------------------------------------------------------
namespace nvt {
namespace {
constexpr int kMtN = 624, kMtM = 397, kMtBS = 256;
__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7FFFFFFFu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}
__global__ __launch_bounds__(kMtBS) void mt19937_folds_kernel(uint32_t seed, uint32_t kfold,
                                                             uint64_t n, uint8_t *__restrict__ out) {
  __shared__ uint32_t mt[kMtN];
  __shared__ unsigned wsum[kMtBS / kWave];
  const unsigned t = threadIdx.x;
  if (t == 0) {
    uint32_t x = seed;
    mt[0] = x;
    for (int i = 1; i < kMtN; ++i) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
      mt[i] = x;
    }
  }
  uint32_t mask = kfold - 1u;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint64_t done = 0;
  __syncthreads();
  while (done < n) {
    // ---- twist: words [0, 227) from the old state, [227, 454) and [454, 624) from new words ----
    {
      uint32_t v = 0;
      if (t < 227) v = mt_twist(mt[t], mt[t + 1], mt[t + kMtM]);
      __syncthreads();
      if (t < 227) mt[t] = v;
      __syncthreads();
      const unsigned i = 227 + t;
      if (t < 227) v = mt_twist(mt[i], mt[i + 1], mt[i - 227]);
      __syncthreads();
      if (t < 227) mt[i] = v;
      __syncthreads();
      const unsigned k = 454 + t;
      if (k < (unsigned)kMtN) v = mt_twist(mt[k], mt[k == kMtN - 1 ? 0 : k + 1], mt[k - 227]);
      __syncthreads();
      if (k < (unsigned)kMtN) mt[k] = v;
      __syncthreads();
    }
    // ---- temper, filter, compact: thread t owns words 3t, 3t+1, 3t+2 (208 threads) ----
    uint32_t val[3];
    unsigned acc = 0, cnt = 0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const unsigned i = 3 * t + e;
      uint32_t y = i < (unsigned)kMtN ? mt[i] : 0xFFFFFFFFu;
      y ^= y >> 11;
      y ^= (y << 7) & 0x9D2C5680u;
      y ^= (y << 15) & 0xEFC60000u;
      y ^= y >> 18;
      val[e] = y & mask;
      const bool ok = i < (unsigned)kMtN && val[e] < kfold;
      acc |= (ok ? 1u : 0u) << e;
      cnt += ok ? 1u : 0u;
    }
    unsigned inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(inc, off, 64);
      if (lane_id() >= (unsigned)off) inc += o;
    }
    if (lane_id() == 63) wsum[t / kWave] = inc;
    __syncthreads();
    unsigned wb = 0, tot = 0;
    for (unsigned q = 0; q < kMtBS / kWave; ++q) {
      if (q < t / kWave) wb += wsum[q];
      tot += wsum[q];
    }
    uint64_t pos = done + wb + inc - cnt;
#pragma unroll
    for (int e = 0; e < 3; ++e)
      if ((acc >> e) & 1u) {
        if (pos < n) out[pos] = (uint8_t)val[e];
        ++pos;
      }
    done += tot;
    __syncthreads();  // wsum / mt are rewritten by the next round
  }
}
// ---- the same stream from G chunks in parallel (jump-ahead) ---------------------------------
// MT19937's word sequence is a linear recurrence over GF(2); with g(x) = x^J mod phi(x) (phi: its
// characteristic polynomial, degree 19937) the 624-word window J words further on is
//     W'[m] = XOR over {i : g_i = 1} of w[i + m]
// where w is the sequence run forward from the current window (tools/mt_jump_polys.py derives
// phi from numpy's own generator, checks this identity against it, and writes the coefficients
// of x^(2^18 * 2^k) mod phi, k = 0 .. 13, into nvt_mt_jump_polys.inc).  Chunk c of 2^18 words
// starts from the seed window jumped by the polynomials of the set bits of c (<= 14 jumps of
// ~0.1 ms in LDS), then generates, tempers, filters and compacts its words exactly like the
// serial kernel; a second launch puts the chunks' accepted values behind each other.  21 M folds:
// 53 ms -> 3.8 ms measured; a 2^28-row partition: 710 ms -> 28 ms (tools/folds_probe.py).
#include "nvt_mt_jump_polys.inc"
constexpr int kMtDeg = 19937;
constexpr int kMtSeq = kMtDeg + kMtN;            // words a jump looks at
constexpr int kMtParBS = 640;                    // 10 waves: thread m < 624 owns window word m
constexpr uint64_t kMtChunk = 1ull << kMtJumpLog2;

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9D2C5680u;
  y ^= (y << 15) & 0xEFC60000u;
  y ^= y >> 18;
  return y;
}

__global__ __launch_bounds__(kMtParBS) void mt19937_chunks_kernel(uint32_t seed, uint32_t kfold,
                                                                 uint64_t nchunks, uint8_t *__restrict__ tmp,
                                                                 unsigned *__restrict__ counts) {
  __shared__ uint32_t w[kMtSeq + 227];   // the sequence a jump reads; its first 624 words = the window
  __shared__ uint32_t gpoly[kMtJumpWords];
  __shared__ unsigned wsum[kMtParBS / kWave];
  const unsigned t = threadIdx.x;
  const uint64_t c = blockIdx.x;
  if (c >= nchunks) return;
  if (t == 0) {
    uint32_t x = seed;
    w[0] = x;
    for (int i = 1; i < kMtN; ++i) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
      w[i] = x;
    }
  }
  __syncthreads();
  // ---- jump to word c * 2^18: one polynomial per set bit of c ----
  for (int k = 0; k < kMtJumpPolys; ++k) {
    if (!((c >> k) & 1ull)) continue;   // (uniform)
    if (t < (unsigned)kMtJumpWords) gpoly[t] = kMtJumpPoly[k][t];
    // the sequence forward of the window, 227 independent words per step
    for (int j0 = 0; j0 < kMtDeg; j0 += 227) {
      const int j = j0 + (int)t;
      if (t < 227 && j < kMtDeg) w[j + kMtN] = mt_twist(w[j], w[j + 1], w[j + kMtM]);
      __syncthreads();
    }
    uint32_t acc = 0;
    if (t < (unsigned)kMtN) {
      for (int i0 = 0; i0 < kMtJumpWords; ++i0) {
        uint32_t g = gpoly[i0];   // (one address for the wave: an LDS broadcast)
        while (g) {
          const int bit = __ffs((int)g) - 1;
          g &= g - 1u;
          acc ^= w[32 * i0 + bit + (int)t];
        }
      }
    }
    __syncthreads();
    if (t < (unsigned)kMtN) w[t] = acc;
    __syncthreads();
  }
  // ---- the chunk's 2^18 words: twist 624 at a time, temper, filter, compact (as the serial kernel) ----
  uint32_t mask = kfold - 1u;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint8_t *dst = tmp + c * kMtChunk;
  unsigned done = 0;
  for (uint64_t base = 0; base < kMtChunk; base += kMtN) {
    uint32_t v = 0;
    if (t < 227) v = mt_twist(w[t], w[t + 1], w[t + kMtM]);
    __syncthreads();
    if (t < 227) w[t] = v;
    __syncthreads();
    const unsigned i2 = 227 + t;
    if (t < 227) v = mt_twist(w[i2], w[i2 + 1], w[i2 - 227]);
    __syncthreads();
    if (t < 227) w[i2] = v;
    __syncthreads();
    const unsigned k2 = 454 + t;
    if (k2 < (unsigned)kMtN) v = mt_twist(w[k2], w[k2 == kMtN - 1 ? 0 : k2 + 1], w[k2 - 227]);
    __syncthreads();
    if (k2 < (unsigned)kMtN) w[k2] = v;
    __syncthreads();
    const unsigned lim = (unsigned)((kMtChunk - base) < (uint64_t)kMtN ? (kMtChunk - base) : (uint64_t)kMtN);
    const uint32_t y = t < lim ? (mt_temper(w[t]) & mask) : 0xFFFFFFFFu;   // thread t owns word t
    const bool ok = t < lim && y < kfold;
    const unsigned long long bal = __ballot(ok);
    const unsigned wcnt = (unsigned)__popcll(bal);
    if (lane_id() == 0) wsum[t / kWave] = wcnt;
    __syncthreads();
    unsigned wb = 0, tot = 0;
    for (unsigned q = 0; q < kMtParBS / kWave; ++q) {
      if (q < t / kWave) wb += wsum[q];
      tot += wsum[q];
    }
    if (ok) dst[done + wb + (unsigned)__popcll(bal & ((1ull << lane_id()) - 1ull))] = (uint8_t)y;
    done += tot;
    __syncthreads();   // wsum / w are rewritten by the next round
  }
  if (t == 0) counts[c] = done;
}

// chunk c's accepted values go behind those of the chunks in front of it
__global__ __launch_bounds__(256) void mt19937_gather_kernel(const uint8_t *__restrict__ tmp,
                                                            const unsigned *__restrict__ counts,
                                                            uint64_t nchunks, uint64_t n,
                                                            uint8_t *__restrict__ out,
                                                            unsigned long long *__restrict__ total) {
  __shared__ unsigned long long s_pre[256 / kWave];
  const uint64_t c = blockIdx.x;
  unsigned long long pre = 0;
  for (uint64_t q = threadIdx.x; q < c; q += 256) pre += counts[q];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pre += __shfl_down(pre, off, 64);
  if (lane_id() == 0) s_pre[threadIdx.x / kWave] = pre;
  __syncthreads();
  unsigned long long base = 0;
  for (int q = 0; q < 256 / kWave; ++q) base += s_pre[q];
  const unsigned cnt = counts[c];
  const uint8_t *src = tmp + c * kMtChunk;
  for (unsigned j = threadIdx.x; j < cnt; j += 256)
    if (base + j < n) out[base + j] = src[j];
  if (c == nchunks - 1 && threadIdx.x == 0) *total = base + cnt;
}
}  // namespace
}  // namespace nvt

extern "C" int nvt_fold_mt19937(uint32_t seed, int kfold, uint64_t n, uint8_t *out, void *stream) {
  using namespace nvt;
  NVT_CHECK_ARG(kfold >= 1 && kfold <= 128, "kfold must be 1 .. 128 (folds are written as uint8)");
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(out != nullptr, "null output");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("fold_mt19937", n, s);
  mt19937_folds_kernel<<<1, kMtBS, 0, s>>>(seed, (uint32_t)kfold, n, out);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

// chunks of 2^18 draws that yield n accepted values with a margin of 8 standard deviations
static uint64_t mt_par_chunks(uint64_t n, int kfold) {
  uint32_t mask = (uint32_t)kfold - 1u;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  const double a = (double)kfold / ((double)mask + 1.0);
  const double draws = (double)n / a, sd = sqrt(draws * (1.0 - a)) / a;
  return (uint64_t)((draws + 8.0 * sd) / (double)nvt::kMtChunk) + 2;
}

extern "C" int nvt_fold_mt19937_par_ws_bytes(uint64_t n, int kfold, uint64_t *bytes) {
  using namespace nvt;
  NVT_CHECK_ARG(bytes, "null out");
  NVT_CHECK_ARG(kfold >= 1 && kfold <= 128, "kfold must be 1 .. 128");
  const uint64_t g = mt_par_chunks(n, kfold);
  *bytes = g * kMtChunk + ((g * 4 + 255) & ~255ull) + 256;
  return NVT_OK;
}

extern "C" int nvt_fold_mt19937_par(uint32_t seed, int kfold, uint64_t n, uint8_t *out, void *ws,
                                    uint64_t ws_bytes, uint64_t *total_out, void *stream) {
  using namespace nvt;
  NVT_CHECK_ARG(kfold >= 1 && kfold <= 128, "kfold must be 1 .. 128 (folds are written as uint8)");
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(out && ws && total_out, "null pointer");
  const uint64_t g = mt_par_chunks(n, kfold);
  NVT_CHECK_ARG(g < (1ull << kMtJumpPolys), "more chunks than the jump polynomials reach");
  NVT_CHECK_ARG(ws_bytes >= g * kMtChunk + ((g * 4 + 255) & ~255ull) + 256, "workspace smaller than nvt_fold_mt19937_par_ws_bytes");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("fold_mt19937", n, s);
  uint8_t *tmp = reinterpret_cast<uint8_t *>(ws);
  unsigned *counts = reinterpret_cast<unsigned *>(tmp + g * kMtChunk);
  mt19937_chunks_kernel<<<(unsigned)g, kMtParBS, 0, s>>>(seed, (uint32_t)kfold, g, tmp, counts);
  NVT_CHECK_LAUNCH();
  mt19937_gather_kernel<<<(unsigned)g, 256, 0, s>>>(tmp, counts, g, n, out,
                                                   reinterpret_cast<unsigned long long *>(total_out));
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

// ---------------------------------------------------------------------------------------------
// Distinct keys of a column PREFIX, estimated: what steers the first path choice of a fit without
// cardinality hints (kernels.py: _presample).  The exact count of the 256 K-row prefix of every
// column went through the hash-partitioned counting path: ~11 launches per column, ~1 ms of a
// fresh 26-column fit for 26 numbers that are tripled before anything is decided on them.  One
// workgroup per column keeps a HyperLogLog sketch in LDS instead (4096 registers: 1.6 % standard
// error; linear counting below 2.5 registers per key, so a handful of keys is counted exactly):
// ONE launch for all columns.  out[c] = {estimated distinct keys, valid rows} of the prefix.
namespace nvt {
namespace {
constexpr int kHllBits = 12, kHllRegs = 1 << kHllBits, kHllBS = 1024;
struct PrefixBatch {
  const void *keys[64];
  const uint8_t *valid[64];
  uint64_t n[64];
  int key_bytes[64];
};
__global__ __launch_bounds__(kHllBS) void prefix_distinct_kernel(PrefixBatch b, uint64_t *__restrict__ out) {
  __shared__ unsigned regs[kHllRegs];
  __shared__ double s_sum[kHllBS / kWave];
  __shared__ unsigned s_zero[kHllBS / kWave];
  __shared__ unsigned long long s_rows;
  const int c = blockIdx.x;
  for (int i = threadIdx.x; i < kHllRegs; i += kHllBS) regs[i] = 0;
  if (threadIdx.x == 0) s_rows = 0;
  __syncthreads();
  const uint64_t n = b.n[c];
  const uint8_t *valid = b.valid[c];
  unsigned long long rows = 0;
  for (uint64_t i = threadIdx.x; i < n; i += kHllBS) {
    if (!bit_valid(valid, i)) continue;
    uint32_t h;
    if (b.key_bytes[c] == 4)
      h = fmix32((uint32_t)reinterpret_cast<const int32_t *>(b.keys[c])[i]);
    else
      h = (uint32_t)(fmix64((uint64_t)reinterpret_cast<const int64_t *>(b.keys[c])[i]) >> 17);
    ++rows;
    const unsigned w = (h << kHllBits) | (1u << (kHllBits - 1));  // (never 0: rho <= 32 - bits + 1)
    atomicMax(&regs[h >> (32 - kHllBits)], (unsigned)__clz((int)w) + 1u);
  }
  if (rows) atomicAdd(&s_rows, rows);
  __syncthreads();
  double sum = 0;
  unsigned zeros = 0;
  for (int i = threadIdx.x; i < kHllRegs; i += kHllBS) {
    const unsigned r = regs[i];
    sum += 1.0 / (double)(1ull << r);
    zeros += r == 0 ? 1u : 0u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, 64);
    zeros += __shfl_down(zeros, off, 64);
  }
  if (lane_id() == 0) {
    s_sum[threadIdx.x / kWave] = sum;
    s_zero[threadIdx.x / kWave] = zeros;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0;
    unsigned z = 0;
    for (int q = 0; q < kHllBS / kWave; ++q) {
      tot += s_sum[q];
      z += s_zero[q];
    }
    const double m = (double)kHllRegs;
    double est = (0.7213 / (1.0 + 1.079 / m)) * m * m / tot;
    if (est <= 2.5 * m && z > 0) est = m * log(m / (double)z);  // linear counting
    out[2 * c] = s_rows == 0 ? 0ull : (uint64_t)(est + 0.5);
    out[2 * c + 1] = s_rows;
  }
}
}  // namespace
}  // namespace nvt

extern "C" int nvt_prefix_distinct(const nvt_prefix_col *cols, int ncols, uint64_t *out, void *stream) {
  using namespace nvt;
  NVT_CHECK_ARG(ncols >= 0 && (ncols == 0 || (cols && out)), "null pointer");
  hipStream_t s = (hipStream_t)stream;
  for (int c0 = 0; c0 < ncols; c0 += 64) {
    const int nc = ncols - c0 < 64 ? ncols - c0 : 64;
    PrefixBatch b;
    memset(&b, 0, sizeof(b));
    for (int j = 0; j < nc; ++j) {
      const nvt_prefix_col &c = cols[c0 + j];
      NVT_CHECK_ARG(c.key_bytes == 4 || c.key_bytes == 8, "key_bytes must be 4 or 8");
      NVT_CHECK_ARG(c.n == 0 || c.keys, "null keys");
      b.keys[j] = c.keys;
      b.valid[j] = c.valid;
      b.n[j] = c.n;
      b.key_bytes[j] = c.key_bytes;
    }
    NVT_PROF("prefix_distinct", 0, s);
    prefix_distinct_kernel<<<nc, kHllBS, 0, s>>>(b, out + 2 * (uint64_t)c0);
    NVT_CHECK_LAUNCH();
  }
  return NVT_OK;
}
