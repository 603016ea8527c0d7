// Categorify.transform / HashBucket on MI355X.
//
// Replaces the pandas `codes.merge(vocab, how="left").sort_values("order")`
// hash join + re-sort of categorify.py:1774-1776 (_encode) with one streaming
// pass: each lane loads 16 B of keys, probes a {key,label} open-addressing
// table and writes the labels with 16/32-byte stores.  Row order is preserved by
// construction, so the reference's O(n log n) re-sort disappears.
//
// Bytes per row (int32 keys, int64 labels): 4 read + 8 written = 12; the table
// probe is extra traffic that stays in L2 / Infinity Cache for all but the
// ~4e7-key columns.
#include <cstdlib>
#include <limits>
#include <type_traits>

#include "nvt_common.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"
#include "nvt_range.hpp"

namespace nvt {

template <typename K>
struct EncSlot;
template <>
struct EncSlot<int32_t> {
  int32_t key;
  int32_t label;
};
template <>
struct EncSlot<int64_t> {
  int64_t key;
  int64_t label;
};
template <typename K>
struct EncTraits;
template <>
struct EncTraits<int32_t> {
  static constexpr int32_t empty = INT32_MIN;
  static constexpr int vec = 4;
  using cas_t = int;
};
template <>
struct EncTraits<int64_t> {
  static constexpr int64_t empty = INT64_MIN;
  static constexpr int vec = 2;
  using cas_t = unsigned long long;
};

template <typename K>
__global__ __launch_bounds__(kBlock) void enc_clear_kernel(EncSlot<K> *table, uint64_t capacity,
                                                           int64_t *sentinel_label) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  EncSlot<K> e;
  e.key = EncTraits<K>::empty;
  e.label = std::numeric_limits<decltype(e.label)>::max();  // atomicMin target
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < capacity; i += stride)
    table[i] = e;
  if (blockIdx.x == 0 && threadIdx.x == 0) *sentinel_label = -1;
}

template <typename K>
__global__ __launch_bounds__(kBlock) void enc_build_kernel(const K *__restrict__ vocab, uint64_t n,
                                                           int64_t first_label, EncSlot<K> *table,
                                                           uint64_t mask, int64_t *sentinel_label) {
  constexpr K EMPTY = EncTraits<K>::empty;
  using C = typename EncTraits<K>::cas_t;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    K key = vocab[i];
    int64_t label = first_label + (int64_t)i;
    if (key == EMPTY) {
      *sentinel_label = label;
      continue;
    }
    uint64_t slot = (uint64_t)slot_hash(key) & mask;
    while (true) {
      K prev = (K)atomicCAS(reinterpret_cast<C *>(&table[slot].key), (C)EMPTY, (C)key);
      if (prev == EMPTY || prev == key) {
        // duplicate vocabulary keys (user-supplied vocabs): first (lowest) label wins,
        // so every writer goes through atomicMin against the cleared max value
        using L = decltype(table[slot].label);
        if constexpr (sizeof(L) == 4)
          atomicMin(reinterpret_cast<int *>(&table[slot].label), (int)label);
        else
          atomicMin(reinterpret_cast<long long *>(&table[slot].label), (long long)label);
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
}

// Vocabulary produced by fit: keys are unique, so an int32 slot {key,label} is claimed
// and filled by ONE 64-bit CAS (half the atomics of the general build above).
__global__ __launch_bounds__(kBlock) void enc_build_unique_i32_kernel(
    const int32_t *__restrict__ vocab, uint64_t n, int64_t first_label, EncSlot<int32_t> *table,
    uint64_t mask, int64_t *sentinel_label) {
  const unsigned long long EMPTY_SLOT =
      ((unsigned long long)(uint32_t)std::numeric_limits<int32_t>::max() << 32) |
      (unsigned long long)(uint32_t)INT32_MIN;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int32_t key = vocab[i];
    int64_t label = first_label + (int64_t)i;
    if (key == INT32_MIN) {
      *sentinel_label = label;
      continue;
    }
    unsigned long long want = ((unsigned long long)(uint32_t)label << 32) | (uint32_t)key;
    uint64_t slot = (uint64_t)slot_hash(key) & mask;
    while (atomicCAS(reinterpret_cast<unsigned long long *>(&table[slot]), EMPTY_SLOT, want) !=
           EMPTY_SLOT)
      slot = (slot + 1) & mask;
  }
}

template <typename K>
__device__ __forceinline__ int64_t probe(const EncSlot<K> *__restrict__ table, uint64_t mask,
                                         K key) {
  constexpr K EMPTY = EncTraits<K>::empty;
  uint64_t slot = (uint64_t)slot_hash(key) & mask;
  while (true) {
    EncSlot<K> s = table[slot];
    if (s.key == key) return (int64_t)s.label;
    if (s.key == EMPTY) return -1;
    slot = (slot + 1) & mask;
  }
}

template <typename K, typename OUT>
__global__ __launch_bounds__(kBlock) void encode_kernel(
    const K *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n,
    const EncSlot<K> *__restrict__ table, uint64_t mask, const int64_t *__restrict__ sentinel_label,
    int64_t null_label, int64_t oov_label, uint32_t num_buckets, OUT *__restrict__ out) {
  constexpr K EMPTY = EncTraits<K>::empty;
  constexpr int VEC = EncTraits<K>::vec;
  const int64_t sent = *sentinel_label;

  auto label_of = [&](K key, bool ok) -> OUT {
    if (!ok) return (OUT)null_label;
    int64_t lab = (key == EMPTY) ? sent : probe<K>(table, mask, key);
    if (lab < 0) {
      lab = oov_label;
      if (num_buckets > 1) lab += (int64_t)(key_hash32((int64_t)key) % num_buckets);
    }
    return (OUT)lab;
  };

  const uint64_t nvec = n / VEC;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  using VecT = typename std::conditional<sizeof(K) == 4, int4, longlong2>::type;
  const VecT *vkeys = reinterpret_cast<const VecT *>(keys);
  for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
    VecT pack = vkeys[v];
    unsigned vbits = 0xF;
    if (valid != nullptr) {
      uint64_t row = v * VEC;
      vbits = (valid[row >> 3] >> (row & 7));
    }
    K k[VEC];
    if constexpr (sizeof(K) == 4) {
      k[0] = pack.x;
      k[1] = pack.y;
      k[2] = pack.z;
      k[3] = pack.w;
    } else {
      k[0] = pack.x;
      k[1] = pack.y;
    }
    OUT r[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) r[j] = label_of(k[j], (vbits >> j) & 1);
    // VEC * sizeof(OUT) is 8, 16 or 32 bytes: store as 1-2 wide vectors
    OUT *dst = out + v * VEC;
    if constexpr (VEC * sizeof(OUT) == 32) {
      int4 a, b;
      memcpy(&a, &r[0], 16);
      memcpy(&b, &r[2], 16);
      reinterpret_cast<int4 *>(dst)[0] = a;
      reinterpret_cast<int4 *>(dst)[1] = b;
    } else if constexpr (VEC * sizeof(OUT) == 16) {
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
    out[i] = label_of(keys[i], bit_valid(valid, i));
}

// Encode with the head of the vocabulary staged in LDS.  The vocabulary is ordered by
// frequency (categorify.py:1316), so its first entries are exactly the keys most rows
// carry: each workgroup copies the first n_hot entries into a private LDS table and only
// rows that miss it go to the global table in HBM.  A vocabulary that fits entirely
// (n_vocab <= n_hot) never touches the global table: pure stream + LDS gathers.
// The key and label streams of encode_hot_kernel are touched once; marking them non-temporal
// keeps the probe table (up to ~130 MB for a 6 M-key vocabulary) resident in L2 / Infinity Cache.
typedef int nvt_v4i __attribute__((ext_vector_type(4)));
template <typename V>
__device__ __forceinline__ V nt_load16(const V *p) {
  static_assert(sizeof(V) == 16, "16-byte vectors only");
#ifdef NVT_NO_NT
  return *p;
#else
  nvt_v4i r = __builtin_nontemporal_load(reinterpret_cast<const nvt_v4i *>(p));
  V v;
  memcpy(&v, &r, 16);
  return v;
#endif
}
template <typename V>
__device__ __forceinline__ void nt_store16(V v, V *p) {
  static_assert(sizeof(V) == 16, "16-byte vectors only");
#ifdef NVT_NO_NT
  *p = v;
#else
  nvt_v4i r;
  memcpy(&r, &v, 16);
  __builtin_nontemporal_store(r, reinterpret_cast<nvt_v4i *>(p));
#endif
}
#define NVT_NT_LOAD(p) nt_load16(p)
#define NVT_NT_STORE(v, p) nt_store16((v), (p))
#ifndef NVT_HOT_DIV
#define NVT_HOT_DIV 4
#endif
constexpr int kEncBS = 1024;
template <typename K>
struct HotCfg;
template <>
struct HotCfg<int32_t> {
  static constexpr int slots = 16384;  // 128 KiB of {key,label} int32 pairs
};
template <>
struct HotCfg<int64_t> {
  static constexpr int slots = 8192;  // 128 KiB of {key,label} int64 pairs
};

// TWO (int32 keys, cache mode only): the staged head of the vocabulary lives in a 2-choice,
// 2-slots-per-bucket table instead of a linear-probing one.  A lookup is exactly two
// independent 16-byte LDS reads whatever the load (a MISS in the linear table walks ~2.5 slots
// at 25 % load and ~8.5 at 75 %, which is why that table is only filled to a quarter), so the
// table can be filled to 7/8: 14336 hot keys instead of 4096, 15-20 % fewer rows go on to the
// table in HBM.  A key that finds both buckets full is simply not cached (the global table
// holds every key).
template <int NB, typename K>
__device__ __forceinline__ void two_buckets(K key, uint32_t &b1, uint32_t &b2) {
#ifndef NVT_ENC_NO_MUL24
  if constexpr (sizeof(K) == 4) {
    // both bucket indices from 24-bit multiplies of the same two words (nvt_common.hpp:
    // mul24_hash; fmix32 + a third 32-bit multiply were ~28 issue slots per key, this is ~10)
    static_assert(NB <= 8192, "13 index bits");
    const uint32_t k = (uint32_t)key, lo = k ^ (k >> 7), hi = k >> 8;
    b1 = ((__umul24(hi, 0x5BD1E9u) + __umul24(lo, 0x9E3779u)) >> 19) & (NB - 1);
    b2 = ((__umul24(hi, 0x7FEB35u) + __umul24(lo, 0x846CA7u)) >> 19) & (NB - 1);
    b2 = b2 == b1 ? b1 ^ 1u : b2;
    return;
  }
#endif
  const uint32_t h = (uint32_t)slot_hash(key);
  b1 = (h >> 13) & (NB - 1);
  b2 = (((h ^ (h >> 15)) * 0x2C1B3C6Du) >> 17) & (NB - 1);
  b2 = b2 == b1 ? b1 ^ 1u : b2;
}
// HEAD16 (int32 keys, cache mode): the head's labels are frequency ranks -- first_label + the
// position of the key among the n_hot most frequent ones -- so a slot needs the key and a 16-bit
// rank, 6 bytes instead of 8: {key0, key1} int2 per bucket + one word of two 16-bit ranks, and
// 12288 buckets (144 KiB of the CU's 160) instead of 8192 (128 KiB): 21504 hot keys at 7/8 load
// instead of 14336.  Fewer rows miss the head, and a miss is what the cache mode pays for (one
// random 64-byte sector each).  Bucket index: 15 hash bits h, (3 h) >> 3.
constexpr int kHead16Buckets = 12288;
constexpr int kHead16Keys = kHead16Buckets * 2 / 8 * 7;   // 21504 < 65536
__device__ __forceinline__ void two_buckets16(int32_t key, uint32_t &b1, uint32_t &b2) {
  const uint32_t k = (uint32_t)key, lo = k ^ (k >> 7), hi = k >> 8;
  const uint32_t h1 = ((__umul24(hi, 0x5BD1E9u) + __umul24(lo, 0x9E3779u)) >> 17) & 0x7FFFu;
  const uint32_t h2 = ((__umul24(hi, 0x7FEB35u) + __umul24(lo, 0x846CA7u)) >> 17) & 0x7FFFu;
  b1 = (h1 + 2u * h1) >> 3;
  b2 = (h2 + 2u * h2) >> 3;
  b2 = b2 == b1 ? (b1 ^ 1u) : b2;   // (12288 is even: b ^ 1 stays in range)
}
// rows that went on to the table in HBM / rows looked up, when the host asks for them
// (NVT_ENC_STATS=1, nvt_encode_stats): a diagnostic, two atomics per wave at the kernel's end
__device__ unsigned long long g_enc_stats[2];

// first slot of the linear-probing LDS table
template <int SLOTS, typename K>
__device__ __forceinline__ uint32_t lds_home(K key) {
#ifndef NVT_ENC_NO_MUL24
  if constexpr (sizeof(K) == 4) {
    static_assert(SLOTS <= 16384, "14 index bits");
    return (mul24_hash((int32_t)key) >> 18) & (SLOTS - 1);
  }
#endif
  return (uint32_t)(slot_hash(key) >> 13) & (SLOTS - 1);
}

struct HeadJob {
  const int32_t *keys;
  unsigned char *image;
  int64_t first_label;
  uint32_t n_hot;
};
constexpr int kHeadJobsMax = 32;
struct HeadJobs {
  HeadJob j[kHeadJobsMax];
};

// The LDS head of a launch: the first n_hot keys of the frequency-ordered vocabulary.  Every
// workgroup of an encode launch used to build it for itself (clear 128-144 KiB, read the keys,
// CAS them in: ~22 us in front of each of the 13 cache-mode launches of a Criteo step, 0.25-0.3 ms
// per step measured by returning right behind it); nvt_vocab_finalize_many now builds it ONCE per
// vocabulary into a global image (enc_head_build_kernel, on the stream that orders the vocabulary:
// off the critical path) and the launch's workgroups copy the image with coalesced loads.
template <typename K, bool TWO, bool HEAD16, int SLOTS>
__device__ __forceinline__ void build_head(unsigned char *lraw, long long *s_sent_p,
                                           const K *__restrict__ hot_keys, uint32_t n_hot,
                                           int64_t first_label) {
  constexpr K EMPTY = EncTraits<K>::empty;
  using L = decltype(EncSlot<K>::label);
  using C = typename EncTraits<K>::cas_t;
  EncSlot<K> *lt = reinterpret_cast<EncSlot<K> *>(lraw);
  int2 *tkeys = reinterpret_cast<int2 *>(lraw);
  uint32_t *tlab = reinterpret_cast<uint32_t *>(lraw + (HEAD16 ? kHead16Buckets * 8 : 0));
  long long &s_sent = *s_sent_p;
  if constexpr (HEAD16) {
    for (int i = threadIdx.x; i < kHead16Buckets; i += kEncBS) {
      tkeys[i] = make_int2((int)EMPTY, (int)EMPTY);
      tlab[i] = 0u;
    }
  } else {
    for (int i = threadIdx.x; i < SLOTS; i += kEncBS) {
      lt[i].key = EMPTY;
      lt[i].label = 0;
    }
  }
  if (threadIdx.x == 0) s_sent = -1;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_hot; i += kEncBS) {
    K key = hot_keys[i];
    if (key == EMPTY) {
      s_sent = first_label + (long long)i;
      continue;
    }
    if constexpr (HEAD16) {
      uint32_t b1, b2;
      two_buckets16((int32_t)key, b1, b2);
      const uint32_t cand[4] = {2 * b1, 2 * b1 + 1, 2 * b2, 2 * b2 + 1};   // key words of the int2 array
      int *kw = reinterpret_cast<int *>(tkeys);
      unsigned short *lw = reinterpret_cast<unsigned short *>(tlab);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int prev = atomicCAS(&kw[cand[c]], (int)EMPTY, (int)key);
        if (prev == (int)EMPTY) {
          lw[cand[c]] = (unsigned short)i;   // rank: i < n_hot <= kHead16Keys < 2^16
          break;
        }
      }
      continue;  // all four slots taken: not cached
    }
    if constexpr (TWO) {
      uint32_t b1, b2;
      two_buckets<SLOTS / 2>(key, b1, b2);
      const uint32_t cand[4] = {2 * b1, 2 * b1 + 1, 2 * b2, 2 * b2 + 1};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        K prev = (K)atomicCAS(reinterpret_cast<C *>(&lt[cand[c]].key), (C)EMPTY, (C)key);
        if (prev == EMPTY) {
          lt[cand[c]].label = (L)(first_label + (int64_t)i);
          break;
        }
      }
      continue;  // all four slots taken: not cached
    }
    uint32_t s = lds_home<SLOTS>(key);
    while (true) {
      K prev = (K)atomicCAS(reinterpret_cast<C *>(&lt[s].key), (C)EMPTY, (C)key);
      if (prev == EMPTY || prev == key) {
        if (prev == EMPTY) lt[s].label = (L)(first_label + (int64_t)i);
        break;
      }
      s = (s + 1) & (SLOTS - 1);
    }
  }
  __syncthreads();
}

template <bool HEAD16>
__global__ __launch_bounds__(kEncBS) void enc_head_build_many_kernel(HeadJobs jobs) {
  constexpr int SLOTS = HotCfg<int32_t>::slots;
  constexpr int kLdsBytes = HEAD16 ? kHead16Buckets * 12 : SLOTS * (int)sizeof(EncSlot<int32_t>);
  __shared__ __align__(16) unsigned char lraw[kLdsBytes];
  __shared__ long long s_sent;
  const HeadJob j = jobs.j[blockIdx.x];
  build_head<int32_t, true, HEAD16, SLOTS>(lraw, &s_sent, j.keys, j.n_hot, j.first_label);
  const int4 *src = reinterpret_cast<const int4 *>(lraw);
  int4 *dst = reinterpret_cast<int4 *>(j.image);
  for (int i = threadIdx.x; i < kLdsBytes / 16; i += kEncBS) dst[i] = src[i];
  if (threadIdx.x == 0) *reinterpret_cast<long long *>(j.image + kLdsBytes) = s_sent;
}

// Small vocabularies (<= SLOTS / 2 keys, int32 keys -> int64 labels: the reference's default dtype):
// the 4-in / 8-out stream of widen_stream (nvt_cont.hip) with an LDS lookup in the middle.  The
// fully staged encode_hot_kernel holds a 128 KiB table whatever the vocabulary, i.e. ONE 1024-thread
// workgroup per CU, and a lane that loads 16 bytes of keys owns 32 bytes of labels, written as two
// 16-byte stores 32 bytes apart: 131 us per 45 M-row column = 4.1 TB/s, of which experiment modes
// attribute 49 us to the stores, 48 us to the bare loop over the keys (3.75 TB/s with nothing but
// loads: 16 waves per CU) and 4 us to the lookups.  Here the table has SLOTS slots (16 / 32 KiB), a
// workgroup is 256 threads (8 / 5 of them per CU), a lane takes two keys per run of 128 and every
// store instruction of a wave covers 1024 contiguous bytes.
template <int SLOTS>
__global__ __launch_bounds__(kBlock) void encode_small_kernel(
    const int32_t *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n, int64_t null_label,
    int64_t oov_label, uint32_t num_buckets, int64_t *__restrict__ out,
    const int32_t *__restrict__ hot_keys, uint32_t n_hot, int64_t first_label) {
  constexpr int32_t EMPTY = EncTraits<int32_t>::empty;
  __shared__ EncSlot<int32_t> lt[SLOTS];
  __shared__ long long s_sent;
  for (int i = threadIdx.x; i < SLOTS; i += kBlock) {
    lt[i].key = EMPTY;
    lt[i].label = 0;
  }
  if (threadIdx.x == 0) s_sent = -1;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_hot; i += kBlock) {
    const int32_t key = hot_keys[i];
    if (key == EMPTY) {
      s_sent = first_label + (long long)i;
      continue;
    }
    uint32_t sl = lds_home<SLOTS>(key);
    while (true) {
      const int32_t prev = atomicCAS(&lt[sl].key, EMPTY, key);
      if (prev == EMPTY || prev == key) {
        if (prev == EMPTY) lt[sl].label = (int32_t)(first_label + (int64_t)i);
        break;
      }
      sl = (sl + 1) & (SLOTS - 1);
    }
  }
  __syncthreads();
  const int64_t sent = (int64_t)s_sent;
  auto encode = [&](int32_t key, bool ok) -> int64_t {
    if (!ok) return null_label;
    int64_t lab = -1;
    if (key == EMPTY) {
      lab = sent;
    } else {
      uint32_t sl = lds_home<SLOTS>(key);
      while (true) {
        const EncSlot<int32_t> e = lt[sl];
        if (e.key == key) {
          lab = (int64_t)e.label;
          break;
        }
        if (e.key == EMPTY) break;
        sl = (sl + 1) & (SLOTS - 1);
      }
    }
    if (lab < 0) {
      lab = oov_label;
      if (num_buckets > 1) lab += (int64_t)(key_hash32((int64_t)key) % num_buckets);
    }
    return lab;
  };
  constexpr int RUN = 2 * kWave, U = 4;
  typedef int v2i_nt __attribute__((ext_vector_type(2)));
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
        raw[u] = __builtin_nontemporal_load(reinterpret_cast<const v2i_nt *>(keys + e));
        if (valid != nullptr) vb[u] = (unsigned)valid[e >> 3] >> (e & 7);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r0 + u >= nruns) break;
      const uint64_t e = (r0 + u) * RUN + 2 * lane;
      int64_t r[2];
      r[0] = encode(raw[u].x, (bool)(vb[u] & 1));
      r[1] = encode(raw[u].y, (bool)((vb[u] >> 1) & 1));
      nvt_v4i o;
      memcpy(&o, r, 16);
      __builtin_nontemporal_store(o, reinterpret_cast<nvt_v4i *>(out + e));
    }
  }
  for (uint64_t i = nruns * RUN + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    out[i] = encode(keys[i], bit_valid(valid, i));
}

// GLOBAL = false: the whole vocabulary is staged (no table in HBM): the probe phases and their
// registers disappear.  (UU = 3 / 4 key vectors per lane in flight, and a 2048-slot table with
// two workgroups per CU for vocabularies <= 1024 keys, were each ~2 % slower: profiles/r02_notes.md)
// RANGE (int32 keys): `table` is a range table (the per-bucket tables of the counting pass,
// dumped; nvt_range.hpp): first slot from the monotone map, probing runs forward without
// wrapping (an empty slot ends every chain).
template <typename K, typename OUT, bool TWO = false, bool GLOBAL = true, int UU = 2,
          int RANGE = 0,  // 0: hashed table, 1: bucket regions dumped by the counting pass, 2: flat
          bool HEAD16 = false,
          bool HALF = false>  // half the head (64 KiB), <= 64 VGPRs: TWO workgroups per CU
__global__ __launch_bounds__(kEncBS, HALF ? 2 : 1) void encode_hot_kernel(
    const K *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n,
    const EncSlot<K> *__restrict__ table, uint64_t mask, const int64_t *__restrict__ sentinel_label,
    int64_t null_label, int64_t oov_label, uint32_t num_buckets, OUT *__restrict__ out,
    const K *__restrict__ hot_keys, uint32_t n_hot, int64_t first_label,
    const int32_t *__restrict__ range_aux = nullptr, int count_stats = 0,
    const unsigned char *__restrict__ head_image = nullptr) {
  static_assert(!HEAD16 || (TWO && sizeof(K) == 4), "HEAD16: the 2-choice head of int32 keys");
  static_assert(!HALF || (TWO && !HEAD16), "HALF: the 2-choice head with 8-byte slots");
  constexpr bool global_needed = GLOBAL;
  RangeMap rmap = {0u, 0u, 0u, 0, 0, 0u, nullptr, nullptr};
  __shared__ uint32_t s_pieces[RANGE == 1 ? kRpPwWords : 1];
  if constexpr (RANGE) rmap = load_map(range_aux);
  if constexpr (RANGE == 1) stage_pieces(rmap, s_pieces, threadIdx.x, kEncBS);  // (barrier below)
  auto first_slot = [&](K key) -> uint64_t {
    if constexpr (RANGE) return rmap.table_slot((int32_t)key);
    return (uint64_t)slot_hash(key) & mask;
  };
  auto next_slot = [&](uint64_t sl) -> uint64_t {
    if constexpr (RANGE) return sl + 1;
    return (sl + 1) & mask;
  };
  constexpr K EMPTY = EncTraits<K>::empty;
  constexpr int VEC = EncTraits<K>::vec;
  constexpr int SLOTS = HALF ? HotCfg<K>::slots / 2 : HotCfg<K>::slots;
  constexpr int kLdsBytes = HEAD16 ? kHead16Buckets * 12 : SLOTS * (int)sizeof(EncSlot<K>);
  __shared__ __align__(16) unsigned char lraw[kLdsBytes];
  EncSlot<K> *lt = reinterpret_cast<EncSlot<K> *>(lraw);
  int2 *tkeys = reinterpret_cast<int2 *>(lraw);                                     // HEAD16: {key0, key1}
  uint32_t *tlab = reinterpret_cast<uint32_t *>(lraw + (HEAD16 ? kHead16Buckets * 8 : 0));  // two 16-bit ranks
  __shared__ long long s_sent;  // label of the sentinel key when it is among the staged keys
  if (head_image != nullptr) {   // (uniform) prebuilt by nvt_vocab_finalize_many: coalesced copy
    const int4 *src = reinterpret_cast<const int4 *>(head_image);
    int4 *dst = reinterpret_cast<int4 *>(lraw);
    for (int i = threadIdx.x; i < kLdsBytes / 16; i += kEncBS) dst[i] = src[i];
    if (threadIdx.x == 0) s_sent = *reinterpret_cast<const long long *>(head_image + kLdsBytes);
    __syncthreads();
  } else {
    build_head<K, TWO, HEAD16, SLOTS>(lraw, &s_sent, hot_keys, n_hot, first_label);
  }
  // a vocabulary staged in full needs neither the global table nor its sentinel word
  const int64_t sent = global_needed ? *sentinel_label : (int64_t)s_sent;

  // LDS lookup: label, or -1 when the key is not in the staged head of the vocabulary
  auto hot_lookup = [&](K key) -> int64_t {
    if constexpr (HEAD16) {
      uint32_t b1, b2;
      two_buckets16((int32_t)key, b1, b2);
      const int2 a = tkeys[b1], c = tkeys[b2];
      const uint32_t la = tlab[b1], lc = tlab[b2];
      int lab = -1;
      lab = a.x == (int)key ? (int)(la & 0xFFFFu) : lab;
      lab = a.y == (int)key ? (int)(la >> 16) : lab;
      lab = c.x == (int)key ? (int)(lc & 0xFFFFu) : lab;
      lab = c.y == (int)key ? (int)(lc >> 16) : lab;
      return lab < 0 ? (int64_t)-1 : first_label + (int64_t)lab;
    }
    if constexpr (TWO) {
      uint32_t b1, b2;
      two_buckets<SLOTS / 2>(key, b1, b2);
      const int4 a = reinterpret_cast<const int4 *>(lt)[b1];
      const int4 c = reinterpret_cast<const int4 *>(lt)[b2];
      int lab = -1;
      lab = a.x == (int)key ? a.y : lab;
      lab = a.z == (int)key ? a.w : lab;
      lab = c.x == (int)key ? c.y : lab;
      lab = c.z == (int)key ? c.w : lab;
      return (int64_t)lab;
    }
    uint32_t s = lds_home<SLOTS>(key);
    while (true) {
      EncSlot<K> e = lt[s];
      if (e.key == key) return (int64_t)e.label;
      if (e.key == EMPTY) return -1;
      s = (s + 1) & (SLOTS - 1);
    }
  };
  auto finish = [&](K key, int64_t lab) -> OUT {
    if (lab < 0) {
      lab = oov_label;
      if (num_buckets > 1) lab += (int64_t)(key_hash32((int64_t)key) % num_buckets);
    }
    return (OUT)lab;
  };

  const uint64_t nvec = n / VEC;
  const uint64_t stride = (uint64_t)gridDim.x * kEncBS;
  using VecT = typename std::conditional<sizeof(K) == 4, int4, longlong2>::type;
  const VecT *vkeys = reinterpret_cast<const VecT *>(keys);
  constexpr int U = UU;
  constexpr int NK = U * VEC;
  // software pipeline: the key vectors (and bitmap bytes) of iteration i+1 are requested
  // before iteration i is processed -- with one 1024-thread workgroup per CU the stream
  // latency is otherwise exposed once per iteration (~20 iterations per column)
  VecT nxt_pack[U];
  unsigned nxt_vb[U];
  unsigned st_miss = 0, st_rows = 0;   // (count_stats)
  auto issue_loads = [&](uint64_t v0, VecT (&pk)[U], unsigned (&bits)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t v = v0 + (uint64_t)u * stride;
      bits[u] = 0;
      if (v < nvec) {
        pk[u] = NVT_NT_LOAD(&vkeys[v]);
        bits[u] = 0x10000u | (valid ? (unsigned)valid[(v * VEC) >> 3] : 0xFFu);  // raw byte
      }
    }
  };
  issue_loads((uint64_t)blockIdx.x * kEncBS + threadIdx.x, nxt_pack, nxt_vb);
  for (uint64_t v0 = (uint64_t)blockIdx.x * kEncBS + threadIdx.x; v0 < nvec; v0 += stride * U) {
    VecT pack[U];
    unsigned vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      pack[u] = nxt_pack[u];
      vb[u] = nxt_vb[u];
    }
    issue_loads(v0 + stride * U, nxt_pack, nxt_vb);
#pragma unroll
    for (int u = 0; u < U; ++u)  // shift after ALL loads are in flight (no early s_waitcnt)
      if (vb[u])
        vb[u] = 0x100u | (((vb[u] & 0xFFu) >> (((v0 + (uint64_t)u * stride) * VEC) & 7)) &
                          ((1u << VEC) - 1u));
    K k[NK];
    int64_t lab[NK];
    bool need[NK];  // still unresolved: must probe the table in HBM
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (sizeof(K) == 4) {
        k[u * VEC + 0] = pack[u].x;
        k[u * VEC + 1] = pack[u].y;
        k[u * VEC + 2] = pack[u].z;
        k[u * VEC + 3] = pack[u].w;
      } else {
        k[u * VEC + 0] = pack[u].x;
        k[u * VEC + 1] = pack[u].y;
      }
    }
    // phase 1: LDS
#pragma unroll
    for (int q = 0; q < NK; ++q) {
      const bool ok = (vb[q / VEC] & 0x100) && ((vb[q / VEC] >> (q % VEC)) & 1);
      lab[q] = -1;
      need[q] = false;
      if (ok) {
        if (k[q] == EMPTY) {
          lab[q] = sent;
        } else {
          lab[q] = hot_lookup(k[q]);
          need[q] = lab[q] < 0 && global_needed;
          if (count_stats) {
            st_rows += 1u;
            st_miss += need[q] ? 1u : 0u;
          }
        }
      } else {
        lab[q] = null_label;
      }
    }
    // phase 2: first probe of every miss issued back to back (independent loads), so the
    // HBM / Infinity-Cache latency is paid once per batch, not once per key
    if constexpr (GLOBAL) {
      uint64_t slot[NK];
      EncSlot<K> e[NK];
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        slot[q] = first_slot(k[q]);
        if (need[q]) e[q] = table[slot[q]];
      }
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        if (!need[q]) continue;
        if constexpr (RANGE == 2) {
          // flat table: a few slots one by one (keys that are spread over their range sit next
          // to home); a key that is still unresolved then -- its vocabulary clusters in its
          // range -- is searched by doubling + bisection behind this loop (runs are in key order)
          int step = 0;
          while (true) {
            if (e[q].key == k[q]) {
              lab[q] = (int64_t)e[q].label;
              need[q] = false;
              break;
            }
            if (e[q].key == EMPTY) {
              need[q] = false;
              break;
            }
            if (++step == kFlatLinear) break;   // need[q] stays set: slot[q] is below the key
            slot[q] = slot[q] + 1;
            e[q] = table[slot[q]];
          }
          continue;
        }
        while (true) {  // collisions continue here (load factor <= 0.5: short chains)
          if (e[q].key == k[q]) {
            lab[q] = (int64_t)e[q].label;
            break;
          }
          if (e[q].key == EMPTY) break;
          slot[q] = next_slot(slot[q]);
          e[q] = table[slot[q]];
        }
      }
      if constexpr (RANGE == 2) {
        bool any = false;
#pragma unroll
        for (int q = 0; q < NK; ++q) any = any || need[q];
        if (any) {  // (never taken for keys that are spread over their range)
#pragma unroll   // (static indices: a rolled loop would put k[] / slot[] / lab[] into scratch)
          for (int q = 0; q < NK; ++q) {
            if (!need[q]) continue;
            if (ukey((int32_t)e[q].key) > ukey((int32_t)k[q])) continue;   // already past it: not there
            const uint64_t at = flat_gallop(reinterpret_cast<const unsigned long long *>(table),
                                            mask + 1, slot[q], (int32_t)k[q]);
            if (at != ~0ull) lab[q] = (int64_t)table[at].label;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!(vb[u] & 0x100)) continue;
      uint64_t v = v0 + (uint64_t)u * stride;
      OUT r[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int q = u * VEC + j;
        const bool ok = (vb[u] >> j) & 1;
        r[j] = ok ? finish(k[q], lab[q]) : (OUT)null_label;
      }
      OUT *dst = out + v * VEC;
      if constexpr (VEC * sizeof(OUT) == 32) {
        int4 a, b;
        memcpy(&a, &r[0], 16);
        memcpy(&b, &r[2], 16);
        NVT_NT_STORE(a, &reinterpret_cast<int4 *>(dst)[0]);
        NVT_NT_STORE(b, &reinterpret_cast<int4 *>(dst)[1]);
      } else if constexpr (VEC * sizeof(OUT) == 16) {
        int4 a;
        memcpy(&a, &r[0], 16);
        NVT_NT_STORE(a, &reinterpret_cast<int4 *>(dst)[0]);
      } else {
        int2 a;
        memcpy(&a, &r[0], 8);
        reinterpret_cast<int2 *>(dst)[0] = a;
      }
    }
  }
  for (uint64_t i = nvec * VEC + (uint64_t)blockIdx.x * kEncBS + threadIdx.x; i < n; i += stride) {
    K key = keys[i];
    OUT r = (OUT)null_label;
    if (bit_valid(valid, i)) {
      int64_t lab = key == EMPTY ? sent : hot_lookup(key);
      if (lab < 0 && key != EMPTY && global_needed) {
        uint64_t sl = first_slot(key);
        while (true) {
          const EncSlot<K> e1 = table[sl];
          if (e1.key == key) {
            lab = (int64_t)e1.label;
            break;
          }
          if (e1.key == EMPTY) break;
          sl = next_slot(sl);
        }
      }
      r = finish(key, lab);
    }
    out[i] = r;
  }
  if (count_stats) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      st_miss += __shfl_down(st_miss, off, 64);
      st_rows += __shfl_down(st_rows, off, 64);
    }
    if (lane_id() == 0) {
      atomicAdd(&g_enc_stats[0], (unsigned long long)st_miss);
      atomicAdd(&g_enc_stats[1], (unsigned long long)st_rows);
    }
  }
}

// Cache mode as a software pipeline (int32 keys; round 6).  What the ISA of encode_hot_kernel
// showed: (1) its "prefetch" of the next iteration's keys never overlapped anything -- the loads sit
// under `if (v < nvec)` / `if (valid)` branches, the validity byte is used right behind its load,
// and the compiler answers every such block with `s_waitcnt vmcnt(0)`; (2) a wave walks its phases
// one after the other per batch of 8 keys per lane: key loads, LDS head lookups, first probes of the
// misses in the table in HBM (one random 128-byte line per missing row), wait, up to 8 probe
// chains one after the other, label stores.  One 1024-thread workgroup per CU (the head takes
// 144 KiB of LDS) is 4 waves per SIMD, so a launch cost about the SUM of its phases (issue ~80 us +
// stream ~96 us + probes ~128 us against 236-258 us measured, profiles/r05_notes.md).
// Here the loop over the FULL steps of a workgroup (every lane has both of its vectors) is
// straight-line code -- unconditional loads, branch-free LDS phase, unconditional probes (a lane whose
// key hit the head reads slot 0: one broadcast line) -- so the compiler's s_waitcnt counts are exact
// and loads stay in flight across phases:
//     step i:  [b] keys(i) (requested in step i-1) through the LDS head
//              [c] probes(i-1) (requested in step i-1) resolved; rare chains walked here, while
//                  nothing else of this wave is in flight
//              [d] keys(i+1) and probes(i) requested
//              [e] labels(i-1) stored
// i.e. the probes of a batch fly during the stores of the batch in front of it and the LDS phase of
// the batch behind it, the key loads during a whole step.  Two stage records (ping-pong, static
// indices: registers).  The few vectors behind the last full step and the rows behind the last
// vector take a plain per-key path.  Results are identical to encode_hot_kernel's: every row is
// stored once, at its own index.
// a wave-uniform GLOBAL pointer the optimiser cannot re-associate with per-thread offsets:
// `base + tid` stays "scalar base + 32-bit vector offset" (the global_load saddr form) instead of a
// hoisted 64-bit per-thread address pair that lives -- or is spilled -- across a whole loop.  (The
// address space is part of the type: a pointer rebuilt from integers is a FLAT pointer otherwise,
// and flat accesses count in vmcnt AND lgkmcnt: every wait becomes vmcnt(0).)
#define NVT_GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ NVT_GLOBAL_AS T *uniform_gptr(T *p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (NVT_GLOBAL_AS T *)(((uint64_t)hi << 32) | lo);
}
template <typename OUT, int RANGE>
__global__ __launch_bounds__(kEncBS, 1) void encode_pipe_kernel(
    const int32_t *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n,
    const EncSlot<int32_t> *__restrict__ table, uint64_t mask,
    const int64_t *__restrict__ sentinel_label, int64_t null_label, int64_t oov_label,
    uint32_t num_buckets, OUT *__restrict__ out, const int32_t *__restrict__ hot_keys,
    uint32_t n_hot, int64_t first_label, const int32_t *__restrict__ range_aux, int count_stats,
    const unsigned char *__restrict__ head_image) {
  static_assert(RANGE == 1 || RANGE == 2, "range tables (slots and byte offsets fit 32 bits)");
  using K = int32_t;
  constexpr K EMPTY = EncTraits<K>::empty;
  constexpr int VEC = 4, U = 2, NK = U * VEC;
  constexpr int SLOTS = HotCfg<K>::slots;
  constexpr int kLdsBytes = kHead16Buckets * 12;
  RangeMap rmap = load_map(range_aux);
  __shared__ uint32_t s_pieces[RANGE == 1 ? kRpPwWords : 1];
  if constexpr (RANGE == 1) stage_pieces(rmap, s_pieces, threadIdx.x, kEncBS);  // (barrier below)
  __shared__ __align__(16) unsigned char lraw[kLdsBytes];
  const int2 *tkeys = reinterpret_cast<const int2 *>(lraw);
  const uint32_t *tlab = reinterpret_cast<const uint32_t *>(lraw + kHead16Buckets * 8);
  __shared__ long long s_sent;
  __shared__ unsigned s_next;   // batches handed out to the waves
  if (threadIdx.x == 0) s_next = 0;
  if (head_image != nullptr) {
    const int4 *src = reinterpret_cast<const int4 *>(head_image);
    int4 *dst = reinterpret_cast<int4 *>(lraw);
    for (int i = threadIdx.x; i < kLdsBytes / 16; i += kEncBS) dst[i] = src[i];
    __syncthreads();
  } else {
    build_head<K, true, true, SLOTS>(lraw, &s_sent, hot_keys, n_hot, first_label);
  }
  const int32_t sent = (int32_t)*sentinel_label;   // (-1: the sentinel key is not in the vocabulary)
  const unsigned long long *tw = reinterpret_cast<const unsigned long long *>(table);
  const char *tbytes = reinterpret_cast<const char *>(table);
#ifdef NVT_ENC_PIPE_TIMING
  const long long t_begin = clock64();
#endif

  unsigned st_miss = 0, st_rows = 0;  // (count_stats)
  // (the linear / piecewise form of the range map is decided once, outside the loops)
  auto run = [&](auto pw_tag) {
    constexpr bool PW = decltype(pw_tag)::value;
    // first slot of a key's probe chain (< 2^29: bucket regions / flat tables of int32 vocabularies)
    auto first_slot = [&](K key) -> uint32_t {
      const uint32_t f = rmap.template fine_staged<PW>(key);
      return RANGE == 2 ? f : f + (f >> 14) * (uint32_t)kRpTail;   // (f >> 14) * kRpRegion + (f & 16383)
    };
    // label in the LDS head, or -1 (labels of the table are < INT32_MAX: build_launch checks)
    auto hot_lookup = [&](K key) -> int32_t {
      uint32_t b1, b2;
      two_buckets16(key, b1, b2);
      const int2 a = tkeys[b1], c = tkeys[b2];
      const uint32_t la = tlab[b1], lc = tlab[b2];
      int lab = -1;
      lab = a.x == key ? (int)(la & 0xFFFFu) : lab;
      lab = a.y == key ? (int)(la >> 16) : lab;
      lab = c.x == key ? (int)(lc & 0xFFFFu) : lab;
      lab = c.y == key ? (int)(lc >> 16) : lab;
      return lab < 0 ? -1 : (int32_t)first_label + lab;
    };
    // the rest of a probe chain whose first slot `e0` (at `sl`) held another key
    auto walk = [&](K key, uint32_t sl0, unsigned long long e0) -> int32_t {
      if constexpr (RANGE == 2) {
        unsigned long long w = 0;
        const uint64_t at = flat_find_from(tw, mask + 1, (uint64_t)sl0, key, e0, &w);
        return at == ~0ull ? -1 : (int32_t)(w >> 32);
      } else {
        uint64_t sl = sl0;
        while (true) {
          const unsigned long long e = tw[++sl];
          if ((int32_t)(uint32_t)e == key) return (int32_t)(e >> 32);
          if ((int32_t)(uint32_t)e == EMPTY) return -1;
        }
      }
    };
    auto finish = [&](K key, int32_t lab) -> OUT {
      int64_t r = (int64_t)lab;
      if (lab < 0) {
        r = oov_label;
        if (num_buckets > 1) r += (int64_t)(key_hash32((int64_t)key) % num_buckets);
      }
      return (OUT)r;
    };
    // one key outside the pipeline (the vectors behind the last full step, the rows behind the last vector)
    auto encode_one = [&](K key, bool ok) -> OUT {
      if (!ok) return (OUT)null_label;
      int32_t lab = key == EMPTY ? sent : hot_lookup(key);
      if (lab < 0 && key != EMPTY) {
        const uint32_t sl = first_slot(key);
        const unsigned long long e = tw[sl];
        if ((int32_t)(uint32_t)e == key) lab = (int32_t)(e >> 32);
        else if ((int32_t)(uint32_t)e != EMPTY) lab = walk(key, sl, e);
      }
      return finish(key, lab);
    };

    const uint64_t nvec = n / VEC;
    const uint64_t stride = (uint64_t)gridDim.x * kEncBS, SU = stride * U;
    const uint64_t vb0 = (uint64_t)blockIdx.x * kEncBS;
    const int4 *vkeys = reinterpret_cast<const int4 *>(keys);
    // validity bytes are ALWAYS loaded (from the keys when the column has no bitmap: any readable
    // bytes) and OR-ed with `vor`: no branch between the loads of a step
    const uint8_t *vsrc = valid != nullptr ? valid : reinterpret_cast<const uint8_t *>(keys);
    const unsigned vor = valid != nullptr ? 0u : 0xFFu;
    // steps in which every lane of this workgroup has all of its U vectors
    uint64_t nfull = 0;
    if (nvec >= vb0 + (uint64_t)(U - 1) * stride + kEncBS)
      nfull = (nvec - (vb0 + (uint64_t)(U - 1) * stride + kEncBS)) / SU + 1;
    // Addresses of the streams: a wave-uniform base per step and vector (vb0, stride and the step
    // index are scalar) + a constant offset per thread -- no 64-bit vector arithmetic, no address
    // registers that live across the phases.  The vector index of a thread is even with its
    // thread index (vb0 and stride are multiples of 1024): its validity nibble is the low or the
    // high half of byte (base >> 1) + (thread >> 1).
    const unsigned tid = threadIdx.x, lane = lane_id();
    const unsigned nib = (tid & 1u) * 4u;
    int4 nxt_pack[U];
    unsigned nxt_vb[U];
    struct Stage {
      int32_t k[NK];
      int32_t lab[NK];
      unsigned long long e[NK];   // first slot of every key's probe chain
      unsigned need;              // bit q: key q is still unresolved (its label is in the table in HBM)
      unsigned vb[U];             // bits 0-3: rows valid
    };
    // batch j of the workgroup = the 64 vectors (x U) of wave slot (j & 15) in its step (j >> 4)
    auto batch_base = [&](uint64_t j, int u) -> uint64_t {   // (scalar)
      return vb0 + (j >> 4) * SU + (uint64_t)u * stride + (j & 15u) * (uint64_t)kWave;
    };
    auto request_keys = [&](uint64_t j) {   // (scalar)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t vs = batch_base(j, u);
        const nvt_v4i raw = __builtin_nontemporal_load(
            reinterpret_cast<const NVT_GLOBAL_AS nvt_v4i *>(uniform_gptr(vkeys + vs)) + lane);
        nxt_pack[u] = make_int4(raw.x, raw.y, raw.z, raw.w);
        nxt_vb[u] = (unsigned)uniform_gptr(vsrc + (vs >> 1))[lane >> 1];   // (used one step later)
      }
    };
    auto lds_phase = [&](Stage &st) {   // [b]
      unsigned need = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        st.k[u * VEC + 0] = nxt_pack[u].x;
        st.k[u * VEC + 1] = nxt_pack[u].y;
        st.k[u * VEC + 2] = nxt_pack[u].z;
        st.k[u * VEC + 3] = nxt_pack[u].w;
        st.vb[u] = ((nxt_vb[u] | vor) >> nib) & 0xFu;
      }
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        const K key = st.k[q];
        const bool ok = (st.vb[q / VEC] >> (q % VEC)) & 1u;
        const bool is_sent = key == EMPTY;
        int32_t lab = hot_lookup(key);
        lab = is_sent ? sent : lab;
        st.lab[q] = lab;
        need |= ((ok & !is_sent & (lab < 0)) ? 1u : 0u) << q;
      }
      st.need = need;
      if (count_stats) {
        st_rows += (unsigned)__popc(st.vb[0]) + (unsigned)__popc(st.vb[1]);
        st_miss += (unsigned)__popc(need);
      }
    };
    auto request_probes = [&](Stage &st) {   // [d]
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        const uint32_t off = ((st.need >> q) & 1u) ? first_slot(st.k[q]) * 8u : 0u;
        st.e[q] = *reinterpret_cast<const unsigned long long *>(tbytes + off);
        // (one slot computation at a time: eight interleaved bisections of the piecewise map spill)
        if constexpr (PW) __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto resolve = [&](Stage &st) {   // [c]
      unsigned more = 0;
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        const bool nd = (st.need >> q) & 1u;
        const int32_t ek = (int32_t)(uint32_t)st.e[q];
        const bool hit = nd & (ek == st.k[q]);
        st.lab[q] = hit ? (int32_t)(st.e[q] >> 32) : st.lab[q];
        more |= ((nd & !hit & (ek != EMPTY)) ? 1u : 0u) << q;
      }
      // chains that go on behind their first slot (~1 in 5 of the probes): every lane walks ITS next
      // one (ffs of its mask; key / slot word picked by a compare chain: static register indices,
      // one copy of the walk in the code instead of eight)
      while (more) {
        const int q = (int)__ffs((int)more) - 1;
        more &= more - 1u;
        K key = st.k[0];
        unsigned long long e0 = st.e[0];
#pragma unroll
        for (int j = 1; j < NK; ++j) {
          key = q == j ? st.k[j] : key;
          e0 = q == j ? st.e[j] : e0;
        }
        const int32_t lab = walk(key, first_slot(key), e0);
#pragma unroll
        for (int j = 0; j < NK; ++j) st.lab[j] = q == j ? lab : st.lab[j];
      }
    };
    auto store = [&](const Stage &st, uint64_t j) {   // [e]
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t vs = batch_base(j, u);
        OUT r[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int q = u * VEC + j;
          r[j] = ((st.vb[u] >> j) & 1u) ? finish(st.k[q], st.lab[q]) : (OUT)null_label;
        }
        NVT_GLOBAL_AS nvt_v4i *dst =
            reinterpret_cast<NVT_GLOBAL_AS nvt_v4i *>(uniform_gptr(out + vs * VEC)) + lane * (VEC * sizeof(OUT) / 16);
        nvt_v4i a;
        memcpy(&a, &r[0], 16);
        __builtin_nontemporal_store(a, dst);
        if constexpr (sizeof(OUT) == 8) {
          nvt_v4i b;
          memcpy(&b, &r[2], 16);
          __builtin_nontemporal_store(b, dst + 1);
        }
      }
    };
    // Round 6: a wave draws its batches from a counter in LDS.  With a fixed share per wave the
    // oldest waves of a SIMD finished 10-43 % of the loop's duration before the youngest (C1 / C11 /
    // C2 / C23: 78 k of 798 k, 117 k of 427 k, 108 k of 258 k, 158 k of 367 k cycles between the first
    // wave's end and the last's), and the tail ran with too few waves to cover its probes.
    const uint64_t WB = nfull * (uint64_t)(kEncBS / kWave);
    auto grab = [&]() -> uint64_t {
      unsigned c = 0;
      if (lane == 0) c = atomicAdd(&s_next, 1u);
      return (uint64_t)__builtin_amdgcn_readfirstlane((int)c);
    };
    auto in_range = [&](uint64_t j) { return j < WB ? j : WB - 1; };   // (a batch to read again, unused)
    if (WB > 0) {
      Stage s0, s1;
#define NVT_PHASE() __builtin_amdgcn_sched_barrier(0)   // (no interleaving across phases: registers)
      // The loop is entered with exactly what a trip through it leaves in flight -- keys(next),
      // probes(current), stores(previous), in this order -- because the compiler's wait counts at the
      // loop header are the minimum over both ways in.
      uint64_t jb = grab();        // the batch whose probes are in flight
      bool last_is_s0 = true, any = jb < WB;
      if (any) {
        request_keys(jb);
        uint64_t ja = grab();      // the batch whose keys are in flight
        NVT_PHASE();
        lds_phase(s0);
        NVT_PHASE();
        request_keys(in_range(ja));
        NVT_PHASE();
        request_probes(s0);
        NVT_PHASE();
        if (ja < WB) {
          uint64_t jn = grab();
          lds_phase(s1);
          NVT_PHASE();
          resolve(s0);
          NVT_PHASE();
          request_keys(in_range(jn));
          NVT_PHASE();
          request_probes(s1);
          NVT_PHASE();
          store(s0, jb);
          NVT_PHASE();
          jb = ja;
          ja = jn;
          last_is_s0 = false;
          while (ja < WB) {
            jn = grab();
            lds_phase(s0);
            NVT_PHASE();
            resolve(s1);
            NVT_PHASE();
            request_keys(in_range(jn));
            NVT_PHASE();
            request_probes(s0);
            NVT_PHASE();
            store(s1, jb);
            NVT_PHASE();
            jb = ja;
            ja = jn;
            last_is_s0 = true;
            if (ja >= WB) break;
            jn = grab();
            lds_phase(s1);
            NVT_PHASE();
            resolve(s0);
            NVT_PHASE();
            request_keys(in_range(jn));
            NVT_PHASE();
            request_probes(s1);
            NVT_PHASE();
            store(s0, jb);
            NVT_PHASE();
            jb = ja;
            ja = jn;
            last_is_s0 = false;
          }
        }
#undef NVT_PHASE
        if (last_is_s0) {
          resolve(s0);
          store(s0, jb);
        } else {
          resolve(s1);
          store(s1, jb);
        }
      }
    }
    for (uint64_t v = vb0 + nfull * SU + tid; v < nvec; v += stride) {
      const int4 pk = vkeys[v];
      const unsigned bits = ((unsigned)vsrc[v >> 1] | vor) >> nib;
      const K kk[VEC] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
      for (int j = 0; j < VEC; ++j) out[v * VEC + j] = encode_one(kk[j], (bits >> j) & 1u);
      if (count_stats) st_rows += (unsigned)__popc(bits & 0xFu);
    }
    for (uint64_t i = nvec * VEC + vb0 + tid; i < n; i += stride)
      out[i] = encode_one(keys[i], bit_valid(valid, i));
  };
  if (RANGE == 1 && rmap.piece_slots != 0) run(std::true_type{});
  else run(std::false_type{});
#ifdef NVT_ENC_PIPE_TIMING
  // (experiment: g_enc_stats[0] += last wave's end - first wave's end, [1] += last wave's end - start)
  {
    __shared__ unsigned long long s_tmin, s_tmax;
    if (threadIdx.x == 0) {
      s_tmin = ~0ull;
      s_tmax = 0;
    }
    const unsigned long long t_end = (unsigned long long)clock64();
    __syncthreads();
    if (lane_id() == 0) {
      atomicMin(&s_tmin, t_end);
      atomicMax(&s_tmax, t_end);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&g_enc_stats[0], s_tmax - s_tmin);
      atomicAdd(&g_enc_stats[1], s_tmax - (unsigned long long)t_begin);
    }
    return;
  }
#endif
  if (count_stats) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      st_miss += __shfl_down(st_miss, off, 64);
      st_rows += __shfl_down(st_rows, off, 64);
    }
    if (lane_id() == 0) {
      atomicAdd(&g_enc_stats[0], (unsigned long long)st_miss);
      atomicAdd(&g_enc_stats[1], (unsigned long long)st_rows);
    }
  }
}

template <typename K>
__global__ __launch_bounds__(kBlock) void hash_bucket_kernel(const K *__restrict__ keys,
                                                             const uint8_t *__restrict__ valid,
                                                             uint64_t n, uint32_t nb,
                                                             int32_t *__restrict__ out,
                                                             const uint64_t *__restrict__ xor_in,
                                                             uint64_t *__restrict__ xor_out) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    // a null row hashes as key 0 whatever bytes its slot holds (Arrow leaves them undefined)
    uint64_t h = key_hash64(bit_valid(valid, i) ? (int64_t)keys[i] : 0);
    if (xor_in) h ^= xor_in[i];
    if (xor_out) xor_out[i] = h;
    if (out) out[i] = (int32_t)((uint32_t)(h >> 32) % nb);
  }
}

// fast path: no XOR chain, vectorised
template <typename K>
__global__ __launch_bounds__(kBlock) void hash_bucket_vec_kernel(const K *__restrict__ keys,
                                                                 const uint8_t *__restrict__ valid,
                                                                 uint64_t n, uint32_t nb,
                                                                 int32_t *__restrict__ out) {
  constexpr int VEC = EncTraits<K>::vec;
  const uint64_t nvec = n / VEC;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  using VecT = typename std::conditional<sizeof(K) == 4, int4, longlong2>::type;
  const VecT *vkeys = reinterpret_cast<const VecT *>(keys);
  for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
    VecT pack = vkeys[v];
    if (valid != nullptr) {  // null rows hash as key 0
      const uint64_t row = v * VEC;
      const unsigned vb = (unsigned)valid[row >> 3] >> (row & 7);
      if (!(vb & 1)) pack.x = 0;
      if (!(vb & 2)) pack.y = 0;
      if constexpr (sizeof(K) == 4) {
        if (!(vb & 4)) pack.z = 0;
        if (!(vb & 8)) pack.w = 0;
      }
    }
    if constexpr (sizeof(K) == 4) {
      int4 r;
      r.x = (int32_t)(key_hash32(pack.x) % nb);
      r.y = (int32_t)(key_hash32(pack.y) % nb);
      r.z = (int32_t)(key_hash32(pack.z) % nb);
      r.w = (int32_t)(key_hash32(pack.w) % nb);
      reinterpret_cast<int4 *>(out)[v] = r;
    } else {
      int2 r;
      r.x = (int32_t)(key_hash32(pack.x) % nb);
      r.y = (int32_t)(key_hash32(pack.y) % nb);
      reinterpret_cast<int2 *>(out)[v] = r;
    }
  }
  for (uint64_t i = nvec * VEC + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    out[i] = (int32_t)(key_hash32(bit_valid(valid, i) ? (int64_t)keys[i] : 0) % nb);
}

template <typename K>
int build_launch(const K *vocab, uint64_t n, int64_t first_label, void *table, uint64_t capacity,
                 int64_t *sentinel_label, int unique_keys, hipStream_t s) {
  NVT_CHECK_ARG(table && sentinel_label, "null table");
  NVT_CHECK_ARG(capacity >= 64 && (capacity & (capacity - 1)) == 0, "capacity must be 2^k >= 64");
  NVT_CHECK_ARG(capacity > n, "capacity must exceed the vocabulary size");
  NVT_CHECK_ARG(sizeof(K) == 8 || first_label + (int64_t)n < INT32_MAX, "labels overflow int32");
  auto *t = reinterpret_cast<EncSlot<K> *>(table);
  NVT_PROF("encode_build", 0, s);
  enc_clear_kernel<K><<<stream_grid(capacity, kBlock * 4), kBlock, 0, s>>>(t, capacity,
                                                                           sentinel_label);
  NVT_CHECK_LAUNCH();
  if (n) {
    NVT_CHECK_ARG(vocab, "null vocabulary");
    if constexpr (sizeof(K) == 4) {
      if (unique_keys) {
        enc_build_unique_i32_kernel<<<stream_grid(n, kBlock), kBlock, 0, s>>>(
            vocab, n, first_label, t, capacity - 1, sentinel_label);
        NVT_CHECK_LAUNCH();
        return NVT_OK;
      }
    }
    enc_build_kernel<K><<<stream_grid(n, kBlock), kBlock, 0, s>>>(vocab, n, first_label, t,
                                                                  capacity - 1, sentinel_label);
    NVT_CHECK_LAUNCH();
  }
  return NVT_OK;
}

// the head-layout switch of this process (the image nvt_vocab_finalize_many builds and the layout
// the cache-mode launch expects are decided by the same word)
static bool enc_head16() {
  static const bool v = ab_env("NVT_ENC_HEAD16") == nullptr || atoi(ab_env("NVT_ENC_HEAD16")) != 0;
  return v;
}

template <typename K>
int encode_launch(const K *keys, const uint8_t *valid, uint64_t n, const void *table,
                  uint64_t capacity, const int64_t *sentinel_label, int64_t null_label,
                  int64_t oov_label, uint32_t num_buckets, void *out, int out_bytes,
                  const K *hot_keys, uint64_t n_vocab, int64_t first_label, hipStream_t s,
                  const int32_t *range_aux = nullptr, const void *head_image = nullptr) {
  // a duplicate-free vocabulary that fits the LDS table is encoded without the global table
  constexpr uint64_t kResident = sizeof(K) == 4 ? NVT_ENCODE_RESIDENT_I32 : NVT_ENCODE_RESIDENT_I64;
  const bool resident = hot_keys != nullptr && n_vocab > 0 && n_vocab <= kResident;
  NVT_CHECK_ARG(resident || (table && sentinel_label), "null table");
  NVT_CHECK_ARG(resident || range_aux || (capacity >= 64 && (capacity & (capacity - 1)) == 0),
                "capacity must be 2^k >= 64");
  NVT_CHECK_ARG(range_aux == nullptr || (sizeof(K) == 4 && hot_keys != nullptr && n_vocab > 0),
                "range tables: int32 keys with the ordered vocabulary (vocab_keys)");
  NVT_CHECK_ARG(out_bytes == 4 || out_bytes == 8, "out_bytes must be 4 or 8");
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(keys && out, "null keys/out");
  NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(keys) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                "keys/out must be 16-byte aligned");
  constexpr int VEC = EncTraits<K>::vec;
  auto *t = reinterpret_cast<const EncSlot<K> *>(table);
  NVT_PROF(sizeof(K) == 4 ? "encode_i32" : "encode_i64", n * (sizeof(K) + (uint64_t)out_bytes), s);
  if (hot_keys != nullptr && n_vocab > 0) {
    // head of the frequency-ordered vocabulary in LDS (load factor <= 0.75: LDS probes are
    // cheap, every extra resident key is a saved trip to L2 / HBM)
    // a vocabulary that fits entirely may fill the table to 75 % (every lookup is a hit);
    // otherwise most lookups of the table MISS, and an unsuccessful linear probe costs
    // ~2.5 slots at 50 % load but ~8.5 at 75 % (measured: 530 -> 1290 us on a 6 M-key column)
    // fully staged (no global table) up to half load: linear probing of a table in which every
    // lookup hits stays short there (a 73 %-full table made an 11.9 k-key column 229 instead
    // of 129 us); larger vocabularies use the cache mode below
    const uint64_t full_cap = kResident;
    const uint64_t part_cap = (uint64_t)HotCfg<K>::slots / NVT_HOT_DIV;
    uint32_t n_hot = (uint32_t)(n_vocab <= full_cap ? n_vocab : part_cap);
    const int global_needed = n_vocab > n_hot;
    unsigned hgrid = stream_grid(n / VEC + 1, kEncBS * 2, 1);
    if constexpr (sizeof(K) == 4) {
      static const bool two = ab_env("NVT_ENC_LINEAR") == nullptr;
      // (range tables: `mask` = capacity - 1 bounds the search of a FLAT table -- capacity slots --
      // and is unused for the dumped bucket tables, capacity 0)
      if (global_needed && (two || range_aux != nullptr)) {  // cache mode: 2-choice table filled to 7/8
        const bool head16 = enc_head16();
        static const int stats = getenv("NVT_ENC_STATS") ? atoi(getenv("NVT_ENC_STATS")) : 0;
        // NVT_ENC_HALF=1 (experiment): half the head, one key vector per lane and <= 64 VGPRs, so
        // that TWO workgroups share a CU -- the waves of a cache-mode launch are parked 77 % of
        // their cycles (SQ_WAIT_ANY), and one workgroup per CU is 4 waves per SIMD
        static const bool half = ab_env("NVT_ENC_HALF") != nullptr && atoi(ab_env("NVT_ENC_HALF")) != 0;
        const uint64_t cap2 = half ? (uint64_t)HotCfg<K>::slots / 16 * 7
                                   : head16 ? (uint64_t)kHead16Keys : (uint64_t)HotCfg<K>::slots / 8 * 7;
        n_hot = (uint32_t)(n_vocab < cap2 ? n_vocab : cap2);
        if (half) hgrid = stream_grid(n / VEC + 1, kEncBS, 2);
        static const bool use_image = ab_env("NVT_ENC_NO_HEAD_IMAGE") == nullptr;   // (A/B switch)
        if (half || !use_image) head_image = nullptr;
#define NVT_ENC_CACHE(OUTT, KIND, H16)                                                            \
  encode_hot_kernel<K, OUTT, true, true, 2, KIND, H16><<<hgrid, kEncBS, 0, s>>>(                   \
      keys, valid, n, t, capacity - 1, sentinel_label, null_label, oov_label, num_buckets,          \
      reinterpret_cast<OUTT *>(out), hot_keys, n_hot, first_label, range_aux, stats,           \
      reinterpret_cast<const unsigned char *>(head_image))
#define NVT_ENC_CACHE_HALF(OUTT, KIND)                                                            \
  encode_hot_kernel<K, OUTT, true, true, 1, KIND, false, true><<<hgrid, kEncBS, 0, s>>>(           \
      keys, valid, n, t, capacity - 1, sentinel_label, null_label, oov_label, num_buckets,          \
      reinterpret_cast<OUTT *>(out), hot_keys, n_hot, first_label, range_aux, stats)
#define NVT_ENC_PIPE(OUTT, KIND)                                                                  \
  encode_pipe_kernel<OUTT, KIND><<<hgrid, kEncBS, 0, s>>>(                                         \
      keys, valid, n, t, capacity - 1, sentinel_label, null_label, oov_label, num_buckets,          \
      reinterpret_cast<OUTT *>(out), hot_keys, n_hot, first_label, range_aux, stats,           \
      reinterpret_cast<const unsigned char *>(head_image))
        // the software pipeline (encode_pipe_kernel) is the cache-mode kernel of the range tables with
        // the 6-byte head; NVT_ENC_PIPE=0 keeps the phase-serial one for A/B runs
        static const bool pipe = ab_env("NVT_ENC_PIPE") == nullptr || atoi(ab_env("NVT_ENC_PIPE")) != 0;
#define NVT_ENC_CACHE_K(KIND)                                                \
  do {                                                                       \
    if (half) {                                                              \
      if (out_bytes == 8) NVT_ENC_CACHE_HALF(int64_t, KIND);                 \
      else NVT_ENC_CACHE_HALF(int32_t, KIND);                                \
    } else if (head16) {                                                     \
      if (out_bytes == 8) NVT_ENC_CACHE(int64_t, KIND, true);                \
      else NVT_ENC_CACHE(int32_t, KIND, true);                               \
    } else {                                                                 \
      if (out_bytes == 8) NVT_ENC_CACHE(int64_t, KIND, false);               \
      else NVT_ENC_CACHE(int32_t, KIND, false);                              \
    }                                                                        \
  } while (0)
#define NVT_ENC_PIPE_K(KIND)                                                 \
  do {                                                                       \
    if (out_bytes == 8) NVT_ENC_PIPE(int64_t, KIND);                         \
    else NVT_ENC_PIPE(int32_t, KIND);                                        \
  } while (0)
        const bool piped = pipe && head16 && !half && range_aux != nullptr;
        // range tables -- capacity > 0: a FLAT range table of `capacity` slots (bounded search);
        // 0: the bucket regions dumped by the counting pass; no range_aux: the hashed table
        if (piped) {
          if (capacity > 0) NVT_ENC_PIPE_K(2); else NVT_ENC_PIPE_K(1);
        } else if (range_aux != nullptr) {
          if (capacity > 0) NVT_ENC_CACHE_K(2); else NVT_ENC_CACHE_K(1);
        } else {
          NVT_ENC_CACHE_K(0);
        }
#undef NVT_ENC_CACHE_HALF
#undef NVT_ENC_CACHE_K
#undef NVT_ENC_PIPE_K
#undef NVT_ENC_PIPE
#undef NVT_ENC_CACHE
        NVT_CHECK_LAUNCH();
        return NVT_OK;
      }
    }
#define NVT_ENC_HOT(OUTT, GL, UUU)                                                               \
  encode_hot_kernel<K, OUTT, false, GL, UUU><<<hgrid, kEncBS, 0, s>>>(                            \
      keys, valid, n, t, capacity - 1, sentinel_label, null_label, oov_label, num_buckets,        \
      reinterpret_cast<OUTT *>(out), hot_keys, n_hot, first_label)
    if constexpr (sizeof(K) == 4) {
      // small vocabularies with int64 labels: smaller tables, 256-thread workgroups, contiguous stores
      static const bool small_on = ab_env("NVT_ENC_NO_SMALL") == nullptr;   // (A/B switch)
      if (!global_needed && out_bytes == 8 && small_on && n_vocab <= 2048) {
        const unsigned sgrid = stream_grid(n / 2 + 1, kBlock * 8, 8);
        if (n_vocab <= 1024)
          encode_small_kernel<2048><<<sgrid, kBlock, 0, s>>>(
              reinterpret_cast<const int32_t *>(keys), valid, n, null_label, oov_label, num_buckets,
              reinterpret_cast<int64_t *>(out), reinterpret_cast<const int32_t *>(hot_keys), n_hot,
              first_label);
        else
          encode_small_kernel<4096><<<sgrid, kBlock, 0, s>>>(
              reinterpret_cast<const int32_t *>(keys), valid, n, null_label, oov_label, num_buckets,
              reinterpret_cast<int64_t *>(out), reinterpret_cast<const int32_t *>(hot_keys), n_hot,
              first_label);
        NVT_CHECK_LAUNCH();
        return NVT_OK;
      }
    }
    if (global_needed) {
      if (out_bytes == 8) NVT_ENC_HOT(int64_t, true, 2); else NVT_ENC_HOT(int32_t, true, 2);
    } else {
      if (out_bytes == 8) NVT_ENC_HOT(int64_t, false, 2); else NVT_ENC_HOT(int32_t, false, 2);
    }
#undef NVT_ENC_HOT
    NVT_CHECK_LAUNCH();
    return NVT_OK;
  }
  unsigned grid = stream_grid(n / VEC + 1, kBlock * 2, 8);
  if (out_bytes == 8)
    encode_kernel<K, int64_t><<<grid, kBlock, 0, s>>>(keys, valid, n, t, capacity - 1,
                                                      sentinel_label, null_label, oov_label,
                                                      num_buckets, reinterpret_cast<int64_t *>(out));
  else
    encode_kernel<K, int32_t><<<grid, kBlock, 0, s>>>(keys, valid, n, t, capacity - 1,
                                                      sentinel_label, null_label, oov_label,
                                                      num_buckets, reinterpret_cast<int32_t *>(out));
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

template <typename K>
int hash_bucket_launch(const K *keys, const uint8_t *valid, uint64_t n, uint32_t nb, int32_t *out,
                       const uint64_t *xor_in, uint64_t *xor_out, hipStream_t s) {
  NVT_CHECK_ARG(nb >= 1 && nb < (1u << 31), "num_buckets must be in [1, 2^31)");
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(keys && (out || xor_out), "null keys/out");
  bool aligned = (reinterpret_cast<uintptr_t>(keys) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  NVT_PROF("hash_bucket", n * (sizeof(K) + 4), s);
  if (!xor_in && !xor_out && aligned)
    hash_bucket_vec_kernel<K><<<stream_grid(n / EncTraits<K>::vec + 1, kBlock * 2), kBlock, 0, s>>>(
        keys, valid, n, nb, out);
  else
    hash_bucket_kernel<K><<<stream_grid(n, kBlock * 4), kBlock, 0, s>>>(keys, valid, n, nb, out,
                                                                       xor_in, xor_out);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int encode_build_any(int key_bytes, const void *vocab, uint64_t n, int64_t first_label, void *table,
                     uint64_t capacity, int64_t *sentinel_label, int unique_keys, hipStream_t s) {
  if (key_bytes == 4)
    return build_launch<int32_t>((const int32_t *)vocab, n, first_label, table, capacity,
                                 sentinel_label, unique_keys, s);
  return build_launch<int64_t>((const int64_t *)vocab, n, first_label, table, capacity,
                               sentinel_label, unique_keys, s);
}

// the two halves of a build, for callers that fill most of the table themselves (the one-pass
// vocabulary ordering, nvt_sort.hip): clear, then insert a duplicate-free int32 key range
int encode_clear_any(int key_bytes, void *table, uint64_t capacity, int64_t *sentinel_label,
                     hipStream_t s) {
  NVT_CHECK_ARG(table && sentinel_label, "null table");
  NVT_CHECK_ARG(capacity >= 64, "capacity must be >= 64");  // (any size: flat range tables)
  NVT_PROF("encode_build", 0, s);
  if (key_bytes == 4)
    enc_clear_kernel<int32_t><<<stream_grid(capacity, kBlock * 4), kBlock, 0, s>>>(
        reinterpret_cast<EncSlot<int32_t> *>(table), capacity, sentinel_label);
  else
    enc_clear_kernel<int64_t><<<stream_grid(capacity, kBlock * 4), kBlock, 0, s>>>(
        reinterpret_cast<EncSlot<int64_t> *>(table), capacity, sentinel_label);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

int encode_insert_any(int key_bytes, const void *vocab, uint64_t n, int64_t first_label,
                      void *table, uint64_t capacity, int64_t *sentinel_label, hipStream_t s) {
  NVT_CHECK_ARG(key_bytes == 4, "encode_insert_any: int32 keys");
  NVT_CHECK_ARG(table && sentinel_label && (n == 0 || vocab), "null pointer");
  NVT_CHECK_ARG(first_label + (int64_t)n < INT32_MAX, "labels overflow int32");
  if (n == 0) return NVT_OK;
  NVT_PROF("encode_build", 0, s);
  enc_build_unique_i32_kernel<<<stream_grid(n, kBlock), kBlock, 0, s>>>(
      (const int32_t *)vocab, n, first_label, reinterpret_cast<EncSlot<int32_t> *>(table),
      capacity - 1, sentinel_label);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // namespace nvt

using namespace nvt;

namespace nvt {
// head images of ORDERED int32 vocabularies (n > NVT_ENCODE_RESIDENT_I32: the launches that encode
// with them run in cache mode), NVT_ENCODE_HEAD_BYTES bytes each: ONE launch, a workgroup per
// vocabulary.  Entries with a null image / too few keys are skipped.
int encode_head_build_many(const int32_t *const *vocab_keys, const uint64_t *n, const int64_t *first_label,
                           void *const *images, int count, hipStream_t s) {
  static const bool half = ab_env("NVT_ENC_HALF") != nullptr && atoi(ab_env("NVT_ENC_HALF")) != 0;
  if (half) return NVT_OK;   // (the experiment builds its smaller head per launch)
  const bool h16 = enc_head16();
  const uint64_t cap = h16 ? (uint64_t)kHead16Keys : (uint64_t)HotCfg<int32_t>::slots / 8 * 7;
  HeadJobs jobs;
  int m = 0;
  auto flush = [&]() -> int {
    if (m == 0) return NVT_OK;
    NVT_PROF("vocab_order", 0, s);
    if (h16) enc_head_build_many_kernel<true><<<m, kEncBS, 0, s>>>(jobs);
    else enc_head_build_many_kernel<false><<<m, kEncBS, 0, s>>>(jobs);
    NVT_CHECK_LAUNCH();
    m = 0;
    return NVT_OK;
  };
  for (int i = 0; i < count; ++i) {
    if (images[i] == nullptr || vocab_keys[i] == nullptr || n[i] <= NVT_ENCODE_RESIDENT_I32) continue;
    jobs.j[m].keys = vocab_keys[i];
    jobs.j[m].image = (unsigned char *)images[i];
    jobs.j[m].first_label = first_label[i];
    jobs.j[m].n_hot = (uint32_t)(n[i] < cap ? n[i] : cap);
    if (++m == kHeadJobsMax) {
      int rc = flush();
      if (rc) return rc;
    }
  }
  return flush();
}
int encode_head_build(const int32_t *vocab_keys, uint64_t n, int64_t first_label, void *image,
                      hipStream_t s) {
  return encode_head_build_many(&vocab_keys, &n, &first_label, &image, 1, s);
}
}  // namespace nvt

extern "C" {

int nvt_encode_stats(uint64_t *out2, int reset, void *stream) {
  NVT_CHECK_ARG(out2, "null out");
  hipStream_t s = (hipStream_t)stream;
  NVT_CHECK_HIP(hipStreamSynchronize(s));
  unsigned long long v[2] = {0, 0};
  NVT_CHECK_HIP(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_enc_stats), sizeof(v)));
  out2[0] = v[0];
  out2[1] = v[1];
  if (reset) {
    const unsigned long long z[2] = {0, 0};
    NVT_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_enc_stats), z, sizeof(z)));
  }
  return NVT_OK;
}

int nvt_encode_table_bytes(int key_bytes, uint64_t capacity, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes && (key_bytes == 4 || key_bytes == 8), "key_bytes must be 4 or 8");
  *bytes = capacity * (key_bytes == 4 ? sizeof(EncSlot<int32_t>) : sizeof(EncSlot<int64_t>));
  return NVT_OK;
}
int nvt_encode_build_i32(const int32_t *vocab_keys, uint64_t n_vocab, int64_t first_label,
                         void *table, uint64_t capacity, int64_t *sentinel_label, int unique_keys,
                         void *stream) {
  return build_launch<int32_t>(vocab_keys, n_vocab, first_label, table, capacity, sentinel_label,
                               unique_keys, (hipStream_t)stream);
}
int nvt_encode_build_i64(const int64_t *vocab_keys, uint64_t n_vocab, int64_t first_label,
                         void *table, uint64_t capacity, int64_t *sentinel_label, int unique_keys,
                         void *stream) {
  return build_launch<int64_t>(vocab_keys, n_vocab, first_label, table, capacity, sentinel_label,
                               unique_keys, (hipStream_t)stream);
}
int nvt_encode_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, const void *table,
                   uint64_t capacity, const int64_t *sentinel_label, int64_t null_label,
                   int64_t oov_label, uint32_t num_buckets, void *out, int out_bytes,
                   const int32_t *vocab_keys, uint64_t n_vocab, int64_t first_label,
                   void *stream) {
  return encode_launch<int32_t>(keys, valid, n, table, capacity, sentinel_label, null_label,
                                oov_label, num_buckets, out, out_bytes, vocab_keys, n_vocab,
                                first_label, (hipStream_t)stream);
}
int nvt_encode_i64(const int64_t *keys, const uint8_t *valid, uint64_t n, const void *table,
                   uint64_t capacity, const int64_t *sentinel_label, int64_t null_label,
                   int64_t oov_label, uint32_t num_buckets, void *out, int out_bytes,
                   const int64_t *vocab_keys, uint64_t n_vocab, int64_t first_label,
                   void *stream) {
  return encode_launch<int64_t>(keys, valid, n, table, capacity, sentinel_label, null_label,
                                oov_label, num_buckets, out, out_bytes, vocab_keys, n_vocab,
                                first_label, (hipStream_t)stream);
}
int nvt_encode_many(const nvt_encode_col *cols, int ncols, void *stream) {
  NVT_CHECK_ARG(ncols == 0 || cols, "null descriptors");
  // the columns are independent (own output, own tables): they go round-robin onto
  // NVT_ENCODE_STREAMS (default 3) internal streams forked from / joined into `stream`.  Round 2
  // measured nothing for it (15.42 / 15.40 / 15.31 ms per step with 1 / 2 / 3 streams: every encode
  // kernel filled the chip with one 128 KiB-LDS workgroup per CU and was bound by the HBM stream for
  // its whole duration); the pipelined cache mode of round 6 waits for table probes for a good part
  // of its launches and leaves stream bandwidth to the launch beside it: 9.55 / 9.25 / 9.20 / 9.20 ms
  // per step with 1 / 2 / 3 / 4 streams.  (Read at every call: bench.py's per-kernel pass sets 1.)
  const int n_side = [] {
    const char *e = getenv("NVT_ENCODE_STREAMS");
    int v = e ? atoi(e) : 3;
    return v < 1 ? 1 : v > kSideStreams ? kSideStreams : v;
  }();
  hipStream_t main_s = (hipStream_t)stream;
  SidePool *pool = nullptr;
  const int lanes = ncols < n_side ? ncols : n_side;
  const bool fork = lanes > 1;
  if (fork) {
    int rc = side_pool(2, &pool);
    if (rc) return rc;
    NVT_CHECK_HIP(hipEventRecord(pool->fork, main_s));
    for (int k = 0; k < lanes; ++k) NVT_CHECK_HIP(hipStreamWaitEvent(pool->s[k], pool->fork, 0));
  }
  int rc_all = NVT_OK;
  for (int i = 0; i < ncols; ++i) {
    const nvt_encode_col &c = cols[i];
    hipStream_t cs = fork ? pool->s[i % lanes] : main_s;
    int rc;
    if (c.wait_event) NVT_CHECK_HIP(hipStreamWaitEvent(cs, (hipEvent_t)c.wait_event, 0));
    if (c.key_bytes == 4)
      rc = encode_launch<int32_t>((const int32_t *)c.keys, c.valid, c.n, c.table, c.capacity,
                                  c.sentinel_label, c.null_label, c.oov_label, c.num_buckets, c.out,
                                  c.out_bytes, (const int32_t *)c.vocab_keys, c.n_vocab,
                                  c.first_label, cs, c.range_aux, c.head_image);
    else if (c.key_bytes == 8)
      rc = encode_launch<int64_t>((const int64_t *)c.keys, c.valid, c.n, c.table, c.capacity,
                                  c.sentinel_label, c.null_label, c.oov_label, c.num_buckets, c.out,
                                  c.out_bytes, (const int64_t *)c.vocab_keys, c.n_vocab,
                                  c.first_label, cs);
    else {
      set_error("nvt_encode_many: key_bytes must be 4 or 8 (column %d)", i);
      rc = NVT_EINVAL;
    }
    if (rc) {
      rc_all = rc;  // (the internal streams are joined below all the same)
      break;
    }
  }
  if (fork)
    for (int k = 0; k < lanes; ++k) {
      NVT_CHECK_HIP(hipEventRecord(pool->join[k], pool->s[k]));
      NVT_CHECK_HIP(hipStreamWaitEvent(main_s, pool->join[k], 0));
    }
  return rc_all;
}
int nvt_hash_bucket_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, uint32_t num_buckets,
                        int32_t *out, const uint64_t *xor_in, uint64_t *xor_out, void *stream) {
  return hash_bucket_launch<int32_t>(keys, valid, n, num_buckets, out, xor_in, xor_out,
                                     (hipStream_t)stream);
}
int nvt_hash_bucket_i64(const int64_t *keys, const uint8_t *valid, uint64_t n, uint32_t num_buckets,
                        int32_t *out, const uint64_t *xor_in, uint64_t *xor_out, void *stream) {
  return hash_bucket_launch<int64_t>(keys, valid, n, num_buckets, out, xor_in, xor_out,
                                     (hipStream_t)stream);
}

}  // extern "C"
