// Monotone key -> slot map of the range path, shared by the counting kernels
// (nvt_range_count.hip), the vocabulary ordering (nvt_sort.hip) and the encode kernel
// (nvt_encode.hip): the per-bucket LDS tables of the counting pass are dumped as they are and
// BECOME the encode table ("range table"), so all three must address it the same way.
#pragma once
#include "nvt_common.hpp"

namespace nvt {

constexpr int kRpSlots = 16384;              // slots per bucket table
constexpr int kRpTail = 128;                 // slots past the table end (no wrap-around)
constexpr int kRpRegion = kRpSlots + kRpTail;  // slots per bucket in the dumped table
constexpr int kRpGuard = 64;                 // empty slots behind the last bucket
// empty slot of an int32 encode table: {key INT32_MIN, label INT32_MAX} (enc_clear_kernel)
constexpr unsigned long long kEncEmptySlot =
    ((unsigned long long)(uint32_t)INT32_MAX << 32) | (unsigned long long)(uint32_t)INT32_MIN;

// order-preserving image of an int32 key in uint32
__device__ __forceinline__ uint32_t ukey(int32_t k) { return (uint32_t)k ^ 0x80000000u; }

// key -> "fine slot" f in [0, NB * 16384): bucket = f >> 14, home slot = f & 16383.
// f = high half of ((min(max(u - ulo, 0), span) << sh) * mul) with (mul, sh) chosen by the sample
// kernel so that the padded span of the sampled keys covers ALL buckets evenly (a power-of-two
// bucket width left up to half of them empty): span + 1 > F = NB * 16384: mul = F * 2^32 /
// (span + 1), sh = 0; smaller spans (dense ids): the integer factor m = F / (span + 1) as
// mul = m << (32 - sh), sh = bit length of m -- exactly d * m, keys spread with gaps.  ONE 32 x 32
// bit multiply per key, the same instruction in both forms (the kernels used to compute the high
// AND the low product and select: integer multiplies are quarter-rate instructions, and these
// loops are bound by instruction issue).
__host__ __device__ inline void range_map_params(uint64_t span, uint64_t F, uint32_t *mul, int *sh) {
  if (span + 1 > F) {
    *mul = (uint32_t)((F << 32) / (span + 1));
    *sh = 0;
    return;
  }
  uint64_t m = F / (span + 1);  // >= 1
  auto bits = [](uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; };
  int b = bits(m);
  // (d << sh) must stay below 2^32 for every d <= span; a smaller factor only spreads less
  while (b > 0 && (b > 31 || (span >> (32 - b)) != 0)) {
    m >>= 1;
    b = bits(m);
  }
  if (m == 0) {  // a span of >= 2^31 keys in a table that large: f = d - 1 (monotone, < F)
    *mul = 0xFFFFFFFFu;
    *sh = 0;
    return;
  }
  *mul = (uint32_t)(m << (32 - b));
  *sh = b;
}
// Piecewise form (round 4): keys that are NOT spread over their range (dense, frequency-ordered
// ids: most of the rows and most of the distinct keys sit in a sliver of [min, max]) overflow the
// equal-width buckets of the linear map.  The sample kernel then cuts the key range into
// kRpPieces pieces at quantiles of its cold sample (rows and distinct keys blended, so that
// neither the rows nor the keys of a piece exceed twice the average) and every piece maps
// linearly onto an equal share of the fine slots: still monotone, buckets balanced for any key
// distribution.  pw (in the column's aux block): u32 splitters[kRpPieces + 1], u32 mul[kRpPieces],
// u32 shflags[2] (bit p set: wide piece, high half of the 32 x 32 product; clear: 16-bit fixed point).
constexpr int kRpPieces = 64;
constexpr int kRpPwMul = kRpPieces + 1, kRpPwSh = 2 * kRpPieces + 1;
struct RangeMap {
  uint32_t ulo, span;
  uint32_t mul;
  int sh;  // pre-shift of the key offset (0: wide spans)
  int flat;  // 1: the table is ONE run of F slots (+ tail), slot = f (vocabulary tables built from a
             //    key-sorted list: flat_build_kernel); 0: bucket regions dumped by the counting pass
  uint32_t piece_slots;      // 0: linear map; else fine slots per piece (the piecewise form)
  const uint32_t *pw;        // piecewise parameters in global memory
  // ... and staged in LDS (stage_pieces): an LDS-address-space pointer, so that the six dependent
  // reads of the splitter search are ds_read, not flat loads that wait for vmcnt(0) each (a
  // generic pointer that may point either way compiled to flat_load_dword + s_waitcnt vmcnt(0)
  // per step: the piecewise map of the dense-id columns)
  const __attribute__((address_space(3))) uint32_t *lpw;
  template <typename P>
  __device__ __forceinline__ uint32_t fine_pieces(P q, uint32_t u) const {
    // piece p: splitters[p] <= u < splitters[p + 1] (keys outside the sampled range clamp)
    unsigned p = 0;
#pragma unroll
    for (unsigned step = kRpPieces / 2; step > 0; step >>= 1)
      p += (u >= q[p + step]) ? step : 0u;
    const uint32_t s0 = q[p], s1 = q[p + 1];
    uint32_t d = u > s0 ? u - s0 : 0u;
    d = d < s1 - s0 - 1u ? d : s1 - s0 - 1u;
    const uint32_t m = q[kRpPwMul + p];
    const bool hi = (q[kRpPwSh + (p >> 5)] >> (p & 31)) & 1u;
    // wide piece (more keys than slots): high half of d * (S << 32) / width; narrow piece
    // (dense ids: fewer keys than slots): 16-bit fixed point, (d * ((S << 16) / width)) >> 16
    uint32_t f = hi ? __umulhi(d, m) : (uint32_t)(((uint64_t)d * m) >> 16);
    f = f < piece_slots - 1u ? f : piece_slots - 1u;
    return p * piece_slots + f;
  }
  __device__ __forceinline__ uint32_t fine(int32_t key) const {
    const uint32_t u = ukey(key);
    if (piece_slots) return lpw ? fine_pieces(lpw, u) : fine_pieces(pw, u);
    uint32_t d = u > ulo ? u - ulo : 0u;
    d = d < span ? d : span;
    return __umulhi(d << sh, mul);
  }
  // the same for kernels that ALWAYS stage the pieces in LDS (stage_pieces) and have hoisted the
  // linear / piecewise decision out of their loops (PW at compile time): no branch, and none of the
  // global-memory form's loads (with their s_waitcnt vmcnt(0)) inside a software-pipelined loop body
  template <bool PW>
  __device__ __forceinline__ uint32_t fine_staged(int32_t key) const {
    const uint32_t u = ukey(key);
    if constexpr (PW) {
      return fine_pieces(lpw, u);
    } else {
      uint32_t d = u > ulo ? u - ulo : 0u;
      d = d < span ? d : span;
      return __umulhi(d << sh, mul);
    }
  }
  // (FLAT at compile time as well: flat tables never use pieces)
  template <bool PW, bool FLAT>
  __device__ __forceinline__ uint64_t table_slot_staged(int32_t key) const {
    const uint32_t f = fine_staged<PW>(key);
    return FLAT ? (uint64_t)f : (uint64_t)(f >> 14) * kRpRegion + (f & (kRpSlots - 1));
  }
  __device__ __forceinline__ uint32_t bucket(int32_t key) const { return fine(key) >> 14; }
  __device__ __forceinline__ uint32_t slot(int32_t key) const { return fine(key) & (kRpSlots - 1); }
  // first slot to look at in the dumped table (bucket regions of kRpRegion slots, probing
  // runs forward without wrapping; the counting pass guarantees an empty slot ends every chain)
  __device__ __forceinline__ uint64_t table_slot(int32_t key) const {
    const uint32_t f = fine(key);
    return flat ? (uint64_t)f : (uint64_t)(f >> 14) * kRpRegion + (f & (kRpSlots - 1));
  }
};

// Slot of `key` in a FLAT range table ({key, label} words laid out from a key-sorted list by
// flat_build_kernel: every run of occupied slots is in ascending key order), or ~0 when the
// key is not there.  The first kFlatLinear slots are probed one by one (keys that are spread
// over their range sit within a few slots of home); beyond that the run is searched by
// doubling steps and bisection -- O(log displacement) loads.  Keys that CLUSTER in their range
// (dense ids with a heavy tail: a million consecutive ids in front of a sparse range) sit
// hundreds of thousands of slots from home; a slot-by-slot walk took minutes per column there.
// `e0` = the word already loaded from `home`.
constexpr int kFlatLinear = 8;
// the rare part:
// table[lo] is occupied and below `key`; doubling steps, then bisection
__device__ __forceinline__ uint64_t flat_gallop(const unsigned long long *__restrict__ table,
                                             uint64_t slots, uint64_t lo, int32_t key) {
  const uint32_t uk = ukey(key);
  auto below = [&](unsigned long long e) {  // occupied and in front of `key`
    const int32_t ek = (int32_t)(uint32_t)e;
    return ek != INT32_MIN && ukey(ek) < uk;
  };
  uint64_t width = kFlatLinear, hi;
  while (true) {
    hi = lo + width < slots - 1 ? lo + width : slots - 1;
    if (!below(table[hi]) || hi == slots - 1) break;
    lo = hi;
    width <<= 1;
  }
  if (below(table[hi])) return ~0ull;  // (ran into the end of the table)
  while (hi - lo > 1) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if (below(table[mid])) lo = mid; else hi = mid;
  }
  return (int32_t)(uint32_t)table[hi] == key ? hi : ~0ull;
}

// (`word`: receives the {key, label} word of the slot that is returned -- the callers that only
// read the label need no second load of it)
__device__ __forceinline__ uint64_t flat_find_from(const unsigned long long *__restrict__ table,
                                                   uint64_t slots, uint64_t home, int32_t key,
                                                   unsigned long long e0,
                                                   unsigned long long *word = nullptr) {
  uint64_t sl = home;
  unsigned long long e = e0;
  int step = 0;
  while (true) {
    const int32_t ek = (int32_t)(uint32_t)e;
    if (ek == key) {
      if (word) *word = e;
      return sl;
    }
    if (ek == INT32_MIN) return ~0ull;          // an empty slot ends every run
    if (++step == kFlatLinear) {                // far from home: the keys cluster in their range
      if (ukey(ek) > ukey(key)) return ~0ull;   // (runs are in key order: already past it)
      const uint64_t at = flat_gallop(table, slots, sl, key);
      if (word && at != ~0ull) *word = table[at];
      return at;
    }
    if (++sl >= slots) return ~0ull;
    e = table[sl];
  }
}

// the piecewise parameters staged in LDS (`lds`: kRpPwWords words; every thread of the workgroup
// calls this, a barrier follows in the caller before the map is used): six dependent reads per
// key come from LDS instead of the vector cache
constexpr int kRpPwWords = 2 * kRpPieces + 3;
__device__ __forceinline__ void stage_pieces(RangeMap &m, uint32_t *lds, unsigned tid, unsigned nthreads) {
  if (!m.piece_slots) return;
  for (unsigned i = tid; i < (unsigned)kRpPwWords; i += nthreads) lds[i] = m.pw[i];
  m.lpw = (const __attribute__((address_space(3))) uint32_t *)lds;
}

__device__ __forceinline__ RangeMap load_map(const int32_t *__restrict__ aux) {
  RangeMap m;
  m.ulo = (uint32_t)aux[NVT_RANGE_AUX_LO];
  m.span = (uint32_t)aux[NVT_RANGE_AUX_LO + 1];
  m.mul = (uint32_t)aux[NVT_RANGE_AUX_LO + 2];  // (word + 3: the high half, always 0)
  m.sh = aux[NVT_RANGE_AUX_LO + 4];
  m.flat = aux[NVT_RANGE_AUX_LO + 5];
  // (word + 6: FlatIndex's has-min flag; word + 7: fine slots per piece, 0 = linear.  Flat tables
  // never use pieces: their aux blocks end before the piecewise parameters)
  m.piece_slots = m.flat ? 0u : (uint32_t)aux[NVT_RANGE_AUX_LO + 7];
  m.pw = reinterpret_cast<const uint32_t *>(aux + NVT_RANGE_AUX_PW);
  m.lpw = nullptr;
  return m;
}

}  // namespace nvt
