// Cross-file entry points of the kernel library (host side), used by the batched C-ABI calls.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvt {

// nvt_util.hip: per-device pools of internal streams that batched calls fork onto from / join
// into the caller's stream (`which` 0: vocabulary finalisation, 1: counting)
constexpr int kSideStreams = 3;
struct SidePool {
  int dev = -1, which = 0;
  hipStream_t s[kSideStreams];
  hipEvent_t fork, join[kSideStreams], aux;
};
int side_pool(int which, SidePool **out);

// nvt_sort.hip
int vocab_sort_any(int key_bytes, void *keys, int64_t *counts, uint64_t n, int64_t max_count,
                   void *tmp, hipStream_t s);
struct SmallSortDesc {
  int32_t *keys;
  int64_t *counts;
  unsigned n;
};
// int32 keys, 2 <= n <= 16384, 0 < max_count < 2^32: sorted by the one-launch batched kernel
bool vocab_sort_small_eligible(int key_bytes, uint64_t n, int64_t max_count);
int vocab_sort_small_batch(const SmallSortDesc *cols, int ncols, hipStream_t s);

// stable LSD radix sort of packed 64-bit words on the bit range [bit_lo, bit_hi); *result is
// `data` or a buffer inside tmp (sort_words_tmp_bytes(n) bytes)
uint64_t sort_words_tmp_bytes(uint64_t n);
int sort_words_bits(uint64_t *data, uint64_t n, int bit_lo, int bit_hi, void *tmp, uint64_t **result,
                    hipStream_t s);

// nvt_encode.hip
int encode_build_any(int key_bytes, const void *vocab, uint64_t n, int64_t first_label, void *table,
                     uint64_t capacity, int64_t *sentinel_label, int unique_keys, hipStream_t s);

}  // namespace nvt
