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
// the packed rows of a key column ((key - bias) image << 32 | fold << rb | row) sorted by bits
// [rb, 64) without writing the unsorted words out first; *result = a buffer inside tmp
int sort_packed_keys(const void *keys, int key_dtype, int64_t key_bias, const uint8_t *fold, int rb,
                     uint64_t n, void *tmp, uint64_t **result, hipStream_t stream);

// nvt_range_count.hip: path NVT_PATH_RANGE of nvt_dense_count_* (aux = hot image + range
// parameters written by hot_sample_kernel + class histogram)
uint64_t range_count_ws_bytes(uint64_t n, int nb_log2);
int range_count_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, int nb_log2, void *ws,
                    int32_t *aux, int32_t *out_keys, int64_t *out_cnt, uint64_t out_cap,
                    void *range_table, uint64_t *state, hipStream_t s, bool pieces);

// nvt_sort_count.hip: path NVT_PATH_SORT of nvt_dense_count_* (radix sort + run lengths; hist =
// the column's uint32[256] class histogram block)
uint64_t sort_count_ws_bytes(uint64_t n);
int sort_count_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, void *ws, unsigned *hist,
                   int32_t *out_keys, int64_t *out_cnt, uint64_t out_cap, uint64_t *state,
                   hipStream_t s);

// nvt_sort.hip: vocabulary order of a KEY-SORTED (key, count) list (range path) in one stable
// counting pass on min(count, 255) + encode table filled in the same pass
uint64_t vocab_order_tmp_bytes(uint64_t n, uint64_t n_big);
int vocab_order_from_sorted(const int32_t *src_keys, const int64_t *src_cnts, uint64_t n,
                            const unsigned *cls_hist, uint64_t n_big, int64_t max_count,
                            int32_t *out_keys, int64_t *out_cnts, void *tmp, int64_t first_label,
                            void *table, uint64_t capacity, int64_t *sentinel_label,
                            const int32_t *range_aux, int range_nb_log2, hipStream_t s,
                            bool *tail_deferred = nullptr, uint64_t flat_slots = 0);
// the same for several vocabularies that own a range table (dumped or flat): one launch per
// stage for ALL of them, class-255 tails included
struct OrderSortedJob {
  const int32_t *src_keys;
  const int64_t *src_cnts;
  uint64_t n;
  const unsigned *cls_hist;
  uint64_t n_big;
  int64_t max_count;
  int32_t *out_keys;
  int64_t *out_cnts;
  void *tmp;
  int64_t first_label;
  void *table;
  uint64_t capacity;
  int64_t *sentinel_label;
  const int32_t *range_aux;
  int range_nb_log2;
  uint64_t flat_slots;
};
int vocab_order_sorted_batch(const OrderSortedJob *jobs, int njobs, hipStream_t s);
// LDS head image of an ordered int32 vocabulary for the cache-mode encode (nvt_encode.hip)
int encode_head_build(const int32_t *vocab_keys, uint64_t n, int64_t first_label, void *image,
                      hipStream_t s);
int encode_head_build_many(const int32_t *const *vocab_keys, const uint64_t *n, const int64_t *first_label,
                           void *const *images, int count, hipStream_t s);
// key-sorted list whose entries carry their 0-based position in the vocabulary order (multi-GPU:
// labelled shard by shard on the owners): ordered arrays + table without any ordering pass
int vocab_from_labels(const int32_t *src_keys, const int64_t *src_cnts, const int32_t *labels, uint64_t n,
                      int32_t *out_keys, int64_t *out_cnts, void *tmp, int64_t first_label, void *table,
                      uint64_t capacity, int64_t *sentinel_label, const int32_t *range_aux,
                      uint64_t flat_slots, hipStream_t s);
// the deferred part: sort of the n_big leading entries (class 255) + their labels
struct OrderTail {
  int32_t *keys;
  int64_t *counts;
  uint64_t n_big;
  int64_t first_label;
  void *table;
  uint64_t capacity;
  int64_t *sentinel_label;
  const int32_t *range_aux;
};
int vocab_order_tail_batch(const OrderTail *t, int nt, hipStream_t s);

// nvt_encode.hip
int encode_clear_any(int key_bytes, void *table, uint64_t capacity, int64_t *sentinel_label,
                     hipStream_t s);
// insert vocab[0 .. n) with labels first_label + i into an already cleared / partly filled table
int encode_insert_any(int key_bytes, const void *vocab, uint64_t n, int64_t first_label,
                      void *table, uint64_t capacity, int64_t *sentinel_label, hipStream_t s);
int encode_build_any(int key_bytes, const void *vocab, uint64_t n, int64_t first_label, void *table,
                     uint64_t capacity, int64_t *sentinel_label, int unique_keys, hipStream_t s);

}  // namespace nvt
