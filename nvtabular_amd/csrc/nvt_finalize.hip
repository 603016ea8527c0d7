// Categorify.fit_end on the device, for every vocabulary of a fit in ONE C-ABI call:
// order each (key, count) list by (count desc, key asc) -- the two sort_values of
// categorify.py:1300,1316 -- and build its encode table (categorify.py:1558-1807 consumes it).
//
// Why one call: the per-vocabulary Python loop (sort call + table allocation + build call,
// ~60 us of host time each) was longer than the kernels of the 13 small Criteo vocabularies,
// and on a slow host it left the GPU idle for half of every step.  Here the host enqueues
// everything back to back:
//   * all small vocabularies (int32 keys, <= 16384 entries) are sorted by ONE launch, one
//     workgroup per vocabulary (LDS bitonic network over packed words);
//   * the large ones go largest-first round-robin onto three internal streams forked from /
//     joined into the caller's stream, so their short tail kernels overlap;
//   * every table build follows its sort on the same stream.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "nvt_common.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"

namespace nvt {
namespace {
constexpr int kSide = kSideStreams;
constexpr uint64_t kOrderBatchMaxEntries = 1ull << 23;
}  // namespace
}  // namespace nvt

using namespace nvt;

static int finalize_impl(const nvt_vocab_col *cols, int ncols, hipStream_t main_s, SidePool *&pool,
                         bool &forked);

extern "C" int nvt_vocab_finalize_many(const nvt_vocab_col *cols, int ncols, void *stream) {
  NVT_CHECK_ARG(ncols == 0 || cols, "null descriptors");
  SidePool *pool = nullptr;
  bool forked = false;
  const int rc = finalize_impl(cols, ncols, (hipStream_t)stream, pool, forked);
  if (rc != NVT_OK && forked && pool != nullptr) {
    // an error after the fork: whatever was launched keeps running on the internal streams; join
    // them into the caller's stream so that it may free / reuse the buffers it handed over
    for (int i = 0; i < kSide; ++i) {
      (void)hipEventRecord(pool->join[i], pool->s[i]);
      (void)hipStreamWaitEvent((hipStream_t)stream, pool->join[i], 0);
    }
  }
  return rc;
}

static int finalize_impl(const nvt_vocab_col *cols, int ncols, hipStream_t main_s, SidePool *&pool,
                         bool &forked) {
  std::vector<int> small, big;
  for (int i = 0; i < ncols; ++i) {
    const nvt_vocab_col &c = cols[i];
    NVT_CHECK_ARG(c.key_bytes == 4 || c.key_bytes == 8, "key_bytes must be 4 or 8");
    NVT_CHECK_ARG(c.n <= 1 || (c.keys && c.counts), "null keys/counts");
    NVT_CHECK_ARG(c.table == nullptr || c.sentinel_label, "table without sentinel_label");
    NVT_CHECK_ARG(c.src_keys == nullptr || (c.key_bytes == 4 && c.src_counts && (c.cls_hist || c.src_labels)),
                  "key-sorted source: int32 keys with src_counts and cls_hist (or src_labels)");
    if (c.src_keys == nullptr && vocab_sort_small_eligible(c.key_bytes, c.n, c.max_count))
      small.push_back(i);
    else
      big.push_back(i);
  }
  std::sort(big.begin(), big.end(), [&](int a, int b) { return cols[a].n > cols[b].n; });
  // large vocabularies: forked onto the internal streams (only when there is something to overlap)
  const bool fork = big.size() + (small.empty() ? 0 : 1) > 1 && !getenv("NVT_FINALIZE_SERIAL");
  if (fork) {
    int rc = side_pool(0, &pool);
    if (rc) return rc;
    forked = true;
    NVT_CHECK_HIP(hipEventRecord(pool->fork, main_s));
    for (int i = 0; i < kSide; ++i) NVT_CHECK_HIP(hipStreamWaitEvent(pool->s[i], pool->fork, 0));
  }
  // with a ready_event the vocabulary's stream is not joined into `stream`: the event is
  // recorded behind its last kernel and the consumer waits on it
  bool need_join = false;
  // the LDS head images of several vocabularies in ONE launch (a workgroup each)
  auto build_heads = [&](const std::vector<int> &which, hipStream_t s) -> int {
    std::vector<const int32_t *> k;
    std::vector<uint64_t> nn;
    std::vector<int64_t> fl;
    std::vector<void *> im;
    for (int i : which) {
      const nvt_vocab_col &c = cols[i];
      if (!(c.head_image && c.key_bytes == 4 && c.unique_keys)) continue;
      k.push_back((const int32_t *)c.keys);
      nn.push_back(c.n);
      fl.push_back(c.first_label);
      im.push_back(c.head_image);
    }
    if (k.empty()) return NVT_OK;
    return encode_head_build_many(k.data(), nn.data(), fl.data(), im.data(), (int)k.size(), s);
  };
  auto finish = [&](const nvt_vocab_col &c, hipStream_t s, bool head = true) -> int {
    if (c.table != nullptr && c.src_keys == nullptr) {
      int rc = encode_build_any(c.key_bytes, c.keys, c.n, c.first_label, c.table, c.capacity,
                                c.sentinel_label, c.unique_keys, s);
      if (rc) return rc;
    }
    if (head && c.head_image && c.key_bytes == 4 && c.unique_keys) {
      int rc = encode_head_build((const int32_t *)c.keys, c.n, c.first_label, c.head_image, s);
      if (rc) return rc;
    }
    if (c.ready_event) {
      NVT_CHECK_HIP(hipEventRecord((hipEvent_t)c.ready_event, s));
      if (s != main_s && ab_env("NVT_FLUSH_QUERY")) (void)hipStreamQuery(s);
    } else if (s != main_s)
      need_join = true;
    return NVT_OK;
  };
  // the small vocabularies first, on the caller's stream: their one-workgroup-per-vocabulary
  // kernel needs a whole CU's LDS and would otherwise wait until the radix passes drain
  if (!small.empty()) {
    std::vector<SmallSortDesc> d(small.size());
    for (size_t j = 0; j < small.size(); ++j) {
      const nvt_vocab_col &c = cols[small[j]];
      d[j].keys = (int32_t *)c.keys;
      d[j].counts = c.counts;
      d[j].n = (unsigned)c.n;
    }
    int rc = vocab_sort_small_batch(d.data(), (int)d.size(), main_s);
    if (rc) return rc;
    for (int i : small) {
      rc = finish(cols[i], main_s);
      if (rc) return rc;
    }
  }
  std::vector<OrderTail> tails;   // class-255 sorts of the key-sorted vocabularies: one batch
  std::vector<int> tail_cols;
  // key-sorted vocabularies with a range table (dumped by the counting pass, or flat): ordered
  // by ONE chain of batched launches on the first internal stream (vocab_order_sorted_batch)
  // instead of ~10 launches per vocabulary spread over the streams -- the host took as long to
  // enqueue those as the GPU to run them
  static const bool batch_order = ab_env("NVT_NO_ORDER_BATCH") == nullptr;
  std::vector<char> batched(ncols > 0 ? ncols : 0, 0);
  if (batch_order) {
    std::vector<OrderSortedJob> jobs;
    std::vector<int> job_cols;
    for (int i : big) {
      const nvt_vocab_col &c = cols[i];
      if (c.src_keys == nullptr || c.table == nullptr || c.range_aux == nullptr || c.n == 0) continue;
      if (c.src_labels != nullptr) continue;   // (labelled already: no ordering pass at all)
      // (tens of millions of entries: the launches are not what such a vocabulary waits for, and
      // on a stream of its own its encode starts while the next one is still being ordered --
      // four 36 M-entry vocabularies: 14.3 ms of GPU time per step against 15.7 in one batch)
      if (c.n > kOrderBatchMaxEntries) continue;
      NVT_CHECK_ARG(c.sort_tmp, "null sort_tmp");
      jobs.push_back({(const int32_t *)c.src_keys, c.src_counts, c.n, c.cls_hist, c.n_big, c.max_count,
                      (int32_t *)c.keys, c.counts, c.sort_tmp, c.first_label, c.table, c.capacity,
                      c.sentinel_label, c.range_aux, c.range_nb_log2, c.flat_slots});
      job_cols.push_back(i);
      batched[i] = 1;
    }
    if (!jobs.empty()) {
      hipStream_t s = fork ? pool->s[0] : main_s;
      int rc = vocab_order_sorted_batch(jobs.data(), (int)jobs.size(), s);
      if (rc) return rc;
      rc = build_heads(job_cols, s);
      if (rc) return rc;
      for (int i : job_cols) {
        rc = finish(cols[i], s, false);
        if (rc) return rc;
      }
    }
  }
  for (size_t j = 0; j < big.size(); ++j) {
    const nvt_vocab_col &c = cols[big[j]];
    if (batched[big[j]]) continue;
    hipStream_t s = fork ? pool->s[j % kSide] : main_s;
    if (c.src_keys != nullptr && c.src_labels != nullptr) {
      // key-sorted list with the vocabulary position of every entry (multi-GPU: the owners
      // labelled their shards): one scatter + the table build
      NVT_CHECK_ARG(c.sort_tmp, "null sort_tmp");
      int rc = vocab_from_labels((const int32_t *)c.src_keys, c.src_counts, c.src_labels, c.n,
                                 (int32_t *)c.keys, c.counts, c.sort_tmp, c.first_label, c.table,
                                 c.capacity, c.sentinel_label, c.range_aux, c.flat_slots, s);
      if (rc) return rc;
      if (c.head_image && c.key_bytes == 4 && c.unique_keys) {
        rc = encode_head_build((const int32_t *)c.keys, c.n, c.first_label, c.head_image, s);
        if (rc) return rc;
      }
      if (c.ready_event) {
        NVT_CHECK_HIP(hipEventRecord((hipEvent_t)c.ready_event, s));
      } else if (s != main_s)
        need_join = true;
      continue;
    }
    if (c.src_keys != nullptr) {
      // key-sorted list of the range path: one stable counting pass orders it and fills the table
      NVT_CHECK_ARG(c.sort_tmp, "null sort_tmp");
      bool deferred = false;
      static const bool batch_tail = ab_env("NVT_NO_TAIL_BATCH") == nullptr;
      int rc = vocab_order_from_sorted((const int32_t *)c.src_keys, c.src_counts, c.n, c.cls_hist,
                                       c.n_big, c.max_count, (int32_t *)c.keys, c.counts,
                                       c.sort_tmp, c.first_label, c.table, c.capacity,
                                       c.sentinel_label, c.range_aux, c.range_nb_log2, s,
                                       batch_tail ? &deferred : nullptr, c.flat_slots);
      if (rc) return rc;
      if (deferred) {
        tails.push_back({(int32_t *)c.keys, c.counts, c.n_big, c.first_label, c.table, c.capacity,
                         c.sentinel_label, c.range_aux});
        tail_cols.push_back(big[j]);
        continue;  // its ready event is recorded behind the batched tail
      }
    } else if (c.n > 1) {
      NVT_CHECK_ARG(c.sort_tmp, "null sort_tmp");
      int rc = vocab_sort_any(c.key_bytes, c.keys, c.counts, c.n, c.max_count, c.sort_tmp, s);
      if (rc) return rc;
    }
    int rc = finish(c, s);
    if (rc) return rc;
  }
  if (!tails.empty()) {
    // every scatter has to be done before the batched sort: internal stream 0 waits for the
    // other two and runs the tail; the caller's stream is NOT joined (joining it here cost
    // 0.3 ms per Criteo step: fill + normalize and the small encodes started later)
    hipStream_t ts = main_s;
    bool all_events = true;
    for (int i : tail_cols) all_events = all_events && cols[i].ready_event != nullptr;
    if (fork) {
      ts = pool->s[0];
      for (int i = 1; i < kSide; ++i) {
        NVT_CHECK_HIP(hipEventRecord(pool->join[i], pool->s[i]));
        NVT_CHECK_HIP(hipStreamWaitEvent(ts, pool->join[i], 0));
      }
      if (!all_events) need_join = true;
    }
    int rc = vocab_order_tail_batch(tails.data(), (int)tails.size(), ts);
    if (rc) return rc;
    rc = build_heads(tail_cols, ts);
    if (rc) return rc;
    for (int i : tail_cols)
      if (cols[i].ready_event) NVT_CHECK_HIP(hipEventRecord((hipEvent_t)cols[i].ready_event, ts));
  }
  if (fork && need_join) {
    for (int i = 0; i < kSide; ++i) {
      NVT_CHECK_HIP(hipEventRecord(pool->join[i], pool->s[i]));
      NVT_CHECK_HIP(hipStreamWaitEvent(main_s, pool->join[i], 0));
    }
  }
  return NVT_OK;
}

extern "C" int nvt_event_create(void **event) {
  NVT_CHECK_ARG(event, "null out pointer");
  hipEvent_t e = nullptr;
  NVT_CHECK_HIP(hipEventCreateWithFlags(&e, getenv("NVT_EVENT_TIMING") ? hipEventDefault
                                                                        : hipEventDisableTiming));
  *event = (void *)e;
  return NVT_OK;
}
extern "C" void nvt_event_destroy(void *event) {
  if (event) (void)hipEventDestroy((hipEvent_t)event);
}
extern "C" int nvt_stream_wait_event(void *stream, void *event) {
  NVT_CHECK_ARG(event, "null event");
  NVT_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return NVT_OK;
}
