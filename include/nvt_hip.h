/*
 * nvt_hip.h -- C ABI of the MI355X (gfx950) kernels behind the NVTabular hot path.
 *
 * This is the drop-in boundary: every O(rows) step that NVTabular's operators
 * hand to a dataframe backend (pandas on CPU, libcudf on GPU) for the
 * Categorify / FillMissing / Normalize / HashBucket / JoinGroupby /
 * TargetEncoding path has one entry point here.  Signatures use plain device
 * pointers and sizes only (no torch / Arrow types).  Conventions:
 *
 *   - every function returns 0 on success, a negative NVT_E* code on failure;
 *     nvt_last_error() returns a thread-local message for the last failure;
 *   - all data pointers are DEVICE pointers (HBM); the caller owns every buffer
 *     (nvt_gb_table objects own their slot arrays);
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous;
 *   - `valid` is an Arrow validity bitmap (LSB-first, 1 = valid, bit i = row i)
 *     or NULL when the column has no nulls.  Float columns additionally treat
 *     NaN as null (pandas isna() semantics);
 *   - re-entrant: no global mutable state.
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   nvtabular/ops/categorify.py:955-1051   _top_level_groupby   -> nvt_count_*          (size-only)
 *   nvtabular/ops/categorify.py:1054-1137  _mid/_bottom_level   -> nvt_count_merge_*, nvt_gb_merge
 *   nvtabular/ops/categorify.py:1149-1337  _write_uniques sorts -> nvt_count_compact_*, nvt_vocab_sort_*
 *   nvtabular/ops/categorify.py:1558-1807  _encode              -> nvt_encode_build_*, nvt_encode_*
 *   nvtabular/ops/categorify.py:1837-1852  _hash_bucket         -> nvt_hash_bucket_* / nvt_encode_* (num_buckets)
 *   nvtabular/ops/hash_bucket.py:86-100    HashBucket.transform -> nvt_hash_bucket_*
 *   nvtabular/ops/moments.py:64-77         _chunkwise_moments   -> nvt_moments
 *   nvtabular/ops/fill.py:49-57            FillMissing          -> nvt_fill_normalize (do_norm = 0)
 *   nvtabular/ops/normalize.py:71-90       Normalize.transform  -> nvt_fill_normalize
 *   nvtabular/ops/normalize.py:150-186     NormalizeMinMax      -> nvt_minmax, nvt_fill_normalize
 *   nvtabular/ops/join_groupby.py:175-217  JoinGroupby.transform-> nvt_gb_lookup + nvt_gather_f64
 *   nvtabular/ops/target_encoding.py:301-384 _op_group_logic    -> nvt_gb_lookup + nvt_te_apply
 *   cpp/nvtabular/inference/categorify.cc:145-252, fill.cc:91-102 (serving-time
 *   encode / fill loops) have the same element semantics as nvt_encode_* /
 *   nvt_fill_normalize.
 */
#ifndef NVT_HIP_H
#define NVT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVT_OK 0
#define NVT_EINVAL (-1)   /* bad argument */
#define NVT_EHIP (-2)     /* a HIP runtime call failed */
#define NVT_ENOMEM (-3)
#define NVT_EUNSUPPORTED (-4) /* valid input this library does not handle (the caller falls back) */

/* element types of continuous columns */
#define NVT_F32 0
#define NVT_F64 1
#define NVT_I32 2
#define NVT_I64 3
#define NVT_U8 4

/* sentinel keys marking an empty hash slot; rows holding this key value are
 * counted in state[NVT_ST_SENTINEL] instead of the table */
#define NVT_EMPTY_I32 INT32_MIN
#define NVT_EMPTY_I64 INT64_MIN

/* layout of the uint64_t state[NVT_STATE_WORDS] block every count/groupby table carries */
#define NVT_ST_NULLS 0     /* rows whose key was null                        */
#define NVT_ST_SENTINEL 1  /* rows whose key equalled the empty sentinel     */
#define NVT_ST_OCCUPIED 2  /* distinct keys currently in the table           */
#define NVT_ST_OVERFLOW 3  /* != 0: table too full, result invalid -> regrow */
/* range path, diagnostic bits beside overflow bit0: WHY the column has to be rerun */
#define NVT_OVF_REGION 16ull  /* a (bucket, workgroup) region of the partition pass filled up  */
#define NVT_OVF_PROBE 32ull   /* a probe chain left the bucket table (keys cluster in a range) */
#define NVT_OVF_FULL 64ull    /* a bucket holds more than 3/4 * 16384 distinct keys            */
#define NVT_ST_ROWS 4      /* rows consumed (nulls included)                 */
/* words 5..7: scratch cursors of nvt_dense_count_*                                  */
#define NVT_ST_MAXCOUNT 8  /* nvt_dense_count_*: largest count in the output list    */
#define NVT_ST_BIG 9       /* range path: entries whose count is >= 255            */
#define NVT_ST_NEED 10     /* range path, overflow bit1: entries the output list needs */
#define NVT_STATE_WORDS 16

int nvt_version(void);
const char *nvt_last_error(void);

/* ---- Categorify.fit: groupby-size tables (open addressing, linear probe) ----
 * i32 table slot = {int32 key, uint32 count} (8 B); i64 slot = {int64 key,
 * uint64 count} (16 B).  capacity must be a power of two. */
int nvt_count_table_bytes(int key_bytes, uint64_t capacity, uint64_t *bytes);
int nvt_count_clear(void *table, int key_bytes, uint64_t capacity, uint64_t *state, void *stream);
int nvt_count_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, void *table,
                  uint64_t capacity, uint64_t *state, void *stream);
int nvt_count_i64(const int64_t *keys, const uint8_t *valid, uint64_t n, void *table,
                  uint64_t capacity, uint64_t *state, void *stream);
/* weighted insert of an (key,count) list -- the tree-merge step (_mid_level_groupby)
 * and the owner-side merge after the multi-GPU exchange */
int nvt_count_merge_i32(const int32_t *keys, const int64_t *counts, uint64_t n, void *table,
                        uint64_t capacity, uint64_t *state, void *stream);
int nvt_count_merge_i64(const int64_t *keys, const int64_t *counts, uint64_t n, void *table,
                        uint64_t capacity, uint64_t *state, void *stream);
/* table -> dense (key,count) arrays, arbitrary order; *out_n (device) = rows written.
 * out arrays must hold state[NVT_ST_OCCUPIED] entries (capacity is always enough) */
int nvt_count_compact_i32(const void *table, uint64_t capacity, int32_t *out_keys,
                          int64_t *out_counts, uint64_t *out_n, void *stream);
int nvt_count_compact_i64(const void *table, uint64_t capacity, int64_t *out_keys,
                          int64_t *out_counts, uint64_t *out_n, void *stream);

/* ---- Categorify.fit, atomic-free: key column -> dense (key,count) list ----
 * Replaces categorify.py:1018 (the groupby-size of _top_level_groupby) and, with weights,
 * the concat + re-groupby of categorify.py:1059-1063.  `path` selects the kernel family:
 *   0  LDS tables, two launches: 256 workgroups count their row slab into a private LDS
 *      table and flush it grouped by home range; 256 workgroups merge one range each
 *      (<= ~11000 distinct int32 keys, ~5000 for int64 keys / weighted input);
 *   6  path 0 for <= 64 distinct keys: hot keys replicated per 8-lane group (same-address
 *      LDS atomics serialise);
 *   7  path 0 with 2 key classes per row slab: each LDS table keeps the keys of one class,
 *      the column is read twice (<= ~21000 distinct keys; 350 us against 420 us on path 1);
 *   1 / 2 / 3  hash-partition the rows into 256 / 64 x 64 / 64 x 256 buckets (exact
 *      per-tile histograms + scan, no cursor atomics), then one LDS table per bucket; a
 *      bucket inflated by a hot key is cut into a primary chunk plus small excess chunks
 *      (dispatched last) whose partial lists are merged per bucket.
 *   path | NVT_PATH_HOT (paths 1 / 2 / 3, int32 keys, no weights): a hot-key filter in front
 *      of the partition -- the histogram pass also counts the rows of a few thousand frequent
 *      keys (picked from a sample of the column) in LDS and switches them off for the
 *      scatter / count stages, which then handle only the remaining rows (60-95 % fewer on
 *      power-law columns).  Exact for any hot set.
 *   NVT_PATH_RANGE | (log2(buckets) << 8)  (int32 keys, no weights, 64 .. 1024 buckets of
 *      <= ~6000 distinct keys each): ONE pass partitions the rows that the hot-key filter does
 *      not absorb by KEY RANGE through per-workgroup write-combining bins in LDS (64-byte lines
 *      into a private region per (bucket, workgroup): no histogram pre-pass, no cursors); one
 *      workgroup per bucket then counts its rows in an LDS table addressed by a monotone
 *      function of the key and emits it in key order.  The output list is SORTED BY KEY
 *      (sentinel key first), and hot_image[NVT_RANGE_AUX_HIST + c] receives the number of
 *      entries with min(count, 255) == c -- what nvt_vocab_finalize_many needs to order the
 *      vocabulary by (count desc, key asc) in ONE stable counting pass (nvt_vocab_col.src_keys).
 *      hot_image must then be the column's own int32[NVT_RANGE_AUX_WORDS].  Ranges are derived
 *      from the sampled min / max: keys that are not spread over their range overflow a region
 *      or a table (overflow bit0) and the column is rerun on a hash path.
 * weights (optional, int64 per row) turns the count into a weighted sum -- the tree-merge
 * of (key,count) lists (_mid_level_groupby).  ws: device scratch of
 * nvt_dense_count_ws_bytes().  state (device uint64[NVT_STATE_WORDS], written):
 * [NVT_ST_NULLS] null rows (weighted), [NVT_ST_SENTINEL] rows whose key is the empty
 * sentinel (NOT in the list), [NVT_ST_OCCUPIED] entries written, [NVT_ST_MAXCOUNT] the
 * largest count, [NVT_ST_OVERFLOW] bit0: an LDS table filled up (rerun on a larger path),
 * bit1: out_capacity too small.  The output list is in no particular order. */
#define NVT_PATH_HOT 16
#define NVT_HOT_IMAGE_WORDS 8192
#define NVT_PATH_RANGE 9
#define NVT_PATH_PIECES 0x10000     /* with NVT_PATH_RANGE: let the sample decide on a piecewise map
                                       (sampled quantile splitters) -- for keys that are not spread
                                       over their range; asked for after the linear map overflowed */
#define NVT_PATH_SORT 10            /* int32 keys, no weights, any number of distinct keys: radix sort of
                                       the rows + run lengths; key-sorted output like the range path;
                                       hot_image = uint32[256] receiving the histogram of min(count, 255) */
#define NVT_RANGE_WGS 256           /* partition workgroups = runs per bucket                 */
#define NVT_RANGE_AUX_LO 8192       /* aux words behind the hot image: range origin (biased),
                                       span, multiplier, 0, pre-shift of the monotone map (nvt_range.hpp) */
#define NVT_RANGE_AUX_HIST 8208     /*   uint32[256] histogram of min(count, 255)             */
#define NVT_RANGE_AUX_HOTSTART 8464 /*   uint32[1025]: hot image slots by bucket (CSR offsets) */
#define NVT_RANGE_AUX_HOTORDER 9504 /*   uint16[8192]: the image slots in bucket order         */
#define NVT_RANGE_AUX_PW 13600      /*   piecewise map (keys not spread over their range): u32
                                       splitters[65], mul[64], shift flags[2]; word LO + 7 = fine
                                       slots per piece (0: the linear map)                      */
#define NVT_RANGE_AUX_WORDS (13600 + 256)
int nvt_dense_count_ws_bytes(int key_bytes, uint64_t n, int path, int weighted, uint64_t *bytes);
int nvt_dense_count_i32(const int32_t *keys, const uint8_t *valid, const int64_t *weights,
                        uint64_t n, int path, void *ws, int32_t *out_keys, int64_t *out_counts,
                        uint64_t out_capacity, uint64_t *state, void *stream);
int nvt_dense_count_i64(const int64_t *keys, const uint8_t *valid, const int64_t *weights,
                        uint64_t n, int path, void *ws, int64_t *out_keys, int64_t *out_counts,
                        uint64_t out_capacity, uint64_t *state, void *stream);

/* ---- vocabulary order (_write_uniques): count descending, key ascending ----
 * LSD radix sort of n (key,count) pairs; tmp must hold nvt_vocab_sort_tmp_bytes(). */
int nvt_vocab_sort_tmp_bytes(int key_bytes, uint64_t n, uint64_t *bytes);
/* max_count: an upper bound on counts[] (e.g. state[NVT_ST_MAXCOUNT]) lets the sort pick
 * its count passes without reading anything back; <= 0 = unknown (the sort then builds
 * per-pass histograms and synchronises the stream once). */
int nvt_vocab_sort_i32(int32_t *keys, int64_t *counts, uint64_t n, int64_t max_count, void *tmp,
                       void *stream);
int nvt_vocab_sort_i64(int64_t *keys, int64_t *counts, uint64_t n, int64_t max_count, void *tmp,
                       void *stream);

/* ---- Categorify.transform (_encode) ----
 * encode table slot: i32 = {int32 key, int32 label}; i64 = {int64 key, int64 label}.
 * Build assigns label first_label + i to vocab_keys[i]. */
int nvt_encode_table_bytes(int key_bytes, uint64_t capacity, uint64_t *bytes);
/* Diagnostic of the cache-mode encode (int32 keys whose vocabulary exceeds the LDS head): with
 * NVT_ENC_STATS=1 in the environment the kernels count out2[0] = rows that went on to the table in
 * HBM and out2[1] = rows looked up, since the last reset (synchronises the stream). */
int nvt_encode_stats(uint64_t *out2, int reset, void *stream);
/* unique_keys != 0 promises vocab_keys holds no duplicates (true for fitted vocabularies):
 * int32 slots are then claimed and filled with one 64-bit CAS.  With duplicates (user
 * vocabs) the lowest label wins. */
int nvt_encode_build_i32(const int32_t *vocab_keys, uint64_t n_vocab, int64_t first_label,
                         void *table, uint64_t capacity, int64_t *sentinel_label, int unique_keys,
                         void *stream);
int nvt_encode_build_i64(const int64_t *vocab_keys, uint64_t n_vocab, int64_t first_label,
                         void *table, uint64_t capacity, int64_t *sentinel_label, int unique_keys,
                         void *stream);
/* out[i] = null_label            if key i is null
 *        = table[key]            if present
 *        = oov_label             if absent and num_buckets <= 1
 *        = oov_label + h32(key) % num_buckets   otherwise      (out_bytes: 4 or 8)
 * vocab_keys / n_vocab / first_label (optional: pass NULL, 0, 0): the ordered, duplicate-free
 * vocabulary the table was built from.  Its head -- the most frequent keys -- is then
 * staged in LDS by every workgroup and only rows that miss it probe the table in HBM.
 * A vocabulary of at most NVT_ENCODE_RESIDENT_I32 (8192) int32 / NVT_ENCODE_RESIDENT_I64
 * (6144) int64 keys is staged in full: table and sentinel_label may then be NULL and
 * nvt_encode_build_* need not be called at all. */
#define NVT_ENCODE_RESIDENT_I32 8192
#define NVT_ENCODE_RESIDENT_I64 6144
int nvt_encode_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, const void *table,
                   uint64_t capacity, const int64_t *sentinel_label, int64_t null_label,
                   int64_t oov_label, uint32_t num_buckets, void *out, int out_bytes,
                   const int32_t *vocab_keys, uint64_t n_vocab, int64_t first_label, void *stream);
int nvt_encode_i64(const int64_t *keys, const uint8_t *valid, uint64_t n, const void *table,
                   uint64_t capacity, const int64_t *sentinel_label, int64_t null_label,
                   int64_t oov_label, uint32_t num_buckets, void *out, int out_bytes,
                   const int64_t *vocab_keys, uint64_t n_vocab, int64_t first_label, void *stream);

/* ---- HashBucket / hashed OOV buckets: out[i] = h32(key) % num_buckets (int32);
 * a null row (valid bit clear) hashes as key 0 whatever bytes its slot holds (Arrow leaves the
 * values under a null undefined; the pandas path hashes fillna(0)).
 * xor_in (optional, uint64 per row) is XORed into the 64-bit hash first and
 * xor_out (optional) receives the 64-bit hash -- the "combo" XOR chain of
 * categorify.py:1846-1851 */
int nvt_hash_bucket_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, uint32_t num_buckets,
                        int32_t *out, const uint64_t *xor_in, uint64_t *xor_out, void *stream);
int nvt_hash_bucket_i64(const int64_t *keys, const uint8_t *valid, uint64_t n, uint32_t num_buckets,
                        int32_t *out, const uint64_t *xor_in, uint64_t *xor_out, void *stream);

/* ---- Normalize.fit (_chunkwise_moments): out[3] (double, device) += {count, sum, sum of
 * squares} over non-null rows; with has_fill, nulls count as fill_val (FillMissing
 * upstream of Normalize).  partials: device scratch of nvt_moments_scratch_bytes(). */
uint64_t nvt_moments_scratch_bytes(void);
int nvt_moments(const void *x, int dtype, const uint8_t *valid, uint64_t n, int has_fill,
                double fill_val, double *out3, void *partials, void *stream);
/* NormalizeMinMax.fit: out2 = {min, max} over non-null rows (NaN if none), merged with
 * the values already in out2 when accumulate != 0 */
int nvt_minmax(const void *x, int dtype, const uint8_t *valid, uint64_t n, int accumulate,
               double *out2, void *partials, void *stream);

/* ---- FillMissing + Normalize.transform, fused:
 *   v      = isnull(x[i]) ? (has_fill ? fill_val : NaN) : x[i]
 *   out[i] = do_norm ? (scale > 0 ? (v - shift) / scale : v - shift) : v
 * out dtype NVT_F32 / NVT_F64 (or the input dtype when do_norm == 0 and ints are
 * filled); filled (optional, uint8 0/1 per row) receives isnull(x[i]) -- the
 * `<col>_filled` column of FillMissing(add_binary_cols=True). */
int nvt_fill_normalize(const void *x, int dtype, const uint8_t *valid, uint64_t n, int has_fill,
                       double fill_val, int do_norm, double shift, double scale, void *out,
                       int out_dtype, uint8_t *filled, void *stream);

/* ---- Clip + LogOp (the reference benchmark's default continuous branch,
 * bench/examples/dask-nvtabular-criteo-benchmark.py:201-204; clip.py:49-55, logop.py:43-53),
 * fused with a pending FillMissing constant:
 *   v = isnull(x) ? (has_fill ? fill_val : null) : x;  v = clamp(v, vmin, vmax);
 *   out = do_log ? logf((float)v + 1) : v          (nulls stay null; NaN for float outputs)
 * out_dtype: NVT_F32 / NVT_F64, or the input dtype for integer clipping without log. */
int nvt_clip_log(const void *x, int dtype, const uint8_t *valid, uint64_t n, int has_fill,
                 double fill_val, int has_min, double vmin, int has_max, double vmax, int do_log,
                 void *out, int out_dtype, void *stream);

/* ---- Bucketize (bucketize.py:76-94): out[i] = np.digitize(x[i], boundaries, right=False),
 * i.e. the number of (ascending, float64, device) boundaries <= x[i]; null / NaN rows get
 * n_boundaries (numpy's answer for NaN).  n_boundaries <= 8192. */
int nvt_bucketize(const void *x, int dtype, const uint8_t *valid, uint64_t n,
                  const double *boundaries, int n_boundaries, int32_t *out, void *stream);

/* ---- JoinGroupby / TargetEncoding / combo-Categorify: multi-key groupby tables ----
 * A table over nkeys (1..3) key columns (each int64 after widening; null components
 * allowed and form their own groups, pandas dropna=False) and nvals (0..8) value
 * columns.  Accumulators per group: size (all rows), count (rows whose FIRST key
 * component is non-null -- categorify.py:995-999), and per value column
 * sum / sum of squares / min / max over its non-null entries. */
typedef struct nvt_gb_table nvt_gb_table;
#define NVT_GB_SUMSQ 1   /* keep sum of squares (std / var requested)  */
#define NVT_GB_MINMAX 2  /* keep min / max                             */
/* The table object owns its device arrays (hipMalloc'ed on the current device). */
int nvt_gb_create(int nkeys, int nvals, int flags, uint64_t capacity, nvt_gb_table **out);
void nvt_gb_destroy(nvt_gb_table *t);
/* The same table in CALLER-provided device memory (nvt_gb_table_bytes() bytes): hipMalloc and
 * hipFree synchronise the device, which a fit that builds a table per partition cannot
 * afford.  nvt_gb_destroy then only frees the handle. */
int nvt_gb_table_bytes(int nkeys, int nvals, int flags, uint64_t capacity, uint64_t *bytes);
int nvt_gb_create_in(int nkeys, int nvals, int flags, uint64_t capacity, void *memory,
                     uint64_t bytes, nvt_gb_table **out);
/* device address of the table's uint64[NVT_STATE_WORDS] state block (read it back with
 * nvt_mailbox_post instead of the blocking nvt_gb_state) */
uint64_t *nvt_gb_state_ptr(nvt_gb_table *t);
/* nvt_gb_update orders the rows by group (assign slot -> radix sort -> segmented reduce; no
 * per-row accumulator atomics) and needs nvt_gb_update_ws_bytes(n) bytes of scratch: owned and
 * grown by the table unless the caller hands one over with nvt_gb_set_workspace. */
int nvt_gb_update_ws_bytes(uint64_t n, uint64_t *bytes);
int nvt_gb_set_workspace(nvt_gb_table *t, void *ws, uint64_t bytes);
int nvt_gb_clear(nvt_gb_table *t, void *stream);
/* keys[k]: int64 device column, key_valid[k]: bitmap or NULL; vals[j]: column of vdtype[j] */
int nvt_gb_update(nvt_gb_table *t, const int64_t *const *keys, const uint8_t *const *key_valid,
                  const void *const *vals, const int *vdtypes, const uint8_t *const *val_valid,
                  uint64_t n, void *stream);
/* merge another table's groups into t (tree reduce / multi-GPU owner merge) given its
 * compacted columns */
int nvt_gb_merge(nvt_gb_table *t, const int64_t *const *keys, const uint8_t *key_null_mask,
                 const int64_t *size, const int64_t *count, const double *const *sum,
                 const double *const *sumsq, const double *const *vmin,
                 const double *const *vmax, uint64_t n, void *stream);
/* host-visible occupancy / overflow: state[NVT_STATE_WORDS] copied to host memory */
int nvt_gb_state(nvt_gb_table *t, uint64_t *host_state, void *stream);
/* compact groups: out_keys[k][g], out_null_mask[g] (bit k = component k is null),
 * out_size/out_count[g], out_sum/sumsq/min/max[j][g]; any out pointer may be NULL.
 * *out_n (device) = number of groups */
int nvt_gb_compact(nvt_gb_table *t, int64_t *const *out_keys, uint8_t *out_null_mask,
                   int64_t *out_size, int64_t *out_count, double *const *out_sum,
                   double *const *out_sumsq, double *const *out_min, double *const *out_max,
                   uint64_t *out_n, void *stream);
/* transform side: build a lookup table over compacted group keys, then map rows to
 * group ids (row i -> index into the compacted arrays, -1 when unseen) */
int nvt_gb_index_build(nvt_gb_table *t, const int64_t *const *keys, const uint8_t *key_null_mask,
                       uint64_t n_groups, void *stream);
int nvt_gb_lookup(nvt_gb_table *t, const int64_t *const *keys, const uint8_t *const *key_valid,
                  uint64_t n, int64_t *out_group, void *stream);
/* ---- Groupby operator (groupby.py:236-263): row order by (group, sort columns) + aggregates.
 * nvt_sort_key_u64: order-preserving 64-bit image of a column (nulls / NaN last in either
 * direction, pandas na_position="last").  nvt_order_rows: stable refinement of a row order by
 * such a key (two 32-bit radix passes) or by group id (-1 = null key, sorted last); perm
 * entries are 64-bit words whose LOW 32 bits are the row index (after a gid refinement the high
 * half is the group id: exactly the `words` nvt_seg_aggregate takes).  nvt_seg_aggregate:
 * out_size uint64[ngroups] rows per group, out_count uint64[nvals][ngroups] non-null values per
 * column, out_sum / out_sumsq / out_min / out_max double[nvals][ngroups] (any of the last
 * three pairs may be NULL) in one wave-level segmented reduction; outputs pre-initialised by
 * the caller (0 / 0 / 0 / 0 / +inf / -inf). */
int nvt_sort_key_u64(const void *x, int dtype, const uint8_t *valid, uint64_t n, int ascending,
                     uint64_t *out, void *stream);
int nvt_order_rows_ws_bytes(uint64_t n, uint64_t *bytes);
int nvt_order_rows(const uint64_t *key64, const int64_t *gid, uint64_t ngroups,
                   const uint64_t *perm_in, uint64_t n, uint64_t *perm_out, void *ws, void *stream);
int nvt_seg_aggregate(const uint64_t *words, uint64_t n, uint64_t ngroups, const void *const *vals,
                      const int *vdtypes, const uint8_t *const *val_valid, int nvals,
                      uint64_t *out_size, uint64_t *out_count, double *out_sum, double *out_sumsq,
                      double *out_min, double *out_max, void *stream);
/* out[i] = group[i] >= 0 ? src[group[i]] : miss   (stat columns joined back onto rows) */
int nvt_gather_f64(const double *src, const int64_t *group, uint64_t n, double miss, void *out,
                   int out_dtype, void *stream);
/* TargetEncoding (target_encoding.py:341-363): with g = group_all[i], f = group_fold[i]
 *   out[i] = (g < 0 || f < 0) ? y_mean     (the reference's unmatched left merges)
 *          : (sum_all[g] - sum_fold[f] + p*y_mean) / (cnt_all[g] - cnt_fold[f] + p)
 * -> float32/float64.  group_fold == NULL means kfold <= 1: no fold terms, only g decides. */
int nvt_te_apply(const int64_t *group_all, const int64_t *group_fold, const double *sum_all,
                 const int64_t *cnt_all, const double *sum_fold, const int64_t *cnt_fold,
                 uint64_t n, double p_smooth, double y_mean, void *out, int out_dtype,
                 void *stream);

/* the same with the DENSE fold statistics of nvt_sgb_reduce: f = g * kfold + fold[i], a pair
 * with cnt_fold[f] == 0 has no rows (unmatched merge: y_mean). */
int nvt_te_apply_folds(const int64_t *group_all, const uint8_t *fold, int kfold,
                       const double *sum_all, const int64_t *cnt_all, const double *sum_fold,
                       const int64_t *cnt_fold, uint64_t n, double p_smooth, double y_mean,
                       void *out, int out_dtype, void *stream);

/* ---- ONE int32 key column without nulls: groupby-aggregate by sorting ------------------
 * (categorify.py:955-1137 _top_level_groupby .. _bottom_level_groupby for JoinGroupby /
 * TargetEncoding, join_groupby.py:150-173, target_encoding.py:254-299.)  The rows are radix-
 * sorted by (key, fold), the groups come out DENSE and ordered by key:
 *   out_keys / out_keys32 [cap]            the group keys, ascending
 *   out_size  uint64[cap * kfold]          rows of (group g, fold f) at g * kfold + f
 *   out_sum / out_sumsq / out_min / out_max  double[nvals][cap * kfold] (NaN / null values
 *                                          skipped; min / max of an empty entry stay +-inf;
 *                                          the last three per flags NVT_GB_SUMSQ / _MINMAX)
 *   tot_size uint64[cap], tot_sum double[nvals][cap]   (kfold > 1 only) sums over the folds
 *   te_records double[nvals][cap][2 * (kfold + 1)] or NULL (kfold > 1 only): per group
 *                                          {sum, count, (sum_f, count_f) for every fold}, the
 *                                          layout nvt_flat_lookup_te reads with one probe
 * Keys: an int32 column (key_bias = INT32_MIN) or an int64 column whose keys span less than
 * 2^32 (key_bias = the smallest key; nvt_key_minmax writes {min, max} to device memory); the
 * 32-bit key image is key - key_bias.  out_keys hold the keys themselves, out_keys32 =
 * key - key_bias - 2^31 (what the flat index stores: pass key_offset = key_bias + 2^31 to the
 * nvt_flat_lookup* functions; 0 for int32 columns).
 * nvt_sgb_sort: words (key image << 32 | fold << row_bits | row) sorted on bits [row_bits, 64);
 * fold: uint8[n] ids < kfold, NULL with kfold == 1; n <= 2^(32 - bits(kfold - 1)), n < 2^30.
 * *sorted_out = device pointer INSIDE ws (nvt_sgb_sort_ws_bytes(n) bytes; keep ws alive),
 * *row_bits_out = 32 - bits(kfold - 1).  The sorted words serve every aggregate on the same key
 * column of the same rows (JoinGroupby after TargetEncoding ignores the fold bits: kfold = 1).
 * nvt_sgb_regroup: run heads of the sorted words -> group ids (out_keys / out_keys32 [cap]) and
 * the words rewritten as ((g * kfold + fold) << 32 | row) into regrouped[n] (not aliasing
 * sorted); cap * kfold < 2^32 - 1.  state[NVT_ST_OCCUPIED] = groups found; more than cap:
 * only the first cap groups are kept (their rows carry slot 0xFFFFFFFF) and
 * state[NVT_ST_NEED] = the capacity a second call needs.  ws: nvt_sgb_regroup_ws_bytes(n).
 * nvt_sgb_reduce: the statistics of the regrouped words (the arrays above; they are
 * initialised here).  kfold = words_kfold: per-(group, fold) entries + totals (+ te_records);
 * kfold = 1 with words_kfold > 1: per-group entries from words regrouped for ANOTHER
 * aggregate's folds (JoinGroupby after TargetEncoding on the same column: one sort, one
 * regroup).  state: the block nvt_sgb_regroup filled (device, read by the kernels only).
 * sorted_out[j] (array may be NULL, entries may be NULL): value column j in the order of the
 * words, in its own dtype [n] -- written while the column is gathered by row; sorted_in[j]: such
 * an array from an earlier aggregate on the same words, read streaming INSTEAD of the gather by
 * row (columns without validity bitmap, not int64).  No host synchronisation anywhere. */
int nvt_sgb_sort_ws_bytes(uint64_t n, uint64_t *bytes);
int nvt_key_minmax(const void *keys, int key_dtype, uint64_t n, int64_t *out2, void *stream);
int nvt_sgb_sort(const void *keys, int key_dtype, int64_t key_bias, const uint8_t *fold, int kfold,
                 uint64_t n, void *ws, uint64_t **sorted_out, int *row_bits_out, void *stream);
int nvt_sgb_regroup_ws_bytes(uint64_t n, uint64_t *bytes);
int nvt_sgb_regroup(const uint64_t *sorted, int row_bits, int kfold, int64_t key_bias, uint64_t n,
                    uint64_t cap, int64_t *out_keys, int32_t *out_keys32, uint64_t *regrouped,
                    uint64_t *state, void *ws, void *stream);
int nvt_sgb_reduce(const uint64_t *regrouped, int words_kfold, int kfold, const void *const *vals,
                   const int *vdtypes, const uint8_t *const *val_valid, int nvals, int flags,
                   uint64_t n, uint64_t cap, uint64_t *out_size, double *out_sum, double *out_sumsq,
                   double *out_min, double *out_max, uint64_t *tot_size, double *tot_sum,
                   double *te_records, const uint64_t *state, const void *const *sorted_in,
                   void *const *sorted_out, void *stream);
/* Owner-side merge of the (key, count) rows of the multi-GPU exchange by sorting
 * (categorify.py:1054-1070 _mid_level_groupby, done on the owner rank): rows = n words
 * (count << 32 | int32 key), 0 <= count < 2^31, in nseg segments [seg_off[s], seg_off[s + 1])
 * (device uint64[nseg + 1]; source-major, column-minor: segment s holds column s % ncol).
 * Output: the (column, key) groups ordered by column, then key: out_keys int32[n], out_col
 * int64[n], out_sum double[n] (summed counts: exact below 2^53); state[NVT_ST_OCCUPIED] =
 * groups.  n < 2^26, ncol <= 64.  ws: nvt_count_merge_sorted_ws_bytes(n).  No host sync. */
int nvt_count_merge_sorted_ws_bytes(uint64_t n, uint64_t *bytes);
int nvt_count_merge_sorted(const int64_t *rows, uint64_t n, const uint64_t *seg_off, int nseg,
                           int ncol, int32_t *out_keys, int64_t *out_col, double *out_sum,
                           uint64_t *state, void *ws, void *stream);
/* Seeded TargetEncoding folds (target_encoding.py:427-439: numpy.random.RandomState(seed)
 * .choice(arange(kfold), n)) on the device: MT19937 seeded by init_genrand(seed), 32-bit draws
 * masked to the next 2^k - 1 and rejected while >= kfold -- out[i] (uint8) = the i-th accepted
 * value, bit for bit numpy's sequence.  One workgroup walks the generator (it is sequential);
 * kfold <= 128. */
int nvt_fold_mt19937(uint32_t seed, int kfold, uint64_t n, uint8_t *out, void *stream);
/* The same values from chunks of 2^18 draws generated in parallel (jump-ahead polynomials of the
 * generator's characteristic polynomial, tools/mt_jump_polys.py): ws = nvt_fold_mt19937_par_ws_bytes
 * bytes of device scratch; *total_out (device word) receives the accepted values the chunks held --
 * the caller checks total >= n (the chunk count carries an 8-sigma margin) and falls back to
 * nvt_fold_mt19937 otherwise. */
int nvt_fold_mt19937_par_ws_bytes(uint64_t n, int kfold, uint64_t *bytes);
int nvt_fold_mt19937_par(uint32_t seed, int kfold, uint64_t n, uint8_t *out, void *ws, uint64_t ws_bytes,
                         uint64_t *total_out, void *stream);
/* Distinct keys of the first n rows of every column, ESTIMATED (HyperLogLog, 4096 registers: 1.6 %
 * standard error, small counts by linear counting), and the valid rows among them: what steers
 * the first counting path of a fit that has no cardinality hints (the role of the reference's
 * `cat_cache` / `tree_width` heuristics around categorify.py:1423-1478: sizing, never results).
 * out uint64[ncols][2] = {estimate, valid rows} (device).  One launch for all columns. */
typedef struct nvt_prefix_col {
  const void *keys;       /* int32 (key_bytes 4) or int64 (8) */
  const uint8_t *valid;   /* Arrow validity bitmap or NULL */
  uint64_t n;             /* rows to look at */
  int key_bytes;
} nvt_prefix_col;
int nvt_prefix_distinct(const nvt_prefix_col *cols, int ncols, uint64_t *out, void *stream);
/* ---- tree merge of KEY-SORTED partial results (multi-partition fit) ------------------------
 * Replaces the concat + re-groupby of _mid_level_groupby (categorify.py:1054-1070) inside the
 * tree of categorify.py:1423-1478 -- and the same tree under join_groupby.py:140-173 /
 * target_encoding.py:171-214 -- for partial results that are already ORDERED BY KEY (range path,
 * sort path, sort-path groupby): out = the union of the ascending, duplicate-free int32 key
 * lists A and B in key order; the counts of a key that occurs in both are summed.  out_keys /
 * out_counts hold na + nb entries, *out_n (device) receives the merged length.  counts are
 * optional (both NULL: keys only).  src_a / src_b (optional, int32[na + nb], on every column of
 * a call or on none): position in A / in B every output entry came from, -1 = none -- the map
 * nvt_merge_payload combines further payload arrays with.  na + nb < 2^31.  One merge-path
 * search launch + one tile launch for ALL columns of the call; ws:
 * nvt_merge_sorted_ws_bytes(cols, ncols) bytes, 16-byte aligned.  No host synchronisation. */
typedef struct nvt_merge_col {
  const int32_t *a_keys;
  const int64_t *a_counts;
  uint64_t na;
  const int32_t *b_keys;
  const int64_t *b_counts;
  uint64_t nb;
  int32_t *out_keys;
  int64_t *out_counts;
  int32_t *src_a;
  int32_t *src_b;
  uint64_t *out_n;
} nvt_merge_col;
int nvt_merge_sorted_ws_bytes(const nvt_merge_col *cols, int ncols, uint64_t *bytes);
int nvt_merge_sorted_many(const nvt_merge_col *cols, int ncols, void *ws, uint64_t ws_bytes,
                          void *stream);
/* out[i * width + j] = op(a[src_a[i] * width + j], b[src_b[i] * width + j]), a side whose index
 * is -1 contributes nothing.  dtype NVT_I64 / NVT_F64; op 0 add, 1 min, 2 max (NaN = no value). */
int nvt_merge_payload(const int32_t *src_a, const int32_t *src_b, uint64_t n, int width, int dtype,
                      int op, const void *a, const void *b, void *out, void *stream);
/* ---- multi-GPU vocabulary exchange: the device work around the collectives (the
 * reference's tree reduce of per-partition frames, categorify.py:1423-1529).  One launch per step
 * for ALL columns of a fit (<= 64 columns of int32 keys + int64 counts, ranks x columns <= 4096):
 *   nvt_exchange_ranges   rng int64[ncol][3] = {-min key, max key, sum of counts}; a column
 *                         without entries: {-INT64_MAX, -INT64_MAX, 0} (one MAX all-reduce follows)
 *   nvt_exchange_hist     send_mat uint64[G][ncol] = rows of column j owned by rank g, with
 *                         owner(key) = min((key - lo[j]) / width[j], G - 1) (key ranges)
 *   nvt_exchange_scatter  rows_out = (count << 32 | key) words grouped by (owner, column);
 *                         cursors uint64[G][ncol] (device) = the start of every group, advanced by
 *                         the call; the order inside a group is unspecified (counts < 2^31)
 *   nvt_exchange_pack_ordered  the same send buffer for KEY-SORTED lists: a group is a contiguous
 *                         slice of its column, row r of the column goes to start[cell] + (r -
 *                         first_row[cell]) -- every group arrives in key order and the owner merges
 *                         sorted runs (nvt_merge_sorted_many) instead of sorting; first_row / start:
 *                         device uint64[G][ncol]
 *   nvt_exchange_unpack   n gathered words in nseg segments [seg_off[s], seg_off[s + 1]) ->
 *                         keys_out / counts_out at dst_off[s] + (position in the segment)
 * lo / width: HOST arrays; seg_off / dst_off: device arrays.  No host synchronisation. */
typedef struct nvt_xcol {
  const int32_t *keys;
  const int64_t *counts;
  uint64_t n;
} nvt_xcol;
int nvt_exchange_ranges(const nvt_xcol *cols, int ncol, int64_t *rng, void *stream);
/* the same for KEY-SORTED lists: {-first key, last key, 0} (no pass over the lists; the caller
 * supplies its own bound of the rows counted) */
int nvt_exchange_ranges_sorted(const nvt_xcol *cols, int ncol, int64_t *rng, void *stream);
int nvt_exchange_hist(const nvt_xcol *cols, int ncol, const int64_t *lo, const uint64_t *width, int G,
                      uint64_t *send_mat, void *stream);
int nvt_exchange_scatter(const nvt_xcol *cols, int ncol, const int64_t *lo, const uint64_t *width, int G,
                         uint64_t *cursors, int64_t *rows_out, void *stream);
int nvt_exchange_pack_ordered(const nvt_xcol *cols, int ncol, const int64_t *lo, const uint64_t *width,
                              int G, const uint64_t *first_row, const uint64_t *start, int64_t *rows_out,
                              void *stream);
int nvt_exchange_unpack(const int64_t *words, uint64_t n, const uint64_t *seg_off, const uint64_t *dst_off,
                        int nseg, int32_t *keys_out, int64_t *counts_out, void *stream);
/* nvt_exchange_unpack with one int32 of payload per word (extra[i] travels with words[i]: the
 * labels of the distributed vocabulary ordering, nvt_vocab_label_shard) */
int nvt_exchange_unpack2(const int64_t *words, const int32_t *extra, uint64_t n, const uint64_t *seg_off,
                         const uint64_t *dst_off, int nseg, int32_t *keys_out, int64_t *counts_out,
                         int32_t *extra_out, void *stream);
/* key -> position in an ascending int32 key list (the group ids of nvt_sgb_regroup) through
 * a flat range table laid out from the list in one pass (no inserts): replaces
 * nvt_gb_index_build + nvt_gb_lookup for such groups (join_groupby.py:198-203,
 * target_encoding.py:350-371).  aux: int32[NVT_FLAT_AUX_WORDS]; table: capacity 8-byte slots,
 * capacity >= slots + n + 64 (slots: home slots, any count from 64 to 2^32 - 1); tmp:
 * nvt_flat_index_tmp_bytes(n).
 * aux[NVT_FLAT_AUX_MAXDISP] = longest displacement (keys clustered in their range make long
 * probe runs: the caller may prefer a hashed index).  nvt_flat_lookup: out[i] = position of
 * keys[i] - key_offset (int32 / int64 column, optional validity bitmap) or -1. */
int nvt_flat_index_tmp_bytes(uint64_t n, uint64_t *bytes);
int nvt_flat_index_build(const int32_t *keys, uint64_t n, uint64_t slots, int32_t *aux, void *table,
                         uint64_t capacity, void *tmp, void *stream);
int nvt_flat_lookup(const void *keys, int dtype, const uint8_t *valid, uint64_t n, const int32_t *aux,
                    const void *table, uint64_t capacity, int64_t key_offset, int64_t *out,
                    void *stream);
/* JoinGroupby.transform in one pass (join_groupby.py:198-217): outs[c][i] = records[g][c] with
 * g = the group of keys[i], miss[c] when the key has no group; records double[groups][ncols]
 * (ncols <= 16), out_dtypes f32 / f64 / i32 / i64 (value-converting stores).  *unseen (device
 * word, may be NULL) is OR-ed with 1 when any row had no group -- the reference's
 * astype(int32) of a count column raises then (join_groupby.py:214). */
int nvt_flat_lookup_gather(const void *keys, int dtype, const uint8_t *valid, uint64_t n,
                           const int32_t *aux, const void *table, uint64_t capacity,
                           int64_t key_offset, const double *records, int ncols, void *const *outs,
                           const int *out_dtypes, const double *miss, uint64_t *unseen, void *stream);
/* TargetEncoding.transform in one pass (target_encoding.py:341-371): probe + nvt_te_apply(_folds)
 * on records double[groups][2 * (kfold + 1)] = {sum, count, (sum_f, count_f) ...} (te_records
 * of nvt_sgb_reduce); fold == NULL (kfold 1): records double[groups][2] = {sum, count}. */
int nvt_flat_lookup_te(const void *keys, int dtype, const uint8_t *valid, uint64_t n, const int32_t *aux,
                       const void *table, uint64_t capacity, int64_t key_offset, const uint8_t *fold,
                       int kfold, const double *records, double p_smooth, double y_mean, void *out,
                       int out_dtype, void *stream);

/* Lookup images: ONE probe and ONE packed record per row for every operator on a key column
 * (JoinGroupby.transform join_groupby.py:198-217 + TargetEncoding.transform
 * target_encoding.py:341-371 on the same key are two left merges on that key in the reference).
 * image = bytes[groups][stride_bytes] (stride a multiple of 8): every operator owns a byte range
 * of the record and stores there what a row receives, already in the OUTPUT dtype.
 * nvt_flat_lookup_image: g = gid_in[i] when gid_in != NULL, else the flat-index probe of keys[i];
 * gid_out (may be NULL) receives g as int32 (-1: no group).  Output c copies `sizes[c]` (4 / 8)
 * bytes from image[g][offs[c] + (folds[c] ? (1 + folds[c][i]) * sizes[c] : 0)] to outs[c][i];
 * rows without group receive miss_bits[c].  *unseen (may be NULL) is OR-ed with 1 when any row had
 * no group.  ncols <= 24.
 * nvt_image_pack: image[g][offs[c]] = (dst dtype) src[c][g], src float64 / int64 arrays [groups].
 * nvt_te_image: (kfold + 1) values per group at image[g][off + slot * size]: slot 0 =
 * (sum + p*mean) / (count + p), slot 1 + f = the out-of-fold value of fold f (mean when the
 * (group, fold) pair has no rows), from the totals tot_count / tot_sum [groups] and the dense fold
 * statistics fold_count / fold_sum [groups * kfold] of nvt_sgb_reduce (kfold = 0: totals only)
 * -- the expression nvt_te_apply_folds evaluates per row, evaluated once per (group, fold):
 * identical bits. */
int nvt_flat_lookup_image(const void *keys, int dtype, const uint8_t *valid, uint64_t n,
                          const int32_t *aux, const void *table, uint64_t capacity, int64_t key_offset,
                          const int32_t *gid_in, int32_t *gid_out, const void *image,
                          uint32_t stride_bytes, int ncols, void *const *outs,
                          const uint8_t *const *folds, const uint32_t *offs, const uint32_t *sizes,
                          const uint64_t *miss_bits, uint64_t *unseen, void *stream);
int nvt_image_pack(const void *const *src, const int *src_dtypes, const int *dst_dtypes,
                   const uint32_t *offs, int ncols, uint64_t groups, void *image,
                   uint32_t stride_bytes, void *stream);
/* JoinGroupby's values from the fit's accumulators (count int64 [groups]; sum / sumsq / mn / mx:
 * arrays of nvals float64 [groups] pointers, NULL arrays / entries when no requested statistic
 * needs them): output c = statistic kinds[c] (0 count, 1 sum, 2 mean, 3 min, 4 max, 5 var, 6 std;
 * categorify.py:1087-1131) of value column vals[c], stored as dst_dtypes[c] at image[g][offs[c]]. */
int nvt_jg_image(const int64_t *count, const double *const *sum, const double *const *sumsq,
                 const double *const *mn, const double *const *mx, int nvals, const int *kinds,
                 const int *vals, const int *dst_dtypes, const uint32_t *offs, int ncols, uint64_t groups,
                 void *image, uint32_t stride_bytes, void *stream);
int nvt_te_image(const int64_t *tot_count, const double *tot_sum, const int64_t *fold_count,
                 const double *fold_sum, int kfold, uint64_t groups, double p_smooth, double y_mean,
                 int out_dtype, void *image, uint32_t stride_bytes, uint32_t off, void *stream);

/* ---- key directory: the key -> group step in ONE 16-byte read -----------------------------------
 * (the same left merges, join_groupby.py:198-217 / target_encoding.py:341-371.)  The flat-index
 * probe above walks {key, group} slots across 64-byte lines: ~2.4 L2 misses per row with the
 * record, and the kernel runs at the rate the fabric takes misses (tools/pmc_lookup.sh).  Here the
 * groups' keys are ascending (group g = position g of keys32) and
 *   dir[b] = uint32[4] {first, k0, k1, k2}, b = 0 .. dir_slots : first = index of the first key
 *            that maps to a bucket >= b under the monotone range map over [keys32[0],
 *            keys32[nkeys - 1]] (dir[dir_slots].first = nkeys); k0..k2 = the bucket's first three
 *            keys (a shorter bucket repeats its last key, an empty one holds the list's next key);
 * a row whose key is k0 / k1 / k2 has its group from that read, a key above k2 walks keys32 from
 * first + 3 (bisection inside the bucket when the keys cluster).  No parameter block, no failure
 * mode to read back.
 * nvt_keydir_build: dir (uint32[4 * (dir_slots + 1)], 16-byte aligned) from the ascending
 * duplicate-free key list.
 * nvt_keydir_lookup_image: nvt_flat_lookup_image's image / outputs / folds / offs / sizes /
 * miss_bits / unseen; keys int32 / int64 (list key = column key - key_offset), rows with a null
 * key take record `null_group` (-1: they miss).
 * nvt_image_build: records [records][stride_bytes] (stride <= 192) written WHOLE in one pass: up
 * to 4 operators' byte ranges, each evaluated exactly as nvt_jg_image / nvt_te_image evaluate it
 * (kind NVT_IMAGE_PART_JG: count / sum / sumsq / mn / mx / nvals / kinds / vals / dst_dtypes /
 * offs / ncols as in nvt_jg_image; NVT_IMAGE_PART_TE: tot_* / fold_* / kfold / p_smooth / y_mean
 * / out_dtype / offset as in nvt_te_image); a part fills its first `groups` records, every other
 * byte of the image is zero. */
#define NVT_IMAGE_PART_JG 0
#define NVT_IMAGE_PART_TE 1
typedef struct nvt_image_part {
  int32_t kind;
  int32_t nvals;               /* JG */
  int32_t ncols;               /* JG */
  int32_t kfold;               /* TE */
  int32_t out_dtype;           /* TE: NVT_F32 / NVT_F64 */
  uint32_t offset;             /* TE: byte offset of slot 0 inside the record */
  uint64_t groups;
  const int64_t *count;        /* JG */
  const double *const *sum;    /* JG: nvals pointers each (or NULL) */
  const double *const *sumsq;
  const double *const *mn;
  const double *const *mx;
  const int32_t *kinds;        /* JG: ncols entries each */
  const int32_t *vals;
  const int32_t *dst_dtypes;
  const uint32_t *offs;
  const int64_t *tot_count;    /* TE */
  const double *tot_sum;
  const int64_t *fold_count;
  const double *fold_sum;
  double p_smooth;
  double y_mean;
  const double *moments;       /* TE: NULL, or {count, sum} of the target on the device: the kernel
                                  takes y_mean = sum / count from there (the fit's mean need not
                                  have reached the host when the image is enqueued) */
} nvt_image_part;
int nvt_keydir_build(const int32_t *keys32, uint64_t n, uint64_t dir_slots, uint32_t *dir, void *stream);
int nvt_keydir_lookup_image(const void *keys, int dtype, const uint8_t *valid, uint64_t n,
                            const uint32_t *dir, uint64_t dir_slots, const int32_t *keys32, uint64_t nkeys,
                            int64_t key_offset, int64_t null_group, const void *image,
                            uint32_t stride_bytes, int ncols, void *const *outs, const uint8_t *const *folds, const uint32_t *offs,
                            const uint32_t *sizes, const uint64_t *miss_bits, uint64_t *unseen,
                            void *stream);
int nvt_image_build(const nvt_image_part *parts, int nparts, uint64_t records, void *image,
                    uint32_t stride_bytes, void *stream);

/* ---- parquet in: PLAIN / uncompressed column chunks of flat numeric columns -------------
 * (merlin.io.Dataset(engine="parquet") under Workflow.fit / transform:
 * tests/unit/workflow/test_cpu_workflow.py:67-81, bench/examples/dask-nvtabular-criteo-benchmark.py
 * :216-237; the reference's dataframe backend decodes the pages.)
 * nvt_pq_decode_chunk (HOST function, no GPU, thread-safe): chunk = the bytes of one column chunk
 * from its first data page on (DataPage v1 / v2, PLAIN values, definition levels in the RLE /
 * bit-packed hybrid, max_def_level 0 = REQUIRED or 1 = OPTIONAL).  Writes the rows' Arrow validity
 * bitmap (LSB first, row i = bit valid_bit_offset + i: consecutive row groups of a partition go
 * into ONE bitmap, by one thread; valid_out needs ceil((offset + rows) / 8) + 8 bytes) and the
 * pages' values -- non-null ones only -- behind each other into values_out.  *rows_out must come
 * out as expect_rows; *values_count = non-null rows.  NVT_EUNSUPPORTED: dictionary / compressed /
 * other encodings / nested columns (the caller reads the file another way).
 * nvt_expand_valid: out[i] = bit i of bitmap ? packed[valid rows in front of i] : 0 for n rows
 * of 4- or 8-byte values (bitmap 8-byte aligned, ceil(n / 64) * 8 bytes readable); ws:
 * nvt_expand_valid_ws_bytes(n). */
int nvt_pq_decode_chunk(const uint8_t *chunk, uint64_t chunk_bytes, int type_size, int max_def_level,
                        uint64_t expect_rows, uint8_t *valid_out, uint64_t valid_bit_offset,
                        uint8_t *values_out, uint64_t values_cap_bytes, uint64_t *rows_out,
                        uint64_t *values_count);
/* nvt_pq_decode_chunk_codec (round 6): the same for what real files hold -- chunk = the bytes of the
 * column chunk from its FIRST page on (the dictionary page when there is one), codec = the chunk's
 * parquet CompressionCodec (0 UNCOMPRESSED, 1 SNAPPY: every page is one raw snappy block; v2 pages
 * keep their levels uncompressed), values PLAIN or dictionary indices (PLAIN_DICTIONARY /
 * RLE_DICTIONARY: RLE / bit-packed hybrid at width <= 32 behind a PLAIN dictionary page).  The
 * decoder writes packed PLAIN values exactly like nvt_pq_decode_chunk.  scratch (host memory the
 * decoder owns during the call: the uncompressed dictionary + one uncompressed page):
 * 2 * total_uncompressed_size of the chunk + 64 bytes always suffice; may be null for codec 0.
 * NVT_EUNSUPPORTED: other codecs (gzip, zstd, lz4, brotli) / encodings (DELTA_*, BYTE_STREAM_SPLIT). */
int nvt_pq_decode_chunk_codec(const uint8_t *chunk, uint64_t chunk_bytes, int codec, int type_size,
                              int max_def_level, uint64_t expect_rows, uint8_t *valid_out,
                              uint64_t valid_bit_offset, uint8_t *values_out, uint64_t values_cap_bytes,
                              uint8_t *scratch, uint64_t scratch_bytes, uint64_t *rows_out,
                              uint64_t *values_count);
int nvt_expand_valid_ws_bytes(uint64_t n, uint64_t *bytes);
int nvt_expand_valid(const void *packed, int type_size, const uint8_t *bitmap, uint64_t n, void *out,
                     void *ws, void *stream);

/* ---- batched entry points: ONE call per operator per partition -----------------------
 * The reference hands a whole dataframe to the backend per operator call
 * (`df[cols].fillna(...)`, `for col in columns: _encode(...)` -- categorify.py:477-537,
 * normalize.py:71-90, moments.py:64-77); the per-column functions above cost one host
 * round trip (ctypes + launch) per column per step, which made the 45 M-row Criteo step
 * host-bound on slow hosts.  These take an array of per-column descriptors (HOST memory,
 * plain pointers and sizes) and enqueue every launch from C++; where the columns are
 * independent streaming passes they run as ONE kernel launch (blockIdx.y = column).
 * Results are bit-identical to calling the per-column function on each descriptor. */
typedef struct nvt_moments_col {
  const void *x;            /* device column                                    */
  const uint8_t *valid;     /* bitmap or NULL                                   */
  uint64_t n;
  int32_t dtype, has_fill;
  double fill_val;
  double *out3;             /* device double[3], accumulated into               */
} nvt_moments_col;
/* partials: device scratch of ncols * nvt_moments_scratch_bytes() */
int nvt_moments_many(const nvt_moments_col *cols, int ncols, void *partials, void *stream);

typedef struct nvt_fillnorm_col {
  const void *x;
  const uint8_t *valid;
  uint64_t n;
  int32_t dtype, has_fill;
  double fill_val;
  int32_t do_norm, out_dtype;
  double shift, scale;
  void *out;
  uint8_t *filled;          /* optional                                         */
  const double *moments;    /* NULL, or the column's {count, sum, sum of squares} on the device
                               (nvt_moments_many's out3): shift = mean and scale = std (0 when it
                               is not > 0) are then finished by the kernel exactly as the host
                               finishes them (moments.py:89-116) -- a fit whose moments have not
                               reached the host yet does not hold the transform back */
} nvt_fillnorm_col;
int nvt_fill_normalize_many(const nvt_fillnorm_col *cols, int ncols, void *stream);

/* one column of Categorify.fit's groupby-size; fields as the arguments of nvt_dense_count_* */
typedef struct nvt_count_col {
  const void *keys;
  const uint8_t *valid;
  const int64_t *weights;
  uint64_t n;
  int32_t key_bytes, path;
  void *ws;                 /* nvt_dense_count_ws_bytes().  Columns of a call that SHARE a workspace
                               run in order; columns given different workspaces (at most 3
                               distinct pointers per call) run concurrently on internal streams
                               forked from / joined into `stream`                          */
  void *out_keys;
  int64_t *out_counts;
  uint64_t out_capacity;
  uint64_t *state;          /* device uint64[NVT_STATE_WORDS]                   */
  int32_t *hot_image;       /* path | NVT_PATH_HOT: device int32[NVT_HOT_IMAGE_WORDS] of this column's
                               own (workspaces may be shared): the hot-key samples of all columns
                               are then taken by ONE launch ahead of the pipelines.  NULL: sampled
                               inside the column's pipeline, image kept in ws                 */
  void *range_table;        /* NVT_PATH_RANGE, optional: nvt_range_table_bytes(buckets) bytes.  The
                               per-bucket count tables are written out as they are: an encode table
                               addressed by the same monotone map, slot = {key, position in the
                               key-ordered output list}.  nvt_vocab_finalize_many turns the positions
                               into labels (nvt_vocab_col.range_table), nvt_encode_many probes it
                               (nvt_encode_col.range_aux).                                        */
} nvt_count_col;
/* bytes of the range table of a column counted with 2^nb_log2 buckets */
int nvt_range_table_bytes(int nb_log2, uint64_t *bytes);
int nvt_dense_count_many(const nvt_count_col *cols, int ncols, void *stream);

/* one vocabulary of Categorify.fit_end: sort (count desc, key asc) in place, then build its
 * encode table (skipped when table == NULL: LDS-resident vocabularies).  Independent
 * vocabularies are spread over a few internal HIP streams forked from / joined into
 * `stream` (the one-workgroup kernels of small vocabularies run beside the radix passes of
 * large ones); all small vocabularies of a call are sorted by ONE launch. */
typedef struct nvt_vocab_col {
  void *keys;               /* device int32/int64[n], sorted in place            */
  int64_t *counts;          /* device int64[n], permuted with the keys           */
  uint64_t n;
  int64_t max_count;        /* upper bound on counts (<= 0: unknown)             */
  int32_t key_bytes, unique_keys;
  void *sort_tmp;           /* nvt_vocab_sort_tmp_bytes(); per column            */
  int64_t first_label;
  void *table;              /* encode table to build, or NULL                    */
  uint64_t capacity;
  int64_t *sentinel_label;
  void *ready_event;        /* optional nvt_event: recorded behind this vocabulary's last
                             * kernel INSTEAD of joining its stream into `stream`; whoever
                             * reads the vocabulary / table next waits on it
                             * (nvt_encode_col.wait_event, nvt_stream_wait_event)        */
  /* KEY-SORTED source list (range path of nvt_dense_count_many; int32 keys): when src_keys is
   * set, (keys, counts) are OUTPUT arrays of n entries -- the list is ordered out of place by
   * ONE stable counting pass on min(count, 255), the n_big entries with count >= 255 are sorted
   * on their own, and the encode table is filled by the same pass.  sort_tmp then needs
   * nvt_vocab_order_tmp_bytes(n, n_big) bytes. */
  const void *src_keys;
  const int64_t *src_counts;
  const uint32_t *cls_hist;  /* device uint32[256]: entries per min(count, 255)           */
  uint64_t n_big;            /* entries with count >= 255 (state[NVT_ST_BIG])             */
  /* with a key-sorted source: `table` is the RANGE TABLE the counting pass dumped
   * (nvt_count_col.range_table; slots {key, position in the source list}): its positions are
   * replaced by labels in one streaming pass -- no clear, no random inserts.  range_aux = the
   * column's aux block (nvt_count_col.hot_image) that holds the map parameters. */
  const int32_t *range_aux;
  int32_t range_nb_log2;
  /* flat_slots > 0 (with a key-sorted source, int32 keys): `table` is BUILT here as a flat
   * range table of flat_slots home slots (any count from 64 to 2^32 - 1; + n + 64 tail slots:
   * capacity = the sum) straight
   * from the sorted keys by a prefix maximum -- no random inserts; range_aux = a device
   * int32[NVT_FLAT_AUX_WORDS] block that RECEIVES the map parameters (pass it to nvt_encode_col)
   * and, in word NVT_FLAT_AUX_MAXDISP, the longest displacement of an entry from its home slot
   * (large: the keys cluster in their range, build an ordinary table with nvt_encode_build_*) */
  uint64_t flat_slots;
  /* with a key-sorted source: the 0-based position of every source entry in the vocabulary order
   * (int32[n]; multi-GPU fits label their shards on the owners, nvt_vocab_label_shard).  cls_hist
   * / n_big are then ignored: (keys, counts) are filled by one scatter and the table (flat, or
   * hashed when flat_slots == 0) is built from the labels -- no ordering pass.  sort_tmp:
   * nvt_vocab_order_tmp_bytes(n, 0) bytes. */
  const int32_t *src_labels;
  /* optional (int32 keys, n > NVT_ENCODE_RESIDENT_I32, unique keys): NVT_ENCODE_HEAD_BYTES bytes that
   * receive the LDS head of the cache-mode encode -- the first keys of the ORDERED vocabulary laid
   * out as the launch's workgroups hold them -- built once here, on the stream that orders the
   * vocabulary, instead of by every workgroup of every encode launch (nvt_encode_col.head_image). */
  void *head_image;
} nvt_vocab_col;
#define NVT_ENCODE_HEAD_BYTES (12288 * 12 + 64)
#define NVT_FLAT_AUX_WORDS (NVT_RANGE_AUX_LO + 16)
#define NVT_FLAT_AUX_MAXDISP (NVT_RANGE_AUX_LO + 8)
#define NVT_FLAT_AUX_NULLGROUP (NVT_RANGE_AUX_LO + 10)  /* nvt_flat_lookup*: 1 + the group of rows whose
                                                          key is null (0: none, such rows miss); set
                                                          by the caller after nvt_flat_index_build */
int nvt_vocab_order_tmp_bytes(uint64_t n, uint64_t n_big, uint64_t *bytes);
/* One rank's SHARD (a key range) of a key-sorted (key, count) list that is ordered "count
 * descending, key ascending" as a whole (categorify.py:1300,1316): label_of[i] = the 0-based
 * position of entry i in the order of the UNION for entries with count < 255, -1 for the others,
 * which are written out compacted in key order (big_keys / big_counts / big_src = their positions
 * in the shard; the caller gathers all ranks' and sorts them exactly).  class_base_diff
 * uint32[256], indexed by class c = min(count, 255): with base(c) = (entries of the union in
 * classes 255 .. c + 1) + (entries of class c on the ranks in front of this one), the value at c is
 * base(c - 1) - base(c) modulo 2^32 for 2 <= c <= 254, base(254) at 255, anything at 1 (the kernel's
 * exclusive prefix over the classes 255, 254, ... reproduces the bases).  tmp:
 * nvt_vocab_order_tmp_bytes(n, 0) bytes; big_* hold as many entries as the shard has with count >=
 * 255. */
int nvt_vocab_label_shard(const int32_t *keys, const int64_t *counts, uint64_t n, const uint32_t *class_base_diff,
                          void *tmp, int32_t *label_of, int32_t *big_keys, int64_t *big_counts,
                          int32_t *big_src, void *stream);
/* hist[c] = entries with min(counts[i], 255) == c, for a key-sorted list that did not come from
 * the range path (multi-GPU: gathered owner shards); hist[255] = n_big */
int nvt_class_hist(const int64_t *counts, uint64_t n, uint32_t *hist, void *stream);
int nvt_vocab_finalize_many(const nvt_vocab_col *cols, int ncols, void *stream);

/* one column of Categorify.transform; fields as the arguments of nvt_encode_* */
typedef struct nvt_encode_col {
  const void *keys;
  const uint8_t *valid;
  uint64_t n;
  const void *table;
  uint64_t capacity;
  const int64_t *sentinel_label;
  int64_t null_label, oov_label;
  uint32_t num_buckets;
  int32_t key_bytes, out_bytes;
  void *out;
  const void *vocab_keys;
  uint64_t n_vocab;
  int64_t first_label;
  void *wait_event;         /* optional nvt_event the launch waits for (stream-side)     */
  const int32_t *range_aux; /* `table` is a range table (nvt_count_col.range_table): probed by
                               the monotone map whose parameters sit in this aux block      */
  const void *head_image;   /* optional: the head image nvt_vocab_finalize_many built for this
                               vocabulary (valid once wait_event has fired)                 */
} nvt_encode_col;
/* The columns are independent (own output, own table): they are spread round-robin over
 * NVT_ENCODE_STREAMS (environment, default 3, 1 = none; read at every call) internal streams forked
 * from / joined into `stream` -- everything the call enqueues is ordered behind what `stream` held
 * and in front of what the caller enqueues next. */
int nvt_encode_many(const nvt_encode_col *cols, int ncols, void *stream);

/* Stream-ordered hand-off between nvt_vocab_finalize_many (which orders the large
 * vocabularies on internal streams) and the calls that consume a vocabulary: with a
 * ready_event per vocabulary the caller's stream is NOT blocked until every sort and table
 * build has finished -- FillMissing + Normalize.transform and the encodes of the small
 * vocabularies run underneath the radix passes of the large ones.  An nvt_event is an
 * opaque handle (a hipEvent_t without timing). */
int nvt_event_create(void **event);
void nvt_event_destroy(void *event);
int nvt_stream_wait_event(void *stream, void *event);

/* ---- device -> host read-back of a few words without a blocking runtime wait ----------
 * The fit needs two tiny read-backs per partition (the counting kernels' state words, the
 * moments).  A hipMemcpy + stream synchronise blocks the host thread in an interrupt-driven
 * wait; on some hosts that wake-up was observed to cost ~10 ms per wait (a 20 ms step became
 * 39 ms with identical kernel times).  A mailbox is coherent, device-mapped pinned host
 * memory: nvt_mailbox_post enqueues ONE small kernel that copies `bytes` (multiple of 8) from
 * device memory into the mailbox and then stores a sequence number with system scope;
 * nvt_mailbox_wait spins on that word on the host (no interrupt, no runtime call) and returns
 * NVT_OK when it reaches `seq` (NVT_EHIP after timeout_s seconds).  nvt_mailbox_data is the
 * host pointer of the payload. */
typedef struct nvt_mailbox nvt_mailbox;
int nvt_mailbox_create(uint64_t bytes, nvt_mailbox **out);
void nvt_mailbox_destroy(nvt_mailbox *mb);
void *nvt_mailbox_data(nvt_mailbox *mb);
uint64_t nvt_mailbox_capacity(nvt_mailbox *mb);
int nvt_mailbox_post(nvt_mailbox *mb, const void *src_device, uint64_t bytes, void *stream,
                     uint64_t *seq_out);
int nvt_mailbox_wait(nvt_mailbox *mb, uint64_t seq, double timeout_s);

/* ---- instrumentation ------------------------------------------------------------------
 * HIP-event timing of every kernel family, recorded on the stream the kernels are launched
 * on.  nvt_prof_begin() arms it; nvt_prof_report() synchronises the device, disarms it and
 * writes a JSON object {"kernels": {name: [total_ms, launches, algorithmic_bytes]},
 * "busy_ms": union of the recorded intervals, "span_ms": first start .. last stop} into buf
 * (*needed = bytes required).  Off = zero cost.  Every scope is also a roctx range, and
 * nvt_range_push/pop let the host layer open ranges named after the reference's @annotate
 * strings (categorify.py:345,477,955,1054,1073,1149). */
int nvt_prof_begin(void);
int nvt_prof_report(char *buf, uint64_t cap, uint64_t *needed);
void nvt_range_push(const char *name);
void nvt_range_pop(void);

/* ---- small utilities used by the host layer ---- */
/* widen an int32/uint8 key column to int64 (multi-key tables take int64 components) */
int nvt_widen_i64(const void *src, int dtype, uint64_t n, int64_t *out, void *stream);
/* count set bits of a validity bitmap over n rows (null bookkeeping, meta.*.parquet) */
int nvt_popcount(const uint8_t *valid, uint64_t n, uint64_t *out_device, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NVT_HIP_H */
