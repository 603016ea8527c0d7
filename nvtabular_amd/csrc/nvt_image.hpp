// Lookup images (include/nvt_hip.h "Lookup images"): what the kernels that BUILD the packed
// per-group records (nvt_sort.hip: nvt_jg_image / nvt_te_image / nvt_image_pack; nvt_keydir.hip:
// nvt_image_build) and the kernels that READ them (nvt_flat_lookup_image, nvt_keydir_lookup_image)
// share -- one definition of every value, so that a record holds the same bits whichever kernel
// wrote it.
#pragma once
#include "nvt_common.hpp"

namespace nvt {

constexpr int kImageMaxCols = 24;
struct ImageOuts {
  void *out[kImageMaxCols];
  const uint8_t *fold[kImageMaxCols];  // fold id column of this output (nullptr: fixed offset)
  uint64_t miss[kImageMaxCols];        // value bits of a row without group
  uint32_t off[kImageMaxCols];         // byte offset inside the record (slot 0)
  uint32_t fstride[kImageMaxCols];     // bytes per fold slot (= the value size)
  uint32_t size[kImageMaxCols];        // 4 or 8
  // nvt_keydir_lookup_image: outputs at fixed offsets that share an aligned 16-byte window of the
  // record are read by ONE 16-byte load (every separate load of a random record is a transaction
  // of its own between the vector cache and the L2: five 4-byte loads of one 64-byte record cost
  // 0.6 ms more than one on 20 M rows).  grp[c] = window of output c (0xFF: its own load),
  // word[c] = its first 32-bit word inside the window, goff[q] = byte offset of window q.
  uint8_t grp[kImageMaxCols], word[kImageMaxCols];
  uint32_t goff[kImageMaxCols / 2];
  int ngroups;
};

__device__ __forceinline__ void image_store(uint8_t *at, int dtype, double x, int64_t xi, bool is_int) {
  switch (dtype) {
    case NVT_F32: *reinterpret_cast<float *>(at) = is_int ? (float)xi : (float)x; break;
    case NVT_F64: *reinterpret_cast<double *>(at) = is_int ? (double)xi : x; break;
    case NVT_I32: *reinterpret_cast<int32_t *>(at) = is_int ? (int32_t)xi : (int32_t)x; break;
    default: *reinterpret_cast<int64_t *>(at) = is_int ? xi : (int64_t)x; break;
  }
}

// JoinGroupby's values from the fit's accumulators (join_groupby.py:175-217 over
// categorify.py:1087-1131 _bottom_level_groupby): count, sum, mean = sum / n, var = (sumsq -
// sum * sum / n) / max(n - 1, 1) (NaN for n = 1), std = sqrt(var), min, max -- evaluated per group
// in float64 like the column-wise path (ops/_groupby.py derive_stats), stored in the output dtype.
constexpr int kJgMaxVals = 8;
struct JgImageArgs {
  const int64_t *count;
  const double *sum[kJgMaxVals], *sumsq[kJgMaxVals], *mn[kJgMaxVals], *mx[kJgMaxVals];
  int kind[kImageMaxCols];   // 0 count, 1 sum, 2 mean, 3 min, 4 max, 5 var, 6 std
  int val[kImageMaxCols];    // value column of the statistic
  int dst_dtype[kImageMaxCols];
  uint32_t off[kImageMaxCols];
};

// statistic c of group g (ni = count[g]); *is_int: the value is the count itself
__device__ __forceinline__ double jg_stat(const JgImageArgs &a, int c, uint64_t g, int64_t ni, bool *is_int) {
  const int j = a.val[c];
  const double n = (double)ni;
  *is_int = false;
  switch (a.kind[c]) {
    case 0: *is_int = true; return 0.0;
    case 1: return a.sum[j][g];
    case 2: return a.sum[j][g] / n;
    case 3: return a.mn[j][g];
    case 4: return a.mx[j][g];
    default: {
      const double s1 = a.sum[j][g], s2 = a.sumsq[j][g];
      const double sq = __dmul_rn(s1, s1);            // (no contraction with the division / subtraction)
      const double num = __dsub_rn(s2, __ddiv_rn(sq, n));
      const double dn = n - 1.0;
      double var = __ddiv_rn(num, dn < 1.0 ? 1.0 : dn);
      if (dn == 0.0) var = __longlong_as_double(0x7FF8000000000000ll);
      return a.kind[c] == 5 ? var : sqrt(var);
    }
  }
}

// TargetEncoding's value of (group g, slot): slot 0 = (sum + p * mean) / (count + p), slot 1 + f
// = the out-of-fold value of fold f (the mean when the (group, fold) pair has no rows) -- the
// expression nvt_te_apply_folds evaluates per row (target_encoding.py:350-371)
__device__ __forceinline__ double te_value(const int64_t *__restrict__ tot_count,
                                           const double *__restrict__ tot_sum,
                                           const int64_t *__restrict__ fold_count,
                                           const double *__restrict__ fold_sum, unsigned kfold,
                                           uint64_t g, unsigned slot, double p, double y_mean) {
  const double c = (double)tot_count[g], d = tot_sum[g];
  if (slot == 0) return (d + p * y_mean) / (c + p);
  const uint64_t f = g * kfold + (slot - 1);
  const double fc = (double)fold_count[f], fs = fold_sum[f];
  return fc > 0.0 ? (d - fs + p * y_mean) / (c - fc + p) : y_mean;
}

}  // namespace nvt
