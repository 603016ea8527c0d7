// Parquet in: the input half of Dataset -> HBM for column chunks of flat int32 / int64 / float /
// double columns: PLAIN or dictionary-encoded (PLAIN_DICTIONARY / RLE_DICTIONARY) values, codec
// UNCOMPRESSED or SNAPPY, data pages v1 / v2 -- what pandas, pyarrow, cuDF and the reference
// write by default (round 6; before: PLAIN + uncompressed only, i.e. parquet_plain.py's own files).  Reference contract: merlin.io.Dataset(engine=
// "parquet") feeding Workflow.fit / transform (tests/unit/workflow/test_cpu_workflow.py:67-81,
// bench/examples/dask-nvtabular-criteo-benchmark.py:216-237); the reference's backend (cuDF /
// pyarrow) decodes pages on its side of that call.
//
//   host (a pool thread per column chunk, GIL released): pread the chunk -> nvt_pq_decode_chunk:
//     walk the thrift-compact page headers, turn the definition levels (RLE / bit-packed hybrid
//     at bit width 1) into the Arrow validity bitmap, and move the pages' values -- PLAIN stores
//     only the non-null ones -- behind each other into a pinned staging buffer;
//   device: values + bitmap arrive by hipMemcpyAsync on the copy stream; a column WITH nulls is
//     expanded to one slot per row there (nvt_expand_valid: rank of a row = popcount prefix of
//     the bitmap), so the PCIe link carries the packed values only.
//
//     snappy blocks are decompressed and dictionary indices (RLE / bit-packed hybrid, width <= 32)
//     resolved through the chunk's dictionary in the same task, so the staging buffer always holds
//     packed PLAIN values.
//
// Anything else in a chunk (other codecs / encodings, nested columns) is reported as
// NVT_EUNSUPPORTED and the caller reads that file with pyarrow.
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/nvt_hip.h"
#include "nvt_common.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"
#include "nvt_scan.hpp"

namespace nvt {
namespace {

// ---- thrift compact protocol, reading side (parquet-format's PageHeader) ----------------------
struct TReader {
  const uint8_t *p, *end;
  bool ok = true;
  uint64_t varint() {
    uint64_t v = 0;
    int sh = 0;
    while (p < end) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7F) << sh;
      if (!(b & 0x80)) return v;
      sh += 7;
      if (sh > 63) break;
    }
    ok = false;
    return 0;
  }
  int64_t zigzag() {
    const uint64_t v = varint();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  void skip_bytes(uint64_t n) {
    if ((uint64_t)(end - p) < n) {
      ok = false;
      p = end;
    } else {
      p += n;
    }
  }
  // compact types: 1 / 2 bool, 3 byte, 4 i16, 5 i32, 6 i64, 7 double, 8 binary, 9 list, 10 set,
  // 11 map, 12 struct
  void skip(int type, int depth = 0) {
    if (!ok || depth > 16) {
      ok = false;
      return;
    }
    switch (type) {
      case 1: case 2: break;
      case 3: skip_bytes(1); break;
      case 4: case 5: case 6: (void)varint(); break;
      case 7: skip_bytes(8); break;
      case 8: skip_bytes(varint()); break;
      case 9: case 10: {
        if (p >= end) { ok = false; return; }
        const uint8_t h = *p++;
        uint64_t n = h >> 4;
        if (n == 15) n = varint();
        const int et = h & 0x0F;
        for (uint64_t i = 0; i < n && ok; ++i) skip(et, depth + 1);
        break;
      }
      case 11: {
        const uint64_t n = varint();
        if (n == 0) break;
        if (p >= end) { ok = false; return; }
        const uint8_t kv = *p++;
        for (uint64_t i = 0; i < n && ok; ++i) {
          skip(kv >> 4, depth + 1);
          skip(kv & 0x0F, depth + 1);
        }
        break;
      }
      case 12: skip_struct(depth + 1); break;
      default: ok = false;
    }
  }
  // next field header of a struct: false at the stop byte
  bool field(int &fid, int &type) {
    if (p >= end) {
      ok = false;
      return false;
    }
    const uint8_t h = *p++;
    if (h == 0) return false;
    const int d = h >> 4;
    type = h & 0x0F;
    if (d == 0) fid = (int)zigzag(); else fid += d;
    return ok;
  }
  void skip_struct(int depth = 0) {
    int fid = 0, type = 0;
    while (ok && field(fid, type)) skip(type, depth);
  }
};

struct PageHead {
  int type = -1;            // 0 data page, 2 dictionary page, 3 data page v2
  int64_t uncompressed = 0, compressed = 0;
  int64_t num_values = 0;   // rows of the page (flat columns)
  int encoding = -1, def_encoding = -1;
  // v2
  int64_t num_nulls = -1, def_bytes = 0, rep_bytes = 0;
  bool v2_compressed = false;
  // dictionary page
  int64_t dict_values = 0;
  int dict_encoding = -1;
};

bool read_page_header(TReader &r, PageHead &h) {
  int fid = 0, type = 0;
  while (r.field(fid, type)) {
    if (fid == 1 && type == 5) h.type = (int)r.zigzag();
    else if (fid == 2 && type == 5) h.uncompressed = r.zigzag();
    else if (fid == 3 && type == 5) h.compressed = r.zigzag();
    else if (fid == 5 && type == 12) {  // DataPageHeader
      int f2 = 0, t2 = 0;
      while (r.field(f2, t2)) {
        if (f2 == 1 && t2 == 5) h.num_values = r.zigzag();
        else if (f2 == 2 && t2 == 5) h.encoding = (int)r.zigzag();
        else if (f2 == 3 && t2 == 5) h.def_encoding = (int)r.zigzag();
        else r.skip(t2);
      }
    } else if (fid == 7 && type == 12) {  // DictionaryPageHeader
      int f2 = 0, t2 = 0;
      while (r.field(f2, t2)) {
        if (f2 == 1 && t2 == 5) h.dict_values = r.zigzag();
        else if (f2 == 2 && t2 == 5) h.dict_encoding = (int)r.zigzag();
        else r.skip(t2);
      }
    } else if (fid == 8 && type == 12) {  // DataPageHeaderV2
      int f2 = 0, t2 = 0;
      h.v2_compressed = true;  // the field's default
      while (r.field(f2, t2)) {
        if (f2 == 1 && t2 == 5) h.num_values = r.zigzag();
        else if (f2 == 2 && t2 == 5) h.num_nulls = r.zigzag();
        else if (f2 == 4 && t2 == 5) h.encoding = (int)r.zigzag();
        else if (f2 == 5 && t2 == 5) h.def_bytes = r.zigzag();
        else if (f2 == 6 && t2 == 5) h.rep_bytes = r.zigzag();
        else if (f2 == 7 && (t2 == 1 || t2 == 2)) h.v2_compressed = (t2 == 1);
        else r.skip(t2);
      }
    } else {
      r.skip(type);
    }
  }
  return r.ok;
}

// ---- validity bitmap writing (LSB first) -----------------------------------------------------
inline void bits_fill(uint8_t *bm, uint64_t pos, uint64_t count, bool one) {
  while (count && (pos & 7)) {  // head
    if (one) bm[pos >> 3] |= (uint8_t)(1u << (pos & 7)); else bm[pos >> 3] &= (uint8_t)~(1u << (pos & 7));
    ++pos;
    --count;
  }
  const uint64_t nbytes = count >> 3;
  if (nbytes) memset(bm + (pos >> 3), one ? 0xFF : 0x00, nbytes);
  pos += nbytes << 3;
  count -= nbytes << 3;
  while (count) {  // tail
    if (one) bm[pos >> 3] |= (uint8_t)(1u << (pos & 7)); else bm[pos >> 3] &= (uint8_t)~(1u << (pos & 7));
    ++pos;
    --count;
  }
}

// copies `count` bits of src (from its bit 0) to bm at bit `pos`; -> number of one bits copied
inline uint64_t bits_copy(uint8_t *bm, uint64_t pos, const uint8_t *src, uint64_t count) {
  uint64_t ones = 0;
  const unsigned sh = (unsigned)(pos & 7);
  const uint64_t full = count >> 3;
  uint8_t *dst = bm + (pos >> 3);
  if (sh == 0) {
    memcpy(dst, src, full);
    for (uint64_t i = 0; i < full; ++i) ones += (uint64_t)__builtin_popcount(src[i]);
    dst += full;
  } else {
    for (uint64_t i = 0; i < full; ++i) {
      const uint8_t b = src[i];
      ones += (uint64_t)__builtin_popcount(b);
      dst[0] = (uint8_t)((dst[0] & ((1u << sh) - 1u)) | (uint8_t)(b << sh));
      dst[1] = (uint8_t)(b >> (8 - sh));
      ++dst;
    }
  }
  const unsigned rem = (unsigned)(count & 7);
  if (rem) {
    const uint8_t b = (uint8_t)(src[full] & ((1u << rem) - 1u));
    ones += (uint64_t)__builtin_popcount(b);
    uint64_t q = pos + (full << 3);
    for (unsigned j = 0; j < rem; ++j, ++q) {
      if ((b >> j) & 1) bm[q >> 3] |= (uint8_t)(1u << (q & 7)); else bm[q >> 3] &= (uint8_t)~(1u << (q & 7));
    }
  }
  return ones;
}

// definition levels of one page (max level 1, RLE / bit-packed hybrid, bit width 1) -> bits
// [pos, pos + rows) of the bitmap; *valid = rows with level 1
bool decode_levels(const uint8_t *lv, uint64_t nbytes, uint64_t rows, uint8_t *bm, uint64_t pos,
                   uint64_t *valid) {
  TReader r{lv, lv + nbytes};
  uint64_t done = 0, ones = 0;
  while (done < rows) {
    const uint64_t head = r.varint();
    if (!r.ok) return false;
    if (head & 1) {  // bit-packed run: (head >> 1) groups of 8 levels, one bit each
      const uint64_t groups = head >> 1, nb = groups;
      if ((uint64_t)(r.end - r.p) < nb) return false;
      uint64_t take = groups * 8;
      if (take > rows - done) take = rows - done;  // (the last group is padded)
      ones += bits_copy(bm, pos + done, r.p, take);
      r.p += nb;
      done += take;
    } else {  // RLE run: (head >> 1) times the value in the next byte (bit width 1 -> 1 byte)
      uint64_t count = head >> 1;
      if (r.p >= r.end || count == 0) return false;
      const uint8_t v = *r.p++;
      if (v > 1) return false;
      if (count > rows - done) count = rows - done;
      bits_fill(bm, pos + done, count, v == 1);
      ones += v ? count : 0;
      done += count;
    }
  }
  *valid = ones;
  return true;
}

// ---- snappy, raw block format (what parquet's SNAPPY codec holds per page) --------------------
// -> uncompressed bytes written, or -1 on malformed input / a block that does not fit `cap`
int64_t snappy_uncompress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap) {
  TReader r{src, src + n};
  const uint64_t want = r.varint();
  if (!r.ok || want > cap) return -1;
  const uint8_t *p = r.p, *end = src + n;
  uint64_t o = 0;
  while (p < end) {
    const uint8_t tag = *p++;
    uint64_t len, off;
    switch (tag & 3) {
      case 0: {   // literal
        len = (uint64_t)(tag >> 2) + 1;
        if (len > 60) {
          const unsigned nb = (unsigned)(len - 60);   // 1 .. 4 length bytes, little endian
          if ((uint64_t)(end - p) < nb) return -1;
          len = 0;
          for (unsigned i = 0; i < nb; ++i) len |= (uint64_t)p[i] << (8 * i);
          len += 1;
          p += nb;
        }
        if ((uint64_t)(end - p) < len || want - o < len) return -1;
        memcpy(dst + o, p, len);
        p += len;
        o += len;
        continue;
      }
      case 1:
        if (p >= end) return -1;
        len = (uint64_t)((tag >> 2) & 7) + 4;
        off = ((uint64_t)(tag >> 5) << 8) | *p++;
        break;
      case 2:
        if (end - p < 2) return -1;
        len = (uint64_t)(tag >> 2) + 1;
        off = (uint64_t)p[0] | ((uint64_t)p[1] << 8);
        p += 2;
        break;
      default:
        if (end - p < 4) return -1;
        len = (uint64_t)(tag >> 2) + 1;
        off = (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24);
        p += 4;
    }
    if (off == 0 || off > o || want - o < len) return -1;
    if (off >= len) {
      memcpy(dst + o, dst + o - off, len);
    } else {   // overlapping copy: a run
      for (uint64_t i = 0; i < len; ++i) dst[o + i] = dst[o - off + i];
    }
    o += len;
  }
  return o == want ? (int64_t)o : -1;
}

// `count` dictionary indices (RLE / bit-packed hybrid behind a one-byte bit width) resolved through
// `dict` (ndict values of T) into out[0 .. count)
template <typename T>
bool decode_dict_indices(const uint8_t *p, const uint8_t *end, uint64_t count, const T *dict, uint64_t ndict,
                         T *out) {
  if (count == 0) return true;
  if (p >= end) return false;
  const unsigned bw = *p++;
  if (bw > 32) return false;
  const uint64_t mask = bw == 32 ? 0xFFFFFFFFull : ((1ull << bw) - 1ull);
  TReader r{p, end};
  uint64_t done = 0;
  while (done < count) {
    const uint64_t head = r.varint();
    if (!r.ok) return false;
    if (head & 1) {   // bit-packed: (head >> 1) groups of 8 values, `bw` bits each, LSB first
      const uint64_t groups = head >> 1;
      if (groups == 0 || groups > (1ull << 40)) return false;
      const uint64_t nb = groups * bw;
      if ((uint64_t)(r.end - r.p) < nb) {
        // (writers may truncate the padding of the last group: take what the values need)
        const uint64_t need_vals = count - done < groups * 8 ? count - done : groups * 8;
        if ((uint64_t)(r.end - r.p) * 8 < need_vals * bw) return false;
      }
      uint64_t take = groups * 8;
      if (take > count - done) take = count - done;
      const uint8_t *q = r.p;
      const uint64_t avail = (uint64_t)(r.end - r.p);
      uint64_t bitpos = 0;
      for (uint64_t i = 0; i < take; ++i, bitpos += bw) {
        const uint64_t byte = bitpos >> 3;
        uint64_t w = 0;   // up to 5 bytes hold a value of <= 32 bits at any bit offset
        const uint64_t nbv = avail - byte < 8 ? avail - byte : 8;
        memcpy(&w, q + byte, nbv);
        const uint64_t idx = (w >> (bitpos & 7)) & mask;
        if (idx >= ndict) return false;
        out[done + i] = dict[idx];
      }
      r.skip_bytes(nb < avail ? nb : avail);
      done += take;
    } else {   // RLE: (head >> 1) times the value in the next ceil(bw / 8) bytes
      uint64_t n = head >> 1;
      const unsigned vb = (bw + 7) / 8;
      if (n == 0 || (uint64_t)(r.end - r.p) < vb) return false;
      uint64_t idx = 0;
      memcpy(&idx, r.p, vb);
      r.p += vb;
      idx &= mask;
      if (idx >= ndict) return false;
      if (n > count - done) n = count - done;
      const T v = dict[idx];
      for (uint64_t i = 0; i < n; ++i) out[done + i] = v;
      done += n;
    }
  }
  return true;
}

}  // namespace

// out[i] = valid(i) ? packed[rank(i)] : 0 with rank(i) = valid rows in front of row i.
// A wave takes 64 bitmap words (4096 rows): lane l owns word l for the rank prefix, then the
// wave walks the 64 words, lane r writing row 64 * w + r -- coalesced stores, and the loads of
// the packed values are consecutive too.
constexpr int kExpWaveRows = 64 * 64;
template <typename T>
__global__ __launch_bounds__(kBlock) void expand_valid_kernel(const T *__restrict__ packed,
                                                              const uint64_t *__restrict__ bitmap,
                                                              uint64_t n,
                                                              const unsigned *__restrict__ tile_base,
                                                              T *__restrict__ out) {
  const unsigned lane = lane_id();
  const uint64_t wave = (uint64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  const uint64_t nwords = (n + 63) / 64;
  const uint64_t w0 = wave * 64;
  if (w0 >= nwords) return;
  uint64_t word = (w0 + lane < nwords) ? bitmap[w0 + lane] : 0ull;
  const uint64_t row0 = (w0 + lane) * 64;
  if (row0 + 64 > n) word &= row0 >= n ? 0ull : ((1ull << (n - row0)) - 1ull);  // bits behind the column
  const unsigned pc = (unsigned)__popcll(word);
  unsigned inc = pc;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(inc, off, 64);
    if (lane >= (unsigned)off) inc += o;
  }
  const uint64_t wbase = (uint64_t)tile_base[wave] + inc - pc;  // valid rows in front of word `lane`
  for (int w = 0; w < 64; ++w) {
    if (w0 + w >= nwords) break;  // (uniform)
    const uint64_t bits = __shfl(word, w, 64);
    const uint64_t base = __shfl(wbase, w, 64);
    const uint64_t row = (w0 + w) * 64 + lane;
    if (row < n) {
      const bool v = (bits >> lane) & 1ull;
      const uint64_t rank = base + (uint64_t)__popcll(bits & ((1ull << lane) - 1ull));
      out[row] = v ? packed[rank] : (T)0;
    }
  }
}

__global__ __launch_bounds__(kBlock) void expand_count_kernel(const uint64_t *__restrict__ bitmap, uint64_t n,
                                                              unsigned *__restrict__ tile_cnt, uint64_t ntiles) {
  const unsigned lane = lane_id();
  const uint64_t wave = (uint64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (wave >= ntiles) return;
  const uint64_t nwords = (n + 63) / 64;
  const uint64_t wi = wave * 64 + lane;
  uint64_t word = wi < nwords ? bitmap[wi] : 0ull;
  const uint64_t row0 = wi * 64;
  if (row0 + 64 > n) word &= row0 >= n ? 0ull : ((1ull << (n - row0)) - 1ull);
  unsigned pc = (unsigned)__popcll(word);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pc += __shfl_down(pc, off, 64);
  if (lane == 0) tile_cnt[wave] = pc;
}

}  // namespace nvt

using namespace nvt;

extern "C" {

// codec: parquet CompressionCodec (0 UNCOMPRESSED, 1 SNAPPY).  scratch: room for the chunk's
// dictionary + its largest page uncompressed (2 * total_uncompressed_size + 64 always suffices);
// may be null for an uncompressed chunk without a dictionary page.
static int pq_decode(const uint8_t *chunk, uint64_t chunk_bytes, int codec, int type_size, int max_def_level,
                     uint64_t expect_rows, uint8_t *valid_out, uint64_t valid_bit_offset,
                     uint8_t *values_out, uint64_t values_cap_bytes, uint8_t *scratch, uint64_t scratch_bytes,
                     uint64_t *rows_out, uint64_t *values_out_count) {
  NVT_CHECK_ARG(chunk && values_out && rows_out && values_out_count, "null pointer");
  NVT_CHECK_ARG(type_size == 4 || type_size == 8, "values are 4 or 8 bytes");
  NVT_CHECK_ARG(max_def_level == 0 || max_def_level == 1, "flat columns: max definition level 0 / 1");
  NVT_CHECK_ARG(max_def_level == 0 || valid_out, "null validity buffer");
  if (codec != 0 && codec != 1) {
    set_error("nvt_pq_decode_chunk: codec %d (UNCOMPRESSED and SNAPPY are decoded here)", codec);
    return NVT_EUNSUPPORTED;
  }
  const uint8_t *p = chunk, *end = chunk + chunk_bytes;
  uint64_t rows = 0, vals = 0;
  const uint8_t *dict = nullptr;   // PLAIN values of the dictionary page (in the chunk or in scratch)
  uint64_t ndict = 0;
  uint64_t scratch_at = 0;         // scratch below this offset holds the dictionary
  // page body -> uncompressed bytes (`want` of them): the body itself, or scratch behind the dictionary
  auto inflate = [&](const uint8_t *src, uint64_t nsrc, uint64_t want, bool compressed,
                     const uint8_t **out) -> int {
    if (!compressed) {
      if (nsrc != want) {
        set_error("nvt_pq_decode_chunk: an uncompressed page of %llu bytes that says %llu",
                  (unsigned long long)nsrc, (unsigned long long)want);
        return NVT_EINVAL;
      }
      *out = src;
      return NVT_OK;
    }
    if (scratch == nullptr || scratch_bytes - scratch_at < want || scratch_bytes < scratch_at) {
      set_error("nvt_pq_decode_chunk: scratch too small for a page of %llu bytes", (unsigned long long)want);
      return NVT_EINVAL;
    }
    if (snappy_uncompress(src, nsrc, scratch + scratch_at, want) != (int64_t)want) {
      set_error("nvt_pq_decode_chunk: malformed snappy block");
      return NVT_EINVAL;
    }
    *out = scratch + scratch_at;
    return NVT_OK;
  };
  while (p < end && rows < expect_rows) {
    TReader r{p, end};
    PageHead h;
    if (!read_page_header(r, h)) {
      set_error("nvt_pq_decode_chunk: malformed page header at byte %llu", (unsigned long long)(p - chunk));
      return NVT_EINVAL;
    }
    const uint8_t *body = r.p;
    if (h.compressed < 0 || h.uncompressed < 0 || (uint64_t)(end - body) < (uint64_t)h.compressed) {
      set_error("nvt_pq_decode_chunk: page of %lld bytes runs past the chunk", (long long)h.compressed);
      return NVT_EINVAL;
    }
    if (h.type == 1) {  // index page: nothing for us
      p = body + h.compressed;
      continue;
    }
    if (h.type == 2) {  // dictionary page: PLAIN values (encoding 0; 2 = PLAIN_DICTIONARY, the old name)
      if (dict != nullptr || (h.dict_encoding != 0 && h.dict_encoding != 2) || h.dict_values < 0) {
        set_error("nvt_pq_decode_chunk: dictionary page (encoding %d, second %d)", h.dict_encoding, dict != nullptr);
        return NVT_EUNSUPPORTED;
      }
      const uint8_t *d = nullptr;
      int rc = inflate(body, (uint64_t)h.compressed, (uint64_t)h.uncompressed, codec != 0, &d);
      if (rc) return rc;
      if ((uint64_t)h.dict_values * (uint64_t)type_size > (uint64_t)h.uncompressed) {
        set_error("nvt_pq_decode_chunk: dictionary page holds fewer values than it says");
        return NVT_EINVAL;
      }
      dict = d;
      ndict = (uint64_t)h.dict_values;
      if (codec != 0) scratch_at = ((uint64_t)h.uncompressed + 15) & ~15ull;   // (kept: pages go behind it)
      p = body + h.compressed;
      continue;
    }
    if (h.type != 0 && h.type != 3) {
      set_error("nvt_pq_decode_chunk: page type %d", h.type);
      return NVT_EUNSUPPORTED;
    }
    const bool by_dict = h.encoding == 2 || h.encoding == 8;   // PLAIN_DICTIONARY / RLE_DICTIONARY
    if (h.encoding != 0 && !by_dict) {
      set_error("nvt_pq_decode_chunk: value encoding %d (PLAIN and dictionary indices are decoded here)", h.encoding);
      return NVT_EUNSUPPORTED;
    }
    if (by_dict && dict == nullptr) {
      set_error("nvt_pq_decode_chunk: dictionary indices without a dictionary page");
      return NVT_EINVAL;
    }
    const uint64_t prow = (uint64_t)h.num_values;
    if (h.num_values < 0 || prow > expect_rows - rows) {
      set_error("nvt_pq_decode_chunk: more rows than the row group holds");
      return NVT_EINVAL;
    }
    const uint8_t *q = nullptr, *pend = nullptr;   // the values (behind the levels), uncompressed
    uint64_t pvalid = prow;
    if (h.type == 0) {   // v1: levels and values are compressed together
      const uint8_t *u = nullptr;
      int rc = inflate(body, (uint64_t)h.compressed, (uint64_t)h.uncompressed, codec != 0, &u);
      if (rc) return rc;
      q = u;
      pend = u + h.uncompressed;
      if (max_def_level == 1) {
        if (h.def_encoding != 3) {  // RLE (the hybrid)
          set_error("nvt_pq_decode_chunk: definition level encoding %d", h.def_encoding);
          return NVT_EUNSUPPORTED;
        }
        if (pend - q < 4) return NVT_EINVAL;
        uint32_t lb;
        memcpy(&lb, q, 4);
        q += 4;
        if ((uint64_t)(pend - q) < lb || !decode_levels(q, lb, prow, valid_out, valid_bit_offset + rows, &pvalid)) {
          set_error("nvt_pq_decode_chunk: malformed definition levels");
          return NVT_EINVAL;
        }
        q += lb;
      }
    } else {  // v2: repetition levels (none for flat columns), then definition levels -- never
              // compressed, no length prefix --, then the values (compressed when the header says so)
      if (h.rep_bytes != 0) {
        set_error("nvt_pq_decode_chunk: repetition levels (nested column)");
        return NVT_EUNSUPPORTED;
      }
      if (h.def_bytes < 0 || (uint64_t)h.compressed < (uint64_t)h.def_bytes ||
          (uint64_t)h.uncompressed < (uint64_t)h.def_bytes) {
        set_error("nvt_pq_decode_chunk: definition levels run past the page");
        return NVT_EINVAL;
      }
      if (max_def_level == 1) {
        if (!decode_levels(body, (uint64_t)h.def_bytes, prow, valid_out, valid_bit_offset + rows, &pvalid)) {
          set_error("nvt_pq_decode_chunk: malformed definition levels");
          return NVT_EINVAL;
        }
      }
      const uint8_t *u = nullptr;
      const uint64_t vcomp = (uint64_t)h.compressed - (uint64_t)h.def_bytes;
      const uint64_t vraw = (uint64_t)h.uncompressed - (uint64_t)h.def_bytes;
      int rc = inflate(body + h.def_bytes, vcomp, vraw, codec != 0 && h.v2_compressed, &u);
      if (rc) return rc;
      q = u;
      pend = u + vraw;
    }
    if ((vals + pvalid) * (uint64_t)type_size > values_cap_bytes) {
      set_error("nvt_pq_decode_chunk: values buffer too small");
      return NVT_EINVAL;
    }
    if (!by_dict) {
      const uint64_t vbytes = pvalid * (uint64_t)type_size;
      if ((uint64_t)(pend - q) < vbytes) {
        set_error("nvt_pq_decode_chunk: page holds fewer values than its levels say");
        return NVT_EINVAL;
      }
      memcpy(values_out + vals * (uint64_t)type_size, q, vbytes);
    } else {
      const bool ok = type_size == 4
          ? decode_dict_indices<uint32_t>(q, pend, pvalid, reinterpret_cast<const uint32_t *>(dict), ndict,
                                          reinterpret_cast<uint32_t *>(values_out) + vals)
          : decode_dict_indices<uint64_t>(q, pend, pvalid, reinterpret_cast<const uint64_t *>(dict), ndict,
                                          reinterpret_cast<uint64_t *>(values_out) + vals);
      if (!ok) {
        set_error("nvt_pq_decode_chunk: malformed dictionary indices");
        return NVT_EINVAL;
      }
    }
    vals += pvalid;
    rows += prow;
    p = body + h.compressed;
  }
  if (rows != expect_rows) {
    set_error("nvt_pq_decode_chunk: %llu rows in the pages, %llu expected", (unsigned long long)rows,
              (unsigned long long)expect_rows);
    return NVT_EINVAL;
  }
  *rows_out = rows;
  *values_out_count = vals;
  return NVT_OK;
}

int nvt_pq_decode_chunk(const uint8_t *chunk, uint64_t chunk_bytes, int type_size, int max_def_level,
                        uint64_t expect_rows, uint8_t *valid_out, uint64_t valid_bit_offset,
                        uint8_t *values_out, uint64_t values_cap_bytes, uint64_t *rows_out,
                        uint64_t *values_out_count) {
  return pq_decode(chunk, chunk_bytes, 0, type_size, max_def_level, expect_rows, valid_out, valid_bit_offset,
                   values_out, values_cap_bytes, nullptr, 0, rows_out, values_out_count);
}

int nvt_pq_decode_chunk_codec(const uint8_t *chunk, uint64_t chunk_bytes, int codec, int type_size,
                              int max_def_level, uint64_t expect_rows, uint8_t *valid_out,
                              uint64_t valid_bit_offset, uint8_t *values_out, uint64_t values_cap_bytes,
                              uint8_t *scratch, uint64_t scratch_bytes, uint64_t *rows_out,
                              uint64_t *values_out_count) {
  return pq_decode(chunk, chunk_bytes, codec, type_size, max_def_level, expect_rows, valid_out, valid_bit_offset,
                   values_out, values_cap_bytes, scratch, scratch_bytes, rows_out, values_out_count);
}

int nvt_expand_valid_ws_bytes(uint64_t n, uint64_t *bytes) {
  NVT_CHECK_ARG(bytes, "null out");
  const uint64_t ntiles = (n + kExpWaveRows - 1) / kExpWaveRows;
  *bytes = ((ntiles * 4 + 255) & ~255ull) + scan_chunks(ntiles) * 8 + 256;
  return NVT_OK;
}

int nvt_expand_valid(const void *packed, int type_size, const uint8_t *bitmap, uint64_t n, void *out,
                     void *ws, void *stream) {
  if (n == 0) return NVT_OK;
  NVT_CHECK_ARG(packed && bitmap && out && ws, "null pointer");
  NVT_CHECK_ARG(type_size == 4 || type_size == 8, "values are 4 or 8 bytes");
  NVT_CHECK_ARG((reinterpret_cast<uintptr_t>(bitmap) & 7) == 0, "bitmap must be 8-byte aligned");
  NVT_CHECK_ARG(n < (1ull << 32), "fewer than 2^32 rows");
  hipStream_t s = (hipStream_t)stream;
  NVT_PROF("parquet_expand", n * (uint64_t)type_size, s);
  const uint64_t ntiles = (n + kExpWaveRows - 1) / kExpWaveRows;
  unsigned *tile = reinterpret_cast<unsigned *>(ws);
  unsigned long long *chunk_tot =
      reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ws) + ((ntiles * 4 + 255) & ~255ull));
  const uint64_t *bm = reinterpret_cast<const uint64_t *>(bitmap);
  const unsigned grid = (unsigned)((ntiles + (kBlock / kWave) - 1) / (kBlock / kWave));
  expand_count_kernel<<<grid, kBlock, 0, s>>>(bm, n, tile, ntiles);
  NVT_CHECK_LAUNCH();
  int rc = exclusive_scan_u32(tile, ntiles, chunk_tot, s);
  if (rc) return rc;
  if (type_size == 4)
    expand_valid_kernel<uint32_t><<<grid, kBlock, 0, s>>>((const uint32_t *)packed, bm, n, tile, (uint32_t *)out);
  else
    expand_valid_kernel<uint64_t><<<grid, kBlock, 0, s>>>((const uint64_t *)packed, bm, n, tile, (uint64_t *)out);
  NVT_CHECK_LAUNCH();
  return NVT_OK;
}

}  // extern "C"
