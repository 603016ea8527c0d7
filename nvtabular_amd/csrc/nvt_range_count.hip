// Categorify.fit groupby-size, path 9 ("range path"): int32 keys, unweighted, ~11 k .. ~6.5 M
// distinct keys.  Replaces categorify.py:955-1051 (_top_level_groupby, size only) like the
// hash-partitioned paths 1-3 of nvt_dense_count.hip, with two differences that matter:
//
//   * ONE pass over the column instead of histogram + scan + 1-2 scatter passes: every
//     workgroup keeps a small write-combining bin per bucket in LDS and flushes whole 64-byte
//     lines into its OWN region of each bucket (no cursors, no global atomics, no histogram
//     pre-pass).  Hot keys are counted in LDS as before (hot-key image of the sample kernel)
//     and never reach the bins.
//   * buckets are KEY RANGES (monotone bucket function) and the per-bucket LDS table is
//     addressed by a monotone function of the key with linear probing, so a table's clusters
//     are in key order and only the (short) clusters themselves have to be ordered when the
//     table is emitted: the (key, count) list leaves this path SORTED BY KEY.  The vocabulary
//     order "count descending, key ascending" (categorify.py:1300,1316) then needs a single
//     stable counting pass on min(count, 255) instead of a 7-pass radix sort
//     (nvt_sort.hip: cls_scatter_kernel).
//
//   rp_partition_kernel  256 workgroups x 1024 threads, contiguous row slabs.  Per row: hot
//                        lookup (one 8-byte LDS read) -> counter, or bin append (one returning
//                        LDS atomic + one LDS write).  A bin that holds >= 16 keys is flushed
//                        (16 lanes x 4 B = one 64-byte line) into region (bucket, workgroup).
//   hot_totals_kernel    column sums of the per-workgroup hot counters -> total per image slot
//   rp_count_kernel      one workgroup per bucket (ticketed, in bucket order): gathers the 256
//                        runs of its bucket + the hot keys that fall into its range into a
//                        monotone LDS table, ranks every entry inside its cluster, finds its
//                        offset in the output by a decoupled look-back over the preceding
//                        buckets and writes (key, count) in key order; also the histogram of
//                        min(count, 255) for the ordering pass.
//
// The ranges come from the sample (min / max of the sampled keys, padded): balanced for keys
// that are spread over their range (hashed ids: Criteo, the bench generator).  Badly balanced
// ranges overflow a region or a table; the overflow bit sends the column to the hash paths,
// which make no assumption about the key distribution.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "nvt_common.hpp"
#include "nvt_internal.hpp"
#include "nvt_prof.hpp"
#include "nvt_range.hpp"

namespace nvt {

namespace {

constexpr int32_t kEmpty = INT32_MIN;
constexpr int kRpBS = 1024;
constexpr int kRpG = NVT_RANGE_WGS;          // partition workgroups (row slabs)
constexpr int kRpBinWords = 20480;           // 80 KiB of bins: CAP = kRpBinWords / NB keys per bin
constexpr int kRpLine = 16;                  // keys per flushed line (64 bytes)
constexpr int kRpMaxNbLog2 = 10;
constexpr int kHotSlotsR = NVT_HOT_IMAGE_WORDS;
constexpr int kHotBucketsR = kHotSlotsR / 2;
constexpr int kRpProbe = 512;
#ifndef NVT_RP_MAX_EPOCH
#define NVT_RP_MAX_EPOCH 8
#endif
constexpr unsigned kRpMaxEpoch = NVT_RP_MAX_EPOCH;   // rounds between two flushes of the partition bins, at most
#ifndef NVT_RANGE_U
#define NVT_RANGE_U 2
#endif
constexpr unsigned long long kStAgg = 1ull << 62, kStPrefix = 2ull << 62,
                             kStMask = (1ull << 62) - 1ull;

__device__ __forceinline__ uint32_t hot_bucket_r(int32_t key) { return hot_image_bucket(key, kHotBucketsR - 1); }

// PW = false: the host did not ask for a piecewise map (no NVT_PATH_PIECES): the linear form,
// straight-line code -- the GB home slots of a gather batch are computed and read together.
// PW = true: the sample kernel may have chosen either form (run-time decision per call).
template <bool PW>
__device__ __forceinline__ uint32_t rp_fine(const RangeMap &map, int32_t key) {
  if constexpr (PW) return map.fine(key);
  else return map.template fine_staged<false>(key);
}


// ---------------------------------------------------------------------------------------------
// pass 1: hot counters + range partition of the cold rows
// ---------------------------------------------------------------------------------------------
// NBL: nb_log2 at compile time (8 / 9 / 10: what the host picks for real columns) so that NB, CAP,
// BPW and the lane-role masks of the flush are constants -- the kernel sits at 100+ scalar
// registers and spent scalar and vector instructions on them every round (SQ counters,
// profiles/r05_sq_counters.json: 28 SALU + 70 VALU per key); 0 = the run-time value.
template <int U, int NBL, bool PW>
__global__ __launch_bounds__(kRpBS) void rp_partition_kernel(
    const int32_t *__restrict__ keys, const uint8_t *__restrict__ valid, uint64_t n,
    const int32_t *__restrict__ aux, int nb_log2_rt, uint32_t region_cap, int32_t *__restrict__ regions,
    uint32_t *__restrict__ fills, unsigned *__restrict__ hot_cnt, uint64_t *state,
    unsigned long long *__restrict__ clr_status, unsigned *__restrict__ clr_hist) {
  // what pass 2 expects cleared (two memset launches per column otherwise): the look-back status
  // words with the ticket behind them, and the class histogram
  const int nb_log2 = NBL ? NBL : nb_log2_rt;
  {
    const unsigned nst = (1u << nb_log2) + 8u;  // + 64 bytes: the ticket
    for (unsigned i = blockIdx.x * kRpBS + threadIdx.x; i < nst; i += kRpG * kRpBS) clr_status[i] = 0ull;
    if (blockIdx.x == kRpG - 1 && threadIdx.x < 256) clr_hist[threadIdx.x] = 0u;
  }
  // hot bucket = {key0, key1, count0, count1}: the counter of a hit sits 8 / 12 bytes behind the keys
  // the lookup has read (one address computation for both)
  __shared__ int4 hot[kHotBucketsR];
  __shared__ int32_t bins[kRpBinWords];
  __shared__ unsigned fill[1 << kRpMaxNbLog2], flushed[1 << kRpMaxNbLog2];
  __shared__ unsigned scratch[kWave];  // one word per lane: target of the adds of non-hits
  __shared__ unsigned long long s_nulls, s_sent;
  __shared__ unsigned s_ovf;
  const unsigned NB = 1u << nb_log2, CAP = (unsigned)kRpBinWords >> nb_log2;
  const unsigned g = blockIdx.x, lane = lane_id();
  // Empty image slots are filled with a key that can never be LOOKED UP in their bucket (a filler
  // whose own bucket is another one), and null rows are looked up as the empty key: nothing in
  // the table equals the empty key, so the classification below needs no validity / sentinel
  // terms per key (round 6: 26 -> 18 vector and ~20 -> ~5 scalar instructions per key).
  int32_t fill0 = 0x5A5A5A5A, fill1 = fill0 + 1;
  while (hot_bucket_r(fill1) == hot_bucket_r(fill0)) ++fill1;
  const uint32_t fb0 = hot_bucket_r(fill0);
  for (int i = threadIdx.x; i < kHotBucketsR; i += kRpBS) {
    const int2 k2 = reinterpret_cast<const int2 *>(aux)[i];
    const int32_t fk = (uint32_t)i != fb0 ? fill0 : fill1;
    hot[i] = make_int4(k2.x == kEmpty ? fk : k2.x, k2.y == kEmpty ? fk : k2.y, 0, 0);
  }
  for (unsigned i = threadIdx.x; i < NB; i += kRpBS) {
    fill[i] = 0;
    flushed[i] = 0;
  }
  if (threadIdx.x == 0) {
    s_nulls = 0;
    s_sent = 0;
    s_ovf = 0;
  }
  RangeMap map = load_map(aux);
  __shared__ uint32_t s_pieces[kRpPwWords];
  stage_pieces(map, s_pieces, threadIdx.x, kRpBS);
#ifdef NVT_RP_PART_TIMING
  long long ptm[8];
  int ptmi = 0;
#define NVT_PTM() do { if (threadIdx.x == 0) ptm[ptmi++] = clock64(); } while (0)
#else
#define NVT_PTM() do {} while (0)
#endif
  NVT_PTM();
  __syncthreads();
  NVT_PTM();

  // flush every bin that holds >= kRpLine keys (or, with `all`, whatever it holds): one
  // 16-lane group per bin, four bins per wave instruction; bin t is owned by thread t
  // ... spread over ALL 16 waves: wave w owns bins [w * NB / 16, (w + 1) * NB / 16), lane l the
  // bookkeeping of the l-th of them.  (With thread t owning bin t, 256 buckets kept 4 waves busy
  // and 12 waiting at the barrier for up to 16 rounds of the loop below.)
  // Round 6: a key that finds its bin full no longer waits for a flush -- it is stored straight to
  // its place in the region (its returned position IS its offset behind what the bin has
  // flushed), so appends never fail, the retry loop with its workgroup-wide OR is gone and the
  // barrier pair + flush run once per EPOCH of R rounds (R adapts: see the loop).  A bin that
  // overflowed in an epoch (fill > CAP) is emptied completely and its region is padded with the
  // empty key up to the next line (rp_count_kernel skips empty keys in its gather).
  const unsigned BPW = NB / (kRpBS / kWave);  // 16 / 32 / 64 bins per wave
  __shared__ unsigned s_epoch[2];             // per epoch parity: bit 0 = a bin above CAP / 2, bits 8.. = bins that overflowed
  if (threadIdx.x < 2) s_epoch[threadIdx.x] = 0;   // (a barrier follows before the first flush)
  auto flush_bins = [&](bool all, unsigned parity) {
    {
      const bool owner = lane < BPW;
      const unsigned wave_base = (threadIdx.x / kWave) * BPW;
      const unsigned t = wave_base + (owner ? lane : 0u);
      const unsigned fr = owner ? fill[t] : 0u;     // appended since the last flush (may exceed CAP)
      const unsigned done0 = owner ? flushed[t] : 0u;
      const bool over = fr > CAP;
      const unsigned f = over ? CAP : fr;           // keys the bin itself holds
      const unsigned nfl = (all || over) ? f : (f / kRpLine) * kRpLine;  // keys leaving the bin
      // an overflowed bin: [done0 + CAP, done0 + fr) is in the region already; pad to a line
      const unsigned padn = over ? ((0u - (done0 + fr)) & (kRpLine - 1u)) : 0u;
      {
        const unsigned long long ob = __ballot(over);
        const unsigned long long hb = __ballot(fr * 2u > CAP);
        if (lane == 0 && (ob | hb))
          atomicAdd(&s_epoch[parity], ((unsigned)__popcll(ob) << 8) | (hb ? 1u : 0u));
      }
      unsigned long long todo = __ballot(nfl > 0);
      const unsigned sub = lane >> 4, l16 = lane & 15;
      while (todo) {
        int sel = -1;
#pragma unroll
        for (unsigned q = 0; q < 4; ++q) {
          const int bit = todo ? (int)__ffsll((long long)todo) - 1 : -1;
          if (todo) todo &= todo - 1;
          sel = (q == sub) ? bit : sel;
        }
        const unsigned src = sel >= 0 ? (unsigned)sel : 0u;
        const unsigned fb = __shfl(f, src, 64), nb_out = __shfl(nfl, src, 64);
        const unsigned done = __shfl(done0, src, 64), frb = __shfl(fr, src, 64), pb = __shfl(padn, src, 64);
        if (sel >= 0) {
          const unsigned bin = wave_base + (unsigned)sel;
          const int32_t *bsrc = bins + bin * CAP;
          const unsigned reach = frb > fb ? done + frb + pb : done + nb_out;  // region words in use afterwards
          if (reach > region_cap) {
            if (l16 == 0) atomicOr(&s_ovf, 1u);
          } else {
            int32_t *dst = regions + ((uint64_t)bin * kRpG + g) * region_cap + done;
#ifndef NVT_RP_NOFLUSHSTORE
            for (unsigned u = l16; u < nb_out; u += kRpLine) dst[u] = bsrc[u];
            if (l16 < pb) dst[frb + l16] = kEmpty;
#else
            if (done == 0xFFFFFFu) dst[0] = bsrc[0];
#endif
          }
          // the keys that stay (< kRpLine of them) move to the front of the bin
          const unsigned rem = fb - nb_out;
          int32_t keep = 0;
          if (l16 < rem) keep = bsrc[nb_out + l16];
          if (l16 < rem) bins[bin * CAP + l16] = keep;
        }
      }
      if (owner) {
        fill[t] = f - nfl;
        flushed[t] = over ? done0 + fr + padn : done0 + nfl;
      }
    }
  };

  // contiguous slab of 16-byte vectors per workgroup; the last vector of the column may be partial
  const uint64_t nvec = (n + 3) / 4, nfull = n / 4;
  const uint64_t per = (nvec + kRpG - 1) / kRpG;
  const uint64_t v_lo = (uint64_t)g * per;
  const uint64_t v_hi = v_lo + per < nvec ? v_lo + per : nvec;
  const uint64_t vf_hi = v_hi < nfull ? v_hi : nfull;   // full vectors of this slab: [v_lo, vf_hi)
  unsigned long long nulls = 0, notvalid = 0;
  unsigned long long empties = 0;   // (wave-uniform: the same in every lane)
  // U vectors per thread per round: the rounds of a workgroup are separated by two barriers, so
  // the load latency of a round is exposed unless the next round's loads are already in flight.
  // Round 6: they never were -- the loads sat under `if (v < v_hi)` / `if (valid)` and the
  // validity byte was shifted right behind its load, which the compiler answers with
  // s_waitcnt vmcnt(0) after EVERY load (ISA of the round-5 kernel: 20 of 20).  The loads of a
  // round are now unconditional (a vector behind the slab reads the slab's last full vector again
  // and is ignored: `present` comes from the index, not from the data; a column without a bitmap
  // reads its bytes from the keys and ORs them with 0xFF) and nothing touches what they return
  // before the round that uses it.  The partial vector at the end of the column (one thread of
  // one workgroup) is appended in a round of its own behind the loop.
  const uint8_t *vsrc = valid != nullptr ? valid : reinterpret_cast<const uint8_t *>(keys);
  const unsigned vor = valid != nullptr ? 0u : 0xFFu;
  const uint64_t v_last = vf_hi > v_lo ? vf_hi - 1 : 0;   // (vf_hi == v_lo: no round runs)
  int4 npack[U];
  unsigned nbyte[U];
  auto issue = [&](uint64_t v0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t v = v0 + (uint64_t)u * kRpBS;
      v = v < v_last ? v : v_last;
      npack[u] = reinterpret_cast<const int4 *>(keys)[v];
      nbyte[u] = (unsigned)vsrc[v >> 1];   // ((v * 4) >> 3)
    }
  };
  bool tail_round = false;   // (the partial vector, once, behind the full vectors)
  int4 tpack = make_int4(0, 0, 0, 0);
  unsigned tok = 0, tin = 0;
  if (nfull < nvec && nfull >= v_lo && nfull < v_hi) {   // (uniform: this workgroup owns it)
    tail_round = true;
    if (threadIdx.x == 0) {
      int32_t kk[4] = {0, 0, 0, 0};
      for (int j = 0; j < 4; ++j) {
        const uint64_t i = nfull * 4 + j;
        if (i < n) {
          tin |= 1u << j;
          if (bit_valid(valid, i)) {
            kk[j] = keys[i];
            tok |= 1u << j;
          }
        }
      }
      tpack = make_int4(kk[0], kk[1], kk[2], kk[3]);
    }
  }
  unsigned R = 1, since = 0, epoch = 0;
  if (vf_hi > v_lo) issue(v_lo + threadIdx.x);
  for (uint64_t v0 = v_lo; v0 < vf_hi || tail_round; v0 += (uint64_t)kRpBS * U) {
    const bool is_tail = !(v0 < vf_hi);   // (uniform)
    int32_t kv[4 * U];
    unsigned oks[U], ins[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t v = v0 + (uint64_t)u * kRpBS + threadIdx.x;
      const bool present = !is_tail && v < vf_hi;
      kv[4 * u + 0] = npack[u].x;
      kv[4 * u + 1] = npack[u].y;
      kv[4 * u + 2] = npack[u].z;
      kv[4 * u + 3] = npack[u].w;
      oks[u] = present ? (((nbyte[u] | vor) >> (((unsigned)v & 1u) * 4u)) & 0xFu) : 0u;
      ins[u] = present ? 0xFu : 0u;
    }
    if (is_tail) {
      tail_round = false;
      kv[0] = tpack.x;
      kv[1] = tpack.y;
      kv[2] = tpack.z;
      kv[3] = tpack.w;
      oks[0] = tok;
      ins[0] = tin;
    } else {
      issue(v0 + (uint64_t)kRpBS * U + threadIdx.x);   // (clamped: the last round reads its own again)
    }
    // Branch-free classification of the round's keys (the per-key if / else ladder compiled to
    // as many exec-mask instructions as there was arithmetic: ~190 instructions per key).  A null
    // (or absent) row is looked up as the empty key, which is in no bucket: every key reads its
    // hot bucket; hits add 1 to their counter, every other lane adds 0 to a scratch word of its
    // own; `pend` collects the misses that are not the empty key.  Empty keys are counted per
    // WAVE (ballot + scalar popcount): sentinel rows = empty keys seen - rows without a valid bit.
    unsigned pend = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned ok = oks[u], in = ins[u];
      nulls += __popc(in & ~ok);
      notvalid += 4u - __popc(ok);
      int32_t kp[4];
      int4 hb[4];
      uint32_t sa[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int32_t m = (int32_t)(ok << (31 - j)) >> 31;   // all ones: row j carries a key
        kp[j] = (kv[4 * u + j] & m) | (kEmpty & ~m);
        sa[j] = hot_bucket_r(kp[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int2 kk2 = *reinterpret_cast<const int2 *>(&hot[sa[j]]);   // (the keys: 8 of the 16 bytes)
        hb[j].x = kk2.x;
        hb[j].y = kk2.y;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = 4 * u + j;
        const int32_t key = kp[j];
        const bool hx = hb[j].x == key, hy = hb[j].y == key;
        const bool hit = hx | hy;
        const bool is_e = key == kEmpty;
        empties += (unsigned)__popcll(__ballot(is_e));
        unsigned *cnt = reinterpret_cast<unsigned *>(&hot[sa[j]]) + (hy ? 3 : 2);
        atomicAdd(hit ? cnt : &scratch[lane], hit ? 1u : 0u);
        pend |= ((!hit & !is_e) ? 1u : 0u) << q;
      }
    }
    // append the cold keys; a key whose bin is full goes straight to its place in the region
    if (s_ovf) pend = 0;  // a region overflowed: the column is rerun on a hash path anyway
#pragma unroll
    for (int q = 0; q < 4 * U; ++q) {
      if ((pend >> q) & 1) {
        const unsigned bkt = rp_fine<PW>(map, kv[q]) >> 14;
        const unsigned pos = atomicAdd(&fill[bkt], 1u);
        if (pos < CAP) {
          bins[bkt * CAP + pos] = kv[q];
        } else {
          const unsigned at = flushed[bkt] + pos;   // (flushed[] only changes between the barriers of a flush)
          if (at < region_cap)
            regions[((uint64_t)bkt * kRpG + g) * region_cap + at] = kv[q];
          else
            atomicOr(&s_ovf, 1u);
        }
      }
    }
    // Epochs: the barrier pair + flush cost 20-25 % of this loop when they ran every round (phase
    // timers, round 6).  R doubles (up to 8) after an epoch in which at most NB / 16 bins overflowed
    // into direct stores and halves after one in which more than NB / 4 did (swept on C1 / C20 /
    // C23: NB / 256 .. NB / 8 to grow; a direct store costs less than the barriers it saves).
    if (++since >= R) {
      __syncthreads();
      flush_bins(false, epoch & 1u);
      if (threadIdx.x == 0) s_epoch[(epoch + 1u) & 1u] = 0;
      __syncthreads();
      const unsigned fl = s_epoch[epoch & 1u];
#ifndef NVT_RP_GROW_DIV
#define NVT_RP_GROW_DIV 16u
#endif
#ifndef NVT_RP_SHRINK_DIV
#define NVT_RP_SHRINK_DIV 4u
#endif
      if ((fl >> 8) > NB / NVT_RP_SHRINK_DIV) R = R > 1u ? R / 2u : 1u;
      else if ((fl >> 8) <= NB / NVT_RP_GROW_DIV) R = R < kRpMaxEpoch ? R * 2u : R;
      ++epoch;
      since = 0;
    }
  }
  __syncthreads();   // (the loop may end inside an epoch)
  NVT_PTM();
  flush_bins(true, epoch & 1u);
  __syncthreads();
  NVT_PTM();
  for (unsigned b = threadIdx.x; b < NB; b += kRpBS) fills[(uint64_t)b * kRpG + g] = flushed[b];
  for (int i = threadIdx.x; i < kHotBucketsR; i += kRpBS) {
    const int4 hbk = hot[i];
    reinterpret_cast<uint2 *>(hot_cnt + (uint64_t)g * kHotSlotsR)[i] = make_uint2((unsigned)hbk.z, (unsigned)hbk.w);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    nulls += __shfl_down(nulls, off, 64);
    notvalid += __shfl_down(notvalid, off, 64);
  }
  if (lane == 0) {
    if (nulls) atomicAdd(&s_nulls, nulls);
    const unsigned long long sent = empties - notvalid;   // (of this wave)
    if (sent) atomicAdd(&s_sent, sent);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_nulls) atomicAdd((unsigned long long *)&state[NVT_ST_NULLS], s_nulls);
    if (s_sent) atomicAdd((unsigned long long *)&state[NVT_ST_SENTINEL], s_sent);
    if (s_ovf) atomicOr((unsigned long long *)&state[NVT_ST_OVERFLOW], 1ull | NVT_OVF_REGION);
    if (g == 0) atomicAdd((unsigned long long *)&state[NVT_ST_ROWS], (unsigned long long)n);
  }
#ifdef NVT_RP_PART_TIMING
  __syncthreads();
  NVT_PTM();
  if (threadIdx.x == 0)
    for (int q = 1; q < ptmi; ++q)
      atomicAdd((unsigned long long *)&state[9 + q], (unsigned long long)(ptm[q] - ptm[q - 1]));
#endif
}

// totals per hot slot = column sums of the per-workgroup counters (64 slots per workgroup x 16
// row groups, every load of a thread in flight; cf. hot_reduce_kernel of nvt_dense_count.hip)
constexpr int kTotGroups = 16;
__global__ __launch_bounds__(64 * kTotGroups) void hot_totals_kernel(
    const unsigned *__restrict__ hot_cnt, int nblocks, unsigned *__restrict__ hot_tot) {
  __shared__ unsigned part[kTotGroups][64];
  const unsigned l = threadIdx.x & 63, g = threadIdx.x >> 6;
  const unsigned slot = blockIdx.x * 64 + l;
  unsigned t = 0;  // < 2^32: the column has fewer rows than that
#pragma unroll 16
  for (int b = (int)g; b < nblocks; b += kTotGroups) t += hot_cnt[(uint64_t)b * kHotSlotsR + slot];
  part[g][l] = t;
  __syncthreads();
  if (g != 0) return;
  unsigned tot = 0;
#pragma unroll
  for (int q = 0; q < kTotGroups; ++q) tot += part[q][l];
  hot_tot[slot] = tot;
}

// ---------------------------------------------------------------------------------------------
// pass 2: one workgroup per bucket -> key-ordered (key, count) entries
// ---------------------------------------------------------------------------------------------
template <bool PW>
__global__ __launch_bounds__(kRpBS) void rp_count_kernel(
    const int32_t *__restrict__ regions, const uint32_t *__restrict__ fills, uint32_t region_cap,
    const int32_t *__restrict__ aux, const unsigned *__restrict__ hot_tot, int nb_log2,
    unsigned long long *status, unsigned *ticket, int32_t *__restrict__ out_keys,
    int64_t *__restrict__ out_cnt, uint64_t out_cap, unsigned *cls_hist,
    unsigned long long *__restrict__ range_table, uint64_t *state) {
  constexpr int NSL = kRpSlots + kRpTail;
  constexpr int NW = kRpBS / kWave;
  __shared__ int32_t lkeys[NSL];
  __shared__ unsigned lcnt[NSL];
  __shared__ unsigned run_len[kRpG];
  __shared__ unsigned hist[256];
  __shared__ unsigned wtot[NW];
  __shared__ unsigned lovf, s_b, s_bad, s_mx;
  __shared__ unsigned long long s_base;
  if (threadIdx.x == 0) {
    s_b = atomicAdd(ticket, 1u);
    lovf = 0;
    s_mx = 0;
  }
  for (int i = threadIdx.x; i < NSL; i += kRpBS) {
    lkeys[i] = kEmpty;
    lcnt[i] = 0;
  }
  if (threadIdx.x < 256) hist[threadIdx.x] = 0;
  __syncthreads();
  const unsigned b = s_b, NB = 1u << nb_log2;
  const unsigned lane = lane_id(), w = threadIdx.x / kWave;
  RangeMap map = load_map(aux);
  __shared__ uint32_t s_pieces[kRpPwWords];
  stage_pieces(map, s_pieces, threadIdx.x, kRpBS);  // (barriers follow before the first use)
#ifdef NVT_RANGE_TIMING
  long long tm[8];
  int tmi = 0;
#define NVT_TM() do { if (threadIdx.x == 0) tm[tmi++] = clock64(); } while (0)
#else
#define NVT_TM() do {} while (0)
#endif
  NVT_TM();
  // (pass 1 overflowed a region: the run lengths are not to be trusted, nothing is gathered)
  const bool skip = (state[NVT_ST_OVERFLOW] & 1ull) != 0;
  if (threadIdx.x < kRpG) {
    const unsigned len = fills[(uint64_t)b * kRpG + threadIdx.x];
    run_len[threadIdx.x] = (skip || len > region_cap) ? 0u : len;
  }
  __syncthreads();
  bool failed = skip;
  // probe chain from slot s on (the home slot has been looked at already when `first` is set)
  auto insert_from = [&](int32_t key, unsigned wgt, uint32_t s) {
#pragma unroll 4
    for (int p = 0; p < kRpProbe; ++p, ++s) {
      if (s >= (uint32_t)NSL) break;
      int32_t cur = lkeys[s];
      if (cur == kEmpty) {
        cur = atomicCAS(&lkeys[s], kEmpty, key);
        if (cur == kEmpty) cur = key;
      }
      if (cur == key) {
        atomicAdd(&lcnt[s], wgt);
        return;
      }
    }
    failed = true;
  };
  // the hot keys of this bucket's range: requested now, inserted behind the gather
  const unsigned hot_h0 = (unsigned)aux[NVT_RANGE_AUX_HOTSTART + b], hot_h1 = (unsigned)aux[NVT_RANGE_AUX_HOTSTART + b + 1];
  int32_t hot_key0 = kEmpty;
  unsigned hot_tot0 = 0;
  if (hot_h0 + threadIdx.x < hot_h1) {
    const unsigned slot = reinterpret_cast<const unsigned short *>(aux + NVT_RANGE_AUX_HOTORDER)[hot_h0 + threadIdx.x];
    hot_key0 = aux[slot];
    hot_tot0 = hot_tot[slot];
  }
  // The bucket's kRpG runs (~60-700 keys each) are gathered through ONE flat index over their
  // concatenation, cut into equal shares for the 16 waves (round 6: wave w used to take runs
  // w*16 .. w*16+15 whatever their lengths, and the workgroup waited 18 k of its 126 k cycles for
  // the wave with the longest runs): every lane has a key in every step, eight loads are in flight
  // per lane, and the home slots of a batch are read together before the (dependent) probe chains
  // start.  (One run after the other was a chain of 16 dependent global loads per wave.)
  {
    // exclusive prefix of the run lengths in LDS: s_pre[q] = keys in front of run q, s_pre[kRpG] =
    // all of them.  Which run a flat index f falls in is followed by a per-lane cursor -- f only
    // grows for a lane, so the cursor moves over each run once in the whole walk (one LDS compare
    // per key) after a binary search for the lane's first index.  The first version counted
    // `f >= pre[q]` over 16 runs and then selected pre[r] out of registers: ~60 vector
    // instructions per gathered key for one address (SQ counters, profiles/r05_notes.md).
    __shared__ unsigned s_pre[kRpG + 2];
    __shared__ unsigned s_wsum[kRpG / kWave];
    static_assert(kRpG % kWave == 0 && kRpG <= kRpBS, "one thread per run");
    {
      unsigned len = 0, inc = 0;
      if (threadIdx.x < kRpG) {
        len = run_len[threadIdx.x];
        inc = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = __shfl_up(inc, off, 64);
          if (lane >= (unsigned)off) inc += o;
        }
        if (lane == 63) s_wsum[w] = inc;
      }
      __syncthreads();
      if (threadIdx.x < kRpG) {
        unsigned wb = 0;
        for (unsigned q = 0; q < w; ++q) wb += s_wsum[q];
        s_pre[threadIdx.x] = wb + inc - len;
        if (threadIdx.x == kRpG - 1) s_pre[kRpG] = wb + inc;
      }
    }
    __syncthreads();
    const unsigned *pre = s_pre;
    const unsigned all_keys = pre[kRpG];
    const unsigned share = (all_keys + NW - 1) / NW;
    const unsigned f_lo = w * share < all_keys ? w * share : all_keys;
    const unsigned total = f_lo + share < all_keys ? f_lo + share : all_keys;   // this wave: [f_lo, total)
    const int32_t *rbase = regions + (uint64_t)b * kRpG * region_cap;
#ifndef NVT_RP_GB
#define NVT_RP_GB 8
#endif
    constexpr int GB = NVT_RP_GB;
    // cursor: run r covers flat indices [r_lo, r_hi); start = the run of this lane's first index
    unsigned r = 0;
    {
      const unsigned f = f_lo + lane < all_keys ? f_lo + lane : (all_keys ? all_keys - 1 : 0);
      unsigned lo = 0, hi = kRpG;   // largest r with pre[r] <= f
      while (hi - lo > 1) {
        const unsigned mid = (lo + hi) >> 1;
        if (pre[mid] <= f) lo = mid; else hi = mid;
      }
      r = lo;
    }
    unsigned r_lo = pre[r], r_hi = pre[r + 1];
    for (unsigned f0 = f_lo; f0 < total && !failed; f0 += GB * kWave) {
      int32_t kk[GB];
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const unsigned f = f0 + u * kWave + lane;
        kk[u] = kEmpty;
        if (f < total) {
          while (f >= r_hi) {   // (f < total <= pre[kRpG]: the cursor stops at the last run at the latest)
            ++r;
            r_lo = r_hi;
            r_hi = pre[r + 1];
          }
          kk[u] = rbase[(uint64_t)r * region_cap + (f - r_lo)];
        }
      }
      // Home slots: a plain read of all GB first (a key that is already at home -- the repeated
      // cold keys of the mid-cardinality columns -- needs nothing else: a returning CAS on a slot
      // that many lanes share serialises, C2 55 -> 74 us with CAS only), then ONE compare-and-swap
      // for the keys that found their home EMPTY (round 6: the first arrivals of a high-cardinality
      // column, 16 M rows over 6 M keys, all went down the per-lane walk below -- an 8-way select,
      // a dependent read and the CAS, ~150 instructions per key and wave).  Only true collisions
      // walk.  (A lane past the end holds kEmpty and is skipped.)
      uint32_t hs[GB];
      int32_t cur[GB];
#pragma unroll
      for (int u = 0; u < GB; ++u) hs[u] = rp_fine<PW>(map, kk[u]) & (kRpSlots - 1);
#pragma unroll
      for (int u = 0; u < GB; ++u) cur[u] = lkeys[hs[u]];
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        if (cur[u] == kEmpty && kk[u] != kEmpty) {
          cur[u] = atomicCAS(&lkeys[hs[u]], kEmpty, kk[u]);
          cur[u] = cur[u] == kEmpty ? kk[u] : cur[u];
        }
      }
#ifdef NVT_RP_SEQ_INSERT
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        if (kk[u] == kEmpty) continue;
        if (cur[u] == kk[u])
          atomicAdd(&lcnt[hs[u]], 1u);  // already at home: fire and forget
        else
          insert_from(kk[u], 1u, hs[u]);
      }
#else
      // Keys that are not at home yet walk their probe chains.  Key position by key position the
      // wave paid the LONGEST chain of its 64 lanes at each of the GB positions (a chain is a
      // loop of dependent LDS round trips); every lane walks ITS keys one after the other
      // instead, one probe per lane and step: as many steps as the unluckiest lane needs in all.
      unsigned todo = 0;
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        if (kk[u] == kEmpty) continue;
        if (cur[u] == kk[u])
          atomicAdd(&lcnt[hs[u]], 1u);  // already at home: fire and forget
        else
          todo |= 1u << u;
      }
      int32_t wkey = 0;
      uint32_t ws = 0;
      int wp = 0;
      bool walking = false;
      while (true) {
        if (!walking && todo) {
          const int u = (int)__ffs((int)todo) - 1;
          wkey = kk[0];
          ws = hs[0];
#pragma unroll
          for (int q = 1; q < GB; ++q) {
            wkey = u == q ? kk[q] : wkey;
            ws = u == q ? hs[q] : ws;
          }
          todo &= todo - 1u;
          ++ws;      // (the home slot holds another key: the CAS above saw it)
          wp = 1;
          walking = true;
        }
        if (!__any(walking)) break;
        if (walking) {
          if (ws >= (uint32_t)NSL || wp >= kRpProbe) {
            failed = true;
            walking = false;
          } else {
            int32_t c = lkeys[ws];
            if (c == kEmpty) {
              c = atomicCAS(&lkeys[ws], kEmpty, wkey);
              if (c == kEmpty) c = wkey;
            }
            if (c == wkey) {
              atomicAdd(&lcnt[ws], 1u);
              walking = false;
            } else {
              ++ws;
              ++wp;
            }
          }
        }
      }
#endif
    }
  }
  NVT_TM();
  // the hot keys of this range (indexed by bucket by the sample kernel), with the totals of
  // their counters
  // (the first of them per thread -- all of them unless a bucket holds > 1024 -- was loaded in front of
  // the gather: three dependent global loads, 15 k cycles of a 123 k-cycle workgroup when they sat here)
  {
    if (hot_key0 != kEmpty && hot_tot0 > 0 && !failed)
      insert_from(hot_key0, hot_tot0, rp_fine<PW>(map, hot_key0) & (kRpSlots - 1));
    const unsigned short *order = reinterpret_cast<const unsigned short *>(aux + NVT_RANGE_AUX_HOTORDER);
    for (unsigned jx = hot_h0 + threadIdx.x + kRpBS; jx < hot_h1; jx += kRpBS) {
      const unsigned slot = order[jx];
      const int32_t key = aux[slot];
      const unsigned tot = hot_tot[slot];
      if (key != kEmpty && tot > 0 && !failed) insert_from(key, tot, rp_fine<PW>(map, key) & (kRpSlots - 1));
    }
  }
  if (failed) atomicOr(&lovf, 1u);
  __syncthreads();
  NVT_TM();
  // workgroup-uniform (the global flag may be raised by another workgroup at any moment: it is
  // read ONCE, by one thread; threads that disagreed here used to split at the return below)
  if (threadIdx.x == 0)
    s_bad = (lovf != 0 ||
             (__hip_atomic_load(&state[NVT_ST_OVERFLOW], __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT) & 1ull))
                ? 1u : 0u;
  __syncthreads();
  const bool bad = s_bad != 0;
  // ---- entries of this bucket; bucket 0 also emits the sentinel key (smallest int32) ----
  const unsigned long long sent_rows = b == 0 ? state[NVT_ST_SENTINEL] : 0ull;
  const unsigned extra = sent_rows > 0 ? 1u : 0u;
  constexpr int ITER = (NSL + kRpBS - 1) / kRpBS;  // slot i = it * 1024 + thread: 17 sweeps
  // occupied slots per (sweep, wave) are recomputed in the second sweep; first the wave totals
  unsigned mine = 0;
  for (int it = 0; it < ITER; ++it) {
    const int i = it * kRpBS + (int)threadIdx.x;
    const bool occ = i < NSL && lkeys[i] != kEmpty;
    mine += (unsigned)__popcll(__ballot(occ));
  }
  if (lane == 0) wtot[w] = mine;  // (every lane of the wave holds the same sum)
  __syncthreads();
  unsigned E = extra;
  for (int q = 0; q < NW; ++q) E += wtot[q];
  // more entries than the table is meant to hold: clusters (and the ranking below) grow
  // quadratically -- report it like a table overflow
  const bool full = E > (unsigned)(kRpSlots / 4 * 3);
  if (bad || full) E = 0;
  // ---- decoupled look-back over the preceding buckets (wave 0) ----
  if (w == 0) {
    unsigned long long excl = 0;
    if (lane == 0)
      __hip_atomic_store(&status[b], (b == 0 ? kStPrefix : kStAgg) | (unsigned long long)E,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b > 0) {
      int top = (int)b - 1;  // look at buckets top, top-1, ... top-63
      while (true) {
        const int idx = top - (int)lane;
        unsigned long long v = kStPrefix;  // virtual bucket -1: prefix 0
        if (idx >= 0) {
          do {
            v = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef NVT_RP_SLEEP
#define NVT_RP_SLEEP 1
#endif
            if ((v >> 62) == 0) __builtin_amdgcn_s_sleep(NVT_RP_SLEEP);
          } while ((v >> 62) == 0);
        }
        const unsigned long long isp = __ballot((v >> 62) == 2);
        // nearest prefix (lowest lane); everything in front of it contributes its aggregate
        const int first = isp ? (int)__ffsll((long long)isp) - 1 : 64;
        unsigned long long add = ((int)lane <= first) ? (v & kStMask) : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) add += __shfl_down(add, off, 64);
        excl += __shfl(add, 0, 64);
        if (isp) break;
        top -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(&status[b], kStPrefix | (excl + E), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_base = excl;
  }
  __syncthreads();
  NVT_TM();
  const unsigned long long base = s_base;
  if (b == NB - 1 && threadIdx.x == 0) {
    if (base + E > out_cap) {
      atomicOr((unsigned long long *)&state[NVT_ST_OVERFLOW], 2ull);
      state[NVT_ST_NEED] = base + E;  // the relaunch can be sized exactly
    } else {
      state[NVT_ST_OCCUPIED] = base + E;
    }
  }
  if (bad || full) {
    if (threadIdx.x == 0 && (lovf != 0 || full))
      atomicOr((unsigned long long *)&state[NVT_ST_OVERFLOW],
               1ull | (lovf != 0 ? NVT_OVF_PROBE : 0ull) | (full ? NVT_OVF_FULL : 0ull));
    return;
  }
  if (base + E > out_cap) return;  // reported by the last bucket
  if (extra && threadIdx.x == 0) {
    out_keys[base] = kEmpty;
    out_cnt[base] = (int64_t)sent_rows;
    const unsigned cls = sent_rows < 255 ? (unsigned)sent_rows : 255u;
    atomicAdd(&hist[cls], 1u);
    unsigned long long *gm = reinterpret_cast<unsigned long long *>(&state[NVT_ST_MAXCOUNT]);
    atomicMax(gm, sent_rows);
  }
  // ---- ranked emission: position = occupied slots before the cluster + rank inside it ----
  // occupied slots in front of slot i = it * 1024 + w * 64 + lane: exclusive prefix over the
  // (sweep, wave) groups in slot order, computed once (17 x 16 groups)
  __shared__ unsigned occ_pre[ITER * NW];
  __shared__ unsigned occ_wsum[8];
  for (int it = 0; it < ITER; ++it) {
    const int i = it * kRpBS + (int)threadIdx.x;
    const bool occ = i < NSL && lkeys[i] != kEmpty;
    const unsigned c = (unsigned)__popcll(__ballot(occ));
    if (lane == 0) occ_pre[it * NW + w] = c;
  }
  __syncthreads();
  {
    constexpr int NG = ITER * NW;                 // 272 groups
    constexpr int SW = (NG + kWave - 1) / kWave;  // scanned by the first 5 waves
    unsigned v = 0, inc = 0;
    if (w < (unsigned)SW) {
      v = threadIdx.x < (unsigned)NG ? occ_pre[threadIdx.x] : 0u;
      inc = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off, 64);
        if (lane >= (unsigned)off) inc += o;
      }
      if (lane == 63) occ_wsum[w] = inc;
    }
    __syncthreads();
    if (w < (unsigned)SW && threadIdx.x < (unsigned)NG) {
      unsigned wb = 0;
      for (unsigned q = 0; q < w; ++q) wb += occ_wsum[q];
      occ_pre[threadIdx.x] = extra + wb + inc - v;
    }
    __syncthreads();
  }
  NVT_TM();
  unsigned mx = 0;
  for (int it = 0; it < ITER; ++it) {
    const int i = it * kRpBS + (int)threadIdx.x;
    const int32_t k = i < NSL ? lkeys[i] : kEmpty;
    const bool occ = k != kEmpty;
    unsigned long long dump = kEncEmptySlot;
    const unsigned long long bal = __ballot(occ);
    const unsigned front = occ_pre[it * NW + w];
    if (occ) {
      const unsigned p_i = front + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
      // cluster of slot i: walk left to its first slot, right to its end, four slots per step
      // (independent LDS reads: the one-slot-at-a-time walk was a chain of dependent reads as
      // long as the longest cluster among the 64 lanes).  rank = smaller keys in the cluster.
      unsigned rank = 0;
      int s = i;
      while (true) {
        int32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = s - 1 - q >= 0 ? lkeys[s - 1 - q] : kEmpty;
        int run = 0;
        bool open = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          open = open && o[q] != kEmpty;
          run += open ? 1 : 0;
          rank += (open && o[q] < k) ? 1u : 0u;
        }
        s -= run;
        if (run < 4) break;
      }
      int e = i + 1;
      while (true) {
        int32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = e + q < NSL ? lkeys[e + q] : kEmpty;
        int run = 0;
        bool open = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          open = open && o[q] != kEmpty;
          run += open ? 1 : 0;
          rank += (open && o[q] < k) ? 1u : 0u;
        }
        e += run;
        if (run < 4) break;
      }
      const uint64_t pos = base + p_i - (unsigned)(i - s) + rank;
      const unsigned c = lcnt[i];
#ifndef NVT_RP_NOWRITE
      out_keys[pos] = k;
      out_cnt[pos] = (int64_t)c;
#else
      if (pos == 0xFFFFFFFFFFull) out_keys[0] = k;
#endif
#ifndef NVT_RP_NOHIST
      atomicAdd(&hist[c < 255u ? c : 255u], 1u);
#endif
      mx = c > mx ? c : mx;
      dump = ((unsigned long long)(uint32_t)pos << 32) | (uint32_t)k;
    }
    // the table as it stands in LDS becomes this bucket's region of the encode table: slot ->
    // {key, position in the key-ordered list}; the ordering pass turns positions into labels
#ifndef NVT_RP_NODUMP
    if (range_table != nullptr && i < NSL) range_table[(uint64_t)b * NSL + i] = dump;
#else
    if (range_table != nullptr && i < NSL && dump == 0x1234567ull) range_table[(uint64_t)b * NSL + i] = dump;
#endif
  }
  NVT_TM();
  // largest count of the bucket: waves -> LDS -> ONE device atomic per workgroup behind the last
  // barrier.  (Round 6, phase timers: every wave read state[NVT_ST_MAXCOUNT] with a device-scope
  // atomic load and raised it with atomicMax -- 16 waves x NB workgroups on one address -- and
  // then waited for it in front of the barrier: 20-36 k of a workgroup's 105-145 k cycles.)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if (lane == 0 && mx > 0) atomicMax(&s_mx, mx);
  if (range_table != nullptr && b == NB - 1 && threadIdx.x < kRpGuard)
    range_table[(uint64_t)NB * NSL + threadIdx.x] = kEncEmptySlot;
  __syncthreads();
  NVT_TM();
#ifdef NVT_RANGE_TIMING
  if (threadIdx.x == 0)   // (summed over the buckets: tools/rp_phase_probe.py divides by NB)
    for (int q = 1; q < tmi; ++q)
      atomicAdd((unsigned long long *)&state[9 + q], (unsigned long long)(tm[q] - tm[q - 1]));
#endif
  if (threadIdx.x == 0 && s_mx > 0)
    atomicMax(reinterpret_cast<unsigned long long *>(&state[NVT_ST_MAXCOUNT]), (unsigned long long)s_mx);
  if (threadIdx.x < 256) {
    const unsigned h = hist[threadIdx.x];
    if (h) atomicAdd(&cls_hist[threadIdx.x], h);
    if (threadIdx.x == 255 && h)
      atomicAdd((unsigned long long *)&state[NVT_ST_BIG], (unsigned long long)h);
  }
}

inline uint64_t al16(uint64_t x) { return (x + 15) & ~15ull; }

}  // namespace

uint32_t range_region_cap(uint64_t n, int nb_log2) {
  const uint64_t rows_per_wg = (n + kRpG - 1) / kRpG;
  const uint64_t c = 2 * (rows_per_wg >> nb_log2) + 64;
  return (uint32_t)((c + kRpLine - 1) / kRpLine * kRpLine);
}

struct RangeWs {
  int32_t *regions;
  uint32_t *fills;
  unsigned *hot_cnt, *hot_tot;
  unsigned long long *status;
  unsigned *ticket;
};

uint64_t range_ws_layout(uint64_t n, int nb_log2, char *base, RangeWs *ws) {
  uint64_t off = 0;
  auto take = [&](uint64_t bytes) {
    char *p = base ? base + off : nullptr;
    off += al16(bytes);
    return p;
  };
  RangeWs w;
  const uint64_t NB = 1ull << nb_log2;
  w.regions = (int32_t *)take(NB * kRpG * range_region_cap(n, nb_log2) * 4);
  w.fills = (uint32_t *)take(NB * kRpG * 4);
  w.hot_cnt = (unsigned *)take((uint64_t)kRpG * kHotSlotsR * 4);
  w.hot_tot = (unsigned *)take((uint64_t)kHotSlotsR * 4);
  w.status = (unsigned long long *)take(NB * 8 + 64);  // + ticket
  w.ticket = (unsigned *)(w.status ? (char *)w.status + NB * 8 : nullptr);
  if (ws) *ws = w;
  return off;
}

uint64_t range_count_ws_bytes(uint64_t n, int nb_log2) {
  return range_ws_layout(n, nb_log2, nullptr, nullptr);
}

// aux = the column's int32[NVT_RANGE_AUX_WORDS]: hot image (sampled, with the range parameters,
// by hot_sample_kernel ahead of this call) and the class histogram (cleared here)
int range_count_i32(const int32_t *keys, const uint8_t *valid, uint64_t n, int nb_log2, void *wsp,
                    int32_t *aux, int32_t *out_keys, int64_t *out_cnt, uint64_t out_cap,
                    void *range_table, uint64_t *state, hipStream_t s, bool pieces) {
  NVT_CHECK_ARG(nb_log2 >= 6 && nb_log2 <= kRpMaxNbLog2, "range path: 64 .. 1024 buckets");
  NVT_CHECK_ARG(aux != nullptr, "range path: the column needs its aux block (hot_image)");
  NVT_PROF("dense_count_r9", n * 4, s);
  RangeWs w;
  range_ws_layout(n, nb_log2, (char *)wsp, &w);
  const unsigned NB = 1u << nb_log2;
  const uint32_t cap = range_region_cap(n, nb_log2);
  static const bool debug = ab_env("NVT_RANGE_DEBUG") != nullptr;
  auto mark = [&](const char *what) {
    if (debug) {
      hipError_t e = hipStreamSynchronize(s);
      fprintf(stderr, "[range n=%llu nb=%u cap=%u] %s: %s\n", (unsigned long long)n, NB, cap, what,
              hipGetErrorString(e));
      fflush(stderr);
    }
  };
  mark("begin");
  if (debug) {
    int32_t prm[5];
    (void)hipMemcpy(prm, aux + NVT_RANGE_AUX_LO, sizeof(prm), hipMemcpyDeviceToHost);
    fprintf(stderr, "[range] map: ulo=%u span=%u mul=%llu sh=%d; ws=%p regions=%p fills=%p hot_cnt=%p aux=%p\n",
            (unsigned)prm[0], (unsigned)prm[1],
            (unsigned long long)(uint32_t)prm[2] | ((unsigned long long)(uint32_t)prm[3] << 32), prm[4],
            wsp, (void *)w.regions, (void *)w.fills, (void *)w.hot_cnt, (void *)aux);
  }
  unsigned *hist = (unsigned *)(aux + NVT_RANGE_AUX_HIST);
#ifdef NVT_RP_MEMSET
  NVT_CHECK_HIP(hipMemsetAsync(w.status, 0, (uint64_t)NB * 8 + 64, s));
  NVT_CHECK_HIP(hipMemsetAsync(aux + NVT_RANGE_AUX_HIST, 0, 256 * 4, s));
#endif
#define NVT_RP_PART(NBL)                                                                            \
  do {                                                                                              \
    if (pieces)                                                                                     \
      rp_partition_kernel<NVT_RANGE_U, NBL, true><<<kRpG, kRpBS, 0, s>>>(                           \
          keys, valid, n, aux, nb_log2, cap, w.regions, w.fills, w.hot_cnt, state, w.status, hist); \
    else                                                                                            \
      rp_partition_kernel<NVT_RANGE_U, NBL, false><<<kRpG, kRpBS, 0, s>>>(                          \
          keys, valid, n, aux, nb_log2, cap, w.regions, w.fills, w.hot_cnt, state, w.status, hist); \
  } while (0)
  static const bool rt_nb = ab_env("NVT_RANGE_RT_NB") != nullptr;  // (A/B: the run-time variant)
  if (rt_nb) NVT_RP_PART(0);
  else if (nb_log2 == 8) NVT_RP_PART(8);
  else if (nb_log2 == 9) NVT_RP_PART(9);
  else if (nb_log2 == 10) NVT_RP_PART(10);
  else NVT_RP_PART(0);
#undef NVT_RP_PART
  NVT_CHECK_LAUNCH();
  mark("partition");
  hot_totals_kernel<<<kHotSlotsR / 64, 64 * kTotGroups, 0, s>>>(w.hot_cnt, kRpG, w.hot_tot);
  NVT_CHECK_LAUNCH();
  mark("totals");
  if (pieces)
    rp_count_kernel<true><<<NB, kRpBS, 0, s>>>(w.regions, w.fills, cap, aux, w.hot_tot, nb_log2, w.status,
                                               w.ticket, out_keys, out_cnt, out_cap,
                                               (unsigned *)(aux + NVT_RANGE_AUX_HIST),
                                               (unsigned long long *)range_table, state);
  else
    rp_count_kernel<false><<<NB, kRpBS, 0, s>>>(w.regions, w.fills, cap, aux, w.hot_tot, nb_log2, w.status,
                                                w.ticket, out_keys, out_cnt, out_cap,
                                                (unsigned *)(aux + NVT_RANGE_AUX_HIST),
                                                (unsigned long long *)range_table, state);
  NVT_CHECK_LAUNCH();
  mark("count");
  return NVT_OK;
}

}  // namespace nvt
