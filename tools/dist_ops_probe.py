"""Cost of the torch plumbing of dist.merge_counts_many on an exclusive GPU: 31 M (key, count)
entries in 26 columns (the bench's per-rank lists), G = 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvtabular_amd import kernels as K

dev = torch.device("cuda", 0)
G, ncol = 8, 26
lens = [6_200_000, 39_000, 17_000, 7_400, 20_000, 3, 7_100, 1_500, 63, 6_100_000, 1_570_000, 368_000, 10,
        2_200, 11_900, 155, 4, 976, 14, 3_000_000, 5_700_000, 5_000_000, 560_000, 12_900, 108, 36]
g = torch.Generator(device=dev).manual_seed(1)
cols = [(torch.randint(-2**31, 2**31 - 1, (n,), device=dev, dtype=torch.int64, generator=g).to(torch.int32).sort().values,
         torch.randint(1, 100, (n,), device=dev, dtype=torch.int64, generator=g)) for n in lens]
tot = sum(lens)


def T(name, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {1e3 * (time.perf_counter() - t0) / reps:8.3f} ms", flush=True)
    return out


k64s = T("to(int64) x26", lambda: [k.to(torch.int64) for k, _ in cols])
T("min/max/sum x26 -> rng", lambda: [(k.min(), k.max(), c.sum()) for (k, c) in cols])
owns = T("range_owner x26", lambda: [((k - int(-2**31)) // (2**32 // G)).clamp_(0, G - 1) for k in k64s])
dest = T("dest = own*ncol+j, cat", lambda: torch.cat([o * ncol + j for j, o in enumerate(owns)]))
rows = T("pack rows (cat, shift, or)", lambda: (torch.cat([c for _, c in cols]) << 32) | (torch.cat(k64s) & 0xFFFFFFFF))
words = T("K.order_rows", lambda: K.order_rows(int(dest.numel()), dev, gid=dest, ngroups=G * ncol))
order = words & 0xFFFFFFFF
T("rows[order]", lambda: rows[order].contiguous())
T("bincount(dest, 208)", lambda: torch.bincount(dest, minlength=G * ncol))
sd = dest[order]
T("searchsorted alternative", lambda: torch.searchsorted(sd, torch.arange(G * ncol + 1, device=dev)))
off = [0]
per = tot // (G * ncol)
for s in range(G * ncol):
    off.append(off[-1] + per)
recv = rows[: off[-1]]
T("merge_counts_sorted", lambda: K.merge_counts_sorted(recv, off, ncol))
merged = K.merge_counts_sorted(recv, off, ncol)
T("pack merged x26 + cat", lambda: torch.cat([(mc << 32) | (mk.to(torch.int64) & 0xFFFFFFFF) for mk, mc in merged]))
every = torch.cat([rows] * 2)[: 60_000_000]
seg = every.numel() // (G * ncol)
T("per-column cat of G slices + unpack", lambda: [
    (lambda s_: (((s_ << 32) >> 32).to(torch.int32), s_ >> 32))(torch.cat([every[(r * ncol + j) * seg:(r * ncol + j + 1) * seg] for r in range(G)]))
    for j in range(ncol)])
cnts = [every[j * seg * G:(j + 1) * seg * G] >> 32 for j in range(ncol)]
T("class_hist x26", lambda: [K.class_hist(c) for c in cnts])

print("---- device path (nvt_exchange_*) ----")
tabs = [(k, c) for k, c in cols]
xb = K.ExchangeBatch(tabs)
T("ExchangeBatch.ranges", lambda: xb.ranges())
los = [int(k.min()) if k.numel() else 0 for k, _ in cols]
his = [int(k.max()) if k.numel() else 0 for k, _ in cols]
widths = [max(1, -(-(h - l + 1) // G)) for l, h in zip(los, his)]
mat = T("ExchangeBatch.hist", lambda: xb.hist(los, widths, G))
flat = mat.reshape(-1).cpu()
starts = torch.zeros(G * ncol, dtype=torch.int64)
starts[1:] = torch.cumsum(flat, 0)[:-1]
T("ExchangeBatch.scatter", lambda: xb.scatter(los, widths, G, starts.to(dev)))
T("merge_counts_sorted(want_packed)", lambda: K.merge_counts_sorted(recv, off, ncol, want_packed=True))
goff = [s * seg for s in range(G * ncol + 1)]
dst = [0] * (G * ncol)
for j in range(ncol):
    for r in range(G):
        dst[r * ncol + j] = (j * G + r) * seg
T("exchange_unpack", lambda: K.exchange_unpack(every[: goff[-1]], goff, dst, goff[-1]))

print("---- ordered exchange (key-sorted lists: slices in key order, owners merge sorted runs) ----")
from nvtabular_amd import dist as D
starts_d = starts.to(dev).view(G, ncol)
first_row = (torch.cumsum(mat, 0) - mat).contiguous()
T("ExchangeBatch.pack_ordered", lambda: xb.pack_ordered(los, widths, G, starts_d, first_row))
# what an owner receives: per column G key-ordered runs of ~n / G entries with interleaved keys
pieces, roff = [], [0]
for src in range(G):
    for k, c in cols:
        pieces.append((c[src::G] << 32) | (k[src::G].to(torch.int64) & 0xFFFFFFFF))
        roff.append(roff[-1] + int(pieces[-1].numel()))
recv_o = torch.cat(pieces)
D.DISTRIBUTED_ORDER = False   # (its collectives need a process group; the labelling is timed below)
T("_merge_sorted_runs (unpack + merge tree + class hist + pack)", lambda: D._merge_sorted_runs(recv_o, roff, G, ncol))
T("  of which exchange_unpack", lambda: K.exchange_unpack(recv_o, roff, [roff[s] for s in range(G * ncol)], roff[-1]))
ka, ca = K.exchange_unpack(recv_o, roff, [roff[s] for s in range(G * ncol)], roff[-1])
lists = [[(ka[roff[src * ncol + j]:roff[src * ncol + j + 1]], ca[roff[src * ncol + j]:roff[src * ncol + j + 1]])
          for src in range(G)] for j in range(ncol)]
T("  of which merge_sorted_tree", lambda: K.merge_sorted_tree(lists))

# the owner's labelling of its merged shards (nvt_vocab_label_shard per column; ~1 / G of the union)
shard = K.merge_sorted_tree(lists)
diffs = torch.zeros(ncol, 256, dtype=torch.int32, device=dev)
labs = [torch.empty(int(k.numel()), dtype=torch.int32, device=dev) for k, _ in shard]
nbig = [int((c >= 255).sum()) for _, c in shard]
T("label_shard x26 (owner's shards)", lambda: [K.label_shard(k, c, diffs[j], nbig[j], labs[j])
                                               for j, (k, c) in enumerate(shard) if k.numel()])

print("---- what every rank does with the gathered union (8 ranks: ~2.7 x the per-rank lists) ----")
from nvtabular_amd import _lib
ulens = [min(int(n * 2.7), 40_000_000) if n > 100_000 else n for n in lens]
g2 = torch.Generator(device=dev).manual_seed(2)
ucols = []
for n in ulens:
    k = torch.randint(-2**31, 2**31 - 1, (int(n * 1.05) + 8,), device=dev, dtype=torch.int64, generator=g2).to(torch.int32)
    k = torch.unique(k)[:n]                                      # key-sorted, duplicate-free
    c = torch.randint(1, 400, (int(k.numel()),), device=dev, dtype=torch.int64, generator=g2)
    ucols.append((k, c))
torch.cuda.synchronize()
print("union entries", sum(int(k.numel()) for k, _ in ucols))


def finalize(with_labels):
    descs = (_lib.VocabCol * len(ucols))()
    keep = []
    for d, (k, c) in zip(descs, ucols):
        n = int(k.numel())
        hist = K.class_hist(c)
        labels = None
        if with_labels:
            labels = torch.arange(n, dtype=torch.int32, device=dev)   # (any permutation: timing only)
        ok, oc = torch.empty_like(k), torch.empty_like(c)
        tab = K.EncodeTable(ok, 3, unique=True, defer_build=True, range_table=None, flat=True)
        tab.fill_vocab_desc(d, oc, 400, src=(k, c, hist, int((c >= 255).sum()), labels))
        keep.append((tab, ok, oc, hist, labels))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K.check(_lib.load().nvt_vocab_finalize_many(descs, len(ucols), K.stream_ptr()), "finalize")
    for tab, *_ in keep:
        tab.wait_ready()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


for mode in (False, True, False, True):
    print(f"nvt_vocab_finalize_many, {'labels from the owners' if mode else 'ordering pass on every rank'}: "
          f"{finalize(mode):8.3f} ms", flush=True)
