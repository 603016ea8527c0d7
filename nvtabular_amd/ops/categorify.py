"""Categorify on MI355X.

Same constructor, semantics and on-disk artefacts as
/root/reference/nvtabular/ops/categorify.py (file:line cited inline); the
groupby-size, vocabulary sort and encode steps run as HIP kernels
(``nvt_count_*``, ``nvt_vocab_sort_*``, ``nvt_encode_*``; multi-column "combo"
groups through ``nvt_gb_*``).

Index layout (categorify.py:53-71): 0 pad, 1 null, [2, 2+nb) OOV / hash buckets,
[2+nb, ...) vocabulary in (count desc, value asc) order.
"""
from __future__ import annotations

import os
import warnings
from copy import deepcopy
from typing import Dict, List

import numpy as np
import pandas as pd
import torch

from .. import kernels as K
from ..device import DeviceColumn, DeviceFrame, as_device_frame, key_view
from ..schema import Tags
from ..selector import ColumnSelector
from .base import StatOperator

PAD_OFFSET = 0
NULL_OFFSET = 1
OOV_OFFSET = 2


def _make_name(*args, sep="_"):
    return sep.join(args)


def _emb_sz_rule(n_cat: int, minimum_size=16, maximum_size=512):
    """categorify.py:687-688"""
    return n_cat, min(max(minimum_size, round(1.6 * n_cat**0.56)), maximum_size)


def _pick(opt, name, default=None):
    if isinstance(opt, dict):
        return opt.get(name, default)
    return opt


MERGE_FAN_IN = 8
LAZY_FINALIZE = os.environ.get("NVT_LAZY_FINALIZE", "1") != "0"
PIPELINE_COUNTS = os.environ.get("NVT_PIPELINE_COUNTS", "1") != "0"
# (opt-in: measured 12.44 vs 12.15 ms per Criteo step -- the ordering kernels and the LDS-resident
# encodes slow each other down by more than the overlap buys; profiles/r04_notes.md)
SPLIT_FINALIZE = os.environ.get("NVT_SPLIT_FINALIZE", "0") == "1"


class _GroupFit:
    """Accumulated groupby-size state of one column group on this GPU."""

    def __init__(self, name: str, cols: List[str], combo: bool):
        self.name = name
        self.cols = cols
        self.combo = combo  # multi-column key tuple (nvt_gb_*) vs single key column
        self.table = None  # (keys, counts, max_count) dense list | K.GroupbyTable (combo)
        self.parts = []    # per-partition dense lists not merged into `table` yet
        self.sorted = None  # (keys tensor, info) of a KEY-SORTED list (range path); applies to
        #                     `table` only while table[0] is that very tensor
        self.nulls = 0
        self.valid_rows = 0  # non-null key rows seen (== sum of counts)
        self.hint = 1 << 12  # expected distinct keys per partition, learned as we go
        self.key_dtype = None
        self.src_dtypes: Dict[str, object] = {}
        self.strings: Dict[str, dict] = {}  # col -> {surrogate: str}


class Categorify(StatOperator):
    """Encode categorical columns as contiguous integers (categorify.py:59-343)."""

    def __init__(
        self,
        freq_threshold=0,
        out_path=None,
        cat_cache="host",
        dtype=None,
        on_host=True,
        encode_type="joint",
        name_sep="_",
        search_sorted=False,
        num_buckets=None,
        vocabs=None,
        max_size=0,
        single_table=False,
        cardinality_memory_limit=None,
        tree_width=None,
        split_out=1,
        split_every=8,
        defer_artifacts=False,
        tie_break="value",
        **kwargs,
    ):
        # categorify.py:226-241
        if "start_index" in kwargs:
            raise ValueError(
                "start_index is now deprecated. `Categorify` will always reserve index `0` for "
                "user-specific purposes, and will use index `1` for null values."
            )
        if "na_sentinel" in kwargs:
            raise ValueError(
                "na_sentinel is now deprecated. `Categorify` will always reserve index `1` for "
                "null values, and the following `num_buckets` indices for out-of-vocabulary values."
            )
        if kwargs:
            raise ValueError(f"Unrecognized key-word arguments: {kwargs}")
        if num_buckets and not (max_size or freq_threshold):
            warnings.warn(
                "You are setting num_buckets without using max_size or freq_threshold to restrict "
                "the number of distinct categories. Are you sure this is what you want?"
            )
        self.storage_name: Dict[str, str] = {}
        if encode_type not in ("joint", "combo"):
            raise ValueError(f"encode_type={encode_type} not supported.")
        if encode_type == "combo" and vocabs is not None:
            raise ValueError("Passing in vocabs is not supported with a combo encoding.")
        super().__init__()
        self.single_table = single_table
        self.freq_threshold = freq_threshold or 0
        self.out_path = out_path or "./"
        self.dtype = dtype
        self.on_host = on_host  # accepted, no effect (categorify.py:164-173: placement only)
        self.cat_cache = cat_cache
        self.encode_type = encode_type
        self.name_sep = name_sep
        self.search_sorted = search_sorted
        self.cardinality_memory_limit = cardinality_memory_limit
        self.split_every = split_every
        self.split_out = split_out
        if tree_width is not None:
            warnings.warn("tree_width is deprecated; use split_out/split_every", FutureWarning)
        if self.search_sorted and self.freq_threshold:
            raise ValueError(
                "cannot use search_sorted=True with anything else than the default freq_threshold"
            )
        if num_buckets == 0:
            raise ValueError(
                "For hashing num_buckets should be an int > 1, otherwise set num_buckets=None."
            )
        if isinstance(num_buckets, (dict, int)) or num_buckets is None:
            self.num_buckets = num_buckets
        else:
            raise ValueError(f"`num_buckets` must be dict or int, got type {type(num_buckets)}")
        if isinstance(max_size, (dict, int)) or max_size is None:
            self.max_size = max_size
        else:
            raise ValueError(f"max_size must be dict or int, got type {type(max_size)}")
        if freq_threshold and max_size:
            raise ValueError("cannot use freq_threshold param together with max_size param")
        if self.num_buckets is not None:
            warnings.warn(
                "Performing a hash-based transformation. Do not expect Categorify to be "
                "consistent on GPU and CPU with this num_buckets setting!"
            )
        # device-side state: storage_name -> encoder (built at fit_end or lazily from parquet)
        self._encoders: Dict[str, object] = {}
        self._cap_hints: Dict[str, int] = {}
        self._pending_counts: Dict[int, tuple] = {}  # id(fit state) -> (CountBatch, owners)
        # Engine extension (not in the reference): with defer_artifacts=True the
        # unique.*/meta.*.parquet files are written by flush_artifacts() /
        # Workflow.save() instead of inside fit, so a device-resident fit does no
        # host I/O.  Default False = reference behaviour (files exist after fit).
        self.defer_artifacts = defer_artifacts
        # Engine extension: order of categories with EQUAL counts.  "value" (default): count
        # descending, value ascending, sorted on the device -- deterministic.  "reference": the
        # (value, count) table is copied to the host and ordered by the reference's literal two
        # pandas sort_values calls (categorify.py:1300,1316; the second is pandas' default
        # unstable sort), so labels equal a reference run on the same host tie for tie.
        if tie_break not in ("value", "reference"):
            raise ValueError("tie_break must be 'value' or 'reference'")
        self.tie_break = tie_break
        self._pending: Dict[str, dict] = {}
        self._last_paths: Dict[str, int] = {}  # counting path of each column's last partition
        self._no_range = set()                 # columns on which the range path overflowed
        self._range_failures = []              # (column, NVT_OVF_* bits, distinct) of each overflow
        self._range_oks: Dict[str, int] = {}   # partitions each column counted on the range path
        self._flat_unchecked: List[str] = []   # flat range tables whose displacement is still unread
        self._no_flat = set()                  # vocabularies whose keys cluster: hashed tables
        self._range_pieces: Dict[str, object] = {}   # column -> int32[65] splitters (NVT_PATH_PIECES)
        self._range_bits_floor: Dict[str, int] = {}  # column -> log2 buckets that held it after an overflow
        self._lazy_finalize = None  # (groups, options, base) of a fit whose ordering is deferred
        self._writer_cache: Dict[str, bool] = {}
        self.vocabs = {}
        if vocabs is not None:
            self.vocabs = self.process_vocabs(vocabs)
        self.categories = deepcopy(self.vocabs)

    # BEFORE the operators whose fit_end is a scalar read-back (Normalize): the count states are
    # read back behind the counting kernels only (CountBatch._read_states), so the host digests
    # them -- a few hundred microseconds for the 26 Criteo columns -- while the moments kernel
    # that was queued behind the counting still runs, instead of after it
    fit_end_priority = -1

    def range_name(self, kind: str) -> str:
        return "Categorify_fit" if kind == "fit" else "Categorify_transform"  # categorify.py:345,477

    # ------------------------------------------------------------------ fit --
    def _groups(self, col_selector: ColumnSelector):
        """[(storage_name, [cols], combo?)] for the groups that still need fitting."""
        flat = []
        for g in col_selector.grouped_names:
            flat += list(g) if isinstance(g, tuple) else [g]
        if sorted(flat) != sorted(set(flat)) and self.encode_type == "joint":
            raise ValueError("Same column name included in multiple groups.")
        for group in col_selector.subgroups:
            if len(group.names) > 1:
                name = _make_name(*group.names, sep=self.name_sep)
                for col in group.names:
                    self.storage_name[col] = name
        out = []
        for g in col_selector.grouped_names:
            cols = list(g) if isinstance(g, tuple) else [g]
            name = _make_name(*cols, sep=self.name_sep)
            if name in self.categories and name in self.vocabs:
                continue  # user supplied vocabulary (categorify.py:382-390)
            combo = self.encode_type == "combo" and len(cols) > 1
            out.append((name, cols, combo))
        return out

    def fit_begin(self, col_selector: ColumnSelector):
        return {name: _GroupFit(name, cols, combo) for name, cols, combo in self._groups(col_selector)}

    def _group_keys(self, g: _GroupFit, frame: DeviceFrame):
        keys, valids = [], []
        for c in g.cols:
            col = frame[c]
            if col.fill is not None:
                col = col.materialize()
            k, v = key_view(col)
            keys.append(k)
            valids.append(v)
            g.src_dtypes.setdefault(c, "str" if col.strings is not None else col.dtype)
            if col.strings is not None:
                g.strings.setdefault(c, {}).update(col.strings)
        return keys, valids

    def fit_partition(self, state, col_selector, frame):
        frame, _ = as_device_frame(frame)
        specs = []   # (group, hkey, keys, valid) of every single-vocabulary column
        combos = []
        for g in state.values():
            keys, valids = self._group_keys(g, frame)
            if g.combo:
                combos.append((g, keys, valids))
                continue
            # joint groups: every column of the group feeds one vocabulary (categorify.py:972-981)
            if len({k.dtype for k in keys}) > 1:
                keys = [K.widen_i64(k) for k in keys]
            g.key_dtype = keys[0].dtype
            for ci, (k, v) in enumerate(zip(keys, valids)):
                specs.append((g, f"{g.name}#{ci}", k, v))

        def launch():
            # every column's count kernels are enqueued by ONE C call (top_level_groupby,
            # categorify.py:955); the one readback of all state words happens in
            # _absorb_pending (next partition / fit_end)
            jobs = [K.DenseCountJob(k, v, None, hint=self._cap_hints.get(hkey, 0),
                                    allow_range=hkey not in self._no_range,
                                    pieces=self._range_pieces.get(hkey),
                                    min_range_bits=self._range_bits_floor.get(hkey, 8))
                    for _, hkey, k, v in specs]
            with K.annotate("top_level_groupby"):
                return K.CountBatch(jobs), [(g, hkey) for g, hkey, _, _ in specs]

        # Once every column has a cardinality hint, this partition's counting is enqueued BEFORE
        # the previous partition's results are read back (the read-back waits for that partition's
        # own kernels only: CountBatch.results): the GPU goes from one partition's counting to the
        # next without waiting for the host to digest the results in between (0.5 ms per
        # partition on the Criteo workload).  A fresh fit reads first: it has no hints to launch on.
        ahead = None
        if specs and PIPELINE_COUNTS and id(state) in self._pending_counts and \
                all(hkey in self._cap_hints for _, hkey, _, _ in specs):
            ahead = launch()
        # the previous partition's counts are read back only now: its kernels (and whatever
        # other operators queued behind them) ran while the host prepared this partition
        self._absorb_pending(state)
        for g, keys, valids in combos:
            self._fit_partition_combo(g, keys, valids)
        if specs:
            self._pending_counts[id(state)] = ahead if ahead is not None else launch()

    def _absorb_pending(self, state):
        item = self._pending_counts.pop(id(state), None)
        if item is None:
            return
        batch, owners = item
        per_group = {}
        for (g, hkey), (dk, dc, nulls, info) in zip(owners, batch.results()):
            old_hint = self._cap_hints.get(hkey, 0)
            self._cap_hints[hkey] = max(64, info["distinct"])
            self._last_paths[hkey] = info["path"]
            if (info.get("range_bits_floor", 8) > self._range_bits_floor.get(hkey, 8)
                    and old_hint > 0 and info["distinct"] <= old_hint + old_hint // 4):
                # the buckets a GOOD hint asked for overflowed (keys not spread evenly): the next
                # partitions / fits of this column start with the bucket count that held it.  (A
                # launch without a hint, or with one the column outgrew, escalates for itself only.)
                self._range_bits_floor[hkey] = info["range_bits_floor"]
            if info.get("range_failed"):
                # one overflow does not condemn a column: the hot-key sample is a heuristic (a
                # frequent key that loses the race for its image bucket floods one region of the
                # partition pass -- about one 45 M-row partition in a hundred on the Criteo-shaped
                # columns).  Keys that are NOT spread over their range fail every time: banned
                # after two overflows making up more than a fifth of the attempts.
                self._range_failures.append((hkey, info.get("range_fail_bits", 0), info["distinct"]))
                fails = sum(1 for h, _, _ in self._range_failures if h == hkey)
                if fails >= 2 and 4 * fails > self._range_oks.get(hkey, 0):
                    self._no_range.add(hkey)
            elif info["path"] == K.PATH_RANGE:
                self._range_oks[hkey] = self._range_oks.get(hkey, 0) + 1
            if info.get("range_failed") and info.get("sorted_by_key") and K.USE_PIECES \
                    and dk.dtype == torch.int32:
                # the linear range map overflowed (keys not spread over their range: dense /
                # frequency-ordered ids) and the sort path delivered the exact key-ordered list:
                # splitters of a piecewise map from it; the column's next partitions / fits take
                # the range path with them, and the overflow is not held against the column
                sp = None if info.get("range_pieces") else K.range_splitters(dk, dc)
                if info.get("range_pieces"):
                    self._range_pieces.pop(hkey, None)   # overflowed WITH splitters: counts as a failure
                if sp is not None:
                    self._range_pieces[hkey] = sp
                    self._no_range.discard(hkey)
                    self._range_failures = [f for f in self._range_failures if f[0] != hkey]
            srt = bool(info.get("sorted_by_key")) and dk.dtype == torch.int32
            g.nulls += nulls
            g.valid_rows += info["rows"] - nulls
            per_group.setdefault(g.name, (g, []))[1].append((dk, dc, info["max_count"], srt, info))
        due = []
        for g, lists in per_group.values():
            # tree merge with the reference's fan-in (_mid_level_groupby, split_every = 8,
            # categorify.py:1054-1070): partial lists pile up and are merged eight at a time, not
            # after every partition
            g.parts.extend(lists)
            if len(g.parts) >= MERGE_FAN_IN:
                due.append(g)
            elif g.table is None and len(g.parts) == 1:
                self._adopt(g, g.parts.pop())
        self._merge_parts_many(due)

    @staticmethod
    def _adopt(g, part):
        """A per-partition list becomes the group's table as it is (with what its counting pass
        left for the finalisation: key order, class histogram, dumped range table)."""
        dk, dc, mx, srt, info = part
        g.table = (dk, dc, mx)
        g.sorted = (dk, info) if srt and info is not None else None

    def _merge_parts_many(self, groups):
        """_mid_level_groupby (categorify.py:1054-1070) for every group in ``groups``: the
        accumulated table and the pending per-partition lists become one table.

        KEY-SORTED int32 lists (range path, sort path -- every large column) are merged by the
        merge-path kernel, all groups in one call per tree level, and stay key-sorted: the
        vocabulary of a multi-partition fit is ordered by the same one-pass class scatter and
        gets the same flat range table as a single-partition fit.  Unordered lists (the
        LDS-resident columns: <= 11 k entries each) are concatenated and re-counted with weights,
        all groups in one nvt_dense_count_many call."""
        sorted_jobs, dense_jobs = [], []
        for g in groups:
            lists = list(g.parts)
            if g.table is not None:
                tab_sorted = g.sorted is not None and g.sorted[0] is g.table[0]
                lists.insert(0, tuple(g.table) + (tab_sorted, g.sorted[1] if tab_sorted else None))
            g.parts = []
            lists = [t for t in lists if int(t[0].numel())]
            if not lists:
                continue
            if len(lists) == 1:
                self._adopt(g, lists[0])
                continue
            if (K.MERGE_SORTED and all(t[3] and t[0].dtype == torch.int32 for t in lists)
                    and sum(int(t[0].numel()) for t in lists) <= K.MERGE_SORTED_MAX_TOTAL):
                sorted_jobs.append((g, lists))
            else:
                dense_jobs.append((g, lists))
        if sorted_jobs:
            merged = K.merge_sorted_tree([[(t[0], t[1]) for t in lists] for _, lists in sorted_jobs])
            for (g, lists), (k, c) in zip(sorted_jobs, merged):
                # (the sum of the lists' maxima bounds the largest merged count)
                g.table = (k, c, sum(int(t[2]) for t in lists))
                # cls_hist / n_big: filled in at fit_end (_complete_sorted_info), once
                g.sorted = (k, dict(sorted_by_key=True, cls_hist=None, n_big=None, merged_parts=True))
        if dense_jobs:
            outs = K.merge_dense_many([[t[:3] for t in lists] for _, lists in dense_jobs],
                                      hints=[self._cap_hints.get(g.name, 0) for g, _ in dense_jobs])
            for (g, _), tab in zip(dense_jobs, outs):
                g.table = tab
        for g in groups:
            if g.table is not None:
                self._cap_hints[g.name] = max(64, int(g.table[0].numel()))

    def _complete_sorted_info(self, groups):
        """Key-sorted tables that came out of the partition merge carry no class histogram yet
        (the counting kernels produce it for single lists): one nvt_class_hist per table, ONE
        read-back of the class-255 sizes."""
        todo = [g for g in groups
                if not g.combo and g.table is not None and g.sorted is not None
                and g.sorted[0] is g.table[0] and g.sorted[1].get("cls_hist") is None
                and int(g.table[0].numel())]
        if not todo:
            return
        for g in todo:
            g.sorted[1]["cls_hist"] = K.class_hist(g.table[1])
        nb = K.read_back(torch.stack([g.sorted[1]["cls_hist"][255] for g in todo]).to(torch.int64))
        for g, v in zip(todo, nb.tolist()):
            g.sorted[1]["n_big"] = int(v) & 0xFFFFFFFF

    def _fit_partition_combo(self, g: _GroupFit, keys, valids):
        hint = self._cap_hints.get(g.name, g.hint)
        n = int(keys[0].numel())
        cap = K.next_pow2(2 * min(max(hint, 32), max(n, 32)))
        while True:
            part = K.GroupbyTable(len(keys), 0, cap)
            part.update(keys, valids, [], [])
            st = part.state()
            if not st[K._lib.ST_OVERFLOW] and st[K._lib.ST_OCCUPIED] * 10 <= part.capacity * 7:
                break
            cap *= 4
        self._cap_hints[g.name] = max(64, st[K._lib.ST_OCCUPIED])
        if g.table is None:
            g.table = part
        else:
            g.table = _merge_groups(g.table, part.compact())

    def fit_end(self, state, col_selector):
        from .. import dist

        self._absorb_pending(state)
        base = os.path.join(self.out_path, "categories")
        os.makedirs(base, exist_ok=True)
        # ranks that share an output directory (the default './') elect ONE writer: they all
        # hold the same merged vocabularies and used to race remove-then-write on the same files
        # (a host collective: done once per output directory, every rank takes the same branch)
        # (keyed by host too: on a multi-node run with node-local output directories every node
        # needs its own copy of the artifacts)
        import socket

        key = (socket.gethostname(), os.path.abspath(str(base)))
        if key not in self._writer_cache:
            self._writer_cache[key] = dist.is_first_rank_with(key)
        self._is_writer = self._writer_cache[key]
        paths = {}
        groups = list(state.values())
        with K.annotate("mid_level_groupby"):
            self._merge_parts_many([g for g in groups if not g.combo and g.parts])
        if dist.world_size() > 1:
            # ONE exchange for every single-vocabulary group of this fit (dist.merge_counts_many)
            singles = [g for g in groups if not g.combo]
            tabs, key_sorted, bounds = [], [], []
            for g in singles:
                # (range / sort path lists and merged partitions are in key order: they travel as
                # contiguous slices and the owners merge sorted runs)
                key_sorted.append(g.table is not None and g.sorted is not None and g.sorted[0] is g.table[0])
                if g.table is None:
                    dev = torch.device("cuda", torch.cuda.current_device())
                    k = torch.empty(0, dtype=torch.int64, device=dev)
                    c = torch.empty(0, dtype=torch.int64, device=dev)
                    mx = 0
                else:
                    k, c, mx = g.table
                # has-strings travels with the summed scalars: whether a group needs the
                # collective string-LUT merge must be the SAME decision on every rank (a rank
                # whose shard was empty has no strings of its own and used to take the
                # non-collective fast path while the others waited in all_gather_object)
                has_str = int(any(c_ in g.strings for c_ in g.cols))
                tabs.append((k, c, [int(g.nulls), int(g.valid_rows), int(mx), has_str]))
                bounds.append(int(g.valid_rows))   # = the sum of this rank's counts of the group
            if tabs:
                for g, (k, c, sc, info) in zip(singles, dist.merge_counts_many(tabs, key_sorted, bounds)):
                    g.table = (k, c, sc[2])  # the sum of the per-rank maxima bounds the max count
                    g.nulls, g.valid_rows = sc[0], sc[1]
                    g.any_rank_strings = sc[3] > 0
                    g.merged = True
                    # owners hold key ranges and gather key-sorted shards: the merged list is
                    # key-sorted, the one-pass ordering applies on every rank (int32 keys)
                    g.sorted = (k, info) if info is not None and info.get("sorted_by_key") else None
        self._complete_sorted_info(groups)
        opts = {}
        for g in groups:
            nb = _pick(self.num_buckets, g.name) if self.num_buckets else None
            oov_count = nb or 1
            max_emb = _pick(self.max_size, g.name) if self.max_size else 0
            freq = _pick(self.freq_threshold, g.name, 0) if self.freq_threshold else 0
            if max_emb and max_emb < oov_count + 2:
                raise ValueError(
                    "`max_size` can never be less than the maximum of `num_buckets + 2` and `3`, "
                    "because we must always reserve pad, null and at least 1 oov-bucket index."
                )
            opts[g.name] = (oov_count, max_emb, freq)
        # device-only vocabularies (no strings, nothing to trim) are ordered and get their
        # encode tables in ONE C call (write_uniques, categorify.py:1149): nvt_vocab_finalize_many
        fast = [g for g in groups if self._fast_finalizable(g, opts[g.name], dist)]
        if fast:
            for g in fast:
                paths[g.name] = "/".join([str(base), f"unique.{g.name}.parquet"])
            if self.defer_artifacts and LAZY_FINALIZE:
                # nothing reads the ordered vocabularies inside fit when the artifacts are
                # deferred: the sorts / table builds are enqueued by the first consumer
                # (transform, flush_artifacts, fitted_vocabulary ...).  In fit -> transform the
                # executor then starts the branches that do not need them (fill + normalize)
                # first, and the ~130 launches below are issued while those kernels already run.
                self._lazy_finalize = (fast, opts, base)
            else:
                with K.annotate("write_uniques"):
                    self._finalize_fast(fast, opts, base)
        for g in groups:
            if g.name in paths:
                continue
            oov_count, max_emb, freq = opts[g.name]
            with K.annotate("write_uniques"):
                if g.combo:
                    vocab = self._finalize_combo(g, dist)
                else:
                    vocab = self._finalize_single(g, dist)
                paths[g.name] = self._save_encodings(
                    g, vocab, base, first_n=max_emb, freq_threshold=freq, oov_count=oov_count
                )
        if not self.defer_artifacts:
            dist.barrier()  # the elected writer's files exist before any rank reads them
        return {name: paths[name] for name in state if name in paths}

    def _fast_finalizable(self, g, opt, dist):
        _, max_emb, freq = opt
        if g.combo or g.table is None or max_emb or freq or self.tie_break == "reference":
            return False
        if any(c in g.strings for c in g.cols) or getattr(g, "any_rank_strings", False):
            return False
        if dist.world_size() > 1 and not getattr(g, "merged", False):
            return False
        return int(g.table[0].numel()) > 0

    def _ensure_finalized(self, small_only=False):
        """Enqueue the deferred ordering + table work of the last fit.  small_only: just the
        vocabularies the one-launch batched sort takes (<= 16384 entries, not key-sorted) -- a
        handful of launches; transform encodes their columns before it spends ~1 ms of host time
        enqueueing the ~130 launches of the large vocabularies, which then run on the internal
        streams UNDERNEATH those encodes instead of in front of them."""
        if self._lazy_finalize is None:
            return
        groups, opts, base = self._lazy_finalize
        if small_only:
            now = [g for g in groups if _small_vocab(g)]
            rest = [g for g in groups if not _small_vocab(g)]
            self._lazy_finalize = (rest, opts, base) if rest else None
        else:
            now, self._lazy_finalize = groups, None
        if now:
            with K.annotate("write_uniques"):
                self._finalize_fast(now, opts, base)

    def _finalize_fast(self, groups, opts, base):
        descs = (K._lib.VocabCol * len(groups))()
        built = []
        for d, g in zip(descs, groups):
            keys, counts, max_count = g.table
            keys, counts = keys.contiguous(), counts.contiguous()
            start = opts[g.name][0] + OOV_OFFSET
            src = rtab = None
            if g.sorted is not None and g.sorted[0] is g.table[0] and keys.dtype == torch.int32:
                # key-sorted list of the range path: ordered out of place in ONE counting pass
                # (write_uniques' two sort_values, categorify.py:1300,1316)
                info = g.sorted[1]
                src = (keys, counts, info["cls_hist"], info["n_big"], info.get("label_of"))
                keys, counts = torch.empty_like(keys), torch.empty_like(counts)
                if info.get("range_table") is not None:
                    rtab = (info["range_table"], info["range_aux"], info["range_bits"])
            # a key-sorted source without a dumped table (sort path, multi-GPU merge): the table
            # is laid out from the sorted keys in one pass (flat range table)
            # (a vocabulary whose keys cluster in their range -- found out by an earlier fit's
            # _check_flat_tables -- gets an ordinary hashed table straight away)
            tab = K.EncodeTable(keys, start, unique=True, defer_build=True, range_table=rtab,
                                flat=src is not None and rtab is None and g.name not in self._no_flat)
            tab.fill_vocab_desc(d, counts, max_count, src=src)
            built.append((g, keys, counts, tab, start))
        K.check(K._lib.load().nvt_vocab_finalize_many(descs, len(groups), K.stream_ptr()),
                "nvt_vocab_finalize_many")
        # flat range tables are used as they are and CHECKED LATER (_check_flat_tables, after the
        # first encode has been queued): reading their largest displacement back here would make
        # the host wait for the whole finalisation before anything of the transform is enqueued
        self._flat_unchecked = [g.name for g, _, _, tab, _ in built if tab.flat_slots]
        for g, keys, counts, tab, start in built:
            if not tab.pending:
                tab.sort_tmp = None  # scratch of work already ordered on this stream
            self._encoders[g.name] = _SingleEncoder(tab, start)
            n = int(counts.numel())
            final = dict(
                name=g.name, cols=list(g.cols), combo=False, keys=[keys], null_mask=None,
                counts=counts, strings=None, start=start, oov_count=opts[g.name][0], oov_size=0,
                unique_count=n, unique_size=int(g.valid_rows), null_size=g.nulls,
                empty_input=False, base=str(base), table=tab,
            )
            if self.defer_artifacts:
                self._pending[g.name] = final
            elif self._is_writer:
                _write_artifacts(final)
            else:
                tab.wait_ready()  # `final` (and the counts it holds) dies here: order the stream first

    def _check_flat_tables(self):
        """Keys that cluster in their range make long probe runs in a monotone (flat) table:
        correct, but slow.  Once per fit -- after the first consumer's kernels are queued -- the
        largest displacement of every flat table is read back (ONE read-back, on a side stream
        behind the tables' finalisation events only) and a table beyond the limit is replaced by an
        ordinary hashed table for everything that follows."""
        names, self._flat_unchecked = self._flat_unchecked, []
        todo = [(n, self._encoders[n]) for n in names
                if n in self._encoders and getattr(self._encoders[n].table, "flat_slots", 0)]
        if not todo:
            return
        for (name, enc), ok in zip(todo, K.flat_tables_ok([e.table for _, e in todo])):
            if ok:
                continue
            self._no_flat.add(name)  # remembered: the next fit builds a hashed table straight away
            final = self._pending.get(name)
            keys = final["keys"][0] if final is not None else enc.table._vk
            tab = K.EncodeTable(keys, enc.first_label_in_file, unique=True)
            self._encoders[name] = _SingleEncoder(tab, enc.first_label_in_file)
            if final is not None:
                final["table"] = tab

    # -- vocabulary finalisation ------------------------------------------------
    def _finalize_single(self, g: _GroupFit, dist):
        if g.table is None:
            dev = torch.device("cuda", torch.cuda.current_device())
            keys = torch.empty(0, dtype=torch.int64, device=dev)
            counts = torch.empty(0, dtype=torch.int64, device=dev)
            max_count = 0
        else:
            keys, counts, max_count = g.table
        nulls = g.nulls
        if dist.world_size() > 1 and not getattr(g, "merged", False):
            keys, counts, nulls = dist.merge_counts(keys, counts, nulls)
            tot = dist.all_reduce_sum(torch.tensor([max_count, g.valid_rows], dtype=torch.int64,
                                                   device=keys.device)).tolist()
            max_count, g.valid_rows = int(tot[0]), int(tot[1])  # sum of maxima bounds the max
        strings = None
        for c in g.cols:
            if c in g.strings:
                strings = {**(strings or {}), **g.strings[c]}
        # collective on every rank (a rank whose shard was empty has no LUT of its own and
        # used to skip the all_gather_object: the other ranks then hung)
        strings = dist.merge_string_luts(strings)
        if self.tie_break == "reference" and keys.numel():
            # the reference's literal ordering (categorify.py:1300,1316) on the host
            hk, hc = K.read_back(keys.to(torch.int64)), K.read_back(counts)
            vals = np.array([strings[int(k)] for k in hk], dtype=object) if strings is not None else hk
            df = pd.DataFrame({"v": vals, "k": hk, "s": hc})
            df = df.sort_values(["v"], na_position="first", ignore_index=True)
            df = df.sort_values("s", ascending=False, ignore_index=True)
            keys = torch.from_numpy(df["k"].to_numpy().astype(hk.dtype)).to(keys.device).to(keys.dtype)
            counts = torch.from_numpy(df["s"].to_numpy().astype(np.int64)).to(counts.device)
        elif strings is not None:
            # string columns: order ties by the string value on the host (O(#uniques))
            hk, hc = keys.cpu().numpy(), counts.cpu().numpy()
            vals = np.array([strings[int(k)] for k in hk], dtype=object)
            order = np.lexsort((vals, -hc))
            keys = torch.from_numpy(hk[order]).to(keys.device)
            counts = torch.from_numpy(hc[order]).to(counts.device)
        else:
            keys, counts = keys.contiguous(), counts.contiguous()
            K.vocab_sort(keys, counts, max_count)
        return dict(keys=[keys], null_mask=None, counts=counts, null_size=nulls, strings=strings,
                    total=g.valid_rows)

    def _finalize_combo(self, g: _GroupFit, dist):
        if g.table is None:  # this rank (or the whole dataset) had no rows
            g.table = K.GroupbyTable(len(g.cols), 0, 64)
        comp = g.table.compact()
        if dist.world_size() > 1:
            comp = dist.merge_groups(comp, len(g.cols), 0)
            # keys seen only on other ranks need their strings too (ordering and artifacts);
            # collective per column, in column order, on every rank
            for c in g.cols:
                merged = dist.merge_string_luts(g.strings.get(c))
                if merged is not None:
                    g.strings[c] = merged
        keys, nm, size = comp["keys"], comp["null_mask"], comp["size"]
        all_null = (1 << len(g.cols)) - 1
        is_all_null = nm == all_null
        null_size = int(size[is_all_null].sum().item())
        keep = ~is_all_null
        keys = [k[keep] for k in keys]
        nm, size = nm[keep], size[keep]
        # order: size desc, then key columns ascending with nulls first (categorify.py:1300,1316)
        if not any(c in g.strings for c in g.cols) and self.tie_break == "value":
            # on the device: stable radix refinements, least significant key first (nvt_order_rows)
            sort_keys = [(size, None, False)]
            for j in range(len(g.cols)):
                notnull = (((nm >> j) & 1) ^ 1).to(torch.uint8)
                sort_keys += [(notnull, None, True), (keys[j], None, True)]
            order = K.order_rows(int(size.numel()), size.device, sort_keys) & 0xFFFFFFFF
            keys = [k[order] for k in keys]
            nm, size = nm[order], size[order]
        else:
            hk = [k.cpu().numpy() for k in keys]
            hnm, hs = nm.cpu().numpy(), size.cpu().numpy()
            sort_cols = []
            for j, c in enumerate(g.cols):
                isnull = (hnm >> j) & 1
                if c in g.strings:
                    lut = g.strings[c]
                    vals = np.array([lut.get(int(k), "") for k in hk[j]], dtype=object)
                else:
                    vals = hk[j]
                sort_cols.append((isnull, vals))
            lex = []
            for isnull, vals in reversed(sort_cols):
                lex += [vals, 1 - isnull]
            lex.append(-hs)
            order = np.lexsort(tuple(lex)) if len(hs) else np.array([], dtype=np.int64)
            dev = size.device
            keys = [torch.from_numpy(k[order]).to(dev) for k in hk]
            nm = torch.from_numpy(hnm[order]).to(dev)
            size = torch.from_numpy(hs[order]).to(dev)
        strings = {c: g.strings[c] for c in g.cols if c in g.strings} or None
        return dict(keys=keys, null_mask=nm, counts=size, null_size=null_size, strings=strings,
                    combo=True)

    def _save_encodings(self, g: _GroupFit, vocab, base, first_n, freq_threshold, oov_count):
        """categorify.py:719-822 on the finalised (ordered) vocabulary."""
        counts = vocab["counts"]
        keys = vocab["keys"]
        nm = vocab["null_mask"]
        start = oov_count + OOV_OFFSET
        oov_size = 0
        if freq_threshold:
            keep = (counts >= freq_threshold) | (counts == 0)
            oov_size += int(counts[~keep].sum().item())
            counts = counts[keep]
            keys = [k[keep] for k in keys]
            nm = nm[keep] if nm is not None else None
        if first_n:
            limit = max(first_n - start, 0)
            if counts.numel() > limit:
                oov_size += int(counts[limit:].sum().item())
                counts = counts[:limit]
                keys = [k[:limit] for k in keys]
                nm = nm[:limit] if nm is not None else None
        unique_count = int(counts.numel())
        if not (freq_threshold or first_n) and vocab.get("total") is not None:
            unique_size = int(vocab["total"])  # nothing dropped: no device readback needed
        else:
            unique_size = int(counts.sum().item()) if unique_count else 0
        # device-side encoder, cached for transform (cat_cache="device" behaviour)
        combo = bool(vocab.get("combo"))
        self._encoders[g.name] = _build_encoder(keys, nm, start, combo)
        final = dict(
            name=g.name, cols=list(g.cols), combo=combo, keys=keys, null_mask=nm, counts=counts,
            strings=vocab["strings"], start=start, oov_count=oov_count, oov_size=oov_size,
            unique_count=unique_count, unique_size=unique_size, null_size=vocab["null_size"],
            empty_input=g.table is None, base=str(base),
        )
        unique_path = "/".join([str(base), f"unique.{g.name}.parquet"])
        if self.defer_artifacts:
            self._pending[g.name] = final
        elif getattr(self, "_is_writer", True):
            _write_artifacts(final)
        return unique_path

    def flush_artifacts(self, force=False):
        """Write any deferred unique.*/meta.*.parquet files (defer_artifacts=True).

        Not a collective: only the writer elected at fit_end writes, everybody else just drops
        the device copies.
        ``force=True`` (set_storage_path, which Workflow.save reaches and which may run on one
        rank only and need the files whoever that rank is) writes on any rank; the files are
        replaced atomically (_write_artifacts), so a forced write concurrent with the elected
        writer's is harmless -- both hold the same vocabularies."""
        self._ensure_finalized()
        for final in self._pending.values():
            # (a non-writer does NOT write because its file is missing: on a shared out_path every
            # rank flushes at about the same time, the file is missing for all of them and all G
            # ranks wrote all vocabularies -- ADVICE r05.  Node-local output directories are covered
            # by the election itself: the writer is the first rank per (host, directory), fit_end)
            if getattr(self, "_is_writer", True) or force:
                _write_artifacts(final)
            elif final.get("table") is not None:
                # dropping the device copies: their buffers go back to the allocator on THIS
                # stream, which must first be ordered behind the internal stream that sorts them
                final["table"].wait_ready()
        self._pending = {}

    def fit_finalize(self, categories):
        """categorify.py:404-415"""
        idx_count = 0
        if self.single_table:
            self.flush_artifacts(force=True)  # every rank re-bases the files it reads below
        for cat in categories:
            self.categories[cat] = categories[cat]
            if self.single_table:
                path = self.categories[cat]
                df = pd.read_parquet(path)
                df.index = df.index + idx_count
                idx_count += df.shape[0]
                df.to_parquet(path, compression=None)
                self._encoders.pop(cat, None)  # rebuilt lazily with the shifted labels

    def clear(self):
        self.categories = deepcopy(self.vocabs)
        self._lazy_finalize = None  # a fit nobody consumed: its ordering is never enqueued
        for enc in self._encoders.values():
            tab = getattr(enc, "table", None)
            if tab is not None:
                tab.wait_ready()  # its buffers are recycled on this stream from here on
        self._encoders = {}
        self._pending = {}

    def fitted_vocabulary(self, name: str):
        """(keys: list of device tensors, counts: device tensor) of a fitted vocabulary, in label
        order, safe to read on the current stream (waits for work still in flight on the
        library's internal streams).  Only available while the fit's device state is kept
        (``defer_artifacts=True`` or before the next ``clear``)."""
        self._ensure_finalized()
        final = self._pending[name]
        if final.get("table") is not None:
            final["table"].wait_ready()
        return final["keys"], final["counts"]

    @property
    def async_pending(self) -> bool:
        """Vocabularies still being ordered on the library's internal streams: the executor
        runs the other branches of the graph first (workflow.py)."""
        return self._lazy_finalize is not None or any(
            getattr(getattr(e, "table", None), "pending", False) for e in self._encoders.values())

    def process_vocabs(self, vocabs):
        """categorify.py:421-454"""
        categories = {}
        if isinstance(vocabs, dict) and all(isinstance(v, pd.Series) for v in vocabs.values()):
            base = os.path.join(self.out_path, "categories")
            os.makedirs(base, exist_ok=True)
            for col, vocab in vocabs.items():
                col_name = _make_name(*col, sep=self.name_sep) if isinstance(col, tuple) else col
                nb = self.num_buckets
                oov_count = 1
                if nb:
                    oov_count = (nb if isinstance(nb, int) else nb[col_name]) or 1
                col_df = pd.DataFrame({col_name: vocab}).dropna()
                col_df.index += NULL_OFFSET + oov_count
                # the reference writes through _save_encodings with its default oov_count=1
                # (categorify.py:439), so the file index restarts at 3
                col_df = col_df.copy()
                col_df.index = pd.RangeIndex(start=3, stop=3 + len(col_df))
                path = "/".join([base, f"unique.{col_name}.parquet"])
                col_df.to_parquet(path, compression=None)
                pd.DataFrame(
                    {
                        "kind": ["pad", "null", "oov", "unique"],
                        "offset": [0, 1, 2, 3],
                        "num_indices": [1, 1, 1, len(col_df)],
                    }
                ).to_parquet("/".join([base, f"meta.{col_name}.parquet"]))
                categories[col_name] = path
        elif isinstance(vocabs, dict) and all(isinstance(v, str) for v in vocabs.values()):
            categories = {
                (_make_name(*col, sep=self.name_sep) if isinstance(col, tuple) else col): path
                for col, path in vocabs.items()
            }
        else:
            raise ValueError(
                "Unrecognized vocab type, please provide either a dictionary with paths to "
                "parquet files or a dictionary with pandas Series objects."
            )
        return categories

    def set_storage_path(self, new_path, copy=False):
        import shutil

        self.flush_artifacts(force=True)
        new = {}
        for col, old in self.categories.items():
            target = old.replace(str(self.out_path), str(new_path))
            if copy and target != old:
                os.makedirs(os.path.dirname(target), exist_ok=True)
                shutil.copy(old, target)
                meta_old = old.replace("unique.", "meta.")
                if os.path.exists(meta_old):
                    shutil.copy(meta_old, target.replace("unique.", "meta."))
            new[col] = target
        self.categories = new
        self.out_path = new_path

    # ------------------------------------------------------------ transform --
    def _encoder_for(self, storage_name: str, cols: List[str], frame: DeviceFrame):
        if self._lazy_finalize is not None and any(g.name == storage_name for g in self._lazy_finalize[0]):
            self._ensure_finalized()
        enc = self._encoders.get(storage_name)
        if enc is not None:
            return enc
        path = self.categories[storage_name]
        nb = _pick(self.num_buckets, storage_name) if self.num_buckets else None
        n_oov = nb or 1
        value = pd.read_parquet(path)
        labels = value.index.to_numpy()
        start = int(labels[0]) if len(labels) else OOV_OFFSET + n_oov
        if len(labels) and start < OOV_OFFSET + n_oov and not self.single_table:
            start += OOV_OFFSET + n_oov  # categorify.py:1641-1643 re-base guard
        dev = torch.device("cuda", torch.cuda.current_device())
        combo = self.encode_type == "combo" and len(cols) > 1
        key_names = cols if combo else [storage_name]
        keys, nm = [], np.zeros(len(value), dtype=np.uint8)
        for j, kn in enumerate(key_names):
            s = value[kn]
            isnull = s.isna().to_numpy()
            if s.dtype == object or pd.api.types.is_string_dtype(s.dtype):
                from ..strings import string_key64

                hk = np.zeros(len(s), dtype=np.int64)
                if (~isnull).any():
                    hk[~isnull] = string_key64(s.to_numpy(dtype=object)[~isnull])
            else:
                hk = s.fillna(0).to_numpy().astype(np.int64)
            nm |= (isnull.astype(np.uint8) << j)
            keys.append(hk)
        if not combo:
            keep = ~(nm.astype(bool))
            if not keep.all():
                if keep.any():
                    raise ValueError(f"vocabulary file {path} mixes null and non-null rows")
                keys = [keys[0][:0]]  # the reference's "empty" file: nothing is ever matched
            src = frame[cols[0]].data.dtype if cols[0] in frame else torch.int64
            i32 = src == torch.int32 and all(
                c not in frame or frame[c].data.dtype == torch.int32 for c in self._group_cols(storage_name, cols)
            )
            kt = torch.from_numpy(keys[0].astype(np.int32 if i32 else np.int64)).to(dev)
            enc = _build_encoder([kt], None, start, False, unique=not pd.Series(keys[0]).duplicated().any())
        else:
            enc = _build_encoder(
                [torch.from_numpy(k).to(dev) for k in keys], torch.from_numpy(nm).to(dev), start, True
            )
        self._encoders[storage_name] = enc
        return enc

    def _group_cols(self, storage_name, cols):
        members = [c for c, sn in self.storage_name.items() if sn == storage_name]
        return members or cols

    def transform(self, col_selector: ColumnSelector, df):
        """categorify.py:477-537"""
        frame, was_pandas = as_device_frame(df)
        new = frame.copy(deep=False)
        if isinstance(self.freq_threshold, dict):
            assert all(x in self.freq_threshold for x in col_selector.names)
        column_mapping = self.column_mapping(col_selector)
        out_dtype = torch.int32 if np.dtype(self.output_dtype) == np.dtype("int32") else torch.int64
        # two rounds while vocabularies of the last fit still wait for their ordering (lazy
        # finalisation): the columns of the SMALL vocabularies are finalised and encoded first,
        # then the large ones are enqueued (the bulk of the host's launch work) and encoded
        names = list(column_mapping)
        rounds = [names]
        if self._lazy_finalize is not None and SPLIT_FINALIZE:
            self._ensure_finalized(small_only=True)
            if self._lazy_finalize is not None:
                late = {g.name for g in self._lazy_finalize[0]}
                first = [n for n in names if self._storage_of(n, column_mapping) not in late]
                rounds = [first, [n for n in names if n not in set(first)]]
        for todo in rounds:
            self._encode_round(todo, column_mapping, frame, new, out_dtype)
        # (column order of the result: as the one-round version produced it)
        ordered = frame.copy(deep=False)
        for name in names:
            ordered[name] = new[name]
        new = ordered
        return new.to_pandas() if was_pandas else new

    def _storage_of(self, name, column_mapping):
        use_name = column_mapping.get(name, name)
        if isinstance(use_name, (list, tuple)) and len(use_name) == 1:
            use_name = use_name[0]
        if use_name != name or self.encode_type == "joint":
            return self.storage_name.get(name, name)
        return name

    def _encode_round(self, names, column_mapping, frame, new, out_dtype):
        batch = []  # single-vocabulary columns: encoded by ONE C call (nvt_encode_many)
        for name in names:
            try:
                use_name = column_mapping.get(name, name)
                if isinstance(use_name, (list, tuple)) and len(use_name) == 1:
                    use_name = use_name[0]
                storage_name = self._storage_of(name, column_mapping)
                cols = list(use_name) if isinstance(use_name, (list, tuple)) else [use_name]
                nb = _pick(self.num_buckets, storage_name) if self.num_buckets else None
                enc = self._encoder_for(storage_name, cols, frame)
                null_off = enc.first_label_in_file if self.single_table else NULL_OFFSET
                if isinstance(enc, _SingleEncoder):
                    col = frame[cols[0]]
                    if col.fill is not None:
                        col = col.materialize()
                    keys, valid = key_view(col)
                    batch.append((name, col, (enc.table, keys, valid, null_off, null_off + 1,
                                              nb or 0)))
                else:
                    new[name] = enc.encode(frame, cols, null_off, null_off + 1, nb or 0, out_dtype)
            except Exception as e:
                raise RuntimeError(f"Failed to categorical encode column {name}") from e
        if batch:
            try:
                outs = K.encode_many([item for _, _, item in batch], out_dtype)
            except Exception as e:
                names = [n for n, _, _ in batch]
                raise RuntimeError(f"Failed to categorical encode column {names[0]}") from e
            for (name, col, _), out in zip(batch, outs):
                new[name] = DeviceColumn(out, None, col.offsets, None, None)
            if self._flat_unchecked:
                self._check_flat_tables()

    # --------------------------------------------------------------- schema --
    def column_mapping(self, col_selector):
        """categorify.py:539-553"""
        if self.encode_type == "combo":
            mapping = {}
            for group in col_selector.grouped_names:
                if isinstance(group, (tuple, list)):
                    name = _make_name(*group, sep=self.name_sep)
                    group = [*group]
                else:
                    name = group
                    group = [group]
                mapping[name] = group
            return mapping
        return super().column_mapping(col_selector)

    def compute_selector(self, input_schema, selector, parents_selector=None,
                         dependencies_selector=None):
        self._validate_matching_cols(input_schema, parents_selector, "computing input selector")
        return parents_selector

    def _compute_properties(self, col_schema, input_schema):
        """categorify.py:555-579"""
        new_schema = super()._compute_properties(col_schema, input_schema)
        col_name = col_schema.name
        category_name = self.storage_name.get(col_name, col_name)
        path = self.categories.get(category_name, None)
        cardinality, dimensions = self.get_embedding_sizes([category_name])[category_name]
        to_add = {
            "num_buckets": _pick(self.num_buckets, col_name),
            "freq_threshold": _pick(self.freq_threshold, col_name),
            "max_size": _pick(self.max_size, col_name),
            "cat_path": path,
            "domain": {"min": 0, "max": cardinality - 1, "name": category_name},
            "embedding_sizes": {"cardinality": cardinality, "dimension": dimensions},
        }
        return col_schema.with_properties({**new_schema.properties, **to_add})

    @property
    def output_tags(self):
        return [Tags.CATEGORICAL]

    @property
    def output_dtype(self):
        return self.dtype or np.int64

    def get_embedding_sizes(self, columns):
        self._ensure_finalized()
        pending = {n: f["unique_count"] for n, f in self._pending.items()}
        return _get_embeddings(self.categories, columns, self.num_buckets, pending)


# ------------------------------------------------------------------ helpers --
def _write_artifacts(final):
    """Host side of categorify.py:719-822: unique.<name>.parquet (columns <name>,
    <name>_size; RangeIndex = label) and meta.<name>.parquet."""
    name, base, start = final["name"], final["base"], final["start"]
    keys, nm, counts = final["keys"], final["null_mask"], final["counts"]
    if final.get("table") is not None:
        final["table"].wait_ready()  # ordered on an internal stream: wait before reading back
    unique_path = "/".join([base, f"unique.{name}.parquet"])
    meta_path = "/".join([base, f"meta.{name}.parquet"])
    combo = final["combo"]
    key_names = final["cols"] if combo else [name]
    unique_count = final["unique_count"]
    hnm = nm.cpu().numpy() if nm is not None else None
    data = {}
    for j, kn in enumerate(key_names):
        col = keys[j].cpu().numpy()
        lut = None
        if final["strings"] is not None:
            lut = final["strings"].get(kn) if combo else final["strings"]
        if lut is not None:
            col = np.array([lut.get(int(k)) for k in col], dtype=object)
        if hnm is not None:
            isnull = ((hnm >> j) & 1).astype(bool)
            if isnull.any():
                if col.dtype == object:
                    col[isnull] = None
                else:
                    col = col.astype(np.float64)
                    col[isnull] = np.nan
        data[kn] = col
    if final["empty_input"]:
        # categorify.py:1318-1324: empty input -> one all-null row
        df = pd.DataFrame({kn: pd.Series([None], dtype=object) for kn in key_names})
        df.index = pd.RangeIndex(start=start, stop=start + 1)
        unique_count = 1
    else:
        df = pd.DataFrame(data)
        df[f"{name}_size"] = counts.cpu().numpy()
        df.index = pd.RangeIndex(start=start, stop=start + len(df))
    os.makedirs(base, exist_ok=True)
    # written beside the target and moved into place: a reader (or a second writer) never sees a
    # missing or half-written file
    tmp = f"{unique_path}.tmp.{os.getpid()}"
    df.to_parquet(tmp, compression=None)
    os.replace(tmp, unique_path)
    meta = {
        "kind": ["pad", "null", "oov", "unique"],
        "offset": [PAD_OFFSET, NULL_OFFSET, OOV_OFFSET, OOV_OFFSET + final["oov_count"]],
        "num_indices": [1, 1, final["oov_count"], unique_count],
        "num_observed": [0, final["null_size"], final["oov_size"], final["unique_size"]],
    }
    tmp = f"{meta_path}.tmp.{os.getpid()}"
    pd.DataFrame(meta).to_parquet(tmp)
    os.replace(tmp, meta_path)
    return unique_path


def _merge_groups(acc: "K.GroupbyTable", comp) -> "K.GroupbyTable":
    st = acc.state()
    need = st[K._lib.ST_OCCUPIED] + comp["n"]
    if 2 * need > acc.capacity:
        old = acc.compact()
        new = K.GroupbyTable(acc.nkeys, acc.nvals, 4 * need,
                             sumsq=bool(acc.flags & K._lib.NVT_GB_SUMSQ),
                             minmax=bool(acc.flags & K._lib.NVT_GB_MINMAX))
        new.merge(old["keys"], old["null_mask"], old["size"], old["count"], old["sum"],
                  old["sumsq"], old["min"], old["max"])
        acc = new
    acc.merge(comp["keys"], comp["null_mask"], comp["size"], comp["count"], comp["sum"],
              comp["sumsq"], comp["min"], comp["max"])
    if acc.state()[K._lib.ST_OVERFLOW]:
        raise K._lib.NvtHipError("groupby table overflow during merge")
    return acc


def _small_vocab(g) -> bool:
    """True for the vocabularies nvt_vocab_finalize_many orders in its one batched launch."""
    if g.table is None or (g.sorted is not None and g.sorted[0] is g.table[0]):
        return False
    keys, _, max_count = g.table
    return keys.dtype == torch.int32 and 2 <= int(keys.numel()) <= 16384 and 0 < int(max_count) < (1 << 32)


class _SingleEncoder:
    """One key column (or a joint group sharing one vocabulary)."""

    def __init__(self, table: "K.EncodeTable", first_label: int):
        self.table = table
        self.first_label_in_file = first_label

    def encode(self, frame, cols, null_label, oov_label, num_buckets, out_dtype):
        col = frame[cols[0]]
        if col.fill is not None:
            col = col.materialize()
        keys, valid = key_view(col)
        out = self.table.encode(keys, valid, null_label, oov_label, num_buckets, out_dtype)
        return DeviceColumn(out, None, col.offsets, None, None)


class _ComboEncoder:
    """Multi-column key tuples: index table over the vocabulary groups."""

    def __init__(self, keys, null_mask, first_label: int):
        self.n = int(keys[0].numel())
        self.first_label_in_file = first_label
        self.index = K.GroupbyTable(len(keys), 0, max(64, 2 * self.n + 1))
        self.index.index_build([k.contiguous() for k in keys], null_mask)

    def encode(self, frame, cols, null_label, oov_label, num_buckets, out_dtype):
        keys, valids = [], []
        for c in cols:
            k, v = key_view(frame[c])
            keys.append(k)
            valids.append(v)
        grp = self.index.lookup(keys, valids)
        labels = grp + self.first_label_in_file
        if num_buckets and num_buckets > 1:
            acc = None
            for k, v in zip(keys, valids):  # XOR chain of categorify.py:1846-1851, nulls as key 0
                _, acc = K.hash_bucket(k, num_buckets, xor_in=acc, want_hash=True, want_bucket=False,
                                       valid=v)
            h32 = (acc >> 32) & 0xFFFFFFFF
            oov = oov_label + (h32 % num_buckets)
        else:
            oov = torch.full_like(labels, oov_label)
        labels = torch.where(grp >= 0, labels, oov)
        # a row is null only when every component is null (categorify.py:1689-1692); the
        # validity used is key_view's, which also covers float-NaN nulls of pandas columns
        all_null = None
        for v in valids:
            if v is None:
                all_null = torch.zeros(labels.numel(), dtype=torch.bool, device=labels.device)
                break
            isnull = ~K.unpack_bitmap(v, labels.numel())
            all_null = isnull if all_null is None else (all_null & isnull)
        labels = torch.where(all_null, torch.full_like(labels, null_label), labels)
        return DeviceColumn(labels.to(out_dtype))


def _build_encoder(keys, null_mask, first_label, combo, unique=True):
    if combo:
        return _ComboEncoder(keys, null_mask, first_label)
    return _SingleEncoder(K.EncodeTable(keys[0].contiguous(), first_label, unique=unique),
                          first_label)


def _get_embeddings(paths, cat_names, buckets=0, pending=None):
    """categorify.py:663-684"""
    import pyarrow.dataset as pa_ds

    embeddings = {}
    if isinstance(buckets, int):
        buckets = {name: buckets for name in cat_names}
    for col in cat_names:
        path = paths.get(col)
        num_rows = OOV_OFFSET
        if pending and col in pending:
            num_rows += pending[col]
        elif path:
            for frag in pa_ds.dataset(path, format="parquet").get_fragments():
                num_rows += frag.metadata.num_rows
        if isinstance(buckets, dict):
            bucket_size = buckets.get(col, 0)
        elif isinstance(buckets, int):
            bucket_size = buckets
        else:
            bucket_size = 1
        num_rows += bucket_size
        embeddings[col] = _emb_sz_rule(num_rows)
    return embeddings


def get_embedding_sizes(source, output_dtypes=None):
    """categorify.py:616-660: {col: (cardinality, dimension)} from a Workflow or node."""
    from ..workflow import Workflow

    if isinstance(source, Workflow):
        _ = source.output_schema  # folds the fitted properties in (lazy after fit)
    output_node = source.output_node if isinstance(source, Workflow) else source
    output = {}
    multihot = set()
    cats_schema = output_node.output_schema.select_by_tag(Tags.CATEGORICAL)
    for col_name, col_schema in cats_schema.column_schemas.items():
        if col_schema.dtype is not None and col_schema.is_list and col_schema.is_ragged:
            multihot.add(col_name)
        sizes = col_schema.properties.get("embedding_sizes", {})
        output[col_name] = (sizes["cardinality"], sizes["dimension"])
    if not multihot:
        return output
    single = {k: v for k, v in output.items() if k not in multihot}
    multi = {k: v for k, v in output.items() if k in multihot}
    return single, multi
