"""Transform side of JoinGroupby / TargetEncoding on the sort path's groups: the flat index, the key
directory, the packed per-group records ("lookup images") and their builders.

Part of the host driver of the C ABI (include/nvt_hip.h); ``kernels.py`` is the facade every
caller imports -- it re-exports these names, holds the run-time switches they read (``K.<FLAG>`` at
call time: tests and A / B runs set them on the facade) and the helpers they share."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import kernels as K
from ._lib import check


class FlatIndex:
    """key -> position in an ascending int32 key list (group ids of sorted_groupby): a flat
    range table laid out from the list in one pass (nvt_flat_index_build).  Same ``lookup``
    as GroupbyTable (the transform side of JoinGroupby / TargetEncoding)."""

    FLAT_AUX_WORDS, FLAT_AUX_MAXDISP = 8192 + 16, 8192 + 8   # include/nvt_hip.h NVT_FLAT_AUX_*
    MAX_DISPLACEMENT = 4096

    def __init__(self, keys32: torch.Tensor, key_offset: int = 0):
        _lib.require_gpu()
        # the list holds column value - key_offset (int64 columns whose keys span < 2^32)
        self.key_offset = int(key_offset)
        self.keys32 = keys32.contiguous()
        self.n = n = int(keys32.numel())
        # home slots: FLAT_INDEX_LOAD of them hold a key (no power of two needed; the smaller
        # the table the more of it the caches keep)
        self.slots = max(64, int(n / K.FLAT_INDEX_LOAD) + 1)
        self.capacity = self.slots + n + 64
        self._table = self._aux = self._dir = None
        self.null_group = -1
        self._ok = None
        # with keyed lookup images (the transform's default path) nothing reads the flat table:
        # it is laid out by the first lookup / gather / te call that needs it
        if not (K.KEYED_IMAGES and K.LOOKUP_IMAGES and n >= 1):
            self._build_table()

    def _build_table(self):
        lib = _lib.load()
        n, dev = self.n, self.keys32.device
        self._table = torch.empty(self.capacity, dtype=torch.int64, device=dev)
        self._aux = torch.zeros(self.FLAT_AUX_WORDS, dtype=torch.int32, device=dev)
        need = C.c_uint64()
        check(lib.nvt_flat_index_tmp_bytes(n, C.byref(need)), "nvt_flat_index_tmp_bytes")
        tmp = torch.empty(need.value, dtype=torch.uint8, device=dev)
        check(lib.nvt_flat_index_build(self.keys32.data_ptr(), n, self.slots, self._aux.data_ptr(),
                                       self._table.data_ptr(), self.capacity, tmp.data_ptr(),
                                       K.stream_ptr()), "nvt_flat_index_build")
        if self.null_group >= 0:
            self._aux[self.FLAT_AUX_MAXDISP + 2] = self.null_group + 1   # NVT_FLAT_AUX_NULLGROUP

    @property
    def table(self) -> torch.Tensor:
        if self._table is None:
            self._build_table()
        return self._table

    @property
    def aux(self) -> torch.Tensor:
        if self._aux is None:
            self._build_table()
        return self._aux

    def set_null_group(self, group: int):
        """Rows whose key is null look up `group` (JoinGroupby / TargetEncoding keep null keys as
        one group, like the reference's groupby(dropna=False)); without it they miss."""
        self.null_group = int(group)
        if self._aux is not None:
            self._aux[self.FLAT_AUX_MAXDISP + 2] = int(group) + 1   # NVT_FLAT_AUX_NULLGROUP

    def ok(self) -> bool:
        """False when the keys cluster in their range (an entry further than MAX_DISPLACEMENT
        slots from its home slot): the caller builds a hashed index instead.  One read-back --
        none while the table is not laid out (keyed lookup images search a crowded bucket by
        bisection, and a table laid out later for a column-wise lookup is searched by galloping
        steps: slower for clustered keys, never wrong)."""
        if self._table is None:
            return True
        if self._ok is None:
            word = self.aux[self.FLAT_AUX_MAXDISP:self.FLAT_AUX_MAXDISP + 2].view(torch.int64)
            d = int(K.read_back(word)[0]) & 0xFFFFFFFF
            self._ok = d <= self.MAX_DISPLACEMENT
        return self._ok

    def lookup(self, keys, key_valid) -> torch.Tensor:
        k = keys[0]
        if k.dtype not in (torch.int32, torch.int64):
            k = K.widen_i64(k)
        k = k.contiguous()
        n = k.numel()
        out = torch.empty(n, dtype=torch.int64, device=k.device)
        K.stat_add("flat_lookups")
        check(_lib.load().nvt_flat_lookup(k.data_ptr(), K.dtype_code(k.dtype), K.ptr(key_valid[0]), n,
                                          self.aux.data_ptr(), self.table.data_ptr(), self.capacity,
                                          self.key_offset, out.data_ptr(), K.stream_ptr()),
              "nvt_flat_lookup")
        return out


    def _key(self, keys):
        k = keys[0]
        if k.dtype not in (torch.int32, torch.int64):
            k = K.widen_i64(k)
        return k.contiguous()

    def gather(self, keys, key_valid, records: torch.Tensor, out_dtypes, miss):
        """JoinGroupby.transform in one launch: (outs, unseen) with outs[c][i] =
        records[group of keys[i], c] (miss[c] for a key without group) and ``unseen`` a device
        word that is non-zero when any row had no group."""
        k = self._key(keys)
        n, ncols = k.numel(), int(records.shape[1])
        assert records.dtype == torch.float64 and records.is_contiguous() and ncols == len(out_dtypes)
        outs = [torch.empty(n, dtype=dt, device=k.device) for dt in out_dtypes]
        unseen = torch.zeros(1, dtype=torch.int64, device=k.device)
        K.stat_add("flat_lookups")
        check(_lib.load().nvt_flat_lookup_gather(
            k.data_ptr(), K.dtype_code(k.dtype), K.ptr(key_valid[0]), n, self.aux.data_ptr(),
            self.table.data_ptr(), self.capacity, self.key_offset, records.data_ptr(), ncols,
            _lib.ptr_array([o.data_ptr() for o in outs]),
            (C.c_int * ncols)(*[K.dtype_code(dt) for dt in out_dtypes]),
            (C.c_double * ncols)(*[float(m) for m in miss]), unseen.data_ptr(), K.stream_ptr()),
            "nvt_flat_lookup_gather")
        return outs, unseen

    def te(self, keys, key_valid, fold, kfold, records: torch.Tensor, p_smooth, y_mean, out_dtype):
        """TargetEncoding.transform in one launch (records: [groups, 2 * (kfold + 1)], or
        [groups, 2] without folds)."""
        k = self._key(keys)
        n = k.numel()
        assert records.dtype == torch.float64 and records.is_contiguous()
        assert int(records.shape[1]) == (2 * (kfold + 1) if fold is not None else 2)
        out = torch.empty(n, dtype=out_dtype, device=k.device)
        K.stat_add("flat_lookups")
        check(_lib.load().nvt_flat_lookup_te(
            k.data_ptr(), K.dtype_code(k.dtype), K.ptr(key_valid[0]), n, self.aux.data_ptr(),
            self.table.data_ptr(), self.capacity, self.key_offset,
            K.ptr(fold.contiguous() if fold is not None else None),
            int(kfold) if fold is not None else 1, records.data_ptr(), float(p_smooth), float(y_mean),
            out.data_ptr(), K.dtype_code(out_dtype), K.stream_ptr()), "nvt_flat_lookup_te")
        return out


    # ---- lookup images: one probe + one packed record per row for ALL operators on this key ----
    def attach(self, consumer: "LookupConsumer"):
        """An operator fitted on these groups registers the values its transform hands a row
        (include/nvt_hip.h, "Lookup images").  Replaces an earlier consumer of the same owner /
        tag (a re-registration after fit_finalize)."""
        cons = getattr(self, "consumers", None)
        if cons is None:
            cons = self.consumers = []
        cons[:] = [c for c in cons if not (c.owner is consumer.owner and c.tag == consumer.tag)]
        cons.append(consumer)
        consumer.index = self
        self._image = None

    def prepare_image(self):
        """End of a fit: enqueue the image the first transform would build (behind the fit's last
        kernels, without a read-back), so that it is computed while the host walks into the
        transform instead of in front of the first lookup."""
        if (K.EAGER_IMAGES and K.LOOKUP_IMAGES and getattr(self, "consumers", None)
                and getattr(self, "_image", None) is None):
            self._build_image()

    def _build_image(self):
        # ranges read at fixed offsets first, 16-byte aligned: the lookup reads every aligned
        # 16-byte window that holds several of a row's values with ONE load (per-fold values are
        # picked by the row's fold id and keep a load each)
        at, place = 0, {}
        for c in sorted(self.consumers, key=lambda c: c.fold_fn is not None):
            align = 16 if (c.width >= 16 and c.fold_fn is None) else 8
            at = (at + align - 1) & ~(align - 1)
            place[id(c)] = at
            at += c.width
        total = max(8, (at + 7) & ~7)
        # records of <= 64 bytes never straddle a 64-byte sector; larger ones are sector-aligned
        stride = K.next_pow2(total) if total <= 64 else (total + 63) & ~63
        # (a consumer's statistics may hold one group more than the key list: the null-key group)
        rows = max([self.n + 1] + [c.groups for c in self.consumers])
        dev = self.keys32.device
        image = torch.empty(rows * stride, dtype=torch.uint8, device=dev)
        plist = []
        if K.ONE_PASS_IMAGES and stride <= _lib.IMAGE_BUILD_MAX_STRIDE \
                and all(c.parts is not None for c in self.consumers):
            for c in self.consumers:
                plist += c.parts(place[id(c)], c.groups)
        if plist and len(plist) <= _lib.IMAGE_BUILD_MAX_PARTS:
            # whole records in ONE pass over every operator's range
            arr = (_lib.ImagePart * len(plist))(*[p[0] for p in plist])
            check(_lib.load().nvt_image_build(arr, len(plist), rows, image.data_ptr(), stride,
                                              K.stream_ptr()), "nvt_image_build")
        else:
            for c in self.consumers:
                c.fill(image, stride, place[id(c)], c.groups)
        keyed = K.KEYED_IMAGES and 1 <= self.n < (1 << 32) - 2
        if keyed and self._dir is None:
            self.dir_slots = max(64, int(self.n / K.KEYDIR_LOAD) + 1)
            self._dir = torch.empty(4 * (self.dir_slots + 1), dtype=torch.int32, device=dev)
            check(_lib.load().nvt_keydir_build(self.keys32.data_ptr(), self.n, self.dir_slots,
                                               self._dir.data_ptr(), K.stream_ptr()), "nvt_keydir_build")
        self._image = (image, stride, place, keyed)
        return self._image

    def image_lookup(self, consumer: "LookupConsumer", keys, key_valid, fold=None):
        """{output name: tensor[n]} of `consumer` for the rows of keys[0], and the device word
        that is non-zero when a row had no group.  Inside a pass (pass_memo) the ONE launch
        serves every attached consumer: the others find their columns in the memo."""
        k = self._key(keys)
        n = int(k.numel())
        valid = key_valid[0]
        memo = K.current_pass_memo()
        mkey = ("image", id(self), k.data_ptr(), n, k._version, K.ptr(valid))
        hit = memo.get(mkey) if memo is not None else None
        if hit is not None and id(consumer) in hit["outs"]:
            return hit["outs"][id(consumer)], hit["unseen"]
        img = getattr(self, "_image", None) or self._build_image()
        image, stride, place, keyed = img
        todo = list(self.consumers) if (memo is not None and hit is None) else [consumer]
        dev = k.device
        outs, ptrs, folds, offs, sizes, miss, keep = {}, [], [], [], [], [], []
        for c in todo:
            f = None
            if c.fold_fn is not None:
                f = fold if (c is consumer and fold is not None) else c.fold_fn(n, dev)
                f = f.contiguous()
                assert f.dtype == torch.uint8 and int(f.numel()) == n
                keep.append(f)
            mine = outs.setdefault(id(c), {})
            for name, dt, rel, per_fold, mv in c.outputs:
                t = torch.empty(n, dtype=dt, device=dev)
                mine[name] = t
                ptrs.append(t.data_ptr())
                folds.append(K.ptr(f) if per_fold else None)
                offs.append(place[id(c)] + rel)
                sizes.append(t.element_size())
                miss.append(_value_bits(mv, dt))
        unseen = torch.zeros(1, dtype=torch.int64, device=dev)
        K.stat_add("image_lookups")
        # nvt_flat_lookup_image takes at most IMAGE_LOOKUP_MAX_OUTPUTS columns per launch: two
        # JoinGroupby operators on one key, or a TargetEncoding with many targets beside one, go
        # out in several launches over the same rows (every launch probes again; `unseen` is only
        # ever raised, so the launches share it)
        for lo in range(0, len(ptrs) if n else 0, K.IMAGE_LOOKUP_MAX_OUTPUTS):
            hi = min(lo + K.IMAGE_LOOKUP_MAX_OUTPUTS, len(ptrs))
            nc = hi - lo
            if keyed:
                check(_lib.load().nvt_keydir_lookup_image(
                    k.data_ptr(), K.dtype_code(k.dtype), K.ptr(valid), n, self._dir.data_ptr(),
                    self.dir_slots, self.keys32.data_ptr(), self.n, self.key_offset, self.null_group,
                    image.data_ptr(),
                    stride, nc, _lib.ptr_array(ptrs[lo:hi]), _lib.ptr_array(folds[lo:hi]),
                    (C.c_uint32 * nc)(*offs[lo:hi]), (C.c_uint32 * nc)(*sizes[lo:hi]),
                    (C.c_uint64 * nc)(*miss[lo:hi]), unseen.data_ptr(), K.stream_ptr()),
                    "nvt_keydir_lookup_image")
                continue
            check(_lib.load().nvt_flat_lookup_image(
                k.data_ptr(), K.dtype_code(k.dtype), K.ptr(valid), n, self.aux.data_ptr(),
                self.table.data_ptr(), self.capacity, self.key_offset, None, None, image.data_ptr(),
                stride, nc, _lib.ptr_array(ptrs[lo:hi]), _lib.ptr_array(folds[lo:hi]),
                (C.c_uint32 * nc)(*offs[lo:hi]), (C.c_uint32 * nc)(*sizes[lo:hi]),
                (C.c_uint64 * nc)(*miss[lo:hi]), unseen.data_ptr(), K.stream_ptr()),
                "nvt_flat_lookup_image")
        if memo is not None and hit is None:
            memo[mkey] = dict(outs=outs, unseen=unseen, keep=(k, valid, keep))
        elif hit is not None:
            hit["outs"].update(outs)   # (a consumer attached after the pass's first launch)
        return outs[id(consumer)], unseen


def _value_bits(value, dt) -> int:
    import struct

    if callable(value):   # (a number that was still on its way to the host when the consumer was made)
        value = value()

    if dt == torch.float32:
        return struct.unpack("<I", struct.pack("<f", float(value)))[0]
    if dt == torch.float64:
        return struct.unpack("<Q", struct.pack("<d", float(value)))[0]
    if dt == torch.int32:
        return int(value) & 0xFFFFFFFF
    return int(value) & 0xFFFFFFFFFFFFFFFF


class LookupConsumer:
    """What one operator's transform hands a row of a key column, as a byte range of the packed
    per-group record of FlatIndex.image_lookup.

    outputs: [(name, torch dtype, byte offset inside the range, per_fold, value of a row without
    group)]; per_fold outputs hold (kfold + 1) consecutive values (slot 0: no fold) and need
    fold_fn(n, device) -> uint8 fold ids.  fill(image, stride, offset, groups) writes the range
    of the first `groups` records."""

    def __init__(self, owner, tag, width, outputs, fill, fold_fn=None, groups=0, parts=None):
        self.owner, self.tag, self.width = owner, tag, int(width)
        self.outputs, self.fill, self.fold_fn = list(outputs), fill, fold_fn
        # parts(offset, groups) -> [(ImagePart, keep-alive)]: the same range as descriptors of the
        # one-pass build (nvt_image_build); None: only fill() can write it
        self.parts = parts
        self.groups = int(groups)   # records this consumer fills (the groups of its statistics)
        self.index = None

    def release(self):
        """The owner clears its fit: leave the index and drop what this consumer holds NOW.  The
        closures reference the operator, the operator references the consumer: without this the
        multi-GB image and statistics of a fit wait for Python's cycle collector, and the next
        fit's allocations miss the cached blocks (hipMalloc of several GB inside a step)."""
        idx, self.index = self.index, None
        if idx is not None:
            cons = getattr(idx, "consumers", None)
            if cons is not None and self in cons:
                cons.remove(self)
            idx._image = None
        self.fill = self.fold_fn = self.owner = self.parts = None


def image_pack(image, stride, columns, groups):
    """columns: [(float64 / int64 tensor [groups], output torch dtype, absolute byte offset)]."""
    if not columns or not groups:
        return
    nc = len(columns)
    srcs = [c[0].contiguous() for c in columns]
    for t in srcs:
        assert t.dtype in (torch.float64, torch.int64) and int(t.numel()) >= groups
    check(_lib.load().nvt_image_pack(
        _lib.ptr_array([t.data_ptr() for t in srcs]), (C.c_int * nc)(*[K.dtype_code(t.dtype) for t in srcs]),
        (C.c_int * nc)(*[K.dtype_code(c[1]) for c in columns]), (C.c_uint32 * nc)(*[int(c[2]) for c in columns]),
        nc, int(groups), image.data_ptr(), int(stride), K.stream_ptr()), "nvt_image_pack")


JG_KINDS = {"count": 0, "sum": 1, "mean": 2, "min": 3, "max": 4, "var": 5, "std": 6}


def jg_image(image, stride, comp, outputs, groups):
    """outputs: [(statistic name, value column index, output torch dtype, absolute byte offset)]
    evaluated per group from the accumulators of `comp` (count / sum / sumsq / min / max)."""
    if not outputs or not groups:
        return
    nvals = len(comp["sum"])
    count = comp["count"].to(torch.int64).contiguous()

    def arr(name):
        lst = comp.get(name) or []
        if len(lst) != nvals:
            return None, []
        keep = [t.to(torch.float64).contiguous() for t in lst]
        return _lib.ptr_array([t.data_ptr() for t in keep]), keep

    ps, ks = arr("sum")
    pq, kq = arr("sumsq")
    pmn, kmn = arr("min")
    pmx, kmx = arr("max")
    nc = len(outputs)
    check(_lib.load().nvt_jg_image(
        count.data_ptr(), ps, pq, pmn, pmx, nvals, (C.c_int * nc)(*[JG_KINDS[o[0]] for o in outputs]),
        (C.c_int * nc)(*[int(o[1]) for o in outputs]), (C.c_int * nc)(*[K.dtype_code(o[2]) for o in outputs]),
        (C.c_uint32 * nc)(*[int(o[3]) for o in outputs]), nc, int(groups), image.data_ptr(), int(stride),
        K.stream_ptr()), "nvt_jg_image")
    del ks, kq, kmn, kmx


def jg_image_part(comp, outputs, groups):
    """The arguments of jg_image as a part of the one-pass build: (ImagePart, keep-alive)."""
    nvals = len(comp["sum"])
    count = comp["count"].to(torch.int64).contiguous()
    keep = [count]

    def arr(name):
        lst = comp.get(name) or []
        if len(lst) != nvals or not nvals:
            return None
        ts = [t.to(torch.float64).contiguous() for t in lst]
        pa = _lib.ptr_array([t.data_ptr() for t in ts])
        keep.extend(ts)
        keep.append(pa)
        return C.cast(pa, C.c_void_p)

    nc = len(outputs)
    kinds = (C.c_int32 * nc)(*[JG_KINDS[o[0]] for o in outputs])
    vals = (C.c_int32 * nc)(*[int(o[1]) for o in outputs])
    dts = (C.c_int32 * nc)(*[K.dtype_code(o[2]) for o in outputs])
    offs = (C.c_uint32 * nc)(*[int(o[3]) for o in outputs])
    keep += [kinds, vals, dts, offs]
    part = _lib.ImagePart(kind=_lib.IMAGE_PART_JG, nvals=nvals, ncols=nc, groups=int(groups),
                          count=count.data_ptr(), sum=arr("sum"), sumsq=arr("sumsq"), mn=arr("min"),
                          mx=arr("max"), kinds=C.cast(kinds, C.c_void_p), vals=C.cast(vals, C.c_void_p),
                          dst_dtypes=C.cast(dts, C.c_void_p), offs=C.cast(offs, C.c_void_p))
    return part, keep


def te_image_part(offset, tot_count, tot_sum, fold_count, fold_sum, kfold, groups, p_smooth, y_mean,
                  out_dtype, moments=None):
    """The arguments of te_image as a part of the one-pass build: (ImagePart, keep-alive).
    ``moments``: float64 {count, sum, ...} of the target on the device -- the kernel then takes
    y_mean = sum / count from there (a fit whose mean has not reached the host yet)."""
    tc, ts = tot_count.contiguous(), tot_sum.contiguous()
    assert tc.dtype == torch.int64 and ts.dtype == torch.float64
    fc = fs = None
    if kfold:
        fc, fs = fold_count.contiguous(), fold_sum.contiguous()
        assert fc.dtype == torch.int64 and fs.dtype == torch.float64
        assert int(fc.numel()) >= groups * kfold and int(fs.numel()) >= groups * kfold
    part = _lib.ImagePart(kind=_lib.IMAGE_PART_TE, kfold=int(kfold), out_dtype=K.dtype_code(out_dtype),
                          offset=int(offset), groups=int(groups), tot_count=tc.data_ptr(),
                          tot_sum=ts.data_ptr(), fold_count=K.ptr(fc), fold_sum=K.ptr(fs),
                          p_smooth=float(p_smooth), y_mean=float(y_mean), moments=K.ptr(moments))
    if moments is not None:
        assert moments.dtype == torch.float64 and moments.is_contiguous() and int(moments.numel()) >= 2
    return part, [tc, ts, fc, fs, moments]


def te_image(image, stride, offset, tot_count, tot_sum, fold_count, fold_sum, kfold, groups, p_smooth,
             y_mean, out_dtype):
    """(kfold + 1) smoothed values per group at `offset` of the records of `image`, from the
    totals [groups] and the dense per-(group, fold) statistics [groups * kfold] (kfold 0: totals
    only)."""
    if not groups:
        return
    tc, ts = tot_count.contiguous(), tot_sum.contiguous()
    assert tc.dtype == torch.int64 and ts.dtype == torch.float64
    fc = fs = None
    if kfold:
        fc, fs = fold_count.contiguous(), fold_sum.contiguous()
        assert fc.dtype == torch.int64 and fs.dtype == torch.float64
        assert int(fc.numel()) >= groups * kfold and int(fs.numel()) >= groups * kfold
    check(_lib.load().nvt_te_image(tc.data_ptr(), ts.data_ptr(), K.ptr(fc), K.ptr(fs), int(kfold), int(groups),
                                   float(p_smooth), float(y_mean), K.dtype_code(out_dtype), image.data_ptr(),
                                   int(stride), int(offset), K.stream_ptr()), "nvt_te_image")
