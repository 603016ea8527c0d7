"""CPU oracle: a pandas/numpy restatement of NVTabular's hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  The reference package
cannot be imported in this image (``merlin-core`` and ``dask`` are absent), so
each function below restates -- call for call, with the very same pandas
methods -- the reference function it cites.  Dask's task graph is replaced by an
explicit loop over a python list of partitions with the same tree shape.

Parity status: PINNED against the reference's own golden vectors
(``tests/test_oracle_golden.py`` ports them from
``/root/reference/tests/unit/ops/*.py``), except for hashed buckets, which the
reference itself leaves unpinned (``tests/unit/ops/test_hash_bucket.py:51-56``;
``merlin.core.dispatch.hash_series`` is not vendored).  For those the oracle
defines the hash documented in DESIGN.md section 4 and the HIP path must match it.

Reference files restated (all under /root/reference/nvtabular/ops/):
  categorify.py:719-822    _save_encodings        -> save_encodings
  categorify.py:955-1051   _top_level_groupby     -> top_level_groupby
  categorify.py:1054-1070  _mid_level_groupby     -> mid_level_groupby
  categorify.py:1073-1137  _bottom_level_groupby  -> bottom_level_groupby
  categorify.py:1149-1337  _write_uniques         -> write_uniques
  categorify.py:1344-1540  _groupby_to_disk       -> category_stats
  categorify.py:1558-1807  _encode                -> encode
  categorify.py:1837-1852  _hash_bucket           -> hash_bucket
  moments.py:28-116                               -> custom_moments & friends
  normalize.py:71-90, 150-161                     -> normalize_transform, minmax_transform
  fill.py:49-57                                   -> fill_missing
  hash_bucket.py:86-100                           -> hash_bucket_op
  join_groupby.py:140-217                         -> join_groupby_fit/transform
  target_encoding.py:171-214, 301-439             -> target_encoding_fit/transform, add_fold
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import pandas as pd

__all__ = [
    "PAD_OFFSET",
    "NULL_OFFSET",
    "OOV_OFFSET",
    "GroupbyOptions",
    "make_name",
    "is_list_series",
    "flatten_list_values",
    "rebuild_list",
    "nvt_hash64",
    "nvt_hash32",
    "hash_values",
    "string_key64",
    "hash_bucket",
    "hash_bucket_op",
    "top_level_groupby",
    "mid_level_groupby",
    "bottom_level_groupby",
    "order_uniques",
    "save_encodings",
    "write_uniques",
    "category_stats",
    "read_vocab",
    "encode",
    "categorify_fit",
    "categorify_transform",
    "emb_sz_rule",
    "embedding_sizes",
    "chunkwise_moments",
    "tree_node_moments",
    "finalize_moments",
    "custom_moments",
    "normalize_transform",
    "minmax_fit",
    "minmax_transform",
    "fill_missing",
    "clip_transform",
    "hashed_cross",
    "groupby_op",
    "bucketize",
    "logop_transform",
    "join_groupby_fit",
    "join_groupby_transform",
    "add_fold",
    "target_encoding_fit",
    "target_encoding_transform",
    "AGG_DTYPES",
]

# categorify.py:53-55
PAD_OFFSET = 0
NULL_OFFSET = 1
OOV_OFFSET = 2

# join_groupby.py:29-34
AGG_DTYPES = {"count": np.int32, "std": np.float32, "var": np.float32, "mean": np.float32}


def make_name(*parts, sep="_"):
    """categorify.py:691-692."""
    return sep.join(parts)


# --------------------------------------------------------------------------
# list-column helpers (merlin.core.dispatch is not vendored; semantics pinned
# by tests/unit/ops/test_categorify.py:128-157 and test_normalize.py:87-107)
# --------------------------------------------------------------------------
def is_list_series(s: pd.Series) -> bool:
    if s.dtype != object:
        return False
    nn = s.dropna()
    return len(nn) > 0 and isinstance(nn.iloc[0], (list, np.ndarray))


def flatten_list_values(s: pd.Series) -> pd.Series:
    leaves = [v for row in s for v in (row if row is not None else [])]
    return pd.Series(leaves, name=s.name)


def rebuild_list(original: pd.Series, flat, dtype=None) -> pd.Series:
    flat = np.asarray(flat)
    if dtype is not None:
        flat = flat.astype(dtype)
    out, pos = [], 0
    for row in original:
        n = len(row)
        out.append(flat[pos : pos + n])
        pos += n
    return pd.Series(out, index=original.index, name=original.name)


def _maybe_flatten(col: str, df: pd.DataFrame) -> pd.DataFrame:
    """categorify.py:1828-1834."""
    if is_list_series(df[col]):
        return flatten_list_values(df[col]).to_frame(col)
    return df


# --------------------------------------------------------------------------
# hashing.  The reference's hash_series is un-vendored and its own tests only
# check range + determinism, so this engine defines its own:
#   h64(key)  = murmur3 fmix64 of the key sign-extended to 64 bits
#   h32(key)  = h64(key) >> 32
#   bucket    = h32 % num_buckets        (combo groups XOR the h64 first)
# --------------------------------------------------------------------------
_M1 = np.uint64(0xFF51AFD7ED558CCD)
_M2 = np.uint64(0xC4CEB9FE1A85EC53)


def nvt_hash64(keys) -> np.ndarray:
    k = np.asarray(keys).astype(np.int64).view(np.uint64).copy()
    with np.errstate(over="ignore"):
        k ^= k >> np.uint64(33)
        k *= _M1
        k ^= k >> np.uint64(33)
        k *= _M2
        k ^= k >> np.uint64(33)
    return k


def nvt_hash32(keys) -> np.ndarray:
    return (nvt_hash64(keys) >> np.uint64(32)).astype(np.uint32)


def string_key64(values) -> np.ndarray:
    """Host front-end for string keys: str -> int64 surrogate key (pandas'
    keyed siphash, the same primitive pandas-backed hash_series builds on).
    The device only ever sees these 64-bit surrogates."""
    arr = np.asarray(values, dtype=object)
    return pd.util.hash_array(arr, categorize=False).view(np.int64)


def hash_values(s: pd.Series) -> np.ndarray:
    """Stand-in for dispatch.hash_series: uint64 per row (nulls hash as 0 key;
    their label is overwritten afterwards, categorify.py:1799-1800)."""
    if s.dtype == object or pd.api.types.is_string_dtype(s.dtype):
        vals = s.fillna("") if s.isna().any() else s
        return nvt_hash64(string_key64(vals.to_numpy()))
    vals = s.fillna(0) if s.isna().any() else s
    return nvt_hash64(vals.to_numpy())


def hash_bucket(df, num_buckets: Dict[str, int], cols: List[str], encode_type="joint"):
    """categorify.py:1837-1852."""
    if encode_type == "joint":
        nb = num_buckets[cols[0]]
        col = df[cols[0]]
        if is_list_series(col):
            col = flatten_list_values(col)
        h = hash_values(col) >> np.uint64(32)
        return (h % np.uint64(nb)).astype(np.int64)
    name = make_name(*cols, sep="_") if len(cols) > 1 else cols[0]
    nb = num_buckets[name]
    acc = np.zeros(len(df), dtype=np.uint64)
    for c in cols:
        acc ^= hash_values(df[c])
    return ((acc >> np.uint64(32)) % np.uint64(nb)).astype(np.int64)


def hashed_cross(df: pd.DataFrame, cols: List[str], num_buckets: int) -> pd.DataFrame:
    """hashed_cross.py:56-67: XOR of the per-column hashes, modulo num_buckets, int32,
    named a_X_b (hash = this engine's documented hash)."""
    acc = np.zeros(len(df), dtype=np.uint64)
    for c in cols:
        acc ^= hash_values(df[c])
    out = pd.DataFrame()
    out["_X_".join(cols)] = ((acc >> np.uint64(32)) % np.uint64(num_buckets)).astype(np.int32)
    return out


def bucketize(df: pd.DataFrame, boundaries: Dict[str, list]) -> pd.DataFrame:
    """bucketize.py:76-94 (use_digitize branch)."""
    out = pd.DataFrame()
    for col, b in boundaries.items():
        out[col] = np.digitize(df[col].values, np.asarray(b), right=False).astype(np.int32)
    return out


def groupby_op(df, selector_names, groupby_cols, sort_cols=None, aggs="list", name_sep="_",
               ascending=True):
    """groupby.py:88-107 (aggregation split), :113-141 (transform), :232-261 (_apply_aggs),
    :287-319 (_first_or_last) -- the pandas branch, literally."""
    import re

    sort_cols = [sort_cols] if isinstance(sort_cols, str) else (sort_cols or [])
    groupby_cols = [groupby_cols] if isinstance(groupby_cols, str) else list(groupby_cols)
    list_aggs, conv_aggs = {}, {}
    if isinstance(aggs, str):
        aggs = {"__all__": [aggs]}
    elif isinstance(aggs, list):
        aggs = {"__all__": aggs}
    for col, v in aggs.items():
        _aggs = v if isinstance(v, list) else [v]
        _conv, _list = [], []
        for a in _aggs:
            if a in ("list", list, "first", "last"):
                a2 = "list" if a == list else a
                if a2 not in _list:
                    _list.append(a2)
                if list not in _conv:
                    _conv.append(list)
            elif a not in _conv:
                _conv.append(a)
        if _conv:
            conv_aggs[col] = _conv
        if _list:
            list_aggs[col] = _list
    allowed = [c for c in selector_names if c not in groupby_cols]

    def ensure(d):
        if "__all__" in d:
            return {c: d["__all__"] for c in allowed}
        return {k: v for k, v in d.items() if k in allowed}

    _list_aggs, _conv_aggs = ensure(list_aggs), ensure(conv_aggs)
    if sort_cols:
        df = df.sort_values(sort_cols, ascending=ascending, ignore_index=True, kind="stable")
    columns = list(dict.fromkeys(list(groupby_cols) + list(_conv_aggs) + list(_list_aggs)))
    out = df[columns].groupby(groupby_cols).agg(_conv_aggs).reset_index()
    out.columns = [name_sep.join([n for n in name if n != ""]) for name in out.columns.to_flat_index()]
    for col, lst in _list_aggs.items():
        for a in lst:
            if a in ("first", "last"):
                src = out[f"{col}{name_sep}list"]
                first = (a == "first" and ascending) or (a == "last" and not ascending)
                out[f"{col}{name_sep}{a}"] = src.apply((lambda y: y[0]) if first else (lambda y: y[-1]))
        if "list" not in lst:
            out.drop(columns=[col + f"{name_sep}list"], inplace=True)
    for col in out.columns:
        if re.search(f"{name_sep}(count|nunique)$", col):
            out[col] = out[col].astype(np.int32)
        elif re.search(f"{name_sep}(mean|median|std|var|sum)$", col):
            out[col] = out[col].astype(np.float32)
    keep = [c for c in out.columns if c not in groupby_cols or c in selector_names]
    return out[keep]


def hash_bucket_op(df: pd.DataFrame, num_buckets: Union[int, Dict[str, int]], cols=None):
    """hash_bucket.py:86-100 -> int32 (hash_bucket.py:129-131); mutates df."""
    if isinstance(num_buckets, int):
        num_buckets = {c: num_buckets for c in cols}
    for col, nb in num_buckets.items():
        if is_list_series(df[col]):
            flat = flatten_list_values(df[col])
            h = (hash_values(flat) >> np.uint64(32)) % np.uint64(nb)
            df[col] = rebuild_list(df[col], h, dtype=np.int32)
        else:
            h = (hash_values(df[col]) >> np.uint64(32)) % np.uint64(nb)
            df[col] = h.astype(np.int32)
    return df


# --------------------------------------------------------------------------
# groupby tree (Categorify / JoinGroupby / TargetEncoding fit)
# --------------------------------------------------------------------------
@dataclass
class GroupbyOptions:
    """Subset of categorify.py:825-899 FitOptions that changes results."""

    col_groups: List[List[str]]
    agg_cols: List[str] = field(default_factory=list)
    agg_list: List[str] = field(default_factory=list)
    freq_limit: Union[int, dict] = 0
    concat_groups: bool = False
    name_sep: str = "-"
    max_size: Optional[Union[int, dict]] = None
    num_buckets: Optional[Union[int, dict]] = None
    split_every: int = 8
    stat_name: str = "categories"

    def __post_init__(self):
        self.col_groups = [[g] if isinstance(g, str) else list(g) for g in self.col_groups]


def _group_key(cols: List[str], opts: GroupbyOptions) -> List[str]:
    if opts.concat_groups and len(cols) > 1:
        return [make_name(*cols, sep=opts.name_sep)]
    return list(cols)


def top_level_groupby(df: pd.DataFrame, opts: GroupbyOptions) -> List[pd.DataFrame]:
    """categorify.py:955-1051, one frame per column group.

    The hash split (:1036-1049) only decides which tree branch a row of the
    partial table travels through; results do not depend on it
    (test_categorify.py:668-704), so it is not restated.
    """
    want_sq = "std" in opts.agg_list or "var" in opts.agg_list
    want_min = "min" in opts.agg_list
    want_max = "max" in opts.agg_list
    out = []
    for cols in opts.col_groups:
        if opts.concat_groups and len(cols) > 1:
            # joint encoding: stack the group's columns into one key column (:972-981)
            name = make_name(*cols, sep=opts.name_sep)
            stacked = pd.concat(
                [_maybe_flatten(c, df)[c] for c in cols], ignore_index=True
            )
            df_gb = pd.DataFrame({name: stacked})
            keys = [name]
        else:
            keys = list(cols)
            df_gb = df[keys + list(opts.agg_cols)].copy(deep=False)

        agg = {}
        base = []
        if "size" in opts.agg_list:
            base.append("size")
        if set(opts.agg_list).difference({"size", "min", "max"}):
            base.append("count")
        agg[keys[0]] = base
        for c in opts.agg_cols:
            agg[c] = ["sum"]
            if want_sq:
                p2 = make_name(c, "pow2", sep=opts.name_sep)
                df_gb[p2] = df_gb[c].pow(2)
                agg[p2] = ["sum"]
            if want_min:
                agg[c].append("min")
            if want_max:
                agg[c].append("max")

        df_gb = _maybe_flatten(keys[0], df_gb)
        gb = df_gb.groupby(keys, dropna=False).agg(agg)
        gb.columns = [
            make_name(*(tuple(keys) + n[1:]), sep=opts.name_sep)
            if n[0] == keys[0]
            else make_name(*(tuple(keys) + n), sep=opts.name_sep)
            for n in gb.columns.to_flat_index()
        ]
        gb.reset_index(inplace=True, drop=False)
        out.append(gb)
    return out


def _agg_kind(col: str) -> str:
    """categorify.py:1140-1146."""
    if col.endswith("_min"):
        return "min"
    if col.endswith("_max"):
        return "max"
    return "sum"


def mid_level_groupby(frames: Sequence[pd.DataFrame], cols: List[str], opts: GroupbyOptions):
    """categorify.py:1054-1070."""
    keys = _group_key(cols, opts)
    df = pd.concat(list(frames), ignore_index=True)
    gb = df.groupby(keys, dropna=False).agg({c: _agg_kind(c) for c in df.columns if c not in keys})
    gb.reset_index(drop=False, inplace=True)
    return gb


def bottom_level_groupby(frames: Sequence[pd.DataFrame], cols: List[str], opts: GroupbyOptions):
    """categorify.py:1073-1137."""
    gb = mid_level_groupby(frames, cols, opts)
    keys = _group_key(cols, opts)
    sep = opts.name_sep
    n_count = make_name(*(keys + ["count"]), sep=sep)
    n_size = make_name(*(keys + ["size"]), sep=sep)
    required = list(keys)
    if "count" in opts.agg_list:
        required.append(n_count)
    if "size" in opts.agg_list:
        required.append(n_size)
    ddof = 1
    for cont in opts.agg_cols:
        n_sum = make_name(*(keys + [cont, "sum"]), sep=sep)
        if "sum" in opts.agg_list:
            required.append(n_sum)
        if "mean" in opts.agg_list:
            n_mean = make_name(*(keys + [cont, "mean"]), sep=sep)
            required.append(n_mean)
            gb[n_mean] = gb[n_sum] / gb[n_count]
        if "min" in opts.agg_list:
            required.append(make_name(*(keys + [cont, "min"]), sep=sep))
        if "max" in opts.agg_list:
            required.append(make_name(*(keys + [cont, "max"]), sep=sep))
        if "var" in opts.agg_list or "std" in opts.agg_list:
            n = gb[n_count]
            x = gb[n_sum]
            x2 = gb[make_name(*(keys + [cont, "pow2", "sum"]), sep=sep)]
            res = x2 - x**2 / n
            div = n - ddof
            div[div < 1] = 1
            res /= div
            res[(n - ddof) == 0] = np.nan
            if "var" in opts.agg_list:
                nm = make_name(*(keys + [cont, "var"]), sep=sep)
                required.append(nm)
                gb[nm] = res
            if "std" in opts.agg_list:
                nm = make_name(*(keys + [cont, "std"]), sep=sep)
                required.append(nm)
                gb[nm] = np.sqrt(res)
    return gb[required]


def order_uniques(df: pd.DataFrame, keys: List[str], tie_break: str = "pandas"):
    """Vocabulary ordering of categorify.py:1296-1324.

    tie_break="pandas": the reference's literal calls (second sort is pandas'
    default *unstable* quicksort, so equal-count order is platform dependent,
    SURVEY HP1).  tie_break="stable": same, but kind="stable" -> (count desc,
    value asc), the deterministic rule the HIP engine implements.
    Returns (frame_to_write, null_size).
    """
    null_size = None
    if len(df):
        df = df.sort_values(keys, na_position="first", ignore_index=True)
        size_col = "_".join(keys + ["size"])
        has_size = size_col in df
        has_nans = df[keys].iloc[0].transpose().isnull().all()
        if hasattr(has_nans, "iloc"):
            has_nans = has_nans[0]
        if has_nans:
            if has_size:
                null_size = df[size_col].iloc[0]
            df = df.iloc[1:]
        else:
            null_size = 0
        if has_size:
            kind = "quicksort" if tie_break == "pandas" else "stable"
            df = df.sort_values(size_col, ascending=False, ignore_index=True, kind=kind)
        return df, null_size
    if hasattr(df, "convert_dtypes"):
        df = df.convert_dtypes()
    df_null = pd.DataFrame({c: [None] for c in keys})
    for c in keys:
        df_null[c] = df_null[c].astype(df[c].dtype)
    return df_null, null_size


def save_encodings(
    df: pd.DataFrame,
    base_path,
    field_name: str,
    preserve_index=False,
    first_n=None,
    freq_threshold=None,
    oov_count=1,
    null_size=None,
) -> str:
    """categorify.py:719-822 for a single (non-collection) frame."""
    os.makedirs(str(base_path), exist_ok=True)
    unique_path = "/".join([str(base_path), f"unique.{field_name}.parquet"])
    meta_path = "/".join([str(base_path), f"meta.{field_name}.parquet"])
    record = True
    oov_size = 0
    unique_count = 0
    unique_size = 0
    size = oov_count + OOV_OFFSET
    _df = df
    _len = len(_df)
    if _len:
        size_col = f"{field_name}_size"
        if size_col not in _df.columns:
            record = False
        if record:
            first_n_local = first_n - size if first_n is not None else _len
            if first_n or freq_threshold:
                removed = None
                if freq_threshold:
                    sizes = _df[size_col]
                    removed = df[(sizes < freq_threshold) & (sizes > 0)]
                    _df = _df[(sizes >= freq_threshold) | (sizes == 0)]
                if first_n and _len > first_n_local:
                    removed = _df.iloc[first_n_local:]
                    _df = _df.iloc[:first_n_local]
                if removed is not None:
                    oov_size += removed[size_col].sum()
                    _len = len(_df)
            unique_size += _df[size_col].sum()
        if not preserve_index:
            _df = _df.copy(deep=False)
            _df.index = pd.RangeIndex(start=size, stop=size + _len, step=1)
        size += _len
        unique_count += _len
        _df.to_parquet(unique_path, compression=None)
    meta = {
        "kind": ["pad", "null", "oov", "unique"],
        "offset": [PAD_OFFSET, NULL_OFFSET, OOV_OFFSET, OOV_OFFSET + oov_count],
        "num_indices": [1, 1, oov_count, unique_count],
    }
    if record:
        meta["num_observed"] = [0, null_size, oov_size, unique_size]
    pd.DataFrame(meta).to_parquet(meta_path)
    return unique_path


def _pick(opt, name):
    if isinstance(opt, dict):
        return opt[name]
    return opt


def write_uniques(frames, base_path, cols: List[str], opts: GroupbyOptions, tie_break="pandas"):
    """categorify.py:1149-1337 (split_out == 1 branch)."""
    keys = _group_key(cols, opts)
    col_name = keys[0]
    max_emb = _pick(opts.max_size, col_name) if opts.max_size else opts.max_size
    nb = _pick(opts.num_buckets, col_name) if opts.num_buckets else opts.num_buckets
    oov_count = nb or 1
    freq = _pick(opts.freq_limit, col_name) if opts.freq_limit else opts.freq_limit
    if max_emb and max_emb < oov_count + 2:
        raise ValueError("`max_size` can never be less than max(num_buckets + 2, 3)")
    df = pd.concat(list(frames), ignore_index=True)
    df_write, null_size = order_uniques(df, keys, tie_break=tie_break)
    return save_encodings(
        df_write,
        base_path,
        make_name(*keys, sep=opts.name_sep),
        first_n=max_emb,
        freq_threshold=freq,
        oov_count=oov_count,
        null_size=null_size,
    )


def category_stats(
    partitions: Sequence[pd.DataFrame],
    opts: GroupbyOptions,
    out_path,
    write="uniques",
    tie_break="pandas",
) -> Dict[str, str]:
    """categorify.py:1344-1540: level-1 groupby per partition, tree of fan-in
    split_every, final node, then write.  write="uniques" -> unique.<col>.parquet
    (Categorify); write="stats" -> cat_stats.<name>.parquet (JoinGroupby/TE)."""
    level = [top_level_groupby(p, opts) for p in partitions]  # [partition][group]
    base = os.path.join(str(out_path), opts.stat_name)
    os.makedirs(base, exist_ok=True)
    paths = {}
    for g, cols in enumerate(opts.col_groups):
        nodes = [lv[g] for lv in level]
        fan = opts.split_every or 8
        while len(nodes) > fan:
            nodes = [
                mid_level_groupby(nodes[i : i + fan], cols, opts) for i in range(0, len(nodes), fan)
            ]
        final = bottom_level_groupby(nodes, cols, opts)
        name = make_name(*_group_key(cols, opts), sep=opts.name_sep)
        if write == "uniques":
            paths[name] = write_uniques([final], base, cols, opts, tie_break=tie_break)
        else:
            # categorify.py:1493-1510: dask to_parquet directory, no index
            full = make_name(*cols, sep=opts.name_sep)
            d = os.path.join(base, f"cat_stats.{full}.parquet")
            os.makedirs(d, exist_ok=True)
            final.to_parquet(os.path.join(d, "part.0.parquet"), index=False)
            paths[full] = d
    return paths


def read_vocab(path, columns=None) -> pd.DataFrame:
    """fetch_table_data(..., cats_only=True): frame [labels, <col>...]."""
    value = pd.read_parquet(path, columns=columns)
    value.index = value.index.rename("labels")
    value.reset_index(drop=False, inplace=True)
    return value


def encode(
    name,
    storage_name,
    path,
    df: pd.DataFrame,
    buckets=None,
    encode_type="joint",
    cat_names=None,
    dtype=None,
    single_table=False,
):
    """categorify.py:1558-1807, pandas branch (merge + sort by 'order')."""
    if isinstance(buckets, int):
        buckets = {n: buckets for n in cat_names}
    sel_l = list(name) if isinstance(name, list) else [name]
    sel_r = list(name) if isinstance(name, list) else [storage_name]
    list_col = any(is_list_series(df[c]) for c in sel_l)
    if list_col and len(sel_l) != 1:
        raise ValueError("Can't categorical encode multiple list columns")
    if buckets and storage_name in buckets:
        n_oov = buckets[storage_name]
    else:
        n_oov = 1
    value = None
    if path:
        if len(df):
            value = read_vocab(path, columns=sel_r)
            if len(value) and value["labels"].iloc[0] < OOV_OFFSET + n_oov:
                value["labels"] += OOV_OFFSET + n_oov
    if value is None:
        value = pd.DataFrame()
        for c in sel_r:
            typ = df[sel_l[0]].dtype if len(sel_l) == 1 else df[c].dtype
            value[c] = pd.Series([None]).astype(typ) if typ != object else pd.Series([None])
        value.index = value.index.rename("labels")
        value.reset_index(drop=False, inplace=True)

    null_off = value["labels"].head(1).iloc[0] if single_table else NULL_OFFSET
    bucket_off = null_off + 1
    expr = df[sel_l[0]].isna()
    for n in sel_l[1:]:
        expr = expr & df[n].isna()
    nulls = np.flatnonzero(expr.to_numpy())

    if list_col:
        codes = flatten_list_values(df[sel_l[0]]).to_frame(sel_l[0])
        codes["order"] = np.arange(len(codes))
    else:
        codes = pd.DataFrame({"order": np.arange(len(df))}, index=df.index)
    for cl, cr in zip(sel_l, sel_r):
        first = df[cl].dropna()
        if len(first) and isinstance(first.iloc[0], (np.ndarray, list)):
            codes[cl] = flatten_list_values(df[cl]).astype(value[cr].dtype).to_numpy()
        else:
            codes[cl] = df[cl].copy().astype(value[cr].dtype)

    indistinct = bucket_off
    if buckets and storage_name in buckets:
        indistinct = hash_bucket(df, buckets, sel_l, encode_type=encode_type) + bucket_off
        merged = codes.merge(value, left_on=sel_l, right_on=sel_r, how="left").sort_values("order")
        merged.reset_index(drop=True, inplace=True)
        lab = merged["labels"]
        labels = lab.where(lab.notna(), pd.Series(indistinct)).to_numpy()
    else:
        lab = codes.merge(value, left_on=sel_l, right_on=sel_r, how="left").sort_values("order")[
            "labels"
        ]
        labels = lab.fillna(indistinct).to_numpy()

    if list_col:
        # nulls inside list leaves: a null *row* has no leaves; element nulls -> 1
        leaf_null = np.flatnonzero(flatten_list_values(df[sel_l[0]]).isna().to_numpy())
        if len(leaf_null):
            labels[leaf_null] = null_off
        return rebuild_list(df[sel_l[0]], labels, dtype=dtype or np.int64)
    if len(nulls):
        labels[nulls] = null_off
    return labels.astype(dtype or np.int64, copy=False)


# convenience drivers ------------------------------------------------------
def categorify_fit(
    partitions,
    cat_groups,
    out_path,
    freq_threshold=0,
    max_size=0,
    num_buckets=None,
    encode_type="joint",
    name_sep="_",
    tie_break="pandas",
    split_every=8,
):
    """Categorify.fit (categorify.py:346-402) + fit_finalize: {storage_name: path}."""
    opts = GroupbyOptions(
        col_groups=cat_groups,
        agg_cols=[],
        agg_list=["size"],
        freq_limit=freq_threshold,
        concat_groups=encode_type == "joint",
        name_sep=name_sep,
        max_size=max_size,
        num_buckets=num_buckets,
        split_every=split_every,
    )
    return category_stats(partitions, opts, out_path, write="uniques", tie_break=tie_break)


def categorify_transform(
    df, cat_groups, categories, num_buckets=None, encode_type="joint", name_sep="_", dtype=None
):
    """Categorify.transform (categorify.py:477-537) for the given groups."""
    out = df.copy(deep=False)
    groups = [[g] if isinstance(g, str) else list(g) for g in cat_groups]
    storage = {}
    for g in groups:
        if len(g) > 1:
            for c in g:
                storage[c] = make_name(*g, sep=name_sep)
    if encode_type == "combo":
        mapping = {make_name(*g, sep=name_sep): g for g in groups}
    else:
        mapping = {c: [c] for g in groups for c in g}
    names = list(mapping)
    for nm, use in mapping.items():
        use_name = use[0] if len(use) == 1 else list(use)
        if use_name != nm or encode_type == "joint":
            sname = storage.get(nm, nm)
        else:
            sname = nm
        out[nm] = encode(
            use_name,
            sname,
            categories[sname],
            df,
            buckets=num_buckets,
            encode_type=encode_type,
            cat_names=names,
            dtype=dtype,
        )
    return out


def emb_sz_rule(n_cat, minimum_size=16, maximum_size=512):
    """categorify.py:687-688."""
    return n_cat, min(max(minimum_size, round(1.6 * n_cat**0.56)), maximum_size)


def embedding_sizes(paths: Dict[str, str], cat_names, buckets=0):
    """categorify.py:663-684."""
    import pyarrow.dataset as pa_ds

    out = {}
    if isinstance(buckets, int):
        buckets = {n: buckets for n in cat_names}
    for col in cat_names:
        path = paths.get(col)
        n = OOV_OFFSET
        if path:
            for frag in pa_ds.dataset(path, format="parquet").get_fragments():
                n += frag.metadata.num_rows
        if isinstance(buckets, dict):
            n += buckets.get(col, 0)
        else:
            n += 1
        out[col] = emb_sz_rule(n)
    return out


# --------------------------------------------------------------------------
# moments / Normalize / FillMissing
# --------------------------------------------------------------------------
def chunkwise_moments(df: pd.DataFrame):
    """moments.py:64-77."""
    vals = {k: pd.DataFrame() for k in ("count", "sum", "squaredsum")}
    for name in df.columns:
        col = df[name]
        if is_list_series(col):
            col = flatten_list_values(col)
        vals["count"][name] = [col.count()]
        vals["sum"][name] = [col.sum().astype("float64")]
        vals["squaredsum"][name] = [col.astype("float64").pow(2).sum()]
    return vals


def tree_node_moments(inputs):
    """moments.py:80-86."""
    out = {}
    for k in ("count", "sum", "squaredsum"):
        parts = [x.get(k) for x in inputs if x.get(k) is not None]
        out[k] = pd.concat(parts, ignore_index=True).sum().to_frame().transpose()
    return out


def finalize_moments(inp, ddof=1):
    """moments.py:89-116."""
    n = inp["count"].iloc[0]
    x = inp["sum"].iloc[0]
    x2 = inp["squaredsum"].iloc[0]
    var = x2 - x**2 / n
    div = n - ddof
    div[div < 1] = 1
    var /= div
    var[(n - ddof) == 0] = np.nan
    out = pd.DataFrame(index=inp["count"].columns)
    out["count"] = n
    out["sum"] = x
    out["sum2"] = x2
    out["mean"] = x / n
    out["var"] = var
    out["std"] = np.sqrt(var)
    return out


def custom_moments(partitions, cols, split_every=32):
    """moments.py:28-61: chunkwise -> tree (fan-in 32) -> finalize."""
    nodes = [chunkwise_moments(p[list(cols)]) for p in partitions]
    while True:
        nodes = [
            tree_node_moments(nodes[i : i + split_every]) for i in range(0, len(nodes), split_every)
        ]
        if len(nodes) == 1:
            break
    return finalize_moments(nodes[0])


def normalize_transform(df, cols, means, stds, out_dtype=None):
    """normalize.py:71-90."""
    new = pd.DataFrame()
    for name in cols:
        values = df[name]
        lst = is_list_series(values)
        if lst:
            values = flatten_list_values(values)
        if stds[name] > 0:
            values = (values - means[name]) / stds[name]
        else:
            values = values - means[name]
        values = values.astype(out_dtype or np.float64)
        if lst:
            values = rebuild_list(df[name], values)
        new[name] = values
    return new


def minmax_fit(partitions, cols):
    """normalize.py:163-186 (dask min/max reductions)."""
    mins = {c: min(p[c].min() for p in partitions) for c in cols}
    maxs = {c: max(p[c].max() for p in partitions) for c in cols}
    return mins, maxs


def minmax_transform(df, cols, mins, maxs, out_dtype=None):
    """normalize.py:150-161."""
    new = pd.DataFrame()
    for name in cols:
        dif = maxs[name] - mins[name]
        if dif > 0:
            new[name] = (df[name] - mins[name]) / dif
        else:
            new[name] = df[name] / (2 * df[name])
        new[name] = new[name].astype(out_dtype or np.float64)
    return new


def clip_transform(df, cols, min_value=None, max_value=None):
    """clip.py:49-55."""
    z = df[list(cols)].copy()
    if min_value is not None:
        z[z < min_value] = min_value
    if max_value is not None:
        z[z > max_value] = max_value
    return z


def logop_transform(df, cols):
    """logop.py:43-53 (float32, log(1 + x))."""
    out = df.copy()
    for name in cols:
        out[name] = np.log(out[name].astype(np.float32) + 1)
    return out


def fill_missing(df, cols, fill_val=0, add_binary_cols=False):
    """fill.py:49-57; mutates df like the reference."""
    if add_binary_cols:
        for c in cols:
            df[f"{c}_filled"] = df[c].isna()
            df[c] = df[c].fillna(fill_val)
    else:
        df[list(cols)] = df[list(cols)].fillna(fill_val)
    return df


# --------------------------------------------------------------------------
# JoinGroupby
# --------------------------------------------------------------------------
def join_groupby_fit(partitions, groups, cont_cols, stats, out_path, name_sep="_"):
    """join_groupby.py:140-173."""
    opts = GroupbyOptions(
        col_groups=groups,
        agg_cols=list(cont_cols),
        agg_list=list(stats),
        concat_groups=False,
        name_sep=name_sep,
    )
    return category_stats(partitions, opts, out_path, write="stats")


def join_groupby_transform(df, groups, categories, name_sep="_"):
    """join_groupby.py:175-217."""
    new_df = pd.DataFrame()
    tmp = "__tmp__"
    df[tmp] = np.arange(len(df), dtype="int32")
    for g in groups:
        sel = [g] if isinstance(g, str) else list(g)
        sname = make_name(*sel, sep=name_sep)
        stat_df = pd.read_parquet(categories[sname])
        tran = df[sel + [tmp]].merge(stat_df, left_on=sel, right_on=sel, how="left")
        tran = tran.sort_values(tmp)
        tran.drop(columns=sel + [tmp], inplace=True)
        new_cols = [c for c in tran.columns if c not in new_df.columns]
        part = tran[new_cols].reset_index(drop=True)
        for col in part.columns:
            for agg in AGG_DTYPES:
                if col.endswith(f"{name_sep}{agg}"):
                    part[col] = part[col].astype(AGG_DTYPES[agg])
        new_df = pd.concat([new_df, part], axis=1)
    df.drop(columns=[tmp], inplace=True)
    return new_df


# --------------------------------------------------------------------------
# TargetEncoding
# --------------------------------------------------------------------------
def add_fold(n: int, kfold: int, fold_seed=None) -> np.ndarray:
    """target_encoding.py:427-439 (re-seeded per partition)."""
    typ = np.min_scalar_type(kfold * 2)
    if fold_seed is None:
        fold = np.arange(n, dtype=typ)
        np.mod(fold, kfold, out=fold)
        return fold
    state = np.random.RandomState(fold_seed)
    return state.choice(np.arange(kfold, dtype=typ), n)


def target_encoding_fit(
    partitions, groups, targets, out_path, kfold=3, fold_seed=42, name_sep="_", target_mean=None
):
    """target_encoding.py:171-214.  Adds '__fold__' to each partition in place."""
    groups = [[g] if isinstance(g, str) else list(g) for g in groups]
    moments = None
    if target_mean is None:
        moments = custom_moments(partitions, targets)
    col_groups = [list(g) for g in groups]
    if kfold > 1:
        for p in partitions:
            if "__fold__" not in p.columns:
                p["__fold__"] = add_fold(len(p), kfold, fold_seed)
        for g in groups:
            col_groups.append(["__fold__"] + g)
    opts = GroupbyOptions(
        col_groups=col_groups,
        agg_cols=list(targets),
        agg_list=["count", "sum"],
        concat_groups=False,
        name_sep=name_sep,
    )
    stats = category_stats(partitions, opts, out_path, write="stats")
    means = {}
    if moments is not None:
        means = {c: float(moments["mean"].loc[c]) for c in moments.index}
    return stats, means


def target_encoding_transform(
    df,
    groups,
    targets,
    stats,
    means,
    kfold=3,
    fold_seed=42,
    p_smooth=20,
    name_sep="_",
    out_dtype=None,
    target_mean=None,
):
    """target_encoding.py:301-420."""
    tmp = "__tmp__"
    df[tmp] = np.arange(len(df), dtype="int32")
    fit_folds = kfold > 1
    if fit_folds:
        df["__fold__"] = add_fold(len(df), kfold, fold_seed)
    y_mean = target_mean or means
    new_df = None
    for g in groups:
        cat_group = [g] if isinstance(g, str) else list(g)
        out_col = [f"TE_{make_name(*cat_group, sep=name_sep)}_{t}" for t in targets]
        agg_all = pd.read_parquet(stats[make_name(*cat_group, sep=name_sep)])
        agg_all.columns = cat_group + ["count_y_all"] + [t + "_sum_y_all" for t in targets]
        if fit_folds:
            cols = ["__fold__"] + cat_group
            agg_f = pd.read_parquet(stats[make_name(*cols, sep=name_sep)])
            agg_f.columns = cols + ["count_y"] + [t + "_sum_y" for t in targets]
            agg_f = agg_f.merge(agg_all, on=cat_group, how="left")
            agg_f["count_y_all"] = agg_f["count_y_all"] - agg_f["count_y"]
            for i, t in enumerate(targets):
                agg_f[t + "_sum_y_all"] = agg_f[t + "_sum_y_all"] - agg_f[t + "_sum_y"]
                agg_f[out_col[i]] = (agg_f[t + "_sum_y_all"] + p_smooth * y_mean[t]) / (
                    agg_f["count_y_all"] + p_smooth
                )
            agg_f = agg_f.drop(
                ["count_y_all", "count_y"]
                + [t + "_sum_y" for t in targets]
                + [t + "_sum_y_all" for t in targets],
                axis=1,
            )
            tran = df[cols + [tmp]].merge(agg_f, on=cols, how="left")
        else:
            cols = cat_group
            for i, t in enumerate(targets):
                agg_all[out_col[i]] = (agg_all[t + "_sum_y_all"] + p_smooth * y_mean[t]) / (
                    agg_all["count_y_all"] + p_smooth
                )
            agg_all = agg_all.drop(["count_y_all"] + [t + "_sum_y_all" for t in targets], axis=1)
            tran = df[cols + [tmp]].merge(agg_all, on=cols, how="left")
        for i, t in enumerate(targets):
            tran[out_col[i]] = tran[out_col[i]].fillna(y_mean[t])
        if out_dtype is not None:
            tran[out_col] = tran[out_col].astype(out_dtype)
        tran = tran.sort_values(tmp, ignore_index=True)
        tran.drop(columns=cols + [tmp], inplace=True)
        tran.index = df.index
        tran = tran.astype(out_dtype or np.float32)
        new_df = tran if new_df is None else pd.concat([new_df, tran], axis=1)
    df.drop(columns=[tmp, "__fold__"] if fit_folds else [tmp], inplace=True)
    return new_df
