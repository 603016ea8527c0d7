"""2 ranks (one GPU, gloo): each rank fits its OWN frame; the merged vocabularies and moments
must equal a single-process fit of the concatenated frames, and every rank must encode its
own rows with them."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch
import torch.distributed as td

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
td.init_process_group("gloo", rank=rank, world_size=world)
import nvtabular_amd as nvt
from nvtabular_amd import ops


def make(r, n=200_000):
    rng = np.random.default_rng(100 + r)
    df = pd.DataFrame({
        "a": (np.minimum(rng.zipf(1.2, n), 50_000) * 7919 % 1_000_003).astype("int32"),
        "b": rng.integers(0, 30, n).astype("int64"),
        "x": rng.normal(3.0, 2.0, n)})
    df.loc[rng.random(n) < 0.1, "x"] = np.nan
    return df


mine = make(rank)
tmp = tempfile.mkdtemp()
wf = nvt.Workflow((["a", "b"] >> ops.Categorify(out_path=os.path.join(tmp, f"r{rank}")))
                  + (["x"] >> ops.FillMissing() >> ops.Normalize()))
got = wf.fit_transform(nvt.Dataset(mine)).to_ddf().compute()


def stat_graph(path):
    # one int32 key column, 200 k rows per rank: every rank fits on the sort path, the ranks merge
    # compacted groups (the dense fold statistics as [fold, key] groups)
    te = ["a"] >> ops.TargetEncoding("x", kfold=5, fold_seed=42, p_smooth=20, out_path=os.path.join(path, "te"))
    jg = ["a"] >> ops.JoinGroupby(cont_cols=["x"], stats=["count", "sum", "mean"], out_path=os.path.join(path, "jg"))
    return nvt.Workflow(te + jg)


got_s = stat_graph(os.path.join(tmp, f"s{rank}")).fit_transform(nvt.Dataset(mine)).to_ddf().compute()


def int32_frame(r, n=300_000):
    # every categorical int32 (what the bench feeds): the exchange takes the device path --
    # nvt_exchange_* launches, (count << 32 | key) rows, owner merge by sorting
    rng = np.random.default_rng(500 + r)
    return pd.DataFrame({
        "p": (np.minimum(rng.zipf(1.15, n), 200_000) * 2654435761 % 2**31).astype("int32"),
        "q": rng.integers(-40, 40, n).astype("int32"),
        "s": (rng.integers(0, 3_000_000, n) - 1_500_000).astype("int32"),
        "t": np.full(n, 7 + r, dtype="int32")})


from nvtabular_amd import dist as _dist

from nvtabular_amd import kernels as _K

before = dict(_dist.STATS)
labelled0 = _K.STATS.get("labelled_vocabularies", 0)
cat32 = ops.Categorify(out_path=os.path.join(tmp, f"i{rank}"))
wf32 = nvt.Workflow(["p", "q", "s", "t"] >> cat32)
got32 = wf32.fit_transform(nvt.Dataset(int32_frame(rank))).to_ddf().compute()
# all four vocabularies were laid out from the labels the owners computed (no ordering pass)
assert _K.STATS.get("labelled_vocabularies", 0) == labelled0 + 4, _K.STATS
# key-sorted lists (range / sort path) + short unsorted ones: groups travel in key order and the
# owners merge sorted runs (no sort of the received rows)
assert _dist.STATS["ordered_exchanges"] == before["ordered_exchanges"] + 1, _dist.STATS
assert _dist.STATS["sorted_merges"] == before["sorted_merges"], _dist.STATS
assert _dist.STATS["packed_exchanges"] == before["packed_exchanges"] + 1, _dist.STATS
assert _dist.STATS["distributed_orders"] == before["distributed_orders"] + 1, _dist.STATS
# the owners ordered their shards (labels travelled with the rows); every rank ordering the whole
# union itself must give the same labels
_dist.DISTRIBUTED_ORDER = False
before = dict(_dist.STATS)
wf32r = nvt.Workflow(["p", "q", "s", "t"] >> ops.Categorify(out_path=os.path.join(tmp, f"ir{rank}")))
got32r = wf32r.fit_transform(nvt.Dataset(int32_frame(rank))).to_ddf().compute()
assert _dist.STATS["distributed_orders"] == before["distributed_orders"], _dist.STATS
assert _dist.STATS["ordered_exchanges"] == before["ordered_exchanges"] + 1, _dist.STATS
_dist.DISTRIBUTED_ORDER = True
for c in ("p", "q", "s", "t"):
    np.testing.assert_array_equal(got32r[c].to_numpy(), got32[c].to_numpy(), err_msg="replicated order " + c)
# the unordered exchange (cursor scatter, owner merge by sorting) stays the fallback: same labels
_dist.ORDERED_EXCHANGE = False
before = dict(_dist.STATS)
wf32u = nvt.Workflow(["p", "q", "s", "t"] >> ops.Categorify(out_path=os.path.join(tmp, f"iu{rank}")))
got32u = wf32u.fit_transform(nvt.Dataset(int32_frame(rank))).to_ddf().compute()
assert _dist.STATS["sorted_merges"] == before["sorted_merges"] + 1, _dist.STATS
_dist.ORDERED_EXCHANGE = True
for c in ("p", "q", "s", "t"):
    np.testing.assert_array_equal(got32u[c].to_numpy(), got32[c].to_numpy(), err_msg="unordered " + c)
# fewer partitions than ranks: a parquet dataset of ONE row group -- rank 1 decodes nothing and
# passes empty tables into the exchange (they used to be int64 empties that switched that rank
# alone to the two-word wire format: mismatched byte counts in the all-to-all)
lone_path = os.path.join(tempfile.gettempdir(), "nvt_multirank_lone.parquet")
if rank == 0:
    int32_frame(9, 50_000).to_parquet(lone_path)
td.barrier()
before = dict(_dist.STATS)
wf_lone = nvt.Workflow(["p", "q", "s"] >> ops.Categorify(out_path=os.path.join(tmp, f"l{rank}")))
lone_ds = nvt.Dataset(lone_path, engine="parquet")
assert lone_ds.npartitions == 1
wf_lone.fit(lone_ds)
assert _dist.STATS["packed_exchanges"] == before["packed_exchanges"] + 1, _dist.STATS   # one-word rows on BOTH ranks
got_lone = wf_lone.transform(int32_frame(9, 50_000))
td.barrier()
# single-process reference on the union (world_size() is 1 inside this block)
td.destroy_process_group()
ref_lone = nvt.Workflow(["p", "q", "s"] >> ops.Categorify(out_path=os.path.join(tmp, f"refl{rank}")))
exp_lone = ref_lone.fit_transform(nvt.Dataset(int32_frame(9, 50_000))).to_ddf().compute()
for c in ("p", "q", "s"):
    np.testing.assert_array_equal(got_lone[c].to_numpy(), exp_lone[c].to_numpy(), err_msg="lone " + c)
full = pd.concat([make(r) for r in range(world)], ignore_index=True)
ref = nvt.Workflow((["a", "b"] >> ops.Categorify(out_path=os.path.join(tmp, f"ref{rank}")))
                   + (["x"] >> ops.FillMissing() >> ops.Normalize()))
exp_all = ref.fit_transform(nvt.Dataset(full)).to_ddf().compute()
lo = sum(len(make(r)) for r in range(rank))
exp = exp_all.iloc[lo: lo + len(mine)].reset_index(drop=True)
for c in ("a", "b"):
    np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy(), err_msg=c)
np.testing.assert_allclose(got["x"].to_numpy(), exp["x"].to_numpy(), rtol=1e-9, atol=1e-12)
# TargetEncoding / JoinGroupby: the reference on the union keeps the ranks' frames as its
# partitions (fold ids are drawn per partition, target_encoding.py:427-439)
ref_s = stat_graph(os.path.join(tmp, f"refs{rank}"))
exp_s = ref_s.fit_transform(nvt.Dataset([make(r) for r in range(world)])).to_ddf().compute()
exp_s = exp_s.iloc[lo: lo + len(mine)].reset_index(drop=True)
assert list(got_s.columns) == list(exp_s.columns)
for c in got_s.columns:
    if c.endswith("_count"):
        np.testing.assert_array_equal(got_s[c].to_numpy(), exp_s[c].to_numpy(), err_msg=c)
    else:
        np.testing.assert_allclose(got_s[c].to_numpy().astype("float64"), exp_s[c].to_numpy().astype("float64"),
                                   rtol=1e-6, atol=1e-7, err_msg=c)
full32 = pd.concat([int32_frame(r) for r in range(world)], ignore_index=True)
refcat32 = ops.Categorify(out_path=os.path.join(tmp, f"refi{rank}"))
ref32 = nvt.Workflow(["p", "q", "s", "t"] >> refcat32)
exp32 = ref32.fit_transform(nvt.Dataset(full32)).to_ddf().compute()
# the vocabulary files (values in label order with their sizes): laid out from the owners' labels
# on the multi-rank side, ordered by the single process on the other
cat32.flush_artifacts(force=True)
refcat32.flush_artifacts(force=True)
for c in ("p", "q", "s", "t"):
    va = pd.read_parquet(os.path.join(tmp, f"i{rank}", "categories", f"unique.{c}.parquet"))
    vb = pd.read_parquet(os.path.join(tmp, f"refi{rank}", "categories", f"unique.{c}.parquet"))
    np.testing.assert_array_equal(va[c].to_numpy(), vb[c].to_numpy(), err_msg="vocabulary " + c)
    np.testing.assert_array_equal(va[f"{c}_size"].to_numpy(), vb[f"{c}_size"].to_numpy(), err_msg="sizes " + c)
lo32 = sum(len(int32_frame(r)) for r in range(rank))
exp32 = exp32.iloc[lo32: lo32 + len(got32)].reset_index(drop=True)
for c in ("p", "q", "s", "t"):
    np.testing.assert_array_equal(got32[c].to_numpy(), exp32[c].to_numpy(), err_msg=c)
print(f"rank {rank}: multi-rank fit == single-process fit of the union "
      f"({len(mine)} of {len(full)} rows, vocab a = {int(exp_all['a'].max()) - 2})", flush=True)
