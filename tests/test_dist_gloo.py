"""Multi-GPU fit-statistics merge, exercised with world_size=2 over gloo on CPU.

The choreography in nvtabular_amd/dist.py (owner hashing -> all-to-all(v) ->
owner merge -> all-gather; all-reduce for moments) is backend-agnostic; the two
device steps are injected here with host implementations (test infrastructure)
so the N>1 path is covered without GPUs.  Results must equal a single-process
groupby over the union of both ranks' data.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_owner(keys_list, G):
    import oracle as O

    acc = np.zeros(keys_list[0].numel(), dtype=np.uint64)
    for k in keys_list:
        acc ^= O.nvt_hash64(k.numpy())
    return torch.from_numpy(((acc >> np.uint64(32)) % np.uint64(G)).astype(np.int32))


def _host_merge(keys, counts):
    import pandas as pd

    s = pd.Series(counts.numpy()).groupby(keys.numpy()).sum()
    return torch.from_numpy(s.index.to_numpy()), torch.from_numpy(s.to_numpy())


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td

    td.init_process_group("gloo", rank=rank, world_size=world)
    from nvtabular_amd import dist

    dist.set_backend_fns(_host_owner, _host_merge)
    rng = np.random.default_rng(100 + rank)
    keys = rng.integers(-50, 300, 5000)
    import pandas as pd

    local = pd.Series(1, index=keys).groupby(level=0).sum()
    k = torch.from_numpy(local.index.to_numpy())
    c = torch.from_numpy(local.to_numpy())
    gk, gc, nulls = dist.merge_counts(k, c, nulls=rank + 3)
    # several columns in ONE exchange: int64 + int32 keys, and a column that is empty on rank 1
    k32 = torch.from_numpy(rng.integers(0, 40, 300).astype("int32"))
    l32 = pd.Series(1, index=k32.numpy()).groupby(level=0).sum()
    tabs = [(k, c, [rank + 3, 10]),
            (torch.from_numpy(l32.index.to_numpy()), torch.from_numpy(l32.to_numpy()), [1, 2, 3]),
            (torch.arange(7 if rank == 0 else 0, dtype=torch.int64),
             torch.ones(7 if rank == 0 else 0, dtype=torch.int64), [rank])]
    dist.reset_traffic()
    many = dist.merge_counts_many(tabs)
    # the byte counters every N > 1 bench line carries (dist_breakdown): host arithmetic on shapes
    tr = dist.TRAFFIC
    assert tr and all(v["calls"] >= 1 and v["bytes_sent"] >= 0 and v["bytes_received"] > 0 for v in tr.values())
    assert "all_to_all_single(uneven)" in tr and "all_reduce" in tr
    assert many[1][0].dtype == torch.int32 and many[0][0].dtype == torch.int64
    tot_r = sum(range(world))
    assert many[0][2] == [3 * world + tot_r, 10 * world] and many[1][2] == [world, 2 * world, 3 * world]
    assert many[2][2] == [tot_r]
    # owners hold key RANGES and gather key-sorted shards: every rank receives each column as
    # ONE key-sorted list (no re-sort of the union), with the histogram of min(count, 255) the
    # one-pass vocabulary ordering needs
    for kk, cc, _, info in many:
        assert (np.diff(kk.numpy().astype(np.int64)) > 0).all()
        assert info["sorted_by_key"]
        np.testing.assert_array_equal(info["cls_hist"].numpy(),
                                      np.bincount(np.minimum(cc.numpy(), 255), minlength=256))
        assert info["n_big"] == int((cc.numpy() >= 255).sum())
    # every column int32 (negative keys, INT32_MIN / MAX, a column empty on rank 1): the rows
    # travel as ONE int64 word (count << 32 | key); same result as the two-word format
    kneg = np.unique(np.concatenate([rng.integers(-2**31, 2**31 - 1, 500),
                                     [-2**31, 2**31 - 1, -1, 0]])).astype("int32")
    cneg = rng.integers(1, 400, kneg.size).astype("int64")
    tabs32 = [(torch.from_numpy(l32.index.to_numpy()), torch.from_numpy(l32.to_numpy()), [1]),
              (torch.from_numpy(kneg), torch.from_numpy(cneg), [2, rank]),
              (torch.arange(5 if rank == 0 else 0, dtype=torch.int32),
               torch.full((5 if rank == 0 else 0,), 9, dtype=torch.int64), [])]
    before = dict(dist.STATS)
    packed = dist.merge_counts_many(tabs32)
    assert dist.STATS["packed_exchanges"] == before["packed_exchanges"] + 1
    dist.PACK_COUNT_ROWS = False
    plain = dist.merge_counts_many(tabs32)
    dist.PACK_COUNT_ROWS = True
    for (pk, pc, ps, pi), (uk, uc, us, ui) in zip(packed, plain):
        assert pk.dtype == torch.int32 and torch.equal(pk, uk) and torch.equal(pc, uc) and ps == us
        assert (np.diff(pk.numpy().astype(np.int64)) > 0).all()
    assert packed[1][0][0].item() == -2**31 and packed[1][0][-1].item() == 2**31 - 1
    assert int(packed[1][1].sum()) >= int(cneg.sum())
    # a rank that received no partition passes int64 EMPTIES for every column (Categorify.fit_end):
    # the one-word wire format and the returned key dtype are decided collectively, not from the
    # rank-local dtypes (the odd ranks here would otherwise send / expect two-word rows)
    if rank % 2 == 0:
        lone = [(torch.from_numpy(kneg), torch.from_numpy(cneg), [1, 5]),
                (torch.from_numpy(l32.index.to_numpy()), torch.from_numpy(l32.to_numpy()), [2, 0])]
    else:
        lone = [(torch.empty(0, dtype=torch.int64), torch.empty(0, dtype=torch.int64), [0, 0]),
                (torch.empty(0, dtype=torch.int64), torch.empty(0, dtype=torch.int64), [0, 0])]
    before = dict(dist.STATS)
    got = dist.merge_counts_many(lone)
    assert dist.STATS["packed_exchanges"] == before["packed_exchanges"] + 1
    assert all(t[0].dtype == torch.int32 and t[3]["sorted_by_key"] for t in got)
    assert (np.diff(got[0][0].numpy().astype(np.int64)) > 0).all()
    assert got[0][2] == [(world + 1) // 2, 5 * ((world + 1) // 2)]
    lone_keys = (got[0][0].numpy().copy(), got[0][1].numpy().copy(), kneg if rank % 2 == 0 else None,
                 cneg if rank % 2 == 0 else None)
    q.put(("lone", rank) + lone_keys)
    m0 = pd.Series(many[0][1].numpy(), index=many[0][0].numpy()).sort_index()
    assert sorted(many[2][0].tolist()) == list(range(7)) and many[2][1].tolist() == [1] * 7
    q.put(("many", rank, m0.index.to_numpy(), m0.to_numpy(), many[1][0].numpy(), many[1][1].numpy(),
           k32.numpy()))
    # a whole multi-column table (JoinGroupby / TargetEncoding merge) in ONE all-to-all(v) and
    # ONE all-gather(v): int64 keys, uint8 null mask, float64 sums
    tk = torch.arange(rank * 1000, rank * 1000 + 50, dtype=torch.int64)
    tcols = [tk, (tk % 3).to(torch.uint8), tk.to(torch.float64) * 0.5 + 0.25]
    recv = dist.exchange_rows(tcols, (tk % world).to(torch.int32))
    assert recv[0].dtype == torch.int64 and recv[1].dtype == torch.uint8 and recv[2].dtype == torch.float64
    assert bool((recv[0] % world == rank).all()) and recv[0].numel() > 0
    assert torch.equal(recv[1], (recv[0] % 3).to(torch.uint8))
    assert torch.equal(recv[2], recv[0].to(torch.float64) * 0.5 + 0.25)
    allrows = dist.gather_rows(recv)
    assert sorted(allrows[0].tolist()) == sorted(t for r in range(world) for t in range(r * 1000, r * 1000 + 50))
    assert torch.equal(allrows[2], allrows[0].to(torch.float64) * 0.5 + 0.25)
    mom = dist.all_reduce_sum(torch.tensor([[1.0 + rank, 2.0, 3.0]], dtype=torch.float64))
    mn = dist.all_reduce_min(torch.tensor([float("nan") if rank == 0 else 4.0 + rank - 1, 2.0 + rank]))
    lut = dist.merge_string_luts({rank: f"s{rank}"})
    got = pd.Series(gc.numpy(), index=gk.numpy()).sort_index()
    q.put((rank, got.index.to_numpy(), got.to_numpy(), nulls, mom.tolist(), mn.tolist(), lut, keys))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4])
def test_merge_counts_world2_gloo(world):
    import pandas as pd

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    items = [q.get(timeout=150) for _ in range(3 * len(procs))]
    many = sorted([t[1:] for t in items if t[0] == "many"], key=lambda t: t[0])
    lone = sorted([t[1:] for t in items if t[0] == "lone"], key=lambda t: t[0])
    res = sorted([t for t in items if t[0] not in ("many", "lone")], key=lambda t: t[0])
    # ranks without entries (int64 empties) took part in the same one-word exchange: every rank
    # holds the union of the even ranks' lists
    src = pd.concat([pd.Series(t[4], index=t[3]) for t in lone if t[3] is not None])
    exp_lone = src.groupby(level=0).sum().sort_index()
    for t in lone:
        np.testing.assert_array_equal(t[1], exp_lone.index.to_numpy())
        np.testing.assert_array_equal(t[2], exp_lone.to_numpy())
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    allkeys = np.concatenate([r[7] for r in res])
    exp = pd.Series(1, index=allkeys).groupby(level=0).sum().sort_index()
    for r in res:
        # every rank ends with the same, complete table
        np.testing.assert_array_equal(r[1], exp.index.to_numpy())
        np.testing.assert_array_equal(r[2], exp.to_numpy())
        assert r[3] == 3 * world + sum(range(world))
        assert r[4] == [[float(world + sum(range(world))), 2.0 * world, 3.0 * world]]
        assert r[5] == [4.0, 2.0]
        assert r[6] == {i: f"s{i}" for i in range(world)}
    # the batched exchange: same int64 table, and the int32 column summed over both ranks
    exp32 = pd.Series(1, index=np.concatenate([m[5] for m in many])).groupby(level=0).sum().sort_index()
    for m in many:
        np.testing.assert_array_equal(m[1], exp.index.to_numpy())
        np.testing.assert_array_equal(m[2], exp.to_numpy())
        got32 = pd.Series(m[4], index=m[3]).sort_index()
        np.testing.assert_array_equal(got32.index.to_numpy(), exp32.index.to_numpy())
        np.testing.assert_array_equal(got32.to_numpy(), exp32.to_numpy())


def test_sort_unsorted_lists_orders_only_the_flagged_lists():
    """dist._sort_unsorted_lists (sender side of the ordered exchange): lists flagged unsorted come
    back in key order with their counts, the others untouched; too many entries -> None."""
    from nvtabular_amd import dist

    rng = np.random.default_rng(3)
    k0 = torch.tensor(np.sort(rng.choice(10**6, 500, replace=False)) - 500_000, dtype=torch.int32)
    k1 = torch.tensor(rng.permutation(2000)[:700] - 1000, dtype=torch.int32)
    k2 = torch.tensor([np.iinfo(np.int32).max, np.iinfo(np.int32).min, 0, -1, 7], dtype=torch.int32)
    tabs = [(k0, torch.arange(500), [1]), (k1, torch.arange(700) * 3, [2]), (k2, torch.tensor([5, 4, 3, 2, 1]), [3]),
            (torch.empty(0, dtype=torch.int32), torch.empty(0, dtype=torch.int64), [4])]
    out = dist._sort_unsorted_lists(tabs, [True, False, False, True])
    assert out[0][0] is k0 and out[3][0] is tabs[3][0]
    for j in (1, 2):
        k, c, sc = out[j]
        order = np.argsort(tabs[j][0].numpy(), kind="stable")
        np.testing.assert_array_equal(k.numpy(), tabs[j][0].numpy()[order])
        np.testing.assert_array_equal(c.numpy(), tabs[j][1].numpy()[order])
        assert k.dtype == torch.int32 and sc == tabs[j][2]
    old = dist.SMALL_SORT_MAX
    try:
        dist.SMALL_SORT_MAX = 100
        assert dist._sort_unsorted_lists(tabs, [True, False, False, True]) is None
    finally:
        dist.SMALL_SORT_MAX = old


def test_class_base_diff_integrates_to_the_label_bases():
    """dist._class_base_diff: the kernel's exclusive prefix over the classes 255, 254, ... of the
    difference array (modulo 2^32) must give, for every owner and class c < 255, the number of
    entries of the union that precede the owner's entries of class c in "count descending, key
    ascending" order."""
    from nvtabular_amd import dist

    rng = np.random.default_rng(12)
    G, ncol = 5, 3
    hists = rng.integers(0, 50_000, (G, ncol, 256)).astype(np.int64)
    hists[:, 1, 200:] = 0                       # a column without large counts
    hists[2] = 0                                # an owner without entries
    hists[:, :, 0] = 0                          # (no entry has count 0)
    H = hists.sum(0)
    for r in range(G):
        diff = dist._class_base_diff(torch.tensor(hists), r).numpy().view(np.uint32).astype(np.uint64)
        P = hists[:r].sum(0)
        for j in range(ncol):
            # what cls_scatter_body computes: digit d = 255 - class, exclusive prefix over d
            v = diff[j][::-1]                   # v[d] = diff[255 - d]
            cbase = (np.cumsum(v) - v) % (1 << 32)
            for c in (1, 2, 3, 100, 199, 253, 254):
                want = H[j, 255] + H[j, c + 1:255].sum() + P[j, c]
                assert int(cbase[255 - c]) == int(want) % (1 << 32), (r, j, c)
            assert int(cbase[0]) == 0           # class 255 is compacted from position 0


def _selfcheck_worker(rank, world, port, q):
    import os

    import torch.distributed as td

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    rep = bench.collective_selfcheck(torch.device("cpu"), "gloo")
    q.put((rank, rep))
    td.destroy_process_group()


def test_bench_collective_selfcheck_runs_every_form_over_gloo():
    """bench.collective_selfcheck (the guard in front of a multi-rank run): every collective form
    dist.py uses -- uneven all_to_all_single, all-gather(v) in int64 and int32, all_gather of equal
    matrices, MAX / SUM all-reduce -- runs and agrees with its gloo reference on 3 ranks."""
    world, port = 3, 29731
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_selfcheck_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    reps = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        rep = reps[r]
        assert rep.get("error") is None, rep
        assert rep["ok"] and rep["ok_on_every_rank"], rep
        names = [c["collective"] for c in rep["checks"]]
        assert "all_gather_v(int32)" in names and "all_gather(equal shapes)" in names
