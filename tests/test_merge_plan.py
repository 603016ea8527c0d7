"""Host logic of the partition tree merge (kernels.merge_sorted_tree): the pairing plan, with the
device merge replaced by numpy.  Reference: the tree of categorify.py:1423-1478 over
_mid_level_groupby (categorify.py:1054-1070)."""
import numpy as np
import pandas as pd
import torch


def _np_pairs(pairs, want_src=False):
    out = []
    for (ka, ca), (kb, cb) in pairs:
        s = pd.concat([pd.Series(ca.numpy(), index=ka.numpy()),
                       pd.Series(cb.numpy(), index=kb.numpy())]).groupby(level=0).sum().sort_index()
        out.append((torch.from_numpy(s.index.to_numpy().astype(np.int32)),
                    torch.from_numpy(s.to_numpy().astype(np.int64))))
    return out


def test_tree_merges_every_list_once_and_streams_the_big_table_last(monkeypatch):
    from nvtabular_amd import kernels as K

    calls = []

    def fake(pairs, want_src=False):
        calls.append([(int(a[0].numel()), int(b[0].numel())) for a, b in pairs])
        return _np_pairs(pairs)

    monkeypatch.setattr(K, "merge_sorted_pairs", fake)
    rng = np.random.default_rng(0)

    def lst(n):
        k = np.unique(rng.integers(0, 10 * n + 10, n)).astype(np.int32)
        return torch.from_numpy(k), torch.from_numpy(rng.integers(1, 50, k.size).astype(np.int64))

    big = lst(50_000)
    cols = [[big] + [lst(1000) for _ in range(8)], [lst(300), lst(200), lst(100)], [lst(10)], []]
    got = K.merge_sorted_tree(cols)
    for lists, (k, c) in zip(cols, got):
        if not lists:
            assert k.numel() == 0 and c.numel() == 0
            continue
        s = pd.concat([pd.Series(cc.numpy(), index=kk.numpy()) for kk, cc in lists]).groupby(level=0).sum().sort_index()
        np.testing.assert_array_equal(k.numpy(), s.index.to_numpy())
        np.testing.assert_array_equal(c.numpy(), s.to_numpy())
    # 9 lists -> 4 levels; the 50 k-entry table takes part in the LAST level only
    nbig = int(big[0].numel())
    assert len(calls) == 4
    assert all(nbig not in pair for lvl in calls[:-1] for pair in lvl)
    assert any(nbig in pair for pair in calls[-1])
    # all columns of a level share one call: level 1 holds column 0's four pairs and column 1's one
    assert len(calls[0]) == 5
