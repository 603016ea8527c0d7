"""Host-side logic that needs neither the GPU nor the shared library."""
import numpy as np


def test_estimate_distinct_model():
    from nvtabular_amd.kernels import _estimate_distinct as est

    m, n = 1 << 18, 45_000_000
    assert est(2, m, n) < 100
    assert 3 * 3500 <= est(3500, m, n) < 12_000            # stays on the LDS-table path
    assert est(20_000, m, n) > 11_000                       # partitioned path
    assert est(m - 10, m, n) == n                           # (almost) all unique
    for D in (50_000, 1_000_000, 20_000_000):               # uniform model, within the 3x margin
        d = int(D * (1 - np.exp(-m / D)))
        assert D <= est(d, m, n) <= min(n, 3.5 * D + 100)


def test_rank_sharding_semantics(tmp_path):
    """Under torch.distributed a parquet dataset is a global list (rank r takes every world-th
    partition); frames handed over in memory are the rank's own shard and are all kept --
    bench.py builds one frame per rank, which an index filter would silently drop on ranks > 0."""
    import pandas as pd

    from nvtabular_amd.io import Dataset

    parts = [pd.DataFrame({"a": np.arange(4) + 10 * i}) for i in range(5)]
    mem = Dataset(parts)
    assert len(list(mem._host_parts(None, None))) == 5
    assert len(list(mem._host_parts(None, (1, 2)))) == 5      # rank-local: nothing is skipped
    one = Dataset(parts[0])
    assert len(list(one._host_parts(None, (3, 8)))) == 1       # the bench.py case
    path = tmp_path / "d.parquet"
    pd.concat(parts).to_parquet(path, row_group_size=4)
    pq = Dataset(str(path), engine="parquet", row_groups_per_part=1)
    assert pq.npartitions == 5
    firsts = lambda shard: [t.column("a")[0].as_py() for t in pq._host_parts(None, shard)]
    assert firsts(None) == [0, 10, 20, 30, 40]
    assert firsts((0, 2)) == [0, 20, 40] and firsts((1, 2)) == [10, 30]
