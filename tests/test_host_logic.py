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
