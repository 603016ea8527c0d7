"""Read-backs that do not leave the device idle (kernels.PendingReadBack): a fit posts its scalars
and the transform that follows takes them from the device -- Normalize's mean / std are finished by
the kernel exactly as ops/normalize.py finalize_moments finishes them on the host
(moments.py:89-116, normalize.py:79-82)."""
import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pending_read_backs_arrive_in_any_order():
    from nvtabular_amd import kernels as K

    dev = torch.device("cuda")
    ts = [torch.arange(i, i + 5 + i, dtype=torch.int64, device=dev) for i in range(6)]
    fs = [torch.linspace(0, 1, 3 + i, dtype=torch.float64, device=dev) for i in range(4)]
    pend = [K.PendingReadBack(t) for t in ts] + [K.PendingReadBack(t) for t in fs]
    x = torch.ones(1 << 20, device=dev).cumsum(0)   # (work queued behind the posts)
    for p, t in reversed(list(zip(pend, ts + fs))):
        np.testing.assert_array_equal(p.get(), t.cpu().numpy())
        np.testing.assert_array_equal(p.get(), t.cpu().numpy())   # (a second get: the same value)
    assert float(x[-1]) == float(1 << 20)
    # the mailboxes went back to the pool: the next round reuses them
    again = [K.PendingReadBack(t) for t in ts]
    for p, t in zip(again, ts):
        np.testing.assert_array_equal(p.get(), t.cpu().numpy())
    assert K.PendingReadBack(torch.empty(0, dtype=torch.int64, device=dev)).get().size == 0
    # a read-back nobody asks for gives its mailbox back (fits without a transform do not leak)
    pool = K.PendingReadBack._free
    before = sum(len(v) for v in pool.values())
    for _ in range(50):
        K.PendingReadBack(ts[0])
    torch.cuda.synchronize()
    assert sum(len(v) for v in pool.values()) <= before + 1
    np.testing.assert_array_equal(K.PendingReadBack(ts[1]).get(), ts[1].cpu().numpy())


@pytest.mark.parametrize("out_dtype", [None, np.float32])
def test_normalize_from_device_moments_equals_host_moments(out_dtype):
    """The first transform after a fit reads {count, sum, sum of squares} on the device, every
    later one the host's numbers: the same bits -- incl. a constant column (std 0: x - mean), a
    column with one valid row (std NaN) and an all-null column (mean NaN)."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(12)
    n = 100_003
    df = pd.DataFrame({
        "a": rng.normal(3.0, 2.0, n),
        "b": rng.integers(-1000, 100_000, n).astype(np.int32),
        "c": np.full(n, 7.25),
        "d": np.where(np.arange(n) == 17, 2.5, np.nan),
        "e": np.full(n, np.nan),
        "f": rng.random(n).astype(np.float32) * 1e6,
    })
    df.loc[rng.random(n) < 0.1, "a"] = np.nan
    norm = ops.Normalize(out_dtype=out_dtype)
    wf = nvt.Workflow(["a", "b", "c", "d", "e", "f"] >> norm)
    wf.fit(nvt.Dataset(df))
    assert norm._pending is not None                      # nobody waited for the moments
    first = wf.transform(nvt.Dataset(df)).to_ddf().compute()   # device moments
    assert norm._pending is None and set(norm.means) == set(df.columns)
    second = wf.transform(nvt.Dataset(df)).to_ddf().compute()  # host moments
    for c in df.columns:
        np.testing.assert_array_equal(first[c].to_numpy().view(np.uint8), second[c].to_numpy().view(np.uint8),
                                      err_msg=c)
    # ... and what the host holds is the reference's arithmetic on the accumulators
    x = df["a"].to_numpy()
    x = x[~np.isnan(x)]
    assert norm.means["a"] == pytest.approx(x.mean(), rel=1e-12)
    assert norm.stds["a"] == pytest.approx(x.std(ddof=1), rel=1e-9)
    assert norm.stds["c"] == 0.0 and np.isnan(norm.stds["d"]) and np.isnan(norm.means["e"])
    np.testing.assert_array_equal(first["c"].to_numpy(), np.zeros(n))


def test_fill_missing_and_normalize_after_fit_reads_means_first():
    """Reading means / stds before the transform resolves the pending read-back: the transform then
    takes the host numbers (the path every test of round 1-5 exercised)."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(1)
    df = pd.DataFrame({"x": rng.normal(size=50_000)})
    df.loc[::7, "x"] = np.nan
    norm = ops.Normalize()
    wf = nvt.Workflow(["x"] >> ops.FillMissing(0.5) >> norm).fit(nvt.Dataset(df))
    m, s = norm.means["x"], norm.stds["x"]
    assert norm._pending is None
    got = wf.transform(nvt.Dataset(df)).to_ddf().compute()["x"].to_numpy()
    filled = df["x"].fillna(0.5).to_numpy()
    np.testing.assert_array_equal(got, (filled - m) / s)
