"""Concurrent transform on ONE fitted workflow (SURVEY 8(b) "Threading": the reference's dask worker
threads call `transform` concurrently on different partitions of the same fitted operators,
categorify.py:1632,1813): every thread's result equals the serial one, the per-thread pass state is
gone afterwards and no device memory is left behind."""
import threading

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _frame(n, seed):
    rng = np.random.default_rng(seed)
    df = pd.DataFrame({
        "C1": pd.array(rng.zipf(1.2, n) % 5000, dtype="Int32"),
        "C2": pd.array(rng.integers(0, 40, n), dtype="Int32"),
        "K": rng.integers(0, 30_000, n).astype("int32"),   # sort-path groups (lookup images)
        "I1": pd.array(np.floor(rng.lognormal(2, 2, n)).astype("int64"), dtype="Int32"),
        "I2": rng.normal(size=n).astype("float32"),
        "y": rng.random(n).astype("float32"),
    })
    df.loc[rng.random(n) < 0.1, "C1"] = pd.NA
    df.loc[rng.random(n) < 0.3, "I1"] = pd.NA
    return df


def _equal(a, b):
    assert list(a.columns) == list(b.columns)
    for c in a.columns:
        x, y = a[c].to_numpy(), b[c].to_numpy()
        assert x.dtype == y.dtype, c
        np.testing.assert_array_equal(x, y, err_msg=c)   # (same kernels, same inputs: bit for bit)


def test_four_threads_transform_one_fitted_workflow(tmp_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops
    from nvtabular_amd.device import as_device_frame

    parts = [_frame(60_000, 100 + i) for i in range(8)]
    cats = ["C1", "C2"] >> ops.Categorify(out_path=str(tmp_path / "cat"))
    conts = ["I1", "I2"] >> ops.FillMissing() >> ops.Normalize()
    jg = ["K"] >> ops.JoinGroupby(cont_cols=["y"], stats=["count", "mean", "std"], out_path=str(tmp_path / "jg"))
    te = ["K"] >> ops.TargetEncoding("y", kfold=4, fold_seed=7, p_smooth=10, out_path=str(tmp_path / "te"))
    wf = nvt.Workflow(cats + conts + jg + te)
    frames = [as_device_frame(p)[0] for p in parts]
    wf.fit(nvt.Dataset(frames))
    serial = [wf.transform(f).to_pandas() for f in frames]
    torch.cuda.synchronize()
    assert K.current_pass_memo() is None
    base_mem = None
    for rep in range(3):
        results, errors, memo_left = {}, [], []

        def work(tid):
            try:
                for i in range(tid, len(frames), 4):      # 4 threads x 8 partitions, interleaved
                    for _ in range(2):                    # ... each partition twice
                        results[(tid, i)] = wf.transform(frames[i]).to_pandas()
                memo_left.append(K.current_pass_memo())
            except Exception as e:  # noqa: BLE001
                errors.append((tid, repr(e)))

        threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert memo_left == [None] * 4 and K.current_pass_memo() is None
        for (tid, i), got in results.items():
            _equal(got, serial[i])
        del results
        torch.cuda.synchronize()
        mem = torch.cuda.memory_allocated()
        if base_mem is None:
            base_mem = mem
        assert mem <= base_mem, (rep, mem, base_mem)   # nothing pinned by a stale memo / consumer
