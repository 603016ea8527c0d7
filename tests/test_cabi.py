"""The C-ABI shared library loads and exports every symbol include/nvt_hip.h declares
(no compute calls: this runs without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "nvt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nvt_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from nvtabular_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in nvt_hip.h but not exported"
    # the ctypes table covers the header exactly (no stale or missing bindings)
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.nvt_version() >= 100


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected on the host side with NVT_EINVAL + a message."""
    import ctypes as C

    from nvtabular_amd import _lib

    lib = _lib.load()
    nbytes = C.c_uint64()
    assert lib.nvt_count_table_bytes(3, 64, C.byref(nbytes)) == -1
    assert b"key_bytes" in lib.nvt_last_error()
    assert lib.nvt_count_table_bytes(4, 1 << 20, C.byref(nbytes)) == 0 and nbytes.value == 8 << 20
    assert lib.nvt_encode_table_bytes(8, 1 << 10, C.byref(nbytes)) == 0 and nbytes.value == 16 << 10
    assert lib.nvt_dense_count_ws_bytes(4, 1000, 9, 0, C.byref(nbytes)) == -1
    # round 6: key directory / one-pass images reject bad arguments before any launch
    assert lib.nvt_keydir_build(None, 10, 10, None, None) == -1 and b"null" in lib.nvt_last_error()
    buf = (C.c_uint32 * 64)()
    keys = (C.c_int32 * 4)(1, 2, 3, 4)
    assert lib.nvt_keydir_build(keys, 0, 10, buf, None) == -1          # no keys
    assert lib.nvt_keydir_build(keys, 4, 0, buf, None) == -1           # no buckets
    part = _lib.ImagePart(kind=7)
    arr = (_lib.ImagePart * 1)(part)
    assert lib.nvt_image_build(arr, 1, 16, buf, 64, None) == -1 and b"kind" in lib.nvt_last_error()
    assert lib.nvt_image_build(arr, 5, 16, buf, 64, None) == -1        # more than 4 parts
    assert lib.nvt_image_build(arr, 0, 16, buf, 200, None) == -1       # stride > 192
    assert lib.nvt_image_build(arr, 0, 0, buf, 64, None) == 0          # nothing to do


def test_ops_fail_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pandas as pd

    import nvtabular_amd as nvt
    from nvtabular_amd import _lib, ops

    wf = nvt.Workflow(["a"] >> ops.Categorify())
    with pytest.raises(_lib.NvtHipError, match="no CPU fallback"):
        wf.fit(nvt.Dataset(pd.DataFrame({"a": [1, 2, 3]})))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nvtabular_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, flags=re.M), f
