"""Host front-end for string categoricals.

The hot path (Criteo) is integer ids; string columns are supported by mapping
every string to a 64-bit surrogate key on the host (pandas' keyed siphash,
``pandas.util.hash_array`` -- the primitive the reference's pandas-backed
``hash_series`` builds on) and letting the HIP kernels count / encode the
surrogates.  The column remembers {surrogate -> string} so vocabularies can be
written with the original values and ordered by them.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch


def string_key64(values) -> np.ndarray:
    arr = np.asarray(values, dtype=object)
    return pd.util.hash_array(arr, categorize=False).view(np.int64)


def string_column_to_device(s: pd.Series, device):
    from .device import DeviceColumn, pack_bitmap

    mask = s.isna().to_numpy()
    vals = s.to_numpy(dtype=object)
    keys = np.zeros(len(s), dtype=np.int64)
    lut = {}
    if (~mask).any():
        uniq = pd.unique(vals[~mask])
        ukeys = string_key64(uniq)
        lut = dict(zip(ukeys.tolist(), uniq.tolist()))
        if len(lut) != len(uniq):
            raise ValueError("64-bit surrogate collision between distinct strings")
        keys[~mask] = string_key64(vals[~mask])
    data = torch.from_numpy(keys).to(device)
    valid = torch.from_numpy(pack_bitmap(~mask)).to(device) if mask.any() else None
    return DeviceColumn(data, valid, None, None, lut)
