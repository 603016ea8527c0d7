"""Device-backed column handed to LambdaOp UDFs.

The reference's LambdaOp (nvtabular/ops/lambdaop.py:19 -> merlin.dag.ops.udf.UDF) calls
``f(col)`` / ``f(col, df)`` with dataframe columns; on its GPU path those are cuDF Series whose
operators run as device kernels.  ``DeviceSeries`` plays that role here for numeric columns:
arithmetic, comparisons, ``astype``, ``fillna``, ``clip``, ``abs``, ``where``, ``isna`` and numpy
ufuncs (``np.log(col + 1)``) run on the HBM-resident buffer through torch-ROCm elementwise
kernels (plumbing: a UDF is arbitrary Python, there is no fixed kernel to hand-write), with pandas'
semantics: an integer column that has nulls behaves as float64 with NaN, ``/`` is float64 true
division, comparisons against a null are False.  Anything else (``.str``, ``.dt``, ``.map``,
``.apply`` ...) raises ``HostFallback`` and LambdaOp re-runs the UDF on pandas on the host.
"""
from __future__ import annotations

import numpy as np
import torch

from .device import DeviceColumn, DeviceFrame, torch_dtype


class HostFallback(Exception):
    """The UDF used something only pandas offers: run it on the host instead."""


def _unpack_valid(col: DeviceColumn) -> torch.Tensor:
    from . import kernels as K

    return K.unpack_bitmap(col.valid, int(col.data.numel()))


class DeviceSeries:
    __array_priority__ = 1000  # numpy defers binary operators to us

    def __init__(self, data: torch.Tensor, name=None):
        self._t = data
        self.name = name

    # ---- construction ------------------------------------------------------------------
    @staticmethod
    def from_column(col: DeviceColumn, name=None) -> "DeviceSeries":
        if col.is_list or col.strings is not None:
            raise HostFallback("list / string column")
        col = col.materialize()
        t = col.data
        if t.dtype == torch.uint8:
            t = t.to(torch.int64)
        if col.valid is not None:
            ok = _unpack_valid(col)
            # pandas' view of a numeric column with nulls: float64 with NaN
            t = torch.where(ok, t.to(torch.float64), torch.full((), float("nan"), dtype=torch.float64,
                                                                device=t.device))
        return DeviceSeries(t, name)

    def to_column(self) -> DeviceColumn:
        t = self._t
        if t.dtype == torch.float16 or t.dtype == torch.bfloat16:
            t = t.to(torch.float32)
        return DeviceColumn(t.contiguous())

    # ---- pandas-ish surface ----------------------------------------------------------------
    @property
    def dtype(self):
        from .device import numpy_dtype

        return numpy_dtype(self._t.dtype)

    @property
    def values(self):
        return self._t

    def __len__(self):
        return int(self._t.numel())

    def _wrap(self, t) -> "DeviceSeries":
        return DeviceSeries(t, self.name)

    @staticmethod
    def _raw(other):
        if isinstance(other, DeviceSeries):
            return other._t
        if isinstance(other, (int, float, bool, np.integer, np.floating, np.bool_)):
            return other.item() if isinstance(other, np.generic) else other
        if isinstance(other, torch.Tensor):
            return other
        raise HostFallback(f"operand of type {type(other).__name__}")

    def _bin(self, other, fn, swap=False):
        a, b = self._t, self._raw(other)
        # numpy / pandas promotion, not torch's: float32 (op) int64 -> float64, and an integer
        # column (op) a Python float -> float64 (torch would give float32 in both cases)
        if isinstance(b, torch.Tensor):
            if a.dtype != b.dtype and a.dtype != torch.bool and b.dtype != torch.bool:
                from .device import numpy_dtype

                dt = torch_dtype(np.result_type(numpy_dtype(a.dtype), numpy_dtype(b.dtype)))
                a, b = a.to(dt), b.to(dt)
        elif isinstance(b, float) and not a.is_floating_point() and a.dtype != torch.bool:
            a = a.to(torch.float64)
        return self._wrap(fn(b, a) if swap else fn(a, b))

    @staticmethod
    def _f64(x):
        return x.to(torch.float64) if isinstance(x, torch.Tensor) else float(x)

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add, True)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return self._bin(o, lambda a, b: torch.sub(torch.as_tensor(a, device=b.device) if not isinstance(a, torch.Tensor) else a, b), True)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul, True)

    def __truediv__(self, o):  # pandas: always float64 true division
        return self._bin(o, lambda a, b: torch.div(self._f64(a), self._f64(b)))

    def __rtruediv__(self, o):
        return self._bin(o, lambda a, b: torch.div(torch.as_tensor(self._f64(a), device=b.device), self._f64(b)), True)

    def _int_divisor_guard(self, o):
        """pandas turns integer // 0 and integer % 0 into inf / NaN (float64); integer division
        by zero on the device is undefined.  A zero scalar, or a whole integer column as the
        divisor (it may hold zeros; checking would be a host synchronisation), goes to the
        host path."""
        b = self._raw(o)
        if self._t.is_floating_point():
            return
        if isinstance(b, torch.Tensor):
            if not b.is_floating_point():
                raise HostFallback("integer // or % by an integer column")
        elif not isinstance(b, float) and b == 0:
            raise HostFallback("integer // or % by zero")

    def __floordiv__(self, o):
        self._int_divisor_guard(o)
        return self._bin(o, lambda a, b: torch.div(a, b, rounding_mode="floor"))

    def __mod__(self, o):
        self._int_divisor_guard(o)
        return self._bin(o, torch.remainder)
    def __pow__(self, o): return self._bin(o, torch.pow)
    def __neg__(self): return self._wrap(-self._t)
    def __abs__(self): return self._wrap(self._t.abs())
    def __invert__(self): return self._wrap(~self._t)
    def __and__(self, o): return self._bin(o, torch.bitwise_and)
    def __or__(self, o): return self._bin(o, torch.bitwise_or)
    def __xor__(self, o): return self._bin(o, torch.bitwise_xor)

    # comparisons: NaN (null) compares False, like pandas
    def __eq__(self, o): return self._bin(o, torch.eq)  # noqa: E704
    def __ne__(self, o): return self._bin(o, torch.ne)
    def __lt__(self, o): return self._bin(o, torch.lt)
    def __le__(self, o): return self._bin(o, torch.le)
    def __gt__(self, o): return self._bin(o, torch.gt)
    def __ge__(self, o): return self._bin(o, torch.ge)
    __hash__ = None

    def abs(self): return abs(self)

    def astype(self, dtype):
        if dtype in (str, "str", object, "object", "string"):
            raise HostFallback("astype(str)")
        try:
            dt = torch_dtype(np.dtype(dtype))
        except Exception:
            raise HostFallback(f"astype({dtype!r})") from None
        t = self._t
        if dt in (torch.int32, torch.int64) and t.is_floating_point() and bool(torch.isnan(t).any()):
            raise ValueError("Cannot convert non-finite values (NA or inf) to integer")  # pandas' error
        return self._wrap(t.to(dt))

    def isna(self):
        t = self._t
        return self._wrap(torch.isnan(t) if t.is_floating_point() else torch.zeros_like(t, dtype=torch.bool))

    isnull = isna

    def notna(self):
        return self._wrap(~self.isna()._t)

    notnull = notna

    def fillna(self, value):
        t = self._t
        if not t.is_floating_point():
            return self
        return self._wrap(torch.where(torch.isnan(t), torch.as_tensor(value, dtype=t.dtype, device=t.device), t))

    def clip(self, lower=None, upper=None):
        return self._wrap(torch.clamp(self._t, min=lower, max=upper))

    def where(self, cond, other=float("nan")):
        c = self._raw(cond)
        o = self._raw(other)
        t = self._t
        if not isinstance(o, torch.Tensor):
            if isinstance(o, float) and not t.is_floating_point():
                t = t.to(torch.float64)
            o = torch.as_tensor(o, dtype=t.dtype, device=t.device)
        return self._wrap(torch.where(c, t, o))

    def mask(self, cond, other=float("nan")):
        return self.where(~DeviceSeries(self._raw(cond)), other)

    def round(self, decimals=0):
        return self._wrap(torch.round(self._t, decimals=decimals))

    # numpy ufuncs: np.log(col), np.sqrt(col), np.exp(col) ...
    _UFUNCS = {
        np.log: torch.log, np.log1p: torch.log1p, np.log2: torch.log2, np.log10: torch.log10,
        np.exp: torch.exp, np.expm1: torch.expm1, np.sqrt: torch.sqrt, np.abs: torch.abs,
        np.absolute: torch.abs, np.negative: torch.neg, np.sin: torch.sin, np.cos: torch.cos,
        np.tanh: torch.tanh, np.floor: torch.floor, np.ceil: torch.ceil, np.sign: torch.sign,
        np.square: torch.square, np.isnan: torch.isnan,
        np.add: torch.add, np.subtract: torch.sub, np.multiply: torch.mul, np.maximum: torch.maximum,
        np.minimum: torch.minimum, np.power: torch.pow,
    }

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        fn = self._UFUNCS.get(ufunc)
        if fn is None or method != "__call__" or kwargs:
            raise HostFallback(f"numpy ufunc {getattr(ufunc, '__name__', ufunc)}")
        args = []
        dev = self._t.device
        for x in inputs:
            r = self._raw(x)
            if not isinstance(r, torch.Tensor):
                r = torch.as_tensor(r, device=dev)
            args.append(r)
        if ufunc in (np.true_divide,):
            args = [a.to(torch.float64) for a in args]
        if len(args) == 1 and not args[0].is_floating_point() and ufunc not in (np.abs, np.absolute, np.negative, np.sign, np.square):
            args[0] = args[0].to(torch.float64)  # numpy promotes integers for transcendental ufuncs
        return self._wrap(fn(*args))

    def __array__(self, *a, **k):
        raise HostFallback("conversion to a numpy array")

    def __getattr__(self, item):
        # .str / .dt / .map / .apply / anything pandas-only
        raise HostFallback(f"Series.{item}")


class DeviceFrameView:
    """The ``df`` argument of a two-parameter UDF: ``df["col"]`` -> DeviceSeries."""

    def __init__(self, frame: DeviceFrame):
        self._frame = frame

    def __getitem__(self, name):
        if not isinstance(name, str):
            raise HostFallback("frame indexing by a non-string key")
        return DeviceSeries.from_column(self._frame[name], name)

    def __contains__(self, name):
        return name in self._frame

    @property
    def columns(self):
        return self._frame.columns

    def __getattr__(self, item):
        raise HostFallback(f"DataFrame.{item}")
