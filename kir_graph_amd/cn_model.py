"""
Copy-number models -- drop-in for ``graphkir/cn_model.py`` (``Dist``, ``CNgroup``, ``KDEcut``,
``loadCNModel``); plotting methods of the reference are out of scope.

``CNgroup.fit`` (124-169) is a grid search: ``bin_num`` candidate bases x ``bin_num`` depth bins x
7 Gaussians, ~1.75 M pdf evaluations per fit and up to 6 fits per sample (KIR3DL3 refit loop).  The
grid runs on the GPU (``csrc/gk_cn.hip``); histogram and grids are produced by numpy exactly as in
the reference so the only arithmetic difference is exp / log to a few ulp.  ``KDEcut`` (257-349) is
a closed-form Gaussian KDE on ~100 points and stays in numpy.
"""
from __future__ import annotations

import json
from typing import Any

import numpy as np

from ._lib import Device, check, lib
from .utils import NumpyEncoder


class Dist:
    """Abstract CN model (save / load / params)."""

    def __init__(self) -> None:
        self.raw_df: list[Any] = []

    def save(self, filename: str) -> None:
        with open(filename, "w") as f:
            json.dump(self.getParams(), f, cls=NumpyEncoder)

    @classmethod
    def load(cls, filename: str) -> "Dist":
        with open(filename) as f:
            return cls.setParams(json.load(f))

    def getParams(self) -> dict[str, Any]:
        raise NotImplementedError

    @classmethod
    def setParams(cls, data: dict[str, Any]) -> "Dist":
        raise NotImplementedError


_device: Device | None = None


def _dev(dev: Device | None) -> Device:
    global _device
    if dev is not None:
        return dev
    if _device is None:
        _device = Device()
    return _device


class CNgroup(Dist):
    """Linear copy-number distributions (LCND): Gaussians at base*n, base found by grid search."""

    def __init__(self, device: Device | None = None) -> None:
        super().__init__()
        self.bin_num: int = 300
        self.max_cn: int = 7
        self.x_max: float = 1
        self.base: float | None = None
        self.base_dev: float = 0.08
        self.y0_dev: float = 1.5
        self.dev_decay: float = 0.5
        self.dev_decay_neg: float = 0.3
        self.start_base: int = 1
        self.data: list[float] = []
        self.likelihood = np.array([])
        self._device = device

    def getParams(self) -> dict[str, Any]:
        return {
            "method": "CNgroup", "x_max": self.x_max, "base": self.base, "base_dev": self.base_dev,
            "y0_dev": self.y0_dev, "dev_decay": self.dev_decay, "dev_decay_neg": self.dev_decay_neg,
            "bin_num": self.bin_num, "max_cn": self.max_cn, "data": self.data, "likelihood": self.likelihood,
            "start_base": self.start_base, "raw_df": self.raw_df,
        }

    @classmethod
    def setParams(cls, data: dict[str, Any]) -> "CNgroup":
        assert data["method"] == "CNgroup"
        self = cls()
        self.base, self.base_dev, self.x_max = data["base"], data["base_dev"], data["x_max"]
        self.y0_dev, self.dev_decay = data["y0_dev"], data["dev_decay"]
        self.bin_num, self.max_cn, self.data = data["bin_num"], data["max_cn"], data["data"]
        self.raw_df = data.get("raw_df", [])
        self.likelihood = np.array(data["likelihood"])
        self.start_base = data.get("start_base", 1)
        self.dev_decay_neg = data.get("dev_decay_neg", self.dev_decay)
        return self

    # ---- model geometry, as the reference computes it
    def _deviations(self) -> np.ndarray:
        if self.start_base == 1:
            return np.array([self.base_dev * self.y0_dev]
                            + [self.base_dev * (self.dev_decay * (n - 1) + 1) for n in np.arange(1, self.max_cn)],
                            dtype=np.float64)
        if self.start_base == 2:
            out = []
            for n in np.arange(0, self.max_cn):
                if n < self.start_base:
                    out.append(self.base_dev * (self.dev_decay_neg * (self.start_base - n) + 1))
                else:
                    out.append(self.base_dev * (self.dev_decay * (n - self.start_base) + 1))
            return np.array(out, dtype=np.float64)
        raise NotImplementedError

    def fit(self, values: list[float], lower_bound: float = 0, upper_bound: float | None = None) -> None:
        if self.base is None:
            max_depth = max(values) * 1.2
            self.base_dev *= max_depth
            self.x_max = max(max_depth, 1e-6)
            self.data = values
        if upper_bound is None:
            upper_bound = self.x_max
        density, _ = np.histogram(values, bins=self.bin_num, range=(0, self.x_max))
        x = np.linspace(0, self.x_max, self.bin_num)
        bases = np.ascontiguousarray(np.linspace(lower_bound, upper_bound, self.bin_num), dtype=np.float64)
        dev = self._deviations()
        dens = np.ascontiguousarray(density, dtype=np.float64)
        loglik = np.empty(self.bin_num, dtype=np.float64)
        d = _dev(self._device)
        check(lib().gk_cn_fit(d.ctx, x.ctypes.data, dens.ctypes.data, self.bin_num, bases.ctypes.data, self.bin_num,
                              dev.ctypes.data, len(dev), 0, self.x_max / self.bin_num, loglik.ctypes.data))
        self.likelihood = np.stack([bases, loglik], axis=1)
        self.base = self.likelihood[np.argmax(self.likelihood[:, 1]), 0]

    def assignCN(self, values: list[float]) -> list[int]:
        assert self.base is not None
        x = np.linspace(0, self.x_max, self.bin_num)
        dev = self._deviations()
        space = self.x_max / self.bin_num
        cn_max = np.empty(self.bin_num, dtype=np.int32)
        d = _dev(self._device)
        check(lib().gk_cn_assign(d.ctx, x.ctypes.data, self.bin_num, float(self.base), dev.ctypes.data, len(dev), 0,
                                 space, cn_max.ctypes.data))
        return [int(cn_max[int(depth / space)]) for depth in values]


class KDEcut(Dist):
    """CN thresholds at the local minima of a Gaussian KDE of the normalised depths (257-349)."""

    def __init__(self) -> None:
        super().__init__()
        self.bandwidth: float = 0.05
        self.points: int = 100
        self.neighbor: int = 5
        self.x_max: float = 0
        self.kde = None
        self.local_min: list[float] = []
        self.data: list[float] = []
        self.prob: list[float] = []

    def getParams(self) -> dict[str, Any]:
        return {"method": "KDEcut", "bandwidth": self.bandwidth, "points": self.points, "neighbor": self.neighbor,
                "x_max": self.x_max, "kde": {"kernel": "gaussian", "bandwidth": self.bandwidth},
                "local_min": self.local_min, "data": self.data, "prob": self.prob, "raw_df": self.raw_df}

    @classmethod
    def setParams(cls, data: dict[str, Any]) -> "KDEcut":
        assert data["method"] == "KDEcut"
        self = cls()
        self.x_max, self.bandwidth = data["x_max"], data["bandwidth"]
        self.neighbor, self.points = data["neighbor"], data["points"]
        self.local_min, self.data = data["local_min"], data["data"]
        self.raw_df, self.prob = data.get("raw_df", []), data["prob"]
        self.kde = True
        return self

    def fit(self, values: list[float]) -> None:
        assert self.kde is None
        self.x_max = np.max(values)
        data = np.array(values, dtype=np.float64) / self.x_max
        x = np.linspace(0, 1.1, self.points)
        # log of the Gaussian kernel density estimate (what sklearn's score_samples returns)
        z = (x[:, None] - data[None, :]) / self.bandwidth
        a = -0.5 * z * z
        m = a.max(axis=1, keepdims=True)
        y = (m[:, 0] + np.log(np.exp(a - m).sum(axis=1))
             - np.log(len(data) * self.bandwidth * np.sqrt(2 * np.pi)))
        self.prob = y
        # strict local minima over +-neighbor points (scipy.signal.argrelextrema(np.less, order), mode clip)
        n = len(y)
        keep = np.ones(n, dtype=bool)
        for s in range(1, self.neighbor + 1):
            plus = y[np.minimum(np.arange(n) + s, n - 1)]
            minus = y[np.maximum(np.arange(n) - s, 0)]
            keep &= (y < plus) & (y < minus)
        self.local_min = list(x[np.nonzero(keep)[0]])
        self.kde = True
        self.data = values

    def assignCN(self, values: list[float]) -> list[int]:
        assert self.kde is not None
        return list(np.searchsorted(self.local_min, np.array(values) / self.x_max))


def loadCNModel(filename: str) -> Dist:
    with open(filename) as f:
        data = json.load(f)
    if data["method"] == "KDEcut":
        return KDEcut.load(filename)
    if data["method"] == "CNgroup":
        return CNgroup.load(filename)
    raise NotImplementedError
