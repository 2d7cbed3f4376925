"""
Oracle: per-position depths -> per-gene depth -> integer copy number.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, from
``/root/reference/graphkir``:

* ``geneDepths``      <- kir_cn.aggrDepths 28-38 (pandas groupby quantile / mean / median)
* ``LCND``            <- cn_model.CNgroup 53-204 (fit 124-169, assignCN 171-177,
                         calcCNGroupProb 179-204)
* ``KDE``             <- cn_model.KDEcut 257-349
* ``depthsToCN``      <- kir_cn.depthToCN 41-123 (incl. the KIR3DL3-diploid refit loop)
* ``predictCN``       <- kir_cn.predictSamplesCN 146-231 (sample-wise and cohort pooling)

pandas / scipy / sklearn calls are the reference's third-party arithmetic and
are used as such.
"""
from __future__ import annotations

from itertools import chain

import numpy as np
import pandas as pd
from scipy.stats import norm


def geneDepths(depths: pd.DataFrame, mode: str = "p75") -> dict[str, float]:
    """gene -> aggregated depth over all positions (28-38); keys in sorted-gene order."""
    grp = depths.groupby(by="gene", as_index=False)["depth"]
    if mode == "median":
        out = grp.median()
    elif mode == "mean":
        out = grp.mean()
    elif mode == "p75":
        out = grp.quantile(0.75)
    else:
        raise NotImplementedError
    return dict(zip(out["gene"], out["depth"]))


class LCND:
    """Linear copy-number distributions: Gaussians at base*n, grid search over base."""

    def __init__(self, base_dev: float = 0.08, start_base: int = 1, bin_num: int = 300):
        self.bin_num, self.max_cn = bin_num, 7
        self.x_max, self.base = 1.0, None
        self.base_dev, self.y0_dev = base_dev, 1.5
        self.dev_decay, self.dev_decay_neg = 0.5, 0.3
        self.start_base = start_base
        self.data: list[float] = []
        self.likelihood = np.array([])

    def groupProb(self, base: float) -> np.ndarray:
        """CN x bins table of pdf * bin width (179-204)."""
        x = np.linspace(0, self.x_max, self.bin_num)
        if self.start_base == 1:
            y0 = norm.pdf(x, loc=0, scale=self.base_dev * self.y0_dev)
            yn = [norm.pdf(x, loc=base * n, scale=self.base_dev * (self.dev_decay * (n - 1) + 1))
                  for n in np.arange(1, self.max_cn)]
            y = np.stack([y0, *yn])
        elif self.start_base == 2:
            rows = []
            for n in np.arange(0, self.max_cn):
                if n < self.start_base:
                    dev = self.base_dev * (self.dev_decay_neg * (self.start_base - n) + 1)
                else:
                    dev = self.base_dev * (self.dev_decay * (n - self.start_base) + 1)
                rows.append(norm.pdf(x, loc=base * n, scale=dev))
            y = np.array(rows)
        else:
            raise NotImplementedError
        return np.array(y * (self.x_max / self.bin_num))

    def fit(self, values: list[float], lower: float = 0, upper: float | None = None) -> None:
        if self.base is None:
            top = max(values) * 1.2
            self.base_dev *= top
            self.x_max = max(top, 1e-6)
            self.data = values
        if upper is None:
            upper = self.x_max
        density, _ = np.histogram(values, bins=self.bin_num, range=(0, self.x_max))
        curve = []
        for base in np.linspace(lower, upper, self.bin_num):
            best = self.groupProb(base).max(axis=0)
            curve.append((base, np.sum(np.log(best + 1e-9) * density)))
        self.likelihood = np.array(curve)
        self.base = self.likelihood[np.argmax(self.likelihood[:, 1]), 0]

    def assign(self, values) -> list[int]:
        assert self.base is not None
        cn_of_bin = self.groupProb(self.base).argmax(axis=0)
        width = self.x_max / self.bin_num
        return [cn_of_bin[int(d / width)] for d in values]


class KDE:
    """KDE local-minimum cuts (257-349)."""

    def __init__(self):
        self.bandwidth, self.points, self.neighbor = 0.05, 100, 5
        self.x_max = 0.0
        self.local_min: list[float] = []

    def fit(self, values: list[float]) -> None:
        from scipy.signal import argrelextrema
        from sklearn.neighbors import KernelDensity
        self.x_max = np.max(values)
        data = np.array(values)[:, None] / self.x_max
        kde = KernelDensity(kernel="gaussian", bandwidth=self.bandwidth).fit(data)
        x = np.linspace(0, 1.1, self.points)
        y = kde.score_samples(x[:, None])
        self.local_min = list(x[argrelextrema(y, np.less, order=self.neighbor)[0]])

    def assign(self, values) -> list[int]:
        return list(np.searchsorted(self.local_min, np.array(values) / self.x_max))


def depthsToCN(samples: list[dict[str, float]], method: str = "CNgroup", kwargs: dict | None = None,
               assume_3DL3_diploid: bool = False, diploid: tuple[float, float] | None = None):
    """Pool all gene depths of all given samples, fit one model, assign CNs (41-123)."""
    values = list(chain.from_iterable(s.values() for s in samples))
    if method == "CNgroup" or method.lower() == "lcnd":
        model = LCND()
        for k, v in (kwargs or {}).items():
            setattr(model, k, v)
        lower, upper = 0.0, None
        if diploid is not None:
            mean, dev = diploid
            lower, upper = (mean - dev) / 2, (mean + dev) / 2
        else:
            model.bin_num += 200
        model.fit(values, lower, upper)
        if assume_3DL3_diploid:
            d3 = [float(s["KIR3DL3*BACKBONE"]) for s in samples]
            cn = model.assign(d3)
            perc, rate, bins0 = float(1), 0.2, model.bin_num
            while not all(c == 2 for c in cn):
                mid = sum(d3) / len(d3)
                model.bin_num = int(bins0 * perc)
                model.fit(values, (mid - perc * 10) / 2, (mid + perc * 10) / 2)
                cn = model.assign(d3)
                perc = perc - rate
                if perc <= 0:
                    break
            assert all(c == 2 for c in cn)
    elif method.lower() == "kde":
        model = KDE()  # type: ignore[assignment]
        model.fit(values)
    else:
        raise NotImplementedError
    out = []
    for s in samples:
        genes, depths = zip(*s.items())
        out.append(dict(zip(genes, model.assign(depths))))
    return out, model


def predictCN(depth_tables: list[pd.DataFrame], mode: str = "p75", method: str = "CNgroup",
              kwargs: dict | None = None, assume_3DL3_diploid: bool = False):
    """predictSamplesCN 146-231 without files: one call = one pooled fit."""
    samples = [geneDepths(df, mode) for df in depth_tables]
    cns, model = depthsToCN(samples, method, kwargs, assume_3DL3_diploid)
    return cns, samples, model


def predictCNPerGene(depth_tables: list[pd.DataFrame], mode: str = "p75", method: str = "CNgroup",
                     kwargs: dict | None = None):
    """predictSamplesCN(per_gene=True), kir_cn.py:195-222: one fit per gene over that gene's depth in every sample
    (no diploid-depth file, no 3DL3 assumption).  Returns (cn dict per sample, {gene: model})."""
    samples = [geneDepths(df, mode) for df in depth_tables]
    cns: list[dict[str, int]] = [{} for _ in samples]
    models = {}
    for gene in sorted({g for s in samples for g in s}):
        keyed = {f"{gene}-{i}": s[gene] for i, s in enumerate(samples) if gene in s}
        got, model = depthsToCN([keyed], method, kwargs)
        models[gene] = model
        for key, cn in got[0].items():
            cns[int(key.rsplit("-", 1)[1])][gene] = cn
    return cns, samples, models
