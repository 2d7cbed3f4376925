"""
Logger, resource globals and the output merge helpers of the typing stage
(``graphkir/utils.py``: logger 33-38, threads 60-74, NumpyEncoder 119-127,
mergeAllele / mergeCN 161-179).  The MSA / download helpers of the reference's
``utils.py`` belong to the index build and are out of scope.
"""
from __future__ import annotations

import dataclasses
import json
import logging
import subprocess
from typing import Any

import numpy as np
import pandas as pd

logger = logging.getLogger("graphkir")
logger.propagate = False
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setLevel(logging.DEBUG)
    _h.setFormatter(logging.Formatter("%(asctime)s [%(name)s] [%(levelname)8s] %(message)s"))
    logger.addHandler(_h)

resources = {"threads": 2, "memory": 7}


def getThreads() -> int:
    return resources["threads"]


def setThreads(threads: int) -> None:
    resources["threads"] = threads


def runShell(cmd: list[str], capture_output: bool = False, cwd: str | None = None):
    """Run a command (always captured, raises on failure) like utils.runShell 88-103."""
    logger.debug(f'[Run] {" ".join(cmd)}')
    proc = subprocess.run(cmd, shell=False, capture_output=True, cwd=cwd, check=True, universal_newlines=True)
    if not capture_output:
        logger.debug(proc.stdout)
    return proc


class NumpyEncoder(json.JSONEncoder):
    def default(self, obj: Any) -> Any:
        if dataclasses.is_dataclass(obj):
            return dataclasses.asdict(obj)
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        if isinstance(obj, np.generic):
            return obj.item()
        return json.JSONEncoder.default(self, obj)


def getGeneName(allele: str) -> str:
    return allele.split("*")[0]


def mergeAllele(allele_result_files: list[str], final_result_file: str) -> pd.DataFrame:
    """Row-concatenate the per-sample allele TSVs into ``cohort.allele.tsv``."""
    df = pd.concat(pd.read_csv(f, sep="\t") for f in allele_result_files)
    df.to_csv(final_result_file, index=False, sep="\t")
    return df


def mergeCN(cn_result_files: list[str], final_result_file: str) -> pd.DataFrame:
    """Pivot the per-sample CN TSVs into ``cohort.cn.tsv`` (gene x cn-file, missing -> 0)."""
    frames = []
    for f in cn_result_files:
        df = pd.read_csv(f, sep="\t")
        df["name"] = f
        frames.append(df)
    df = pd.pivot_table(pd.concat(frames), values="cn", index="gene", columns=["name"])
    df = df.fillna(0).astype(int)
    df.to_csv(final_result_file, sep="\t")
    return df


def _envList(var: str) -> dict[str, str]:
    """``name`` / ``name=value`` entries of a comma-separated environment variable."""
    import os
    out: dict[str, str] = {}
    for item in os.environ.get(var, "").split(","):
        item = item.strip()
        if item:
            name, _, value = item.partition("=")
            out[name] = value
    return out


def traceOn(what: str) -> bool:
    """GK_TRACE lists the development traces that are on (pool, search, ingest, bench): csrc/gk_env.h."""
    return what in _envList("GK_TRACE")


def testHook(name: str, default: str | None = None) -> str | None:
    """GK_TEST_HOOKS: switches the tests use to force rarely taken paths (csrc/gk_env.h); the value of ``name=value``, ""
    for a bare ``name``, ``default`` when it is not listed."""
    return _envList("GK_TEST_HOOKS").get(name, default)
