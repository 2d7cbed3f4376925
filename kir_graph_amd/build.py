"""Compile ``libgraphkir_hip.so`` in-tree with hipcc for gfx950 (no GPU needed to build)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SOURCES = ["gk_runtime.hip", "gk_scan.hip", "gk_tabulate.hip", "gk_typing.hip", "gk_lut.hip",
           "gk_search.hip", "gk_em.hip", "gk_cn.hip", "gk_depth.hip", "gk_sampack.cpp"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def outOfDate(lib: Path) -> bool:
    if not lib.exists():
        return True
    t = lib.stat().st_mtime
    deps = [PKG / "csrc" / s for s in SOURCES] + [PKG / "csrc" / "gk_common.h", ROOT / "include" / "graphkir_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


def buildNative(force: bool = False, verbose: bool = False) -> Path:
    lib = PKG / "libgraphkir_hip.so"
    if not force and not outOfDate(lib):
        return lib
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value", "-Wno-unused-result",
           f"-I{ROOT / 'include'}", f"-I{PKG / 'csrc'}", "-o", str(lib)]
    cmd += [str(PKG / "csrc" / s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    return lib


if __name__ == "__main__":
    print(buildNative(force=True, verbose=True))
