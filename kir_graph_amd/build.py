"""Compile ``libgraphkir_hip.so`` in-tree with hipcc for gfx950 (no GPU needed to build)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SOURCES = ["gk_runtime.hip", "gk_scan.hip", "gk_tabulate.hip", "gk_typing.hip", "gk_lut.hip",
           "gk_search.hip", "gk_bound.hip", "gk_em.hip", "gk_cn.hip", "gk_depth.hip", "gk_sampack.cpp", "gk_bamread.cpp", "gk_textout.cpp",
           "gk_comm.cpp", "gk_hostsearch.cpp"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def outOfDate(lib: Path) -> bool:
    if not lib.exists():
        return True
    t = lib.stat().st_mtime
    deps = [PKG / "csrc" / s for s in SOURCES] + list((PKG / "csrc").glob("*.h")) + [ROOT / "include" / "graphkir_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


# device source files a kernel's code comes from (everything under csrc/ for a kernel that is not listed)
KERNEL_SOURCES = {
    "compat_kernel": ["gk_typing.hip", "gk_lut.h", "gk_common.h"],
    "patch_pending": ["gk_typing.hip", "gk_lut.h", "gk_common.h"],
    "count_ids_genes": ["gk_typing.hip", "gk_common.h"],
    "flag_nonempty": ["gk_typing.hip", "gk_common.h"],
    "flag_pairs": ["gk_typing.hip", "gk_common.h"],
    "gather_pair_flags": ["gk_typing.hip", "gk_common.h"],
    "tab_count": ["gk_tabulate.hip", "gk_common.h"],
    "tab_emit": ["gk_tabulate.hip", "gk_common.h"],
    "tab_expand": ["gk_tabulate.hip", "gk_common.h"],
    "minsum_sad": ["gk_bound.hip", "gk_common.h"],
    "minsum_finish": ["gk_bound.hip", "gk_common.h"],
    "select_hist1": ["gk_bound.hip", "gk_common.h"],
    "select_hist2": ["gk_bound.hip", "gk_common.h"],
    "select_append": ["gk_bound.hip", "gk_common.h"],
    "setmin_u8": ["gk_bound.hip", "gk_common.h"],
    "fraction_chunks": ["gk_search.hip", "gk_common.h"],
    "setsum_leaves": ["gk_search.hip", "gk_common.h"],
    "fold_leaves": ["gk_search.hip", "gk_common.h"],
    "colsum_chunks": ["gk_search.hip", "gk_common.h"],
    "maxsum_chunks": ["gk_search.hip", "gk_common.h"],
    "combine_chunks": ["gk_search.hip", "gk_common.h"],
    "em_sets_groups": ["gk_em.hip", "gk_common.h"],
    "em_sets_hash": ["gk_em.hip", "gk_common.h"],
    "em_sets_verify": ["gk_em.hip", "gk_common.h"],
    "em_sets_emit": ["gk_em.hip", "gk_common.h"],
    "em_kernel_genes": ["gk_em.hip", "gk_common.h"],
}


def sourceDigest(kernel: str | None = None) -> str:
    """sha256 (first 16 hex digits) of the device sources ``kernel`` is compiled from (``KERNEL_SOURCES``; all of
    ``csrc/*.hip`` and ``csrc/*.h`` when the kernel is not listed): what a measurement of that kernel belongs to.
    ``bench.py`` only reports a committed HBM-traffic figure whose recorded digest equals the running code's
    (tools/pmc_traffic.py writes it)."""
    import hashlib
    h = hashlib.sha256()
    names = KERNEL_SOURCES.get(kernel or "")
    if names is None:
        files = sorted((PKG / "csrc").glob("*.hip")) + sorted((PKG / "csrc").glob("*.h"))
    else:
        files = [PKG / "csrc" / n for n in names]
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


# Per-source extra flags (none needed at present).
EXTRA_FLAGS: dict[str, list[str]] = {}


def buildNative(force: bool = False, verbose: bool = False) -> Path:
    """Compile every source to an object (in parallel) under ``csrc/build/`` and link the library."""
    lib = PKG / "libgraphkir_hip.so"
    if not force and not outOfDate(lib):
        return lib
    objdir = PKG / "csrc" / "build"
    objdir.mkdir(exist_ok=True)
    base = [hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value",
            "-Wno-unused-result", f"-I{ROOT / 'include'}", f"-I{PKG / 'csrc'}"]
    jobs = []
    for src in SOURCES:
        obj = objdir / (src.rsplit(".", 1)[0] + ".o")
        cmd = base + EXTRA_FLAGS.get(src, []) + ["-c", str(PKG / "csrc" / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        jobs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    failed = []
    for src, obj, proc in jobs:
        out, err = proc.communicate()
        if proc.returncode != 0:
            failed.append(f"{src}:\n{out}\n{err}")
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib)] + [str(o) for _, o, _ in jobs] + ["-lz", "-ldl"]
    if verbose:
        print(" ".join(link))
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return lib


if __name__ == "__main__":
    print(buildNative(force=True, verbose=True))
