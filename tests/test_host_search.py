"""The HOST half of the native search (csrc/gk_hostsearch.cpp: gk_search_run -- first occurrences of the candidate
multisets, the top_n cut, the stable three-key ranking, the bounded branch with its hand-back to the exact one) on the
CPU: the four device entry points it calls are numpy implementations of their contracts (tests/asan/search_stub.cpp
forwards them), so the run needs no GPU and is compared, field by field, with the oracle's search on the same table.

The table is made of values whose sums are exact in float64 (L = -3 * mismatch count), so every summation order gives
the same bits and ties abound -- the tie handling (numpy.argsort's order handed in as a callback, ties across a cut,
rows equal in every ranking key) is exactly what this host code is about."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from kir_graph_amd import _lib
from oracle import typing as oty

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kir_graph_amd", "csrc")


@pytest.fixture(scope="module")
def host_search(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("hostsearch") / "libsearch_host.so")
    cmd = [hipcc, "--cuda-host-only", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", f"-I{ROOT}/include", f"-I{CSRC}",
           "-x", "hip", f"{CSRC}/gk_hostsearch.cpp", f"{ROOT}/tests/asan/search_stub.cpp", "-o", out]
    if os.environ.get("GK_HOSTSEARCH_SANITIZE") == "1":     # set by test_host_search_under_sanitizers for its child process
        cmd[2:2] = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared-libsan",
                    "-fno-omit-frame-pointer"]
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}     # the compiler must not run under the preloaded runtime
    res = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lib = C.CDLL(out)
    lib.gk_last_error.restype = C.c_char_p
    lib.gk_test_ctx.restype = C.c_void_p
    return lib


MAXSUM = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                     C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_double))
FRACTION = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                       C.POINTER(C.c_double))
SETSUM = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                     C.POINTER(C.c_double), C.POINTER(C.c_double))
BOUND = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_uint64, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                    C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.POINTER(C.c_uint32),
                    C.POINTER(C.c_int32), C.POINTER(C.c_uint32))


class Table:
    """A gene's log-likelihood table and mismatch table with the device calls of the search done in numpy."""

    def __init__(self, rng, n_rows, n_allele):
        groups = rng.integers(0, max(2, n_allele // 4), n_allele)        # alleles come in near-identical families
        base = rng.integers(0, 3, (n_rows, groups.max() + 1))
        miss = base[:, groups] + (rng.random((n_rows, n_allele)) < 0.05)
        self.miss = miss.astype(np.int64)                                 # [R][A]
        self.L = -3.0 * self.miss                                         # exact in float64, like its sums
        self.R, self.A = n_rows, n_allele
        self.Lf = np.asfortranarray(self.L)                               # column-major [A][R], as in HBM
        self.ldm = (n_rows + 63) // 64 * 64
        self.miss8 = np.zeros((n_allele, self.ldm), dtype=np.uint8)
        self.miss8[:, :n_rows] = self.miss.T
        self.msum = self.miss.sum(axis=0).astype(np.uint32)
        self.calls = {"maxsum": 0, "fraction": 0, "setsum": 0, "bound": 0}
        self.callbacks = (MAXSUM(self._maxsum), FRACTION(self._fraction), SETSUM(self._setsum), BOUND(self._bound))

    @staticmethod
    def _arr(ptr, n, dtype):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True) if n else np.zeros(0, dtype)

    def _shares(self, ids):
        g = self.L[:, ids]                                                # R x K x c
        owns = np.equal(g, g.max(axis=2)[:, :, None])
        return g.max(axis=2).sum(axis=0), (owns / owns.sum(axis=2)[:, :, None]).sum(axis=0) / self.R

    def _maxsum(self, d_L, n_rows, ld, ids, n_sets, c_prev, cols, n_cols, out):
        self.calls["maxsum"] += 1
        assert d_L == self.Lf.ctypes.data and n_rows == self.R and ld == self.R
        cols = self._arr(cols, n_cols, np.int64)
        if c_prev == 0:
            res = self.L[:, cols].sum(axis=0)[None, :]
        else:
            prev = self._arr(ids, n_sets * c_prev, np.int64).reshape(n_sets, c_prev)
            best = self.L[:, prev].max(axis=2)                            # R x T
            res = np.maximum(self.L[:, cols][:, None, :], best[:, :, None]).sum(axis=0)
        np.ctypeslib.as_array(out, shape=(n_sets * n_cols,))[:] = res.ravel()
        return 0

    def _fraction(self, d_L, n_rows, ld, ids, n_sets, c, frac_out):
        self.calls["fraction"] += 1
        sets = self._arr(ids, n_sets * c, np.int64).reshape(n_sets, c)
        np.ctypeslib.as_array(frac_out, shape=(n_sets * c,))[:] = self._shares(sets)[1].ravel()
        return 0

    def _setsum(self, d_L, n_rows, ld, ids, n_sets, c, value_out, frac_out):
        self.calls["setsum"] += 1
        sets = self._arr(ids, n_sets * c, np.int64).reshape(n_sets, c)
        value, frac = self._shares(sets)
        np.ctypeslib.as_array(value_out, shape=(n_sets,))[:] = value
        np.ctypeslib.as_array(frac_out, shape=(n_sets * c,))[:] = frac.ravel()
        return 0

    def _bound(self, d_miss8, ldm, n_rows, d_msum, ids, n_sets, c_prev, cols, n_cols, first, top_n, cap, hdr, idx_out, m_out):
        self.calls["bound"] += 1
        assert d_miss8 == self.miss8.ctypes.data and ldm == self.ldm and d_msum == self.msum.ctypes.data
        prev = self._arr(ids, n_sets * c_prev, np.int64).reshape(n_sets, c_prev)
        cols = self._arr(cols, n_cols, np.int64)
        keep = self._arr(first, n_sets * n_cols, np.uint8).reshape(n_sets, n_cols) != 0
        low = self.miss[:, prev].min(axis=2)                              # R x T
        M = np.minimum(self.miss[:, cols][:, None, :], low[:, :, None]).sum(axis=0)
        h = np.ctypeslib.as_array(hdr, shape=(4,))
        h[:] = 0
        n_cand = int(keep.sum())
        if n_cand:
            t = min(top_n, n_cand)
            cut = int(np.partition(M[keep], t - 1)[t - 1])
            chosen = np.flatnonzero((keep & (M <= cut)).ravel())
            chosen = chosen[np.random.default_rng(len(chosen)).permutation(len(chosen))]   # "in no particular order"
            h[0], h[1], h[2] = n_cand, cut, len(chosen)
            n_out = min(len(chosen), cap)
            np.ctypeslib.as_array(idx_out, shape=(max(n_out, 1),))[:n_out] = chosen[:n_out]
            np.ctypeslib.as_array(m_out, shape=(max(n_out, 1),))[:n_out] = M.ravel()[chosen[:n_out]]
        return 0


def native_steps(lib, tab, cols, n_steps, top_n, bound):
    lib.gk_test_device_calls(*tab.callbacks)
    handle = C.c_void_p()
    cols32 = np.ascontiguousarray(cols, dtype=np.int32)
    rc = lib.gk_search_run(C.c_void_p(lib.gk_test_ctx()), C.c_uint64(tab.Lf.ctypes.data), C.c_int64(tab.R), C.c_int64(tab.R),
                           C.c_int32(tab.A), C.c_uint64(tab.miss8.ctypes.data if bound else 0), C.c_int64(tab.ldm),
                           C.c_uint64(tab.msum.ctypes.data if bound else 0), cols32.ctypes.data_as(C.c_void_p),
                           C.c_int32(len(cols32)), C.c_int32(n_steps), C.c_int32(top_n), _lib.NUMPY_ARGSORT, None,
                           C.byref(handle))
    assert rc == 0, lib.gk_last_error()
    steps = []
    n = C.c_int32()
    assert lib.gk_search_steps(handle, C.byref(n)) == 0 and n.value == n_steps
    for s in range(n_steps):
        width, rows, bounded = C.c_int32(), C.c_int64(), C.c_int32()
        assert lib.gk_search_info(handle, C.c_int32(s), C.byref(width), C.byref(rows), C.byref(bounded)) == 0
        k, r = width.value, rows.value
        value, sums = np.empty(r), np.empty((r, k))
        ids, frac = np.empty((r, k), dtype=np.int32), np.empty((r, k))
        assert lib.gk_search_copy(handle, C.c_int32(s), value.ctypes.data_as(C.c_void_p), sums.ctypes.data_as(C.c_void_p),
                                  ids.ctypes.data_as(C.c_void_p), frac.ctypes.data_as(C.c_void_p)) == 0
        steps.append((value, sums, ids, frac, bool(bounded.value)))
    lib.gk_search_destroy(handle)
    return steps


@pytest.mark.parametrize("bound", [False, True])
def test_native_search_host_half_equals_the_oracle(host_search, bound):
    rng = np.random.default_rng(5 + bound)
    names = {i: f"a{i}" for i in range(400)}
    bounded_steps = exact_steps = 0
    for trial in range(40):
        n_rows = int(rng.choice([7, 64, 129, 500]))
        n_allele = int(rng.choice([3, 9, 33, 70, 130]))
        top_n = int(rng.choice([2, 5, 17, 60, 200]))      # 200 >= alleles: the second step's bound runs on the upper triangle only
        cn = int(rng.integers(1, 5))
        tab = Table(rng, n_rows, n_allele)
        cols = np.arange(n_allele) if trial % 3 else np.sort(rng.choice(n_allele, size=max(1, n_allele // 2), replace=False))
        want = [oty.firstStep(tab.L, cols, top_n, names)]
        for _ in range(cn - 1):
            want.append(oty.nextStep(tab.L, want[-1], cols, top_n, names))
        got = native_steps(host_search, tab, cols, cn, top_n, bound)
        for s, (w, (value, sums, ids, frac, was_bounded)) in enumerate(zip(want, got)):
            where = (trial, s, n_rows, n_allele, top_n)
            assert np.array_equal(ids, w.allele_id), where
            assert np.array_equal(value, w.value), where
            assert np.array_equal(sums, w.value_sum_indv), where
            assert np.array_equal(frac, w.fraction), where
            if s:
                bounded_steps += was_bounded
                exact_steps += not was_bounded
        if not bound:
            assert tab.calls["bound"] == 0 and tab.calls["setsum"] == 0
    # both branches of the bounded search are walked: steps the bound serves and steps handed back to the exact sums
    assert exact_steps > 0 and (bounded_steps > 0) == bound


def test_site_verdicts_of_the_host_only_build(host_search):
    """gk_site_verdict_tallies of the host-only build (the one the sanitizers see) against the plain loop of
    typing_mulit_allele.py:829-857 on random tallies: substitutions, insertions of one and several bases, deletions."""
    from collections import defaultdict
    from kir_graph_amd.index import packKey
    rng = np.random.default_rng(3)
    strings = ["A", "G", "AT", "CCG"]
    ins_code = np.array([ord(x) if len(x) == 1 else 256 + i for i, x in enumerate(strings)], dtype=np.int64)
    seen = set()
    for trial in range(200):
        n_keys = int(rng.integers(4, 60))
        keys, labels, positions, dels = [], [], [], []
        for _ in range(n_keys):
            pos, typ = int(rng.integers(0, 8)), int(rng.choice([0, 1, 1, 2]))
            val = int(rng.choice([65, 67, 71, 84])) if typ == 1 else int(rng.integers(len(strings))) if typ == 0 else int(rng.integers(1, 6))
            keys.append(packKey(0, pos, typ, val))
            labels.append(chr(val) if typ == 1 else strings[val] if typ == 0 else str(val))
            positions.append(pos); dels.append(typ == 2)
        keys = np.array(keys, dtype=np.uint64)
        n = int(rng.integers(1, n_keys + 1))
        ords = np.sort(rng.choice(n_keys, size=n, replace=False)).astype(np.int32)
        pc = np.where(rng.random(n) < 0.6, rng.integers(1, 50, n), 0).astype(np.uint32)
        nc = np.where(rng.random(n) < 0.6, rng.integers(1, 50, n), 0).astype(np.uint32)
        cn = int(rng.integers(2, 5))
        site = defaultdict(lambda: defaultdict(int))
        for o, a, b in zip(ords.tolist(), pc.tolist(), nc.tolist()):
            if dels[o]:
                continue
            if a:
                site[positions[o]][labels[o]] += a
            if b:
                site[positions[o]]["*" + labels[o]] += b
        hits = 0
        for obs in site.values():
            if len(obs) <= 1 or all("*" in k for k in obs):
                continue
            counts = [c for c in sorted(obs.values(), reverse=True) if c > 3]
            total = sum(counts)
            if total < 20:
                continue
            major = [c / total for c in counts if c / total > 0.1]
            if len(major) > 1 and major[1] > (1 / (cn * 2)):
                hits += 1
        verdict = C.c_int32()
        rc = host_search.gk_site_verdict_tallies(keys.ctypes.data_as(C.c_void_p), C.c_int64(len(keys)),
                                                 ins_code.ctypes.data_as(C.c_void_p), C.c_int64(len(ins_code)),
                                                 ords.ctypes.data_as(C.c_void_p), pc.ctypes.data_as(C.c_void_p),
                                                 nc.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int32(cn), C.byref(verdict))
        assert rc == 0 and bool(verdict.value) == (hits == 0), trial
        seen.add(bool(verdict.value))
    assert seen == {True, False}


def _clang_asan():
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


def test_host_search_under_sanitizers():
    """The same comparison with the host half built with AddressSanitizer + UndefinedBehaviorSanitizer (host compile
    of the ROCm clang; its runtime is preloaded into a child interpreter): no out-of-bounds access, no overflow, no
    misaligned or invalid value on any of the 80 searches and 200 site verdicts."""
    import sys
    asan = _clang_asan()
    if asan is None or os.environ.get("GK_HOSTSEARCH_SANITIZE") == "1":
        pytest.skip("no clang sanitizer runtime (or already inside the sanitized run)")
    env = dict(os.environ, GK_HOSTSEARCH_SANITIZE="1", LD_PRELOAD=asan,      # (the UBSan checks live in the ASan runtime)
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k", "equals_the_oracle or host_only_build",
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0 and "3 passed" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])
