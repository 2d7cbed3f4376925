"""GPU parity of the search reductions called straight through the C ABI on tie-heavy random tables:
gk_maxsum / gk_fraction / gk_setmax vs numpy (whose add.reduce order they reproduce bit for bit)."""
import numpy as np
import pytest

from kir_graph_amd._lib import check, lib

pytestmark = pytest.mark.gpu


def table(rng, n_rows, n_allele):
    # few distinct values per row -> many exact ties, like log10 of products of 0.999 / 0.001
    L = -rng.integers(0, 4, (n_rows, n_allele)).astype(np.float64) * 2.9995654882259823
    return L - rng.integers(20, 60, (n_rows, 1)) * 0.00043451177401769


@pytest.mark.parametrize("n_rows", [5, 129, 1000, 8200, 20011])
@pytest.mark.parametrize("c", [1, 2, 3, 4, 5, 8])
def test_fraction_matches_numpy(device, n_rows, c):
    rng = np.random.default_rng(100 * c + n_rows)
    n_allele, n_sets = 70, 150
    L = table(rng, n_rows, n_allele)
    dL = device.put(np.ascontiguousarray(L.T))
    ids = np.ascontiguousarray(rng.integers(0, n_allele, (n_sets, c)), dtype=np.int32)
    ids[:40, 0] = 3                                     # hub allele shared by many sets
    out = np.empty((n_sets, c))
    check(lib().gk_fraction(device.ctx, dL.ptr, n_rows, n_rows, ids.ctypes.data, n_sets, c, out.ctypes.data))
    gathered = L[:, ids]                                # rows x sets x c   (typing_mulit_allele.py:575-580)
    owns = np.equal(gathered, gathered.max(axis=2)[:, :, None])
    want = (owns / owns.sum(axis=2)[:, :, None]).sum(axis=0) / n_rows
    assert np.array_equal(out, want)
    dL.free()


@pytest.mark.parametrize("n_rows,n_allele,n_sets,c_prev", [(5, 9, 3, 1), (1000, 33, 40, 1), (8200, 70, 100, 2),
                                                           (20011, 45, 33, 3), (129, 17, 16, 0)])
def test_maxsum_and_setmax_match_numpy(device, n_rows, n_allele, n_sets, c_prev):
    rng = np.random.default_rng(n_rows + n_allele)
    L = table(rng, n_rows, n_allele)
    dL = device.put(np.ascontiguousarray(L.T))
    cols = np.arange(n_allele, dtype=np.int32)
    if c_prev == 0:
        out = np.empty((1, n_allele))
        check(lib().gk_maxsum(device.ctx, dL.ptr, n_rows, n_rows, None, 1, 0, cols.ctypes.data, n_allele,
                              out.ctypes.data))
        assert np.array_equal(out[0], L[:, cols].sum(axis=0))      # typing_mulit_allele.py:514
    else:
        ids = np.ascontiguousarray(rng.integers(0, n_allele, (n_sets, c_prev)), dtype=np.int32)
        out = np.empty((n_sets, n_allele))
        check(lib().gk_maxsum(device.ctx, dL.ptr, n_rows, n_rows, ids.ctypes.data, n_sets, c_prev, cols.ctypes.data,
                              n_allele, out.ctypes.data))
        prev = L[:, ids].max(axis=2)                    # allele_prob of the previous sets (line 569)
        # exactly the reference's expression (540-542): the fancy-indexed operands are laid out
        # allele-major, which makes numpy reduce over reads with its pairwise tree; the same maths on
        # a C-ordered (sets, reads, alleles) array would be summed sequentially and differ in the last bits
        want = np.maximum(L[:, cols], prev.T[:, :, None]).sum(axis=1)
        assert np.array_equal(out, want)
        buf = device.alloc((n_sets, n_rows), np.float64)
        check(lib().gk_setmax(device.ctx, dL.ptr, n_rows, n_rows, ids.ctypes.data, n_sets, c_prev, buf.ptr))
        assert np.array_equal(buf.download().reshape(n_sets, n_rows).T, prev)
        buf.free()
    dL.free()


@pytest.mark.parametrize("n_allele", [33, 70, 200])
def test_second_allele_table_is_mirrored_exactly(device, n_allele):
    """Sets = every single allele in rank order, columns = every allele: the library computes the upper
    triangle in column order and mirrors it; the result must still be numpy's, bit for bit."""
    n_rows = 3000
    rng = np.random.default_rng(n_allele)
    L = table(rng, n_rows, n_allele)
    dL = device.put(np.ascontiguousarray(L.T))
    cols = np.arange(n_allele, dtype=np.int32)
    ids = np.ascontiguousarray(rng.permutation(n_allele)[:, None], dtype=np.int32)
    out = np.empty((n_allele, n_allele))
    check(lib().gk_maxsum(device.ctx, dL.ptr, n_rows, n_rows, ids.ctypes.data, n_allele, 1, cols.ctypes.data,
                          n_allele, out.ctypes.data))
    prev = L[:, ids.flatten()]
    want = np.maximum(L[:, cols], prev.T[:, :, None]).sum(axis=1)
    assert np.array_equal(out, want)
    dL.free()


@pytest.mark.parametrize("n_rows,n_allele,n_sets,c_prev,top_n", [(5, 9, 3, 1, 4), (1000, 33, 40, 1, 25), (8200, 70, 100, 2, 60),
                                                                 (20011, 130, 129, 3, 600), (129, 17, 16, 1, 1000),
                                                                 (70001, 200, 200, 1, 600)])
def test_bound_step_selects_what_numpy_selects(device, n_rows, n_allele, n_sets, c_prev, top_n):
    """gk_miss_colsum / gk_bound_step straight through the C ABI on a random u8 mismatch table:
    M = sum_r min(miss) for every candidate, the top_n-th smallest among the masked candidates, and the
    selection {masked, M <= M_T} -- against numpy."""
    rng = np.random.default_rng(7 * n_rows + n_allele)
    miss = rng.integers(0, 4, (n_allele, n_rows)).astype(np.uint8)
    miss[:, rng.integers(0, n_rows, max(1, n_rows // 50))] += 40         # a few far-off reads
    ldm = (n_rows + 63) // 64 * 64
    table8 = np.zeros((n_allele, ldm), dtype=np.uint8)
    table8[:, :n_rows] = miss
    d_miss = device.put(table8)
    d_msum = device.alloc(n_allele, np.uint32)
    check(lib().gk_miss_colsum(device.ctx, d_miss.ptr, ldm, n_allele, d_msum.ptr))
    assert np.array_equal(d_msum.download(), miss.sum(axis=1, dtype=np.uint64).astype(np.uint32))
    cols = np.ascontiguousarray(rng.permutation(n_allele)[:max(1, n_allele - 3)], dtype=np.int32)
    ids = np.ascontiguousarray(rng.integers(0, n_allele, (n_sets, c_prev)), dtype=np.int32)
    first = (rng.random((n_sets, len(cols))) < 0.8).astype(np.uint8)
    cap = 4 * top_n + 4096
    hdr = np.zeros(4, dtype=np.uint32)
    idx = np.empty(cap, dtype=np.int32)
    mm = np.empty(cap, dtype=np.uint32)
    check(lib().gk_bound_step(device.ctx, d_miss.ptr, ldm, n_rows, d_msum.ptr, ids.ctypes.data, n_sets, c_prev,
                              cols.ctypes.data, len(cols), first.ctypes.data, top_n, cap, hdr.ctypes.data,
                              idx.ctypes.data, mm.ctypes.data))
    prev = miss[ids].min(axis=1).astype(np.int64)                        # [set][read]
    M = np.stack([np.minimum(prev, miss[c].astype(np.int64)[None, :]).sum(axis=1) for c in cols], axis=1)
    flat, mask = M.ravel(), first.ravel().astype(bool)
    n_cand = int(mask.sum())
    assert int(hdr[0]) == n_cand
    cut = np.sort(flat[mask])[min(top_n, n_cand) - 1]
    assert int(hdr[1]) == cut
    want = np.flatnonzero(mask & (flat <= cut))
    assert int(hdr[2]) == len(want)
    if len(want) <= cap:
        order = np.argsort(idx[:len(want)])
        assert np.array_equal(idx[:len(want)][order], want)
        assert np.array_equal(mm[:len(want)][order], flat[want])
    d_miss.free()
    d_msum.free()


@pytest.mark.parametrize("n_rows,c", [(5, 2), (1000, 2), (8200, 3), (20011, 4), (129, 1)])
def test_setsum_gives_the_bits_of_maxsum_and_fraction(device, n_rows, c):
    """gk_setsum = value of gk_maxsum (one set at a time) + shares of gk_fraction, same bits."""
    rng = np.random.default_rng(n_rows + c)
    n_allele, n_sets = 60, 90
    L = table(rng, n_rows, n_allele)
    dL = device.put(np.ascontiguousarray(L.T))
    ids = np.ascontiguousarray(rng.integers(0, n_allele, (n_sets, c)), dtype=np.int32)
    value, frac, frac0 = np.empty(n_sets), np.empty((n_sets, c)), np.empty((n_sets, c))
    check(lib().gk_setsum(device.ctx, dL.ptr, n_rows, n_rows, ids.ctypes.data, n_sets, c, value.ctypes.data,
                          frac.ctypes.data))
    check(lib().gk_fraction(device.ctx, dL.ptr, n_rows, n_rows, ids.ctypes.data, n_sets, c, frac0.ctypes.data))
    assert np.array_equal(frac, frac0)
    best = np.asfortranarray(L[:, ids].max(axis=2))      # reads x sets, the read axis contiguous per set
    assert np.array_equal(value, best.sum(axis=0))
    if c >= 2:      # the same number as the search's own expression (540-542): previous set + one more column
        cols = np.arange(n_allele, dtype=np.int32)
        out = np.empty((n_sets, n_allele))
        prev = np.ascontiguousarray(ids[:, :c - 1])
        check(lib().gk_maxsum(device.ctx, dL.ptr, n_rows, n_rows, prev.ctypes.data, n_sets, c - 1, cols.ctypes.data,
                              n_allele, out.ctypes.data))
        assert np.array_equal(value, out[np.arange(n_sets), ids[:, c - 1]])
    dL.free()


@pytest.mark.parametrize("n_rows,n_allele,n_sets,c", [
    (3, 11, 5, 2),            # a leaf shorter than 8 rows: all tail
    (135, 40, 130, 2),        # two leaves (64 + 71) and a tail, two sets per lane group
    (8192, 70, 600, 2),       # one full chunk, the shape of a search step
    (8199, 70, 700, 3),       # a second chunk of 7 rows
    (20011, 230, 650, 2),     # chunks with uneven last leaves; 230 columns staged
    (20011, 300, 300, 4),     # more columns than ride in registers (the direct staging loop)
    (9001, 64, 1000, 2),      # more sets than one launch carries: two batches
    (9001, 64, 600, 3),       # three alleles: two batches of 4 sets per lane group
    (5000, 50, 500, 4),       # four alleles: two batches of 3
])
def test_leafwise_setsum_equals_the_tiled_form_and_numpy(device, monkeypatch, n_rows, n_allele, n_sets, c):
    """gk_setsum through setsum_leaves / fold_leaves (every column through LDS once, numpy's tree folded leaf by leaf) gives
    the bits of the tiled kernel (fraction_chunks with the value riding along) and of numpy."""
    rng = np.random.default_rng(n_rows * 7 + n_sets + c)
    L = table(rng, n_rows, n_allele)
    dL = device.put(np.ascontiguousarray(L.T))
    ids = np.ascontiguousarray(rng.integers(0, n_allele, (n_sets, c)), dtype=np.int32)
    ids[:40, 0] = 3
    ids[-1, :] = n_allele - 1                         # the last column is used
    got = {}
    for form in ("leaves", "tiles"):
        monkeypatch.setenv("GK_TEST_HOOKS", f"setsum={form}")
        value, frac = np.empty(n_sets), np.empty((n_sets, c))
        check(lib().gk_setsum(device.ctx, dL.ptr, n_rows, n_rows, ids.ctypes.data, n_sets, c, value.ctypes.data,
                              frac.ctypes.data))
        got[form] = (value, frac)
    assert np.array_equal(got["leaves"][0], got["tiles"][0])
    assert np.array_equal(got["leaves"][1], got["tiles"][1])
    gathered = L[:, ids]
    best = np.asfortranarray(gathered.max(axis=2))
    assert np.array_equal(got["leaves"][0], best.sum(axis=0))
    owns = np.equal(gathered, gathered.max(axis=2)[:, :, None])
    want = (owns / owns.sum(axis=2)[:, :, None]).sum(axis=0) / n_rows
    assert np.array_equal(got["leaves"][1], want)
    dL.free()
