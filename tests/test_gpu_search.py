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
