"""A fixed-seed slice of the randomised parity run (tests/fuzz_parity.py) under the driver's `pytest -m gpu`:
HIP path against the oracle on random indices and samples -- lists, novel numbering, every field of every
copy-number step of three strategies, the EM report -- including wide genes (several allele slots per lane) and
pairs in the wide record format.  Both routes of a search step must have been taken: served by the integer
bound, and handed back to the exact float64 kernels (typing_mulit_allele.py:534-598)."""
import importlib.util
import os
import time

import pytest

pytestmark = pytest.mark.gpu


def _fuzz():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("gk_fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fixed_seed_slice_of_the_randomised_parity_run(device, monkeypatch):
    from kir_graph_amd.typing_mulit_allele import SEARCH_STATS
    monkeypatch.setenv("GK_SEARCH", "bound")
    fp = _fuzz()
    before = dict(SEARCH_STATS)
    t0 = time.time()
    done = []
    # (first seed, cases, wide genes, pairs beyond the 128-byte record)
    for first, count, wide, spill in ((3030001, 26, False, False), (3031001, 4, True, False), (3032001, 10, False, True)):
        fp.WIDE, fp.SPILL = wide, spill
        for seed in range(first, first + count):
            done.append((seed, fp.one(seed, device)))
    fp.WIDE = fp.SPILL = False
    bounded = SEARCH_STATS["bounded"] - before["bounded"]
    redone = SEARCH_STATS["redone_exactly"] - before["redone_exactly"]
    print(f"[fuzz slice] {len(done)} cases in {time.time() - t0:.0f}s, search steps: bounded {bounded}, redone exactly {redone}, "
          f"pairs in the wide record format: {fp.WIDE_SEEN[0]}")
    assert len(done) == 40
    assert bounded > 0 and redone > 0, (bounded, redone)      # both branches of gk_search_run ran
    assert fp.WIDE_SEEN[0] > 0                                 # tab_count_wide / tab_emit_wide ran
