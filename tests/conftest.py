import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    from kir_graph_amd import _lib
    if _lib.deviceCount() == 0:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box (there is no CPU fallback)")
    dev = _lib.Device(0)
    yield dev
    dev.close()


@pytest.fixture(scope="session")
def small_case():
    """Seeded synthetic index + sample shared by the parity tests (4 genes, 4000 pairs)."""
    from kir_graph_amd import synth
    from kir_graph_amd.index import GkIndex
    sidx = synth.makeIndex(seed=2022, n_genes=4, var_range=(300, 600), allele_range=(20, 40))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=1031, n_pairs=4000)
    return sidx, gidx, sample
