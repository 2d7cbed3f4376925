"""The RCCL backend of kir_graph_amd/comm.py on the one GPU of the test box: a communicator of ONE rank goes
through the same calls as a node-wide one (dlopen of librccl, ncclCommInitRank from the unique id passed through
the rendezvous directory, gk_allgather_f64 / gk_allreduce_max_f64 / barrier staged through HBM).  More ranks need
more GPUs (RCCL refuses two ranks on one device); the driver's multi-GPU run covers that."""
import numpy as np
import pytest

from kir_graph_amd import cohort
from kir_graph_amd.comm import Comm, FileStore

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_communicator(device, tmp_path):
    comm = Comm(0, 1, FileStore(str(tmp_path / "rdzv")), dev=device, backend="rccl")
    assert comm._handle is not None
    v = np.array([1.5, -2.0, 3.25, 0.0])
    got = comm.allgatherF64(v)
    assert got.shape == (1, 4) and np.array_equal(got[0], v)
    big = np.arange(5000, dtype=np.float64) * 0.5          # larger than the first staging buffer
    assert np.array_equal(comm.allgatherF64(big)[0], big)
    assert comm.maxF64(-7.25) == -7.25
    comm.barrier()
    assert comm.allgatherObject({"a": [1, 2]}) == [{"a": [1, 2]}]
    # the cohort layer on top of it: depths of this rank's samples come back in cohort order
    c = cohort.Comm(3, comm)
    assert c.mine == [0, 1, 2]
    depths = [{"KIR2DL1*BACKBONE": 10.0 + s, "KIR3DL3*BACKBONE": 20.0 + s} for s in range(3)]
    assert c.allgatherDepths(depths) == [10.0, 20.0, 11.0, 21.0, 12.0, 22.0]
    comm.close()
    assert not (tmp_path / "rdzv").exists()
