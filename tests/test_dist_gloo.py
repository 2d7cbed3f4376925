"""Cohort sharding over ranks with world_size 2 on CPU: the N > 1 path of --cn-cohort.

Two transports carry the same cohort logic (``cohort.Comm``): the product's own host-side one
(``comm.Comm`` with the ``file`` backend -- the ``rccl`` backend needs GPUs and is covered by the GPU tests)
and, as a cross-check of the collective semantics, ``torch.distributed`` with ``gloo`` wrapped here in the
same four calls."""
import os
import subprocess
import sys
import textwrap

from kir_graph_amd.cohort import sampleWeights, shardSamples

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, os.environ["GK_ROOT"])
    import numpy as np
    from kir_graph_amd import cohort, comm

    if os.environ["GK_TEST_TRANSPORT"] == "gloo":
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")

        class Transport:
            rank, world = dist.get_rank(), dist.get_world_size()
            def allgatherF64(self, vec):
                mine = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
                out = torch.empty(self.world * len(vec), dtype=torch.float64)
                dist.all_gather_into_tensor(out, mine)
                return out.numpy().reshape(self.world, len(vec))
            def maxF64(self, x):
                t = torch.tensor([x], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())
            def allgatherObject(self, obj):
                out = [None] * self.world
                dist.all_gather_object(out, obj)
                return out
            def barrier(self):
                dist.barrier()
            def close(self):
                dist.destroy_process_group()
        transport = Transport()
    elif os.environ["GK_TEST_TRANSPORT"] == "rccl_hang":
        # the communicator's set-up never returns on rank 1 (a bootstrap that cannot reach its peers) and fails on
        # rank 0: after the deadline both ranks carry on over the file backend
        import threading
        real = comm.Comm.__init__
        def stuck(self, rank, world, store, dev=None, backend="rccl"):
            if backend == "rccl" and rank == 1:
                threading.Event().wait()          # for ever
            real(self, rank, world, store, dev=dev, backend=backend)
        comm.Comm.__init__ = stuck
        os.environ["GK_RCCL_INIT_TIMEOUT"] = "2"
        transport = comm.initFromEnv(dev=object(), backend="rccl")
        comm.Comm.__init__ = real
        assert transport.backend == "file" and transport.world == 2
    elif os.environ["GK_TEST_TRANSPORT"] == "rccl_fallback":
        # no GPU here: the RCCL communicator cannot be made on any rank, and all ranks agree on the file backend
        transport = comm.initFromEnv(backend="rccl")
        assert transport.backend == "file" and transport.world == 2
    else:
        transport = comm.initFromEnv(backend="file")
        assert transport.backend == "file" and transport.world == 2

    n_samples = 5
    weights = [5.0, 1.0, 1.0, 1.0, 4.0]
    c = cohort.Comm(n_samples, transport, weights=weights)
    assert c.shards == [[0, 2], [1, 3, 4]], c.shards          # LPT: 5 | 4, then the 1s to the lighter rank
    genes = [f"KIR{g}*BACKBONE" for g in ("2DL1", "2DL4", "3DL3")]
    # sample s, gene j has depth 100*s + j + 0.25 (distinct, order-revealing)
    local = [{g: 100.0 * s + j + 0.25 for j, g in enumerate(genes)} for s in c.mine]
    pooled = c.allgatherDepths(local)
    # the pooled-depth fit input must be identical on every rank and in cohort order
    want = [100.0 * s + j + 0.25 for s in range(n_samples) for j in range(len(genes))]
    assert pooled == want, (c.rank, pooled)
    # per-rank file lists merge back into cohort order (main.py logic)
    merged = c.gatherInCohortOrder([f"s{s}.cn.tsv" for s in c.mine])
    assert merged == [f"s{s}.cn.tsv" for s in range(n_samples)]
    assert transport.maxF64(1.5 + c.rank) == 2.5
    for k in range(20):                                        # many rounds: keys are recycled, order holds
        got = transport.allgatherF64(np.array([c.rank * 10.0 + k, k]))
        assert got.tolist() == [[k, k], [10.0 + k, k]]
    c.barrier()
    if c.rank == 0:
        print("OK", json.dumps(pooled[:4]))
    c.close()
""")


def test_shard_assignment_is_a_partition():
    for n in (1, 5, 8, 64):
        for w in (1, 2, 4, 8):
            shards = shardSamples(n, w)
            assert sorted(i for s in shards for i in s) == list(range(n))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
            lpt = shardSamples(n, w, weights=[(i * 7919) % 13 + 1 for i in range(n)])
            assert sorted(i for s in lpt for i in s) == list(range(n))
            assert all(s == sorted(s) for s in lpt)


def test_longest_processing_time_first_balances_read_counts():
    # 64 samples of 5 M reads with a few 20 M ones: no rank gets two of the big ones before every rank has one
    weights = [20.0 if i % 9 == 0 else 5.0 for i in range(64)]
    shards = shardSamples(64, 8, weights)
    load = [sum(weights[i] for i in s) for s in shards]
    assert max(load) - min(load) <= 20.0
    assert max(load) <= sum(weights) / 8 + 15.0
    round_robin = [sum(weights[i] for i in s) for s in shardSamples(64, 8)]
    assert max(load) <= max(round_robin)


def test_sample_weights_from_file_sizes(tmp_path):
    a, b = tmp_path / "a.bam", tmp_path / "b.bam"
    a.write_bytes(b"x" * 10)
    b.write_bytes(b"x" * 30)
    assert sampleWeights([str(a), str(b)]) == [10.0, 30.0]
    assert sampleWeights([[str(a), str(b)], [str(b), ""]]) == [40.0, 30.0]
    assert sampleWeights([str(a), str(tmp_path / "missing")]) is None


def _run_two_ranks(tmp_path, transport):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 500)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GK_ROOT=ROOT, GK_TEST_TRANSPORT=transport,
                   GK_RDZV_DIR=str(tmp_path / "rdzv"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    assert "OK" in outs[0][0]


def test_allgather_depths_world_size_2_file_backend(tmp_path):
    _run_two_ranks(tmp_path, "file")
    assert not (tmp_path / "rdzv").exists()          # rank 0 removed the rendezvous directory on close


def test_allgather_depths_world_size_2_gloo(tmp_path):
    _run_two_ranks(tmp_path, "gloo")


def test_rccl_failure_falls_back_to_the_file_backend_on_every_rank(tmp_path):
    _run_two_ranks(tmp_path, "rccl_fallback")


def test_rccl_set_up_that_never_returns_falls_back_after_the_deadline(tmp_path):
    _run_two_ranks(tmp_path, "rccl_hang")
