"""Cohort sharding over ranks with world_size 2 on CPU (gloo): the N > 1 path of --cn-cohort."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from kir_graph_amd.cohort import shardSamples

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, os.environ["GK_ROOT"])
    import torch.distributed as dist
    from kir_graph_amd import cohort
    dist.init_process_group("gloo")
    n_samples = 5
    comm = cohort.Comm(n_samples)
    genes = [f"KIR{g}*BACKBONE" for g in ("2DL1", "2DL4", "3DL3")]
    # sample s, gene j has depth 100*s + j + 0.25 (distinct, order-revealing)
    local = [{g: 100.0 * s + j + 0.25 for j, g in enumerate(genes)} for s in comm.mine]
    pooled = comm.allgatherDepths(local)
    # the pooled-depth fit input must be identical on every rank and in cohort order
    want = [100.0 * s + j + 0.25 for s in range(n_samples) for j in range(len(genes))]
    assert pooled == want, (comm.rank, pooled)
    # per-rank file lists merge back into cohort order on rank 0 (main.py logic)
    mine = [f"s{s}.cn.tsv" for s in comm.mine]
    gathered = [None] * comm.world
    dist.all_gather_object(gathered, mine)
    merged = [""] * n_samples
    for r, idxs in enumerate(comm.shards):
        for k, gi in enumerate(idxs):
            merged[gi] = gathered[r][k]
    assert merged == [f"s{s}.cn.tsv" for s in range(n_samples)]
    comm.barrier()
    if comm.rank == 0:
        print("OK", json.dumps(pooled[:4]))
    dist.destroy_process_group()
""")


def test_shard_assignment_is_a_partition():
    for n in (1, 5, 8, 64):
        for w in (1, 2, 4, 8):
            shards = shardSamples(n, w)
            assert sorted(i for s in shards for i in s) == list(range(n))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_allgather_depths_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 500)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GK_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    assert "OK" in outs[0][0]
