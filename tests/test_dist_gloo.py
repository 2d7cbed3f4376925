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
        def stuck(self, rank, world, store, dev=None, backend="rccl", **kw):
            if backend == "rccl" and rank == 1:
                threading.Event().wait()          # for ever
            real(self, rank, world, store, dev=dev, backend=backend, **kw)
        comm.Comm.__init__ = stuck
        os.environ["GK_RCCL_INIT_TIMEOUT"] = "2"
        transport = comm.initFromEnv(dev=object(), backend="rccl")
        comm.Comm.__init__ = real
        assert transport.backend == "file" and transport.world == 2
    elif os.environ["GK_TEST_TRANSPORT"] == "rccl_fallback":
        # no GPU here: the RCCL communicator cannot be made on any rank, and all ranks agree on the file backend
        transport = comm.initFromEnv(backend="rccl")
        assert transport.backend == "file" and transport.world == 2
    elif os.environ["GK_TEST_TRANSPORT"] == "rccl_strict":
        # bench.py --gpus N: no quiet fallback -- every rank raises, none hangs
        try:
            comm.initFromEnv(backend="rccl", fallback=False)
        except comm.CommError as e:
            assert "RCCL communicator failed on ranks [0, 1]" in str(e), e
            if os.environ["RANK"] == "0":
                print("OK strict")
            sys.exit(0)
        raise AssertionError("strict mode fell back")
    elif os.environ["GK_TEST_TRANSPORT"] == "dies":
        # rank 1 fails before its first collective: rank 0 must learn of it at once, not after the 600 s timeout
        import time
        transport = comm.initFromEnv(backend="file")
        if transport.rank == 1:
            transport.store.abort("rank 1: ValueError: all samples of a cohort must report the same genes")
            sys.exit(3)
        t0 = time.time()
        try:
            transport.allgatherF64(np.array([1.0]))
        except comm.CommError as e:
            assert "another rank gave up" in str(e) and time.time() - t0 < 30, (e, time.time() - t0)
            print("OK abort seen")
            sys.exit(0)
        raise AssertionError("the collective returned although rank 1 never joined it")
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


def _run_two_ranks(tmp_path, transport, codes=(0, 0)):
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
    for p, (out, err), want in zip(procs, outs, codes):
        assert p.returncode == want, err[-2000:]
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


def test_strict_mode_raises_on_every_rank_instead_of_falling_back(tmp_path):
    _run_two_ranks(tmp_path, "rccl_strict")


def test_a_rank_that_gives_up_ends_the_wait_of_the_others(tmp_path):
    _run_two_ranks(tmp_path, "dies", codes=(0, 3))


def test_keys_of_a_dead_run_in_a_reused_directory_are_not_read(tmp_path):
    """GK_RDZV_DIR reused after a crash: the old run's payloads (named by round and rank only) must not be taken
    for this launch's -- every key carries the launch token."""
    import numpy as np
    rdzv = tmp_path / "rdzv"
    rdzv.mkdir()
    for name in ("x0.r1", "x1.r1", "x2.r1", "rccl_id", "rccl_ok.r1", "bye.r1",
                 "12345_999_29500_0.x0.r1", "12345_999_29500_0.x1.r1"):
        (rdzv / name).write_bytes(np.array([-777.0, -777.0, -777.0, -777.0]).tobytes())
    _run_two_ranks(tmp_path, "file")         # pooled depths are checked inside the workers: no -777 anywhere


def test_supervisor_ends_the_launch_when_one_rank_fails(tmp_path):
    """comm.superviseRanks (main.spawnRanks, bench.launch_ranks): the surviving rank is told through the abort key,
    the rendezvous directory is removed, the launch reports failure -- within seconds."""
    import time
    import uuid
    from kir_graph_amd.comm import superviseRanks
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, os.environ["GK_ROOT"])
        import numpy as np
        from kir_graph_amd import comm
        if os.environ["RANK"] == "1":
            os._exit(9)                      # dies without a word
        t = comm.initFromEnv(backend="file")
    """))
    rdzv, token = tmp_path / "rdzv", uuid.uuid4().hex
    rdzv.mkdir()
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), GK_ROOT=ROOT,
                                       GK_RDZV_DIR=str(rdzv), GK_RDZV_TOKEN=token), stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    t0 = time.time()
    assert superviseRanks(procs, str(rdzv), token, grace=60.0) == 1
    assert time.time() - t0 < 30                       # rank 0 left on the abort key, not on a timeout or a kill
    assert "another rank gave up (rank 1 exited with code 9)" in procs[0].stderr.read()
    assert not rdzv.exists()


def test_a_communicator_that_comes_up_after_its_deadline_is_destroyed_unused(tmp_path, monkeypatch):
    """the helper thread that outlives the RCCL-init deadline must not run a collective (its peers
    are on the file backend: it would never complete) and must not leak the communicator."""
    import threading
    import pytest
    from kir_graph_amd import _lib, comm
    calls = []
    release = threading.Event()

    class FakeLib:
        def gk_comm_unique_id(self, buf, n):
            return 0
        def gk_comm_create(self, ctx, uid, n, rank, world, out):
            release.wait(10)                  # ncclCommInitRank returns late
            calls.append("create")
            return 0
        def gk_comm_destroy(self, h):
            calls.append("destroy")
            return 0
        def gk_comm_barrier(self, h):
            calls.append("barrier")
            return 0

    class FakeDevice:
        ordinal, ctx = 0, None
        def __init__(self, ordinal=0):
            pass
        def sync(self):
            calls.append("sync")
        def close(self):
            pass

    monkeypatch.setattr(_lib, "lib", lambda: FakeLib())
    monkeypatch.setattr(_lib, "check", lambda rc: None)
    monkeypatch.setattr(_lib, "Device", FakeDevice)
    store = comm.FileStore(str(tmp_path / "rdzv"), token="t")
    abandoned = threading.Event()
    box = {}

    def make():
        try:
            comm.Comm(0, 1, store, dev=FakeDevice(), backend="rccl", abandoned=abandoned)
        except comm.CommError as e:
            box["error"] = str(e)

    th = threading.Thread(target=make)
    th.start()
    th.join(0.3)
    assert th.is_alive()                      # the deadline passes ...
    abandoned.set()
    release.set()                             # ... then the communicator comes up
    th.join(10)
    assert not th.is_alive()
    assert "after its deadline" in box.get("error", "")
    assert calls == ["create", "destroy"]     # no barrier, no leak


def test_purge_only_removes_the_keys_of_dead_launches(tmp_path):
    """A shared rendezvous directory: rank 0 of a new launch removes what a DEAD launch left (its owner pid is gone),
    never the keys of a launch whose owner is alive -- however old they are: a rank parked in a round while a peer types
    for more than ten minutes holds an old key -- and never files that are not this package's keys."""
    import time
    from kir_graph_amd import comm
    d = tmp_path / "rdzv"
    alive = comm.FileStore(str(d), token="alive1")
    alive.claim()                                   # this process: alive
    alive.set("x3.r1", b"payload")
    alive.set("rccl_ok.r0", b"1")
    gone = subprocess.Popen([sys.executable, "-c", "pass"])
    gone.wait()
    dead = comm.FileStore(str(d), token="dead22")
    dead.set("owner", f"{gone.pid} {comm._hostIdentity()}".encode())       # a launch whose rank 0 has exited
    dead.set("x0.r0", b"old")
    dead.set("abort", b"why")
    (d / "notes.txt").write_text("not a key of this package")
    (d / "orphan9.x1.r0").write_bytes(b"no owner key, fresh")
    old = time.time() - 3600
    for name in ("alive1.x3.r1", "alive1.rccl_ok.r0", "alive1.owner"):
        os.utime(d / name, (old, old))              # old, but its launch lives
    mine = comm.FileStore(str(d), token="mine33")
    removed = mine.purgeOthers()
    left = sorted(p.name for p in d.iterdir())
    assert removed == 3 and not any(n.startswith("dead22.") for n in left)
    assert {"alive1.x3.r1", "alive1.rccl_ok.r0", "alive1.owner", "notes.txt", "orphan9.x1.r0"} <= set(left)
    assert alive.get("x3.r1") == b"payload"


def test_purge_trusts_a_pid_only_on_its_own_host(tmp_path):
    """The directory may be shared between nodes or containers: a pid of another host / pid namespace says nothing
    here, so such a launch's keys go by age alone; and a pid that exists protects keys for a day, not for ever (the
    number may have been handed to another process)."""
    import time
    from kir_graph_amd import comm
    d = tmp_path / "rdzv"
    gone = subprocess.Popen([sys.executable, "-c", "pass"])
    gone.wait()
    far = comm.FileStore(str(d), token="far001")              # a LIVE launch on another node: its pid does not exist here
    far.set("owner", f"{gone.pid} othernode pid:[4026531836]".encode())
    far.set("rccl_id", b"id")
    far.set("x2.r1", b"fresh")
    legacy = comm.FileStore(str(d), token="old002")           # an owner key without an identity (an older launch)
    legacy.set("owner", str(gone.pid).encode())
    legacy.set("x0.r0", b"fresh")
    stale = comm.FileStore(str(d), token="far003")            # another node's launch, silent for an hour
    stale.set("owner", f"{gone.pid} othernode pid:[4026531836]".encode())
    stale.set("x9.r0", b"old")
    reused = comm.FileStore(str(d), token="pid004")           # this process's pid, keys untouched for two days
    reused.claim()
    reused.set("x1.r0", b"ancient")
    now = time.time()
    for name, age in (("far003.owner", 3600), ("far003.x9.r0", 3600), ("pid004.owner", 2 * 86400), ("pid004.x1.r0", 2 * 86400)):
        os.utime(d / name, (now - age, now - age))
    removed = comm.FileStore(str(d), token="mine05", timeout=60.0).purgeOthers()
    left = {p.name for p in d.iterdir()}
    assert removed == 4
    assert {"far001.owner", "far001.rccl_id", "far001.x2.r1", "old002.owner", "old002.x0.r0"} <= left
    assert not any(n.startswith(("far003.", "pid004.")) for n in left)
