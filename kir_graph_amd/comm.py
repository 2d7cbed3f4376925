"""
Ranks of one node: rendezvous and the few collectives the typing path needs.

One process per GPU.  The data path has no exchange; the control path has three:

* ``--cn-cohort`` pools the gene depths of all samples before ONE copy-number fit
  (graphkir/kir_cn.py:61, 167-177; main.py:572-589) -> ``Comm.allgatherF64``;
* rank 0 merges the per-sample TSV names in cohort order (main.py:590-603) -> ``Comm.allgatherObject``;
* ``bench.py`` brackets its timed region with ``Comm.barrier`` and takes ``Comm.maxF64`` of the times.

Backends:

* ``rccl`` -- ``gk_comm_*`` of the C ABI on librccl (xGMI inside the node): needs one GPU per rank;
* ``file`` -- the same calls through the rendezvous directory, for runs that place several ranks on one
  GPU (RCCL refuses two ranks on a device) and for the CPU tests of the multi-rank logic.  It moves a few
  hundred bytes of control data between host processes; nothing of the typing path computes on the CPU.

There is no ``torch`` here: the launcher only has to export RANK / WORLD_SIZE / LOCAL_RANK (``torchrun``
does, ``bench.py --gpus N`` does it itself).  The rendezvous is a directory of small files on the node
(``GK_RDZV_DIR``, else a name derived from the launcher's pid + start time + MASTER_PORT, which all ranks of a
launch share and no other launch does); rank 0 removes it on ``close``.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import shutil
import time

import numpy as np


class CommError(RuntimeError):
    pass


def _parentStamp() -> str:
    ppid = os.getppid()
    start = "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]    # field 22: start time in clock ticks since boot
    except (OSError, IndexError):
        pass
    return f"{ppid}_{start}"


def rendezvousDir() -> str:
    d = os.environ.get("GK_RDZV_DIR")
    if d:
        return d
    port = os.environ.get("MASTER_PORT", "0")
    gen = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"gk_rdzv_{_parentStamp()}_{port}_{gen}")


class FileStore:
    """Write-once keys in a directory: ``set`` is atomic (rename), ``get`` waits for the key."""

    def __init__(self, path: str, timeout: float = 600.0):
        self.path, self.timeout = path, timeout
        os.makedirs(path, exist_ok=True)

    def set(self, key: str, value: bytes) -> None:
        tmp = os.path.join(self.path, f".{key}.{os.getpid()}.tmp")
        with open(tmp, "wb") as f:
            f.write(value)
        os.replace(tmp, os.path.join(self.path, key))

    def get(self, key: str) -> bytes:
        target = os.path.join(self.path, key)
        t0, nap = time.monotonic(), 0.0002
        while True:
            try:
                with open(target, "rb") as f:
                    return f.read()
            except FileNotFoundError:
                if time.monotonic() - t0 > self.timeout:
                    raise CommError(f"rendezvous: {key} did not appear in {self.path} within {self.timeout:.0f}s "
                                    "(a rank died or never started)") from None
                time.sleep(nap)
                nap = min(nap * 1.5, 0.02)

    def drop(self, key: str) -> None:
        try:
            os.remove(os.path.join(self.path, key))
        except FileNotFoundError:
            pass


class Comm:
    """The ranks of one launch.  ``dev``: the rank's ``_lib.Device`` (needed by the ``rccl`` backend)."""

    def __init__(self, rank: int, world: int, store: FileStore, dev=None, backend: str = "rccl"):
        if backend not in ("rccl", "file"):
            raise ValueError(f"unknown backend {backend!r}")
        self.rank, self.world, self.store, self.backend = rank, world, store, backend
        self._seq = 0
        self._handle = None
        self._dev = dev
        if backend == "rccl":
            from ._lib import check, lib
            if dev is None:
                raise CommError("the rccl backend needs the rank's device context")
            if rank == 0:
                uid = C.create_string_buffer(128)
                check(lib().gk_comm_unique_id(uid, 128))
                store.set("rccl_id", uid.raw)
            uid = store.get("rccl_id")
            h = C.c_void_p()
            check(lib().gk_comm_create(dev.ctx, uid, len(uid), rank, world, C.byref(h)))
            self._handle = h
        self.barrier()      # everyone is here (and, with rccl, the communicator works) before anything is removed

    # ---- host-side exchange through the store (every rank writes one key per round, reads all)
    def _round(self, payload: bytes) -> list[bytes]:
        k = self._seq
        self._seq += 1
        self.store.set(f"x{k}.r{self.rank}", payload)
        out = [self.store.get(f"x{k}.r{r}") for r in range(self.world)]
        if k >= 2:                       # every rank has finished round k-1, hence read all keys of round k-2
            self.store.drop(f"x{k - 2}.r{self.rank}")
        return out

    def allgatherObject(self, obj) -> list:
        """``[obj of rank 0, ..., obj of rank world-1]`` on every rank (JSON-serialisable objects)."""
        return [json.loads(b.decode()) for b in self._round(json.dumps(obj).encode())]

    def allgatherF64(self, vec: np.ndarray) -> np.ndarray:
        """float64 ``[world, n]``: row r = the vector of rank r (same n on every rank)."""
        vec = np.ascontiguousarray(vec, dtype=np.float64).ravel()
        n = len(vec)
        if n == 0:
            return np.zeros((self.world, 0))
        if self._handle is None:
            rows = [np.frombuffer(b, dtype=np.float64) for b in self._round(vec.tobytes())]
            if any(len(r) != n for r in rows):
                raise CommError("all-gather: ranks sent vectors of different lengths")
            return np.stack(rows)
        from ._lib import check, lib
        out = np.empty((self.world, n), dtype=np.float64)
        check(lib().gk_allgather_f64(self._handle, vec.ctypes.data, out.ctypes.data, n))
        return out

    def maxF64(self, x: float) -> float:
        if self._handle is None:
            return float(max(np.frombuffer(b, dtype=np.float64)[0] for b in self._round(np.float64(x).tobytes())))
        from ._lib import check, lib
        v = np.array([x], dtype=np.float64)
        check(lib().gk_allreduce_max_f64(self._handle, v.ctypes.data, 1))
        return float(v[0])

    def barrier(self) -> None:
        if self._handle is None:
            self._round(b"")
            return
        from ._lib import check, lib
        check(lib().gk_comm_barrier(self._handle))

    def close(self) -> None:
        if self.store is None:
            return
        self._round(b"")
        if self._handle is not None:
            from ._lib import lib
            lib().gk_comm_destroy(self._handle)
            self._handle = None
        # rank 0 removes the directory once every other rank has said it will not read from it again
        if self.rank:
            self.store.set(f"bye.r{self.rank}", b"")
        else:
            for r in range(1, self.world):
                self.store.get(f"bye.r{r}")
            shutil.rmtree(self.store.path, ignore_errors=True)
        self.store = None


def worldFromEnv() -> tuple[int, int, int]:
    """(rank, world, local rank) from the launcher's environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))))


def initFromEnv(dev=None, backend: str | None = None, fallback: bool = True) -> Comm | None:
    """The launch's communicator, or None for a single process.  ``backend`` None: ``GK_COMM_BACKEND``, else
    rccl when every local rank has a GPU of its own, file otherwise.  ``fallback``: when the RCCL communicator
    cannot be set up on some rank (no librccl, no device, initialisation error) ALL ranks agree -- through the
    rendezvous directory -- to carry the few control messages over the file backend instead, and say so."""
    rank, world, _ = worldFromEnv()
    if world <= 1:
        return None
    backend = backend or os.environ.get("GK_COMM_BACKEND")
    if backend is None:
        from ._lib import deviceCount
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        backend = "rccl" if 0 < local_world <= deviceCount() else "file"
    store = FileStore(rendezvousDir())
    if backend != "rccl":
        return Comm(rank, world, store, dev=dev, backend=backend)
    made, why = None, ""
    try:
        if dev is None:
            from .kir_typing import defaultDevice
            dev = defaultDevice()
        # The communicator is made on a helper thread with a deadline (GK_RCCL_INIT_TIMEOUT seconds, default 300):
        # an initialisation that never returns (a bootstrap that cannot reach its peers) must not hang the run --
        # it counts as a failure, and the ranks fall back together like for any other failure.
        import threading
        box: dict = {}

        def make():
            try:
                box["comm"] = Comm(rank, world, store, dev=dev, backend="rccl")
            except Exception as e:    # noqa: BLE001 -- reported below
                box["error"] = e

        worker = threading.Thread(target=make, name="gk-rccl-init", daemon=True)
        worker.start()
        worker.join(float(os.environ.get("GK_RCCL_INIT_TIMEOUT", "300")))
        if worker.is_alive():
            raise CommError("RCCL communicator initialisation did not return in time")
        if "error" in box:
            raise box["error"]
        made = box["comm"]
    except Exception as e:            # noqa: BLE001 -- whatever went wrong, the ranks must agree on what to do next
        if not fallback:
            raise
        why = f"{type(e).__name__}: {e}"
    store.set(f"rccl_ok.r{rank}", b"1" if made is not None else b"0")
    everyone = [store.get(f"rccl_ok.r{r}") == b"1" for r in range(world)]
    if all(everyone):
        return made
    if not fallback:
        raise CommError("RCCL communicator failed on ranks " + str([r for r, ok in enumerate(everyone) if not ok]))
    import sys
    print(f"[comm] rank {rank}: RCCL communicator unavailable ({why or 'another rank failed'}); "
          "control messages go through the rendezvous directory", file=sys.stderr, flush=True)
    if made is not None and made._handle is not None:
        from ._lib import lib
        lib().gk_comm_destroy(made._handle)
        made._handle = None
    other = Comm.__new__(Comm)
    other.rank, other.world, other.store, other.backend = rank, world, store, "file"
    other._seq, other._handle, other._dev = 0, None, dev     # no rank has used the store's rounds yet
    other.barrier()
    return other
