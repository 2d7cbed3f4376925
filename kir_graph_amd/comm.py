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
launch share and no other launch does); rank 0 removes it on ``close``.  Every key carries the launch's token
(``launchToken``), so keys that a crashed run left in a reused ``GK_RDZV_DIR`` are never read (rank 0 removes them).
A rank that fails writes an ``abort`` key: peers waiting in ``FileStore.get`` stop at once instead of after the timeout.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import re
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


def launchToken() -> str:
    """What every rank of ONE launch can compute and no other launch shares: ``GK_RDZV_TOKEN`` when the ranks are
    started by hand, else the launcher's pid + start time, MASTER_PORT and the elastic restart count."""
    tok = os.environ.get("GK_RDZV_TOKEN")
    if tok:
        return "".join(c if c.isalnum() or c in "_-" else "_" for c in tok)
    port = os.environ.get("MASTER_PORT", "0")
    gen = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    return f"{_parentStamp()}_{port}_{gen}"


def rendezvousDir() -> str:
    d = os.environ.get("GK_RDZV_DIR")
    if d:
        return d
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"gk_rdzv_{launchToken()}")


def _hostIdentity() -> str:
    """Where a pid means something: host name + pid namespace (a rendezvous directory may be shared between nodes or
    mounted into several containers)."""
    import socket
    try:
        ns = os.readlink("/proc/self/ns/pid")
    except OSError:
        ns = "?"
    return f"{socket.gethostname()} {ns}"


class FileStore:
    """Write-once keys in a directory: ``set`` is atomic (rename), ``get`` waits for the key.  File names start with
    the launch token: a directory that still holds the keys of a dead run (a reused ``GK_RDZV_DIR``) cannot feed
    them to this one; ``purgeOthers`` (rank 0) removes them."""

    def __init__(self, path: str, timeout: float = 600.0, token: str | None = None):
        self.path, self.timeout = path, timeout
        self.token = launchToken() if token is None else token
        os.makedirs(path, exist_ok=True)

    def _name(self, key: str) -> str:
        return os.path.join(self.path, f"{self.token}.{key}")

    def set(self, key: str, value: bytes) -> None:
        tmp = os.path.join(self.path, f".{self.token}.{key}.{os.getpid()}.tmp")
        with open(tmp, "wb") as f:
            f.write(value)
        os.replace(tmp, self._name(key))

    def get(self, key: str) -> bytes:
        target, abort = self._name(key), self._name("abort")
        t0, nap, polls = time.monotonic(), 0.0002, 0
        while True:
            try:
                with open(target, "rb") as f:
                    return f.read()
            except FileNotFoundError:
                polls += 1
                if polls % 16 == 0 and os.path.exists(abort):
                    try:
                        why = open(abort, "rb").read().decode(errors="replace")
                    except OSError:
                        why = ""
                    raise CommError(f"rendezvous: another rank gave up ({why or 'no reason recorded'})") from None
                if time.monotonic() - t0 > self.timeout:
                    raise CommError(f"rendezvous: {key} did not appear in {self.path} within {self.timeout:.0f}s "
                                    "(a rank died or never started)") from None
                time.sleep(nap)
                nap = min(nap * 1.5, 0.02)

    def drop(self, key: str) -> None:
        try:
            os.remove(self._name(key))
        except FileNotFoundError:
            pass

    def abort(self, why: str) -> None:
        """Tell the ranks that wait in ``get`` that this one will not write again."""
        try:
            if not os.path.exists(self._name("abort")):
                self.set("abort", why.encode()[:400])
        except OSError:
            pass

    # every key of this package is "<token>.<key>" (or ".<token>.<key>.<pid>.tmp" while it is written); a launch marks
    # itself alive with "<token>.owner" = the pid of its rank 0 on this node
    _KEY = re.compile(r"^\.?(?P<token>[0-9A-Za-z_-]+)\.(?:x\d+\.r\d+|bye\.r\d+|rccl_ok\.r\d+|rccl_id|abort|owner)(?:\.\d+\.tmp)?$")

    def claim(self) -> None:
        """Rank 0: record the pid that owns this launch's keys (what ``purgeOthers`` of a later launch looks at)."""
        self.set("owner", f"{os.getpid()} {_hostIdentity()}".encode())

    def purgeOthers(self, older_than: float = 600.0) -> int:
        """Remove the keys a DEAD launch left behind in a shared rendezvous directory; returns how many.  Hygiene only
        -- keys of other tokens are never read.  Only files that look like this package's keys are touched, and only
        those of a token whose owner no longer exists.  The ``owner`` key holds rank 0's pid AND where that pid means
        something (host name + pid namespace): the pid is asked only on that host; for a launch of another host or
        container (a shared directory), and for a token without an owner key, only age counts -- all its files older
        than ``older_than`` AND than twice the ``get`` timeout (a live launch may hold a key for as long as a peer
        types).  A pid that still exists protects its keys for a day (pid numbers are reused)."""
        n = 0
        now = time.time()
        try:
            names = os.listdir(self.path)
        except OSError:
            return 0
        by_token: dict[str, list[str]] = {}
        for name in names:
            m = self._KEY.match(name)
            if m and m.group("token") != self.token:
                by_token.setdefault(m.group("token"), []).append(name)
        for token, files in by_token.items():
            owner = os.path.join(self.path, f"{token}.owner")
            try:
                fields = open(owner, "rb").read().decode().split()
                pid, where = int(fields[0]), " ".join(fields[1:])
            except (OSError, ValueError, IndexError):
                pid, where = 0, ""
            try:
                newest = max(os.path.getmtime(os.path.join(self.path, f)) for f in files)
            except OSError:
                continue
            aged = now - newest >= max(older_than, 2 * self.timeout)
            if pid > 0 and where == _hostIdentity():
                # same host and pid namespace: the pid says whether the launch lives -- unless the number was handed
                # to another process since, which the age of the keys catches (a live launch touches its keys)
                try:
                    os.kill(pid, 0)
                    alive = True
                except ProcessLookupError:
                    alive = False
                except OSError:
                    alive = True
                if alive and now - newest < max(86400.0, 4 * self.timeout):
                    continue                      # a day without a key touched: the pid is somebody else's by now
            elif not aged:
                continue                          # another host / namespace, or no owner key: only the age rule
            for f in files:
                try:
                    os.remove(os.path.join(self.path, f))
                    n += 1
                except OSError:
                    pass
        return n


class Comm:
    """The ranks of one launch.  ``dev``: the rank's ``_lib.Device`` (needed by the ``rccl`` backend)."""

    def __init__(self, rank: int, world: int, store: FileStore, dev=None, backend: str = "rccl", abandoned=None):
        """``abandoned`` (a ``threading.Event``): set by a caller that has stopped waiting for this constructor; a
        communicator that comes up afterwards is destroyed at once and NO collective is issued on it (its peers
        have moved on: the collective would never complete)."""
        if backend not in ("rccl", "file"):
            raise ValueError(f"unknown backend {backend!r}")
        self.rank, self.world, self.store, self.backend = rank, world, store, backend
        self._seq = 0
        self._handle = None
        self._dev = dev
        self._cdev = None
        if rank == 0:
            store.claim()
            store.purgeOthers()
        try:
            if backend == "rccl":
                from ._lib import Device, check, lib
                if dev is None:
                    raise CommError("the rccl backend needs the rank's device context")
                # a context (stream, staging) of its own on the rank's GPU: collectives never queue behind, or in front
                # of, the typing kernels, and a collective that cannot complete cannot block the compute stream
                self._cdev = Device(dev.ordinal)
                if rank == 0:
                    uid = C.create_string_buffer(128)
                    check(lib().gk_comm_unique_id(uid, 128))
                    store.set("rccl_id", uid.raw)
                uid = store.get("rccl_id")
                h = C.c_void_p()
                check(lib().gk_comm_create(self._cdev.ctx, uid, len(uid), rank, world, C.byref(h)))
                if abandoned is not None and abandoned.is_set():
                    lib().gk_comm_destroy(h)
                    raise CommError("RCCL communicator came up after its deadline; destroyed unused")
                self._handle = h
            self.barrier()      # everyone is here (and, with rccl, the communicator works) before anything is removed
        except BaseException:
            self._teardown()    # the private context (stream, pinned rings, pool) does not outlive a failed set-up
            raise

    def _teardown(self) -> None:
        """Destroy the communicator and the context it lives on (idempotent; the store is left alone)."""
        if self._handle is not None:
            from ._lib import lib
            lib().gk_comm_destroy(self._handle)
            self._handle = None
        if self._cdev is not None:
            self._cdev.close()
            self._cdev = None

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
        """Every rank has reached this point; with a device, the work queued on the rank's own stream is done."""
        if hasattr(self._dev, "sync"):
            self._dev.sync()
        if self._handle is None:
            self._round(b"")
            return
        from ._lib import check, lib
        check(lib().gk_comm_barrier(self._handle))

    def close(self) -> None:
        if self.store is None:
            return
        self._round(b"")
        self._teardown()
        # rank 0 removes the directory once every other rank has said it will not read from it again
        if self.rank:
            self.store.set(f"bye.r{self.rank}", b"")
        else:
            for r in range(1, self.world):
                self.store.get(f"bye.r{r}")
            shutil.rmtree(self.store.path, ignore_errors=True)
        self.store = None


def superviseRanks(procs: list, rdzv: str, token: str, grace: float = 20.0) -> int:
    """Wait for the rank processes of a launch (``subprocess.Popen`` objects whose environment carries
    ``GK_RDZV_DIR=rdzv`` and ``GK_RDZV_TOKEN=token``).  When one exits non-zero the others are told at once
    (``abort`` key: their ``FileStore.get`` raises), given ``grace`` seconds to leave and then terminated; the
    rendezvous directory is removed whatever happened.  Returns 0 when every rank returned 0, else 1 (a rank killed
    by a signal counts as a failure, not as a small positive code)."""
    store = None
    failed_at = None
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if failed_at is None and any(c not in (None, 0) for c in codes):
                failed_at = time.monotonic()
                bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
                store = store or FileStore(rdzv, token=token)
                store.abort(f"rank {bad[0][0]} exited with code {bad[0][1]}")
            if failed_at is not None and time.monotonic() - failed_at > grace:
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                for p in procs:
                    try:
                        p.wait(timeout=10)
                    except Exception:      # noqa: BLE001
                        p.kill()
                break
            time.sleep(0.05)
        codes = [p.wait() for p in procs]
        return 0 if all(c == 0 for c in codes) else 1
    finally:
        shutil.rmtree(rdzv, ignore_errors=True)


def worldFromEnv() -> tuple[int, int, int]:
    """(rank, world, local rank) from the launcher's environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))))


def initFromEnv(dev=None, backend: str | None = None, fallback: bool = True) -> Comm | None:
    """The launch's communicator, or None for a single process.  ``backend`` None: ``GK_COMM_BACKEND``, else
    rccl when every local rank has a GPU of its own, file otherwise.  ``fallback``: when the RCCL communicator
    cannot be set up on some rank (no librccl, no device, initialisation error) ALL ranks agree -- through the
    rendezvous directory -- to carry the few control messages over the file backend instead, and say so; with
    ``fallback=False`` every rank raises ``CommError`` instead.  (``bench.py --gpus N`` falls back too and says so in
    its line -- ``config.rank_barrier`` / ``rank_barrier_note``: the timed data path has no collective.)"""
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
        abandoned = threading.Event()

        def make():
            try:
                box["comm"] = Comm(rank, world, store, dev=dev, backend="rccl", abandoned=abandoned)
            except Exception as e:    # noqa: BLE001 -- reported below
                box["error"] = e

        worker = threading.Thread(target=make, name="gk-rccl-init", daemon=True)
        worker.start()
        worker.join(float(os.environ.get("GK_RCCL_INIT_TIMEOUT", "300")))
        if worker.is_alive():
            # the helper is told that nobody waits for it: if ncclCommInitRank returns later it destroys the
            # communicator and issues nothing on it (Comm.__init__); it works on a context of its own, so the
            # rank's compute stream is not involved either way
            abandoned.set()
            raise CommError("RCCL communicator initialisation did not return in time")
        if "error" in box:
            raise box["error"]
        made = box["comm"]
    except Exception as e:            # noqa: BLE001 -- whatever went wrong, the ranks must agree on what to do next
        why = f"{type(e).__name__}: {e}"
    store.set(f"rccl_ok.r{rank}", b"1" if made is not None else b"0")
    everyone = [store.get(f"rccl_ok.r{r}") == b"1" for r in range(world)]
    if all(everyone):
        return made
    if not fallback:
        if made is not None:
            made._teardown()
        raise CommError("RCCL communicator failed on ranks " + str([r for r, ok in enumerate(everyone) if not ok])
                        + (f" (this rank: {why})" if why else ""))
    import sys
    print(f"[comm] rank {rank}: RCCL communicator unavailable ({why or 'another rank failed'}); "
          "control messages go through the rendezvous directory", file=sys.stderr, flush=True)
    if made is not None:
        made._teardown()
    other = Comm.__new__(Comm)
    other.rank, other.world, other.store, other.backend = rank, world, store, "file"
    other._seq, other._handle, other._dev, other._cdev = 0, None, dev, None     # no rank has used the store's rounds yet
    other.barrier()
    return other
