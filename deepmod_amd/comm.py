"""Multi-GPU plumbing without torch: one process per GPU, a persistent RCCL communicator behind the C ABI
(dm_comm_*), and a file rendezvous for the 128-byte unique id and small JSON metadata.

The reference has no comms backend: "multi-node" = run separate processes on separate input folders and add the BED
files afterwards (docs/Usage.md:22-27, DeepMod_tools/sum_chr_mod.py:47-52).  Here the same additive merge of the
per-position (touch, cov, mod) counters is one integer RCCL reduce per contig x strand at the end of the run
(SURVEY.md 8e); reads shard across ranks with no data-path collective.
"""
from __future__ import annotations

import ctypes
import json
import os
import time
from typing import List, Optional, Sequence

from . import _lib


def shard(items: Sequence, rank: int, world: int) -> List:
    """Static round-robin shard of work items (the reference's queue gives the same "any worker takes the next batch"
    distribution; round-robin makes it deterministic when ranks are launched independently)."""
    return [it for i, it in enumerate(items) if i % world == rank]


def rccl_unique_id() -> bytes:
    buf = ctypes.create_string_buffer(128)
    _lib.check(_lib.load().dm_rccl_unique_id(buf))
    return buf.raw


def rccl_info():
    """(path of the collective library this process bound, ncclGetVersion code - 22707 = 2.27.7) - binds it on first use (dm_rccl_info)."""
    buf = ctypes.create_string_buffer(1024)
    version = ctypes.c_int(0)
    _lib.check(_lib.load().dm_rccl_info(buf, len(buf), ctypes.byref(version)))
    return buf.value.decode('utf-8', 'replace'), int(version.value)


class FileRendezvous:
    """Byte / JSON exchange between the ranks of one run through files in a directory every rank can see (the run's
    output folder).  Writes are atomic (tmp + rename); readers poll."""

    def __init__(self, directory: str, rank: int, world: int, timeout: float = 600.0, fresh_after: float = 0.0):
        self.dir, self.rank, self.world, self.timeout = directory, rank, world, timeout
        self.fresh_after = fresh_after        # files last written before this time belong to an earlier run: ignored
        os.makedirs(directory, exist_ok=True)

    def _path(self, name: str) -> str:
        return os.path.join(self.dir, name)

    def put(self, name: str, data: bytes) -> None:
        tmp = self._path('.%s.tmp.%d' % (name, os.getpid()))
        with open(tmp, 'wb') as fh:
            fh.write(data)
        os.replace(tmp, self._path(name))

    ABORT = 'ABORT'

    def abort(self, reason: str) -> None:
        """A rank that fails says so: every other rank's next (or current) wait ends within milliseconds instead of the timeout."""
        try:
            self.put(self.ABORT, ('rank %d: %s' % (self.rank, reason)).encode('utf-8', 'replace'))
        except OSError:
            pass

    def _aborted(self):
        path = self._path(self.ABORT)
        if os.path.exists(path) and os.path.getmtime(path) >= self.fresh_after:
            with open(path, 'rb') as fh:
                return fh.read().decode('utf-8', 'replace')
        return None

    def get(self, name: str) -> bytes:
        deadline = time.time() + self.timeout
        path = self._path(name)
        while not (os.path.exists(path) and os.path.getmtime(path) >= self.fresh_after):
            why = self._aborted()
            if why is not None:
                raise RuntimeError('rendezvous aborted by ' + why)
            if time.time() > deadline:
                raise TimeoutError('rendezvous: %s did not appear within %.0f s' % (path, self.timeout))
            time.sleep(0.005)
        with open(path, 'rb') as fh:
            return fh.read()

    def broadcast(self, name: str, data: Optional[bytes], root: int = 0) -> bytes:
        if self.rank == root:
            self.put(name, data)
            return data
        return self.get(name)

    def all_gather_json(self, name: str, obj) -> list:
        self.put('%s.%d' % (name, self.rank), json.dumps(obj).encode())
        return [json.loads(self.get('%s.%d' % (name, r)).decode()) for r in range(self.world)]


class Communicator:
    """dm_comm handle: created once per process (collectively), reused for every reduce, destroyed at the end."""

    def __init__(self, device: int, unique_id: bytes, rank: int, nranks: int):
        self._lib = _lib.load()
        self.device, self.rank, self.nranks = device, rank, nranks
        buf = ctypes.create_string_buffer(unique_id, 128)
        # ncclCommInitRank blocks until every rank has joined.  A peer that died before it got here, or a collective library that cannot reach it,
        # would leave this process waiting for ever inside the library: a watchdog ends it (exit code 3, the reason on stderr - the manager then
        # stops the other ranks) after DEEPMOD_COMM_TIMEOUT seconds (default 600; 0 = wait for ever)
        import os
        import sys
        import threading
        timeout = float(os.environ.get('DEEPMOD_COMM_TIMEOUT', '600') or 0)
        watchdog = None
        if timeout > 0:
            def give_up():
                sys.stderr.write("dm_comm_create (ncclCommInitRank, rank %d of %d on device %d) has not returned after %.0f s: a rank that never joined, or a "
                                 "collective library that cannot reach its peers; giving up (DEEPMOD_COMM_TIMEOUT)\n" % (rank, nranks, device, timeout))
                sys.stderr.flush()
                os._exit(3)
            watchdog = threading.Timer(timeout, give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            self._h = self._lib.dm_comm_create(device, buf, rank, nranks)
        finally:
            if watchdog is not None:
                watchdog.cancel()
        if not self._h:
            raise _lib.DeepModHipError("dm_comm_create: " + _lib.last_error())

    @classmethod
    def from_rendezvous(cls, device: int, rdv: FileRendezvous, name: str = 'rccl_id') -> "Communicator":
        uid = rdv.broadcast(name, rccl_unique_id() if rdv.rank == 0 else None)
        return cls(device, uid, rdv.rank, rdv.world)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dm_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def barrier(self):
        _lib.check(self._lib.dm_comm_barrier(self._h))

    def max(self, value: float) -> float:
        v = ctypes.c_double(value)
        _lib.check(self._lib.dm_comm_max_f64(self._h, ctypes.byref(v)))
        return v.value

    def stats(self):
        n, b = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._lib.dm_comm_stats(self._h, ctypes.byref(n), ctypes.byref(b)))
        return {"collectives": n.value, "bytes": b.value, "rccl_nranks": int(self._lib.dm_comm_size(self._h))}
