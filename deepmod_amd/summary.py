"""Per-position (coverage, mod-count) summary on the GPU and the BED writer.

Counterpart of the accumulation loop and writer of the reference's sum_handler
(bin/DeepMod_scripts/myDetect.py:1089-1120): dense int32 counters per contig x strand
(`touch`, `cov`, `mod`) updated with integer atomics, so the result is independent of read order
and of how reads were sharded over GPUs; the cross-GPU merge is one integer RCCL reduce over a persistent
communicator (deepmod_amd/comm.py; the additive merge the reference does across runs with
DeepMod_tools/sum_chr_mod.py:47-52).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import _lib
from .model import DeviceArray, _ptr

FLAG_IS_BASE = 1   # refbase == Base and refbase not in '-Nn'   (myDetect.py:1091-1092)
FLAG_NOT_GAP = 2   # readbase != '-'                            (myDetect.py:1097)
FLAG_MOD = 4       # mod_pred == 1                              (myDetect.py:1099)


class PositionSummary:
    def __init__(self, length: int, device: int = 0):
        self._lib = _lib.load()
        self.length = int(length)
        self.device = device
        self._h = self._lib.dm_summary_create(device, self.length)
        if not self._h:
            raise _lib.DeepModHipError("dm_summary_create: " + _lib.last_error())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dm_summary_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def follow(self, model=None):
        """Enqueue device-resident adds on `model`'s stream (in order with its launches); None detaches."""
        _lib.check(self._lib.dm_summary_follow(self._h, model._h if model is not None else None))
        self._followed = model          # keep the model alive as long as the summary follows it

    def add(self, pos, flags, n: Optional[int] = None):
        if not isinstance(pos, DeviceArray):
            pos = np.ascontiguousarray(pos, dtype=np.int64)
            flags = np.ascontiguousarray(flags, dtype=np.uint8)
            n = pos.size
        _lib.check(self._lib.dm_summary_add(self._h, _ptr(pos), _ptr(flags), n))

    def add_classified(self, pos, flags, cls, n: Optional[int] = None):
        if not isinstance(pos, DeviceArray):
            pos = np.ascontiguousarray(pos, dtype=np.int64)
            n = pos.size
        if not isinstance(flags, DeviceArray):
            flags = np.ascontiguousarray(flags, dtype=np.uint8)
        if not isinstance(cls, DeviceArray):
            cls = np.ascontiguousarray(cls, dtype=np.uint8)
        _lib.check(self._lib.dm_summary_add_classified(self._h, _ptr(pos), _ptr(flags), _ptr(cls), n))

    def add_device(self, pos_ptr: int, flags_ptr: int, n: int):
        """dm_summary_add on raw device addresses."""
        _lib.check(self._lib.dm_summary_add(self._h, pos_ptr, flags_ptr, n))

    def add_classified_device(self, pos_ptr: int, flags_ptr: int, cls_ptr: int, n: int):
        _lib.check(self._lib.dm_summary_add_classified(self._h, pos_ptr, flags_ptr, cls_ptr, n))

    def sync(self):
        _lib.check(self._lib.dm_summary_sync(self._h))

    def fetch(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        touch = np.empty(self.length, np.int32)
        cov = np.empty(self.length, np.int32)
        mod = np.empty(self.length, np.int32)
        _lib.check(self._lib.dm_summary_fetch(self._h, touch.ctypes.data, cov.ctypes.data, mod.ctypes.data))
        return touch, cov, mod

    def grow(self, new_length: int):
        """Extend the counters to new_length positions (kept counts, zeros beyond); no-op if not larger."""
        _lib.check(self._lib.dm_summary_grow(self._h, int(new_length)))
        self.length = max(self.length, int(new_length))

    # -- multi-GPU merge --------------------------------------------------------------------
    def reduce(self, comm, root: int = 0):
        """In-place int32 sum of touch|cov|mod over the ranks of `comm` (deepmod_amd.comm.Communicator): one RCCL
        reduce to `root` (result valid there), or an all-reduce with root < 0.  Collective: every rank calls it."""
        _lib.check(self._lib.dm_summary_reduce(self._h, comm._h, root))

    def reduce_scatter(self, comm) -> Tuple[int, int]:
        """The merge that scales: afterwards this rank owns the all-rank sums of positions [first, first + count) - one
        ncclReduceScatter per counter array (dm_summary_reduce_scatter).  Collective.  -> (first, count)"""
        import ctypes
        first, count = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._lib.dm_summary_reduce_scatter(self._h, comm._h, ctypes.byref(first), ctypes.byref(count)))
        self._slice = (first.value, count.value)
        return self._slice

    def fetch_slice(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(touch, cov, mod) of this rank's slice after reduce_scatter."""
        n = self._slice[1]
        out = [np.empty(n, np.int32) for _ in range(3)]
        _lib.check(self._lib.dm_summary_fetch_slice(self._h, *[a.ctypes.data for a in out]))
        return tuple(out)


def bed_lines(chrom: str, strand: str, base: str, touch: np.ndarray, cov: np.ndarray, mod: np.ndarray, first_pos: int = 0) -> bytes:
    """BED text exactly as the reference writes it (myDetect.py:1112-1120): one line per position
    whose key was created (touch > 0), sorted by position, single spaces, trailing space before the
    newline, column 5 = min(cov, 1000), pct = trunc(100*mod/max(cov,1)).  Formatted by dm_bed_format_at (host C);
    first_pos: the position of element 0 (a rank's slice of the contig after reduce_scatter)."""
    lib = _lib.load()
    touch = np.ascontiguousarray(touch, np.int32)
    cov = np.ascontiguousarray(cov, np.int32)
    mod = np.ascontiguousarray(mod, np.int32)
    n = len(touch)
    args = (chrom.encode("ascii"), strand.encode("ascii"), base.encode("ascii"), int(first_pos), touch.ctypes.data, cov.ctypes.data, mod.ctypes.data, n)
    bound = lib.dm_bed_format_at(*args, None, 0)
    if bound < 0:
        raise _lib.DeepModHipError("dm_bed_format: " + _lib.last_error())
    buf = np.empty(max(int(bound), 1), np.uint8)
    got = lib.dm_bed_format_at(*args, buf.ctypes.data, int(bound))
    if got < 0 or got > bound:
        raise _lib.DeepModHipError("dm_bed_format: " + _lib.last_error())
    return buf[:got].tobytes()


def bed_parts(chrom: str, strand: str, base: str, touch: np.ndarray, cov: np.ndarray, mod: np.ndarray, first_pos: int = 0,
              slice_positions: int = 1 << 22, threads: int = 8):
    """The same text as bed_lines, as uint8 arrays in position order: the table is cut into slices of `slice_positions` positions
    that `threads` threads format side by side (dm_bed_format_at takes the position of its first element; compiled code, no
    interpreter lock), `threads` slices at a time - a chr1-sized table is 6e7 lines / 5 GB of text, and one thread formatting
    it into one buffer that is then copied was half of a low-coverage run."""
    from concurrent.futures import ThreadPoolExecutor
    lib = _lib.load()
    touch = np.ascontiguousarray(touch, np.int32)
    cov = np.ascontiguousarray(cov, np.int32)
    mod = np.ascontiguousarray(mod, np.int32)
    n = len(touch)
    head = (chrom.encode("ascii"), strand.encode("ascii"), base.encode("ascii"))

    def one(lo):
        hi = min(n, lo + slice_positions)
        args = head + (int(first_pos) + lo, touch[lo:hi].ctypes.data, cov[lo:hi].ctypes.data, mod[lo:hi].ctypes.data, hi - lo)
        bound = lib.dm_bed_format_at(*args, None, 0)
        if bound < 0:
            raise _lib.DeepModHipError("dm_bed_format: " + _lib.last_error())
        if bound == 0:
            return None
        buf = np.empty(int(bound), np.uint8)
        got = lib.dm_bed_format_at(*args, buf.ctypes.data, int(bound))
        if got < 0 or got > bound:
            raise _lib.DeepModHipError("dm_bed_format: " + _lib.last_error())
        return buf[:got]

    starts = list(range(0, n, max(1, int(slice_positions))))
    if len(starts) <= 1 or threads <= 1:
        for lo in starts:
            part = one(lo)
            if part is not None:
                yield part
        return
    with ThreadPoolExecutor(threads) as pool:
        for g in range(0, len(starts), threads):
            for part in pool.map(one, starts[g:g + threads]):
                if part is not None:
                    yield part


def bed_lines_py(chrom: str, strand: str, base: str, touch: np.ndarray, cov: np.ndarray, mod: np.ndarray) -> bytes:
    """The same text from a Python loop (a line-by-line restatement of the reference's writer; tests compare the two)."""
    idx = np.flatnonzero(touch > 0)
    out = []
    for pos, cv, md in zip(idx.tolist(), cov[idx].tolist(), mod[idx].tolist()):
        pct = "%d" % (100 * md / (cv if cv > 0 else 1))
        out.append(" ".join([chrom, str(pos), str(pos + 1), base, str(1000 if cv > 1000 else cv), strand,
                             str(pos), str(pos + 1), "0,0,0", str(cv), pct, str(md), "\n"]))
    return "".join(out).encode("ascii")
