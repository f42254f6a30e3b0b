"""ORACLE — TEST INFRASTRUCTURE ONLY (numpy twin of oracle/deepmod_oracle.c + ctypes loader).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product (deepmod_amd/) never does.

Restates the reference graph of /root/reference/bin/DeepMod_scripts/myMultiBiRNN.py:21-61
(see the header of deepmod_oracle.c for the line-by-line mapping).  Pinned against the
numpy-interpreted reference GraphDef through tests/golden/bilstm_*.npz; arithmetic parity
vs real TensorFlow kernels is otherwise UNPINNED (TF absent, reference ships no tests).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

NFEAT, HID, WIN, LIVE = 7, 100, 21, 11

HEAD_W, HEAD_B = "Variable", "Variable_1"


def cell_name(direction: str, layer: int, what: str) -> str:
    return "bidirectional_rnn/%s/multi_rnn_cell/cell_%d/basic_lstm_cell/%s" % (direction, layer, what)


def _sigmoid(x):
    return (np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def predict_windows_np(weights: Dict[str, np.ndarray], x: np.ndarray, dtype=np.float32
                       ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """numpy restatement, fp32 like the graph (default) or with every operation in `dtype` = float64: the EXACT value of
    the graph on the fp32 weights and inputs, the yardstick where fp32 round-off itself is amplified (weight scale 16).
    x: [n,21,7] -> (prob [n,2], cls [n] int64, hcat [n,200])."""
    F = np.dtype(dtype).type
    sig = lambda t: (F(1) / (F(1) + np.exp(-t, dtype=F))).astype(F)
    x = np.asarray(x, dtype=np.float32).astype(F)
    n = x.shape[0]
    finals = []
    for d, direction in enumerate(("fw", "bw")):
        h = [np.zeros((n, HID), F) for _ in range(3)]
        c = [np.zeros((n, HID), F) for _ in range(3)]
        for s in range(LIVE):
            row = s if d == 0 else WIN - 1 - s
            inp = x[:, row, :]
            for l in range(3):
                kern = weights[cell_name(direction, l, "kernel")].astype(F)
                bias = weights[cell_name(direction, l, "bias")].astype(F)
                g = (np.concatenate([inp, h[l]], axis=1) @ kern).astype(F) + bias
                gi, gj, gf, go = np.split(g, 4, axis=1)
                c[l] = (c[l] * sig(gf + F(1.0)) + sig(gi) * np.tanh(gj)).astype(F)
                h[l] = (np.tanh(c[l]) * sig(go)).astype(F)
                inp = h[l]
        finals.append(h[2])
    hcat = np.concatenate(finals, axis=1)
    logits = (hcat @ weights[HEAD_W].astype(F)).astype(F) + weights[HEAD_B].astype(F)
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    prob = (e / e.sum(axis=1, keepdims=True)).astype(F)
    cls = np.argmax(prob, axis=1).astype(np.int64)
    return prob, cls, hcat


def flatten_weights(weights: Dict[str, np.ndarray]) -> np.ndarray:
    """Canonical flat blob of deepmod_oracle.c (408,402 floats)."""
    parts = []
    for direction in ("fw", "bw"):
        for l in range(3):
            parts.append(np.asarray(weights[cell_name(direction, l, "kernel")], np.float32).ravel())
            parts.append(np.asarray(weights[cell_name(direction, l, "bias")], np.float32).ravel())
    parts.append(np.asarray(weights[HEAD_W], np.float32).ravel())
    parts.append(np.asarray(weights[HEAD_B], np.float32).ravel())
    flat = np.concatenate(parts)
    assert flat.size == 408402, flat.size
    return np.ascontiguousarray(flat)


# ---------------------------------------------------------------------------------
# C oracle (gcc, OpenMP) through ctypes
# ---------------------------------------------------------------------------------
_LIB: Optional[ctypes.CDLL] = None
LIB_PATH = os.path.join(HERE, "_build", "libdeepmod_oracle.so")


def build_c_oracle(force: bool = False) -> str:
    src = os.path.join(HERE, "deepmod_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB_PATH


def c_oracle() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            build_c_oracle()
        lib = ctypes.CDLL(LIB_PATH)
        lib.dmo_weight_count.restype = ctypes.c_int64
        lib.dmo_predict_windows.restype = ctypes.c_int
        lib.dmo_predict_windows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.dmo_summary_add.restype = ctypes.c_int
        lib.dmo_summary_add.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p,
                                                                 ctypes.c_void_p, ctypes.c_int64]
        _LIB = lib
    return _LIB


def usable_cores() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota.  A container that shows 256 logical CPUs but is
    throttled to 16 would otherwise run the oracle's OpenMP loops with 256 threads on 16 CPUs (measured on the GPU boxes: the config-1
    test's oracle pass 6x slower)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def predict_windows_c(weights: Dict[str, np.ndarray], x: np.ndarray, nthreads: int = 0,
                      want_hcat: bool = False):
    """nthreads = 0: as many OpenMP threads as this process has usable cores."""
    lib = c_oracle()
    if nthreads <= 0:
        nthreads = usable_cores()
    flat = flatten_weights(weights)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    prob = np.empty((n, 2), np.float32)
    cls = np.empty(n, np.uint8)
    hcat = np.empty((n, 2 * HID), np.float32) if want_hcat else None
    rc = lib.dmo_predict_windows(flat.ctypes.data, x.ctypes.data, n, prob.ctypes.data, cls.ctypes.data,
                                 hcat.ctypes.data if want_hcat else None, nthreads)
    if rc != 0:
        raise RuntimeError("dmo_predict_windows rc=%d" % rc)
    if want_hcat:
        return prob, cls.astype(np.int64), hcat
    return prob, cls.astype(np.int64)


def summary_add_c(touch, cov, mod, pos, flags):
    lib = c_oracle()
    pos = np.ascontiguousarray(pos, np.int64)
    flags = np.ascontiguousarray(flags, np.uint8)
    rc = lib.dmo_summary_add(touch.ctypes.data, cov.ctypes.data, mod.ctypes.data, touch.size,
                             pos.ctypes.data, flags.ctypes.data, pos.size)
    if rc != 0:
        raise RuntimeError("dmo_summary_add rc=%d" % rc)
