"""ORACLE — TEST INFRASTRUCTURE ONLY (numpy twin of oracle/deepmod_oracle.c + ctypes loader).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product (deepmod_amd/) never does.

Restates the reference graph of /root/reference/bin/DeepMod_scripts/myMultiBiRNN.py:21-61
(see the header of deepmod_oracle.c for the line-by-line mapping).  Pinned against the
numpy-interpreted reference GraphDef through tests/golden/bilstm_*.npz; arithmetic parity
vs real TensorFlow kernels is otherwise UNPINNED (TF absent, reference ships no tests).
The activation kernels TF's CPU build uses (Eigen's rational tanh / logistic) are restated
below from the published algorithm (eigen_fast_tanh, eigen_logistic): the graph evaluated
with them bounds what that unpinned part can move (tests/test_oracle_golden.py).
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


def eigen_fast_tanh(x: np.ndarray) -> np.ndarray:
    """float32 tanh as the CPU kernels of TensorFlow 1.x compute it: Eigen 3.3's `generic_fast_tanh_float` (Eigen/src/Core/MathFunctionsImpl.h; the
    `scalar_tanh_op` packet path under EIGEN_FAST_MATH, which TF builds with) - the argument clamped to [-9, 9], then a 13th-degree odd over a 6th-degree
    even polynomial in Horner form, every operation in float32.  Eigen is a third-party dependency of TensorFlow, itself a dependency the reference
    does not vendor (docs/Install.md:26): the published algorithm is restated here; measured against libm over [-20, 20]: 3.5e-7 at most."""
    f = np.float32
    x = np.clip(np.asarray(x, f), f(-9.0), f(9.0))
    alpha = [f(4.89352455891786e-03), f(6.37261928875436e-04), f(1.48572235717979e-05), f(5.12229709037114e-08), f(-8.60467152213735e-11),
             f(2.00018790482477e-13), f(-2.76076847742355e-16)]
    beta = [f(4.89352518554385e-03), f(2.26843463243900e-03), f(1.18534705686654e-04), f(1.19825839466702e-06)]
    x2 = x * x
    p = x2 * alpha[6] + alpha[5]
    for c in (alpha[4], alpha[3], alpha[2], alpha[1], alpha[0]):
        p = x2 * p + c
    p = x * p
    q = x2 * beta[3] + beta[2]
    q = x2 * q + beta[1]
    q = x2 * q + beta[0]
    return (p / q).astype(f)


def eigen_logistic(x: np.ndarray) -> np.ndarray:
    """float32 sigmoid as Eigen's `scalar_logistic_op<float>` computes it (Eigen/src/Core/functors/UnaryFunctors.h, the Eigen revisions TensorFlow
    1.1x pins): the argument clamped to [-18, 18], a 9th-degree odd over a 10th-degree even polynomial, + 0.5.  Against libm: 2.1e-7 at most."""
    f = np.float32
    x = np.clip(np.asarray(x, f), f(-18.0), f(18.0))
    alpha = [f(2.48287947061529e-01), f(8.51377133304701e-03), f(6.08574864600143e-05), f(1.15627324459942e-07), f(4.37031012579801e-11)]
    beta = [f(9.93151921023180e-01), f(1.16817656904453e-01), f(1.70198817374094e-03), f(6.29106785017040e-06), f(5.76102136993427e-09),
            f(6.10247389755681e-13)]
    x2 = x * x
    p = x2 * alpha[4] + alpha[3]
    for c in (alpha[2], alpha[1], alpha[0]):
        p = x2 * p + c
    p = x * p
    q = x2 * beta[5] + beta[4]
    for c in (beta[3], beta[2], beta[1], beta[0]):
        q = x2 * q + c
    return (p / q + f(0.5)).astype(f)


def predict_windows_np(weights: Dict[str, np.ndarray], x: np.ndarray, dtype=np.float32, activations: str = 'libm'
                       ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """numpy restatement, fp32 like the graph (default) or with every operation in `dtype` = float64: the EXACT value of
    the graph on the fp32 weights and inputs, the yardstick where fp32 round-off itself is amplified (weight scale 16).
    activations = 'eigen' (fp32 only): tanh and sigmoid by the rational approximations of TensorFlow's CPU kernels (eigen_fast_tanh,
    eigen_logistic) instead of libm's - how far the one part of the reference's arithmetic that no fixture pins can move a result.
    x: [n,21,7] -> (prob [n,2], cls [n] int64, hcat [n,200])."""
    F = np.dtype(dtype).type
    sig = lambda t: (F(1) / (F(1) + np.exp(-t, dtype=F))).astype(F)
    tanh = np.tanh
    if activations == 'eigen':
        if F is not np.float32:
            raise ValueError("activations='eigen' is the float32 arithmetic of TensorFlow's CPU kernels")
        sig, tanh = eigen_logistic, eigen_fast_tanh
    elif activations != 'libm':
        raise ValueError("activations: 'libm' or 'eigen'")
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
                c[l] = (c[l] * sig(gf + F(1.0)) + sig(gi) * tanh(gj)).astype(F)
                h[l] = (tanh(c[l]) * sig(go)).astype(F)
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
