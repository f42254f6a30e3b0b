"""Host-side model object over the C ABI: weights -> canonical blob -> dm_model, and the
TF1-Session-shaped adapter that makes the build's `mPredict1` a behavioural twin of the
reference's (bin/DeepMod_scripts/myDetect.py:805-820).

Mirrors, for the inference path only:
  * myMultiBiRNN.mCreateSession(num_input, num_hidden, timesteps, moptions)
        (bin/DeepMod_scripts/myMultiBiRNN.py:21-91) -> same 12-tuple arity, handles are tokens
  * tf.Session / tf.train.import_meta_graph / Saver.restore / tf.train.latest_checkpoint as used at
        bin/DeepMod_scripts/myDetect.py:951-956 ("load tensors by variable name")
"""
from __future__ import annotations

import ctypes
import os
import sys
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib, tfbundle
from .synth import HEAD_B, HEAD_W, cell_name

NFEAT, HID, WIN, LAYERS = 7, 100, 21, 3


def flatten_weights(tensors: Dict[str, np.ndarray], nfeat: int = NFEAT, hidden: int = HID) -> np.ndarray:
    """Canonical flat blob expected by dm_model_create (include/deepmod_hip.h)."""
    parts = []
    for direction in ("fw", "bw"):
        for layer in range(LAYERS):
            kern = np.asarray(tensors[cell_name(direction, layer, "kernel")], dtype=np.float32)
            bias = np.asarray(tensors[cell_name(direction, layer, "bias")], dtype=np.float32)
            kin = nfeat if layer == 0 else hidden
            if kern.shape != (kin + hidden, 4 * hidden) or bias.shape != (4 * hidden,):
                raise ValueError("unexpected shape for %s layer %d: %s / %s" % (direction, layer, kern.shape, bias.shape))
            parts += [kern.ravel(), bias.ravel()]
    head_w = np.asarray(tensors[HEAD_W], dtype=np.float32)
    head_b = np.asarray(tensors[HEAD_B], dtype=np.float32)
    if head_w.shape != (2 * hidden, 2) or head_b.shape != (2,):
        raise ValueError("unexpected head shapes %s / %s" % (head_w.shape, head_b.shape))
    parts += [head_w.ravel(), head_b.ravel()]
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


class DeviceArray:
    """A typed block of device memory owned through the C ABI (no torch, no hip-python)."""

    def __init__(self, shape, dtype, device: int = 0):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.device = device
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        lib = _lib.load()
        self.ptr = lib.dm_device_alloc(device, max(self.nbytes, 1))
        if not self.ptr:
            raise _lib.DeepModHipError(_lib.last_error())

    @classmethod
    def from_host(cls, arr: np.ndarray, device: int = 0) -> "DeviceArray":
        arr = np.ascontiguousarray(arr)
        out = cls(arr.shape, arr.dtype, device)
        _lib.check(_lib.load().dm_memcpy_h2d(device, out.ptr, arr.ctypes.data, arr.nbytes))
        return out

    def to_host(self) -> np.ndarray:
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            _lib.check(_lib.load().dm_memcpy_d2h(self.device, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            _lib.load().dm_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """Page-locked host bytes (hipHostMalloc through the C ABI) with numpy views: staging memory uploads run from as DMA."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = device, int(nbytes)
        self.ptr = _lib.load().dm_host_alloc(device, max(self.nbytes, 1))
        if not self.ptr:
            raise _lib.DeepModHipError(_lib.last_error())
        self._buf = (ctypes.c_char * max(self.nbytes, 1)).from_address(self.ptr)

    def view(self, dtype, count: int, offset: int = 0) -> np.ndarray:
        return np.frombuffer(self._buf, dtype, count, offset)

    def free(self):
        if getattr(self, "ptr", None):
            self._buf = None
            _lib.load().dm_host_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return a.ptr
    return a.ctypes.data


class BiLSTMModel:
    """One dm_model on one GPU (one per process, like the reference's one TF session per process)."""

    PRECISIONS = {"f32": _lib.DM_PREC_F32, "f16x3": _lib.DM_PREC_F16X3, "f16x3r": _lib.DM_PREC_F16X3_ROLES, "f16i8": _lib.DM_PREC_F16I8}

    def __init__(self, tensors: Dict[str, np.ndarray], device: int = 0, precision: Optional[str] = None):
        self._lib = _lib.load()
        self.device = device
        flat = flatten_weights(tensors)
        self._h = self._lib.dm_model_create(device, flat.ctypes.data, flat.size, NFEAT, HID, WIN, LAYERS)
        if not self._h:
            raise _lib.DeepModHipError("dm_model_create: " + _lib.last_error())
        # library default is "f16x3" (split-f16 MFMA, fp32-class results); DEEPMOD_PRECISION=f32 selects the fp32 MFMA kernel,
        # "auto" runs the load-time calibration gate of the int8 mode (calibrate_i8)
        precision = precision or os.environ.get("DEEPMOD_PRECISION")
        self.calibration = None
        if precision:
            self.set_precision(precision)

    # the gate of "auto": 2^18 synthetic windows through the fp32 and the int8 kernel (~35 ms), int8 only if they agree to 4e-5 - a
    # quarter of the sample and a stricter bound than the 5e-5 on 10^6 windows it stands for (profiles/r04/i8_tail.txt: weights with
    # trained statistics 1.2e-5 in the tail of 10^6 windows; U(-a, a) kernels at scale 4 7e-5 - 1.1e-4: refused)
    CALIB_WINDOWS, CALIB_BOUND = 1 << 18, 4e-5

    def calibrate_i8(self, n_windows: Optional[int] = None, bound: Optional[float] = None):
        """dm_model_calibrate_i8 -> (largest |p(f32) - p(f16i8)| on the calibration windows, whether DM_PREC_F16I8 was selected)."""
        err, sel = ctypes.c_double(), ctypes.c_int()
        _lib.check(self._lib.dm_model_calibrate_i8(self._h, int(n_windows or self.CALIB_WINDOWS), float(bound or self.CALIB_BOUND),
                                                   ctypes.byref(err), ctypes.byref(sel)))
        self.calibration = {"max_abs_dp": err.value, "selected_f16i8": bool(sel.value), "windows": int(n_windows or self.CALIB_WINDOWS),
                            "bound": float(bound or self.CALIB_BOUND)}
        return err.value, bool(sel.value)

    def set_precision(self, name: str):
        if name == "auto":
            if self.get_info(_lib.DM_INFO_F16_REPRESENTABLE):
                self.set_option(_lib.DM_OPT_PRECISION, _lib.DM_PREC_F16X3)
                self.calibrate_i8()
            return
        if name not in self.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(list(self.PRECISIONS) + ["auto"]))
        self.set_option(_lib.DM_OPT_PRECISION, self.PRECISIONS[name])

    @classmethod
    def from_checkpoint(cls, prefix: str, device: int = 0) -> "BiLSTMModel":
        return cls(tfbundle.load_bundle(prefix), device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dm_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- options / profiling --------------------------------------------------------------
    def set_option(self, key: int, value: int):
        _lib.check(self._lib.dm_model_set_option(self._h, key, value))

    def profile_reset(self):
        _lib.check(self._lib.dm_profile_reset(self._h))

    def profile_get(self) -> Tuple[float, int, int]:
        ms = ctypes.c_double()
        launches = ctypes.c_int64()
        windows = ctypes.c_int64()
        _lib.check(self._lib.dm_profile_get(self._h, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(windows)))
        return ms.value, launches.value, windows.value

    def sync(self):
        _lib.check(self._lib.dm_model_sync(self._h))

    def get_info(self, key: int) -> int:
        v = ctypes.c_int64()
        _lib.check(self._lib.dm_model_get_info(self._h, key, ctypes.byref(v)))
        return v.value

    def upload_async(self, dst_ptr: int, arr: np.ndarray):
        """Host array -> device address, queued on the model's stream (keep `arr` alive until the next sync())."""
        _lib.check(self._lib.dm_model_h2d_async(self._h, dst_ptr, arr.ctypes.data, arr.nbytes))

    def mark(self, i: int):
        """Record marker i on the model's stream (after the launches queued so far)."""
        _lib.check(self._lib.dm_model_mark(self._h, i))

    def wait_mark(self, i: int):
        """Block until marker i has passed (at once if never recorded)."""
        _lib.check(self._lib.dm_model_wait_mark(self._h, i))

    def predict_rows_device(self, rows_ptr: int, m_rows: int, first: int, count: int, cls_ptr: int, prob_ptr=None):
        """dm_predict_read on raw device addresses (staging buffers of the streaming worker)."""
        _lib.check(self._lib.dm_predict_read(self._h, rows_ptr, m_rows, first, count, prob_ptr, cls_ptr))

    def predict_rows_at_device(self, rows_ptr: int, m_rows: int, centre_ptr: int, count: int, cls_ptr: int, prob_ptr=None):
        """dm_predict_read_at on raw device addresses: only the windows centred on the given rows."""
        _lib.check(self._lib.dm_predict_read_at(self._h, rows_ptr, m_rows, centre_ptr, count, prob_ptr, cls_ptr))

    def assemble_rows_device(self, rows_ptr: int, code_ptr: int, ev3_ptr: int, rdesc_ptr: int, n_reads: int, n_rows: int):
        """dm_rows_assemble on raw device addresses: the device form of a batch of raw reads -> feature rows [n_rows][7], on the model's stream."""
        _lib.check(self._lib.dm_rows_assemble(self._h, rows_ptr, code_ptr, ev3_ptr, rdesc_ptr, n_reads, n_rows))

    def predict_read_at(self, rows, centres, prob=None, cls=None, want_prob: bool = True):
        """Classify the windows centred on rows[centres[i]] of a feature matrix rows float[m,7] (numpy or DeviceArray)."""
        if isinstance(rows, DeviceArray):
            m = rows.shape[0]
        else:
            rows = np.ascontiguousarray(rows, dtype=np.float32)
            m = rows.shape[0]
        if not isinstance(centres, DeviceArray):
            centres = np.ascontiguousarray(centres, dtype=np.int32)
        count = centres.shape[0]
        if cls is None:
            cls = np.empty(count, np.uint8)
        if prob is None and want_prob:
            prob = np.empty((count, 2), np.float32)
        _lib.check(self._lib.dm_predict_read_at(self._h, _ptr(rows), m, _ptr(centres), count, _ptr(prob), _ptr(cls)))
        return prob, cls

    # -- inference ------------------------------------------------------------------------
    def predict_windows(self, x, prob=None, cls=None, want_prob: bool = True):
        """x: float[n,21,7] numpy (any float dtype; cast to fp32 like the TF placeholder feed) or a
        DeviceArray.  Returns (prob float32[n,2] or None, cls uint8[n])."""
        if isinstance(x, DeviceArray):
            n = x.shape[0]
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            if x.ndim != 3 or x.shape[1:] != (WIN, NFEAT):
                raise ValueError("expected [n,%d,%d] windows, got %s" % (WIN, NFEAT, x.shape))
            n = x.shape[0]
        if cls is None:
            cls = np.empty(n, np.uint8)
        if prob is None and want_prob:
            prob = np.empty((n, 2), np.float32)
        _lib.check(self._lib.dm_predict_windows(self._h, _ptr(x), n, _ptr(prob), _ptr(cls)))
        return prob, cls

    def predict_read(self, rows, first: int, count: int, prob=None, cls=None, want_prob: bool = True):
        """Classify windows centred on rows[first .. first+count) of a per-read feature matrix
        rows float[m,7] (windows are assembled on the device)."""
        if isinstance(rows, DeviceArray):
            m = rows.shape[0]
        else:
            rows = np.ascontiguousarray(rows, dtype=np.float32)
            if rows.ndim != 2 or rows.shape[1] != NFEAT:
                raise ValueError("expected [m,%d] feature rows, got %s" % (NFEAT, rows.shape))
            m = rows.shape[0]
        if cls is None:
            cls = np.empty(count, np.uint8)
        if prob is None and want_prob:
            prob = np.empty((count, 2), np.float32)
        _lib.check(self._lib.dm_predict_read(self._h, _ptr(rows), m, first, count, _ptr(prob), _ptr(cls)))
        return prob, cls


# ---------------------------------------------------------------------------------------------
# TF1-shaped adapter (the drop-in seam of SURVEY.md 8b)
# ---------------------------------------------------------------------------------------------
class _Token:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<deepmod_amd %s>" % self.name


class Session:
    """Stands in for tf.Session in `sp_options['rnn'] = (sess, X, Y, init_l, mfpred)`
    (reference myDetect.py:972).  `run(init_l)` is a no-op (it only resets tf.metrics local
    variables); `run([mfpred], feed_dict={X: x, Y: y})` returns
    [int64[n]] exactly like the reference call at myDetect.py:816-820 (Y is accepted and
    ignored, quirk Q7).  `run([prediction], ...)` additionally exposes the probabilities."""

    def __init__(self, graph: "Graph", device: int = 0):
        self.graph = graph
        self.device = device
        self.model: Optional[BiLSTMModel] = None

    # the command-line path (every model the detect command loads comes through here): DM_PREC_F16X3, the mode that meets the path's 1e-4
    # tolerance by construction (round 5: the int8 cross-term mode is opt-in again - its documented bound is 2e-4).  DEEPMOD_PRECISION =
    # f32 / f16i8 overrides; DEEPMOD_PRECISION=auto lets the model's own calibration run decide between f16x3 and f16i8
    # (BiLSTMModel.calibrate_i8) and says on stderr what it chose
    def _load(self, tensors):
        want = os.environ.get("DEEPMOD_PRECISION", "f16x3")
        self.model = BiLSTMModel(tensors, self.device, precision=want)
        if want == "auto" and self.model.calibration is not None:
            c = self.model.calibration
            sys.stderr.write("deepmod_amd: DEEPMOD_PRECISION=auto: calibration max|dp| %.3g on %d windows (bound %.3g) -> %s\n"
                             % (c["max_abs_dp"], c["windows"], c["bound"], "f16i8 (int8 cross terms)" if c["selected_f16i8"] else "f16x3"))

    def restore(self, prefix: str):
        self._load(tfbundle.load_bundle(prefix))

    def load_tensors(self, tensors: Dict[str, np.ndarray]):
        self._load(tensors)

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        if all(f is self.graph.init_l or f is self.graph.init for f in flist):
            return None if single else [None] * len(flist)
        if self.model is None:
            raise _lib.DeepModHipError("Session.run before restore(): no weights loaded")
        if feed_dict is None or self.graph.X not in feed_dict:
            raise ValueError("feed_dict must provide X")
        x = np.asarray(feed_dict[self.graph.X])
        want_prob = any(f is self.graph.prediction for f in flist)
        try:
            prob, cls = self.model.predict_windows(x, want_prob=want_prob)
        except _lib.DeepModRangeError:        # the split-f16 kernel refuses what it cannot represent; TensorFlow computes it in fp32
            keep = self.model.get_info(_lib.DM_INFO_PRECISION)
            self.model.set_option(_lib.DM_OPT_PRECISION, _lib.DM_PREC_F32)
            try:
                prob, cls = self.model.predict_windows(x, want_prob=want_prob)
            finally:
                self.model.set_option(_lib.DM_OPT_PRECISION, keep)
        out = []
        for f in flist:
            if f is self.graph.mfpred:
                out.append(cls.astype(np.int64))
            elif f is self.graph.prediction:
                out.append(prob)
            else:
                raise ValueError("cannot fetch %r (inference-only build)" % (f,))
        return out[0] if single else out

    def close(self):
        if self.model is not None:
            self.model.close()
            self.model = None


class Graph:
    def __init__(self, num_input, num_hidden, timesteps):
        if (num_input, num_hidden, timesteps) != (NFEAT, HID, WIN):
            raise ValueError("this build supports fnum=7 hidden=100 windowsize=21 only (got %s)" %
                             ((num_input, num_hidden, timesteps),))
        self.X = _Token("X")
        self.Y = _Token("Y")
        self.init = _Token("init")
        self.init_l = _Token("init_l")
        self.mfpred = _Token("mfpred")
        self.prediction = _Token("prediction")
        self.saver = Saver()


class Saver:
    """`new_saver.restore(sess, prefix)` -> load tensors by variable name."""

    def restore(self, sess: Session, save_path: str):
        if save_path is None:
            raise ValueError("no checkpoint found (latest_checkpoint returned None)")
        sess.restore(save_path)


_last_graph: Optional[Graph] = None


def mCreateSession(num_input, num_hidden, timesteps, moptions):
    """Same arity/positions as the reference's 12-tuple
    (init, init_l, loss_op, accuracy, train_op, X, Y, saver, auc_op, mpre, mspf, mfpred);
    training-side entries are None (inference-only build)."""
    global _last_graph
    if moptions.get("outputlayer", "") in ("sigmoid",):
        raise ValueError("--outputlayer sigmoid is not used by any shipped model and is not built")
    g = Graph(num_input, num_hidden, timesteps)
    _last_graph = g
    return (g.init, g.init_l, None, None, None, g.X, g.Y, g.saver, None, None, None, g.mfpred)


def import_meta_graph(meta_path: str) -> Saver:
    """The .meta is not needed to run (the graph is compiled in); its presence is still checked
    so a wrong --modfile fails as early as it does in the reference."""
    if not os.path.exists(meta_path) and not os.path.exists(meta_path[:-5] + ".index"):
        raise FileNotFoundError(meta_path)
    return Saver()


def new_session(device: int = 0) -> Session:
    if _last_graph is None:
        raise RuntimeError("call mCreateSession first")
    return Session(_last_graph, device)


latest_checkpoint = tfbundle.latest_checkpoint
