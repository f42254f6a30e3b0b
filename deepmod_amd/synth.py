"""Seeded synthetic inputs for the BiLSTM path (no network, no real checkpoints/reads here).

* ``synthetic_weights(seed, scale)``: tensors with the exact names/shapes of the reference
  checkpoints; glorot-uniform kernels scaled by ``scale`` (1 keeps
  p1 mid-range, 4 saturates - both regimes are tested), biases U(-0.5,0.5), head N(0,1).
* ``synthetic_windows(n, seed)``: BASELINE.json config 2 windows ``float32[n,21,7]`` in the
  feature layout of reference bin/DeepMod_scripts/myDetect.py:894-900 (fnum=7):
  [A,C,G,T one-hot of the reference base | mean | stdv | length].
* ``write_synthetic_checkpoint(prefix, seed, scale)``: a TF bundle with the real checkpoint's
  byte layout (4,900,832-byte .data), restorable by ``deepmod_amd.model``.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import tfbundle

NFEAT, HID, WIN = 7, 100, 21
HEAD_W, HEAD_B = "Variable", "Variable_1"


def cell_name(direction: str, layer: int, what: str) -> str:
    return "bidirectional_rnn/%s/multi_rnn_cell/cell_%d/basic_lstm_cell/%s" % (direction, layer, what)


def variable_shapes(nfeat: int = NFEAT, hidden: int = HID):
    shapes = [(HEAD_W, (2 * hidden, 2)), (HEAD_B, (2,))]
    for d in ("bw", "fw"):
        for l in range(3):
            kin = nfeat if l == 0 else hidden
            shapes.append((cell_name(d, l, "bias"), (4 * hidden,)))
            shapes.append((cell_name(d, l, "kernel"), (kin + hidden, 4 * hidden)))
    return shapes


# byte offsets of the real checkpoints' .data-00000-of-00001 (identical in all five BiLSTM models;
# gaps hold the Adam slots).  Parsed from the reference .index files with tfbundle.read_index.
REAL_LAYOUT = {
    HEAD_W: 0, HEAD_B: 4800,
    cell_name("bw", 0, "bias"): 4832, cell_name("bw", 0, "kernel"): 9632,
    cell_name("bw", 1, "bias"): 523232, cell_name("bw", 1, "kernel"): 528032,
    cell_name("bw", 2, "bias"): 1488032, cell_name("bw", 2, "kernel"): 1492832,
    cell_name("fw", 0, "bias"): 2452832, cell_name("fw", 0, "kernel"): 2457632,
    cell_name("fw", 1, "bias"): 2971232, cell_name("fw", 1, "kernel"): 2976032,
    cell_name("fw", 2, "bias"): 3936032, cell_name("fw", 2, "kernel"): 3940832,
}
REAL_DATA_SIZE = 4900832


def synthetic_weights(seed: int = 7, scale: float = 1.0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in variable_shapes():
        if name == HEAD_W:
            out[name] = rng.standard_normal(shape).astype(np.float32)
        elif name == HEAD_B:
            out[name] = rng.standard_normal(shape).astype(np.float32)
        elif name.endswith("bias"):
            out[name] = rng.uniform(-0.5, 0.5, shape).astype(np.float32)
        else:
            a = scale * np.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rng.uniform(-a, a, shape).astype(np.float32)
    return out


def synthetic_windows(n: int, seed: int = 20260928) -> np.ndarray:
    rng = np.random.default_rng(seed)
    x = np.zeros((n, WIN, NFEAT), dtype=np.float32)
    cat = rng.choice(5, size=(n, WIN), p=[0.24, 0.24, 0.24, 0.24, 0.04])
    for b in range(4):
        x[:, :, b] = (cat == b)
    x[:, :, 4] = np.round(np.clip(rng.normal(0.0, 1.2, (n, WIN)), -5, 5), 3)
    x[:, :, 5] = np.round(np.abs(rng.normal(0.25, 0.15, (n, WIN))), 3)
    x[:, :, 6] = rng.geometric(0.12, (n, WIN)).astype(np.float32)
    return x


def write_synthetic_checkpoint(prefix: str, seed: int = 7, scale: float = 1.0) -> Dict[str, np.ndarray]:
    w = synthetic_weights(seed, scale)
    tfbundle.write_bundle(prefix, w, layout=REAL_LAYOUT, total_size=REAL_DATA_SIZE)
    return w


def write_synthetic_data_for_index(index_path: str, prefix: str, seed: int = 7, scale: float = 1.0) -> Dict[str, np.ndarray]:
    """A shipped model directory rebuilt around its REAL `.index` (copied next to `prefix`): a `.data-00000-of-00001` of the size
    and byte layout that index describes (Adam slots and all), holding synthetic weights where the model's variables live and
    zeros elsewhere.  What a user who has the real shard does with --modfile is then exercised byte for byte except the values."""
    import os
    import shutil
    os.makedirs(os.path.dirname(prefix) or '.', exist_ok=True)
    shutil.copyfile(index_path, prefix + ".index")
    entries = tfbundle.read_index(prefix + ".index")
    w = synthetic_weights(seed, scale)
    total = max(e.offset + e.size for e in entries.values())
    blob = np.zeros(total, np.uint8)
    for name, arr in w.items():
        e = entries[name]
        if tuple(e.shape) != arr.shape or e.size != arr.nbytes:
            raise ValueError("variable %s: the index says %s, the model wants %s" % (name, e.shape, arr.shape))
        blob[e.offset:e.offset + e.size] = np.frombuffer(np.ascontiguousarray(arr, "<f4").tobytes(), np.uint8)
    with open(tfbundle.data_path(prefix), "wb") as fh:
        fh.write(blob.tobytes())
    return w
