"""Dev-time tool (runs only in the build container, never on the GPU box):
interpret the reference's serialized TF1 graph (`train_deepmod/*/*.meta`) with numpy.

The `.meta` MetaGraphDef is the reference's own implementation of the path in
serialized form (graph built by bin/DeepMod_scripts/myMultiBiRNN.py:21-91).  TensorFlow
is not installed, so this file walks the protobuf by hand and evaluates the nodes
needed for `Softmax:0` / `ArgMax:0` with a memoised recursion.  Its outputs are the
golden vectors under tests/golden/ (see tests/golden/make_golden.py) that pin the
oracle in oracle/.

Nothing in here is shipped or imported by the product.
"""
from __future__ import annotations

import os
import struct
import sys
from typing import Dict, List, Optional

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmod_amd.tfbundle import pb_fields, _get_varint  # noqa: E402


class Node:
    __slots__ = ("name", "op", "inputs", "attr")

    def __init__(self):
        self.name = ""
        self.op = ""
        self.inputs: List[str] = []
        self.attr: Dict[str, bytes] = {}


def _parse_node(buf: bytes) -> Node:
    n = Node()
    for fno, wt, val in pb_fields(buf):
        if fno == 1:
            n.name = val.decode()
        elif fno == 2:
            n.op = val.decode()
        elif fno == 3:
            n.inputs.append(val.decode())
        elif fno == 5:  # map<string, AttrValue> entry
            key = None
            av = b""
            for f2, _, v2 in pb_fields(val):
                if f2 == 1:
                    key = v2.decode()
                elif f2 == 2:
                    av = v2
            n.attr[key] = av
    return n


def load_meta(path: str):
    """Return (nodes_by_name, tf_version) from a MetaGraphDef file."""
    buf = open(path, "rb").read()
    nodes: Dict[str, Node] = {}
    version = None
    for fno, wt, val in pb_fields(buf):
        if fno == 1:  # meta_info_def
            for f2, _, v2 in pb_fields(val):
                if f2 == 5:
                    version = v2.decode()
        elif fno == 2:  # graph_def
            for f2, _, v2 in pb_fields(val):
                if f2 == 1:
                    nd = _parse_node(v2)
                    nodes[nd.name] = nd
    return nodes, version


# ---- AttrValue helpers -------------------------------------------------------
def _attr_int(av: bytes, default=None):
    for fno, wt, val in pb_fields(av):
        if fno == 3:
            return val if val < (1 << 63) else val - (1 << 64)
    return default


def _attr_bool(av: bytes, default=False):
    for fno, wt, val in pb_fields(av):
        if fno == 5:
            return bool(val)
    return default


_DT = {1: np.float32, 3: np.int32, 9: np.int64, 2: np.float64, 10: np.bool_}


def _attr_tensor(av: bytes) -> np.ndarray:
    for fno, wt, val in pb_fields(av):
        if fno == 8:
            dtype = 1
            dims: List[int] = []
            content = None
            fvals: List[float] = []
            ivals: List[int] = []
            for f2, w2, v2 in pb_fields(val):
                if f2 == 1:
                    dtype = v2
                elif f2 == 2:
                    for f3, _, v3 in pb_fields(v2):
                        if f3 == 2:
                            d = 0
                            for f4, _, v4 in pb_fields(v3):
                                if f4 == 1:
                                    d = v4
                            dims.append(d)
                elif f2 == 4:
                    content = v2
                elif f2 == 5:
                    if w2 == 5:
                        fvals.append(struct.unpack("<f", v2)[0])
                    else:
                        fvals.extend(struct.unpack("<%df" % (len(v2) // 4), v2))
                elif f2 in (7, 10):
                    if w2 == 0:
                        ivals.append(v2)
                    else:
                        p = 0
                        while p < len(v2):
                            x, p = _get_varint(v2, p)
                            ivals.append(x)
            npdt = _DT[dtype]
            if content is not None:
                arr = np.frombuffer(content, dtype=npdt).copy()
            elif fvals:
                arr = np.array(fvals, dtype=npdt)
            elif ivals:
                ivals = [x if x < (1 << 63) else x - (1 << 64) for x in ivals]
                arr = np.array(ivals, dtype=npdt)
            else:
                arr = np.zeros(1, dtype=npdt)
            count = int(np.prod(dims)) if dims else 1
            if arr.size == 1 and count != 1:
                arr = np.full(count, arr[0], dtype=npdt)
            return arr.reshape(dims)
    raise ValueError("no tensor in attr")


# ---- evaluator ------------------------------------------------------------------
class GraphRunner:
    """Evaluate tensors of a TF1 GraphDef with numpy (forward inference subset)."""

    def __init__(self, nodes: Dict[str, Node], variables: Dict[str, np.ndarray]):
        self.nodes = nodes
        self.vars = variables
        self.op_counts: Dict[str, int] = {}

    def run(self, fetches: List[str], feeds: Dict[str, np.ndarray]) -> List[np.ndarray]:
        self.cache: Dict[str, object] = {}
        self.feeds = feeds
        self.op_counts = {}
        sys.setrecursionlimit(100000)
        return [self._tensor(f) for f in fetches]

    def _tensor(self, ref: str):
        if ref.startswith("^"):
            return None
        name, _, idx = ref.partition(":")
        idx = int(idx) if idx else 0
        out = self._node(name)
        return out[idx] if isinstance(out, (list, tuple)) else out

    def _node(self, name: str):
        if name in self.cache:
            return self.cache[name]
        nd = self.nodes[name]
        val = self._eval(nd)
        self.cache[name] = val
        self.op_counts[nd.op] = self.op_counts.get(nd.op, 0) + 1
        return val

    def _eval(self, nd: Node):
        op = nd.op
        ins = [i for i in nd.inputs if not i.startswith("^")]
        g = self._tensor
        if op == "Placeholder":
            key = nd.name if nd.name in self.feeds else nd.name + ":0"
            return np.asarray(self.feeds[key], dtype=np.float32)   # TF casts the feed to the placeholder dtype
        if op in ("VariableV2", "Variable"):
            return np.asarray(self.vars[nd.name], dtype=np.float32)
        if op == "Identity":
            return g(ins[0])
        if op == "Const":
            return _attr_tensor(nd.attr["value"])
        if op == "Unpack":
            x = g(ins[0])
            axis = _attr_int(nd.attr.get("axis", b""), 0)
            return [np.take(x, i, axis=axis) for i in range(x.shape[axis])]
        if op == "ConcatV2":
            axis = int(g(ins[-1]))
            return np.concatenate([g(i) for i in ins[:-1]], axis=axis)
        if op == "MatMul":
            a, b = g(ins[0]), g(ins[1])
            if _attr_bool(nd.attr.get("transpose_a", b"")):
                a = a.T
            if _attr_bool(nd.attr.get("transpose_b", b"")):
                b = b.T
            return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)
        if op == "BiasAdd":
            return (g(ins[0]) + g(ins[1])).astype(np.float32)
        if op in ("Add", "AddV2"):
            return (g(ins[0]) + g(ins[1])).astype(np.float32)
        if op == "Mul":
            return (g(ins[0]) * g(ins[1])).astype(np.float32)
        if op == "Split":
            axis = int(g(ins[0]))
            num = _attr_int(nd.attr["num_split"])
            return np.split(g(ins[1]), num, axis=axis)
        if op == "Relu":
            return np.maximum(g(ins[0]), np.float32(0)).astype(np.float32)
        if op == "RealDiv":
            return (g(ins[0]) / np.asarray(g(ins[1]), np.float32)).astype(np.float32)
        if op == "Sub":
            return (g(ins[0]) - g(ins[1])).astype(np.float32)
        if op == "Floor":
            return np.floor(g(ins[0])).astype(np.float32)
        if op == "RandomUniform":
            shape = np.asarray(g(ins[0])).astype(int).tolist()
            return np.random.default_rng(0).random(shape, dtype=np.float32)   # [0,1): with keep_prob = 1 the mask is all ones
        if op == "Sigmoid":
            x = g(ins[0]).astype(np.float32)
            return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)
        if op == "Tanh":
            return np.tanh(g(ins[0]).astype(np.float32)).astype(np.float32)
        if op == "Softmax":
            x = g(ins[0]).astype(np.float32)
            e = np.exp(x - x.max(axis=-1, keepdims=True))
            return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)
        if op == "ArgMax":
            x = g(ins[0])
            axis = int(g(ins[1]))
            return np.argmax(x, axis=axis).astype(np.int64)
        if op == "Shape":
            return np.array(g(ins[0]).shape, dtype=np.int32)
        if op == "StridedSlice":
            x = g(ins[0])
            b, e, s = g(ins[1]), g(ins[2]), g(ins[3])
            shrink = _attr_int(nd.attr.get("shrink_axis_mask", b""), 0)
            bm = _attr_int(nd.attr.get("begin_mask", b""), 0)
            em = _attr_int(nd.attr.get("end_mask", b""), 0)
            sl = []
            for i in range(len(b)):
                if shrink & (1 << i):
                    sl.append(int(b[i]))
                else:
                    sl.append(slice(None if bm & (1 << i) else int(b[i]),
                                    None if em & (1 << i) else int(e[i]), int(s[i])))
            return np.asarray(x[tuple(sl)])
        if op == "ExpandDims":
            return np.expand_dims(g(ins[0]), int(g(ins[1])))
        if op == "Fill":
            dims = np.asarray(g(ins[0])).astype(int).tolist()
            return np.full(dims, g(ins[1]), dtype=np.asarray(g(ins[1])).dtype)
        if op == "Pack":
            axis = _attr_int(nd.attr.get("axis", b""), 0)
            return np.stack([np.asarray(g(i)) for i in ins], axis=axis)
        raise NotImplementedError("op %s (node %s)" % (op, nd.name))


BILSTM_VARIABLES = (
    [("Variable", (200, 2)), ("Variable_1", (2,))]
    + [("bidirectional_rnn/%s/multi_rnn_cell/cell_%d/basic_lstm_cell/%s" % (d, l, k),
        ((107 if l == 0 else 200, 400) if k == "kernel" else (400,)))
       for d in ("fw", "bw") for l in range(3) for k in ("kernel", "bias")]
)


if __name__ == "__main__":
    import glob
    for meta in sorted(glob.glob("/root/reference/train_deepmod/rnn_*/*.meta")):
        nodes, ver = load_meta(meta)
        rng = np.random.default_rng(0)
        weights = {n: rng.uniform(-0.1, 0.1, s).astype(np.float32) for n, s in BILSTM_VARIABLES}
        gr = GraphRunner(nodes, weights)
        X = rng.normal(size=(8, 21, 7)).astype(np.float32)
        p, c = gr.run(["Softmax:0", "ArgMax:0"], {"Placeholder": X})
        print(os.path.basename(meta), ver, len(nodes), "nodes; executed", sum(gr.op_counts.values()),
              "MatMul", gr.op_counts.get("MatMul"), p[0], c[:4])
