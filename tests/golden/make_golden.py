"""Generate the committed golden fixtures.  Runs ONLY in the build container (needs
/root/reference); the fixtures it writes are plain data (inputs + expected outputs).

    python tests/golden/make_golden.py [bilstm] [host]

bilstm  G1: for every shipped BiLSTM .meta, interpret the reference's serialized graph
        (tools/graphdef_interp.py) on seeded synthetic weights (deepmod_amd.synth) and
        windows -> {seed, scale, X, prob, cls, weight checksum}.  Also records the parsed
        .index tables (names/shapes/offsets) as JSON.
host    G2-G4: stub-import the reference's own Python (fake tensorflow/h5py modules) and
        record inputs/outputs of get_Feature, mPredict1 (window order, batch split, scatter)
        and sum_handler (exact BED bytes).
"""
from __future__ import annotations

import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

REF = "/root/reference"


def make_bilstm():
    from graphdef_interp import load_meta, GraphRunner
    from deepmod_amd import synth, tfbundle

    index_tables = {}
    for meta in sorted(glob.glob(REF + "/train_deepmod/rnn_*/*.meta")):
        model = os.path.basename(os.path.dirname(meta))
        nodes, ver = load_meta(meta)
        entries = tfbundle.read_index(meta[:-5] + ".index")
        index_tables[model] = {
            "prefix": os.path.basename(meta)[:-5], "tf_version": ver,
            "entries": {n: {"shape": list(e.shape), "offset": e.offset, "size": e.size, "crc32c": e.crc32c}
                        for n, e in entries.items()}}
        for scale in (1.0, 4.0):
            seed_w = 7 + int(scale)
            w = synth.synthetic_weights(seed_w, scale)
            x = synth.synthetic_windows(64, seed=1000 + int(scale))
            gr = GraphRunner(nodes, w)
            prob, cls = gr.run(["Softmax:0", "ArgMax:0"], {"Placeholder": x})
            assert gr.op_counts.get("MatMul") == 67, gr.op_counts
            wsum = float(sum(np.abs(v.astype(np.float64)).sum() for v in w.values()))
            out = os.path.join(HERE, "bilstm_%s_s%d.npz" % (model, int(scale)))
            np.savez_compressed(out, seed_w=seed_w, scale=scale, X=x, prob=prob, cls=cls, weight_abs_sum=wsum,
                                matmuls=67, tf_version=ver)
            print("wrote", out, "p1 range", prob[:, 1].min(), prob[:, 1].max(), "cls1", int(cls.sum()))
    with open(os.path.join(HERE, "index_tables.json"), "w") as fh:
        json.dump(index_tables, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["bilstm", "host"]
    if "bilstm" in what:
        make_bilstm()
    if "host" in what:
        from make_golden_host import make_host
        make_host()
