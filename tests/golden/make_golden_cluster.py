"""Golden vectors for the CpG-cluster second stage, produced by running the
REFERENCE's own script /root/reference/DeepMod_tools/hm_cluster_predict.py here, end to end, with a
stub `tensorflow` whose Session.run evaluates the reference's serialized graph
(train_deepmod/na12878_cluster_train_mod-keep_prob0.7-nb25-chr1/Cg.cov5.nb25.meta) with the REAL
checkpoint weights through tools/graphdef_interp.py.

Outputs: tests/golden/cluster_case.json {pred BED text, motif BED text, expected *_clusterCpG BED text}
         tests/golden/cluster_case.npz  {X features fed to sess.run, MLP outputs returned}
Run only here:  python tests/golden/make_golden_cluster.py
"""
import io
import json
import os
import runpy
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from graphdef_interp import GraphRunner, load_meta  # noqa: E402
from deepmod_amd import tfbundle  # noqa: E402

MODEL = "/root/reference/train_deepmod/na12878_cluster_train_mod-keep_prob0.7-nb25-chr1/Cg.cov5.nb25"


def synth_inputs(rng, chrom="chr1", length=6000):
    """CpG motif table + a pred BED in the dialect hm_cluster_predict.py parses (lsp[1]=pos, [5]=strand,
    [9]=cov, [10]=pct, [11]=mod count)."""
    seq = rng.choice(list("ACGT"), length, p=[.2, .3, .3, .2])
    motif, bed = [], []
    for p in range(length - 1):
        if seq[p] == "C" and seq[p + 1] == "G":
            motif.append("%s %d + C\n" % (chrom, p))
            motif.append("%s %d - C\n" % (chrom, p + 1))
    for line in motif:
        c, p, s, _ = line.split()
        if rng.random() < 0.85:
            cov = int(rng.integers(0, 40))
            mod = int(rng.integers(0, cov + 1)) if cov else 0
            pct = int(100 * mod / cov) if cov else 0
            bed.append("%s %s %d C %d %s %s %d 0,0,0 %d %d %d\n" % (c, p, int(p) + 1, min(cov, 1000), s, p, int(p) + 1, cov, pct, mod))
    # a non-CpG C row (must be ignored) and rows sorted by position as DeepMod writes them
    bed.append("%s %d %d C 9 + %d %d 0,0,0 9 55 5\n" % (chrom, length + 10, length + 11, length + 10, length + 11))
    return "".join(motif), "".join(bed)


def main():
    rng = np.random.default_rng(2026)
    tmp = tempfile.mkdtemp()
    motif_txt, bed_txt = synth_inputs(rng)
    os.makedirs(tmp + "/motif")
    open(tmp + "/motif/motif_chr1_C.bed", "w").write(motif_txt)
    open(tmp + "/pred.chr1.C.bed", "w").write(bed_txt)

    nodes, _ = load_meta(MODEL + ".meta")
    weights = tfbundle.load_bundle(MODEL, verify_crc=True)
    runner = GraphRunner(nodes, weights)
    seen = {"X": [], "out": []}

    class FakeTensor:
        def __init__(self, name): self.name = name
    class FakeGraph:
        def get_tensor_by_name(self, name): return FakeTensor(name)
    class FakeSession:
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def run(self, fetches, feed_dict=None):
            feeds = {k.name.split(":")[0]: np.asarray(v, np.float32) for k, v in feed_dict.items()}
            assert float(feeds["keep_prob"]) == 1.0
            out = runner.run(["output:0"], {"X": feeds["X"], "keep_prob": feeds["keep_prob"]})[0]
            seen["X"].append(feeds["X"]); seen["out"].append(out)
            return [out]
    class FakeSaver:
        def restore(self, sess, path): return None
    tf = types.ModuleType("tensorflow")
    tf.train = types.SimpleNamespace(import_meta_graph=lambda p: FakeSaver(), latest_checkpoint=lambda d: MODEL)
    tf.Session = FakeSession
    tf.get_default_graph = lambda: FakeGraph()
    fake_locale = types.ModuleType("locale")
    fake_locale.LC_ALL = 0
    fake_locale.setlocale = lambda *a, **k: None
    sys.modules.update({"tensorflow": tf, "locale": fake_locale})
    argv, stdout = sys.argv, sys.stdout
    sys.argv = ["hm_cluster_predict.py", tmp + "/pred", tmp + "/motif"]
    sys.stdout = io.StringIO()
    try:
        runpy.run_path("/root/reference/DeepMod_tools/hm_cluster_predict.py", run_name="__main__")
    finally:
        sys.argv, sys.stdout = argv, stdout
    out_txt = open(tmp + "/pred_clusterCpG.chr1.C.bed").read()
    X = np.concatenate(seen["X"]); out = np.concatenate(seen["out"]).ravel()
    json.dump({"chrom": "chr1", "motif": motif_txt, "pred_bed": bed_txt, "expected": out_txt},
              open(os.path.join(HERE, "cluster_case.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "cluster_case.npz"), X=X.astype(np.float64), output=out.astype(np.float32))
    print("sites", len(out_txt.splitlines()), "X", X.shape, "out range", out.min(), out.max())
    print(out_txt.splitlines()[0])


if __name__ == "__main__":
    main()
