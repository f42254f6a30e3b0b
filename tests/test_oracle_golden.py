"""CPU: the oracle (C + numpy) against the golden vectors produced by interpreting the
reference's own serialized graph (tests/golden/make_golden.py)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from deepmod_amd import synth, tfbundle
from oracle import oracle_np

FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "bilstm_*.npz")))


def test_fixtures_present():
    assert len(FIXTURES) == 10  # 5 shipped BiLSTM models x 2 weight regimes


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[7:-4] for p in FIXTURES])
def test_oracle_matches_interpreted_reference_graph(path):
    g = np.load(path)
    w = synth.synthetic_weights(int(g["seed_w"]), float(g["scale"]))
    wsum = sum(np.abs(v.astype(np.float64)).sum() for v in w.values())
    assert abs(wsum - float(g["weight_abs_sum"])) < 1e-6 * wsum, "synthetic weight generator drifted"
    assert int(g["matmuls"]) == 67  # 11 live steps x 3 layers x 2 dirs + head
    prob_c, cls_c = oracle_np.predict_windows_c(w, g["X"])
    assert np.abs(prob_c - g["prob"]).max() <= 5e-6
    assert np.array_equal(cls_c, g["cls"])
    if path.endswith("conmodC_P100wd21_f7ne1u0_4_s4.npz"):
        prob_n, cls_n, _ = oracle_np.predict_windows_np(w, g["X"])
        assert np.abs(prob_n - g["prob"]).max() <= 1e-6
        assert np.array_equal(cls_n, g["cls"])


def test_oracle_matches_interpreted_reference_graph_on_trained_like_weights():
    """The same pin on weights with TRAINED statistics (tests/golden/make_trained_like.py: outlier kernel entries up to 1.7 beside a
    median of 0.05, learnt biases): the reference's serialized graph, evaluated by tools/graphdef_interp.py, on 64 planted + 64 config-2
    windows."""
    from conftest import trained_like_weights
    w = trained_like_weights()
    assert set(w) == set(n for n, _ in synth.variable_shapes())
    for n, shape in synth.variable_shapes():
        assert w[n].shape == shape
    g = np.load(os.path.join(GOLDEN, "trained_like_case.npz"))
    prob_c, cls_c = oracle_np.predict_windows_c(w, g["X"])
    assert np.abs(prob_c - g["prob"]).max() <= 5e-6
    assert np.array_equal(cls_c, g["cls"])
    prob_n, cls_n, _ = oracle_np.predict_windows_np(w, g["X"])
    assert np.abs(prob_n - g["prob"]).max() <= 1e-6
    assert 0.02 < g["cls"].mean() < 0.98           # both classes occur


def test_oracle_matches_interpreted_reference_graph_on_read_shaped_windows():
    """Round 5: the pin on READ-SHAPED inputs (tests/golden/make_golden_tail.py): 192 windows of synthetic reads whose tail events carry
    normalised means on the +-5 clip and lengths up to 27,000 samples, evaluated by the reference's serialized graph on the trained-like
    weights - the kind of input on which the split-f16 kernels had a 2e-5 error no synthetic-window fixture showed (profiles/HISTORY.md 4.1')."""
    from conftest import trained_like_weights
    w = trained_like_weights()
    g = np.load(os.path.join(GOLDEN, "trained_like_tail_case.npz"))
    X = g["X"]
    assert X.shape == (192, 21, 7) and X[:, :, 6].max() > 20000 and (np.abs(X[:, :, 4]) == 5.0).any()
    prob_c, cls_c = oracle_np.predict_windows_c(w, X)
    assert np.abs(prob_c - g["prob"]).max() <= 5e-6
    assert np.array_equal(cls_c, g["cls"])
    prob_n, cls_n, _ = oracle_np.predict_windows_np(w, X)
    assert np.abs(prob_n - g["prob"]).max() <= 1e-6


def test_tensorflow_cpu_activation_kernels_move_no_probability_by_1e_5():
    """The part of the reference's arithmetic no fixture pins is inside TensorFlow's kernels.  Its largest piece, restated from the published
    algorithm: on CPU TensorFlow 1.x computes tanh and sigmoid with Eigen's float32 rational approximations (`generic_fast_tanh_float`,
    `scalar_logistic_op<float>`; oracle_np.eigen_fast_tanh / eigen_logistic), not libm.  They are within 4e-7 of libm, and the whole graph evaluated
    with them stays within 5e-6 of the interpreted reference graph on every fixture (classes equal) and within 1e-5 of the libm oracle on 4,000 windows
    at weight scale 4 - a twentieth of the path's tolerance (1e-4)."""
    from conftest import trained_like_weights
    x = np.linspace(-20.0, 20.0, 400001).astype(np.float32)
    assert np.abs(oracle_np.eigen_fast_tanh(x) - np.tanh(x.astype(np.float64))).max() < 4e-7
    assert np.abs(oracle_np.eigen_logistic(x) - 1.0 / (1.0 + np.exp(-x.astype(np.float64)))).max() < 4e-7
    assert oracle_np.eigen_fast_tanh(np.float32([-50, 50, 0])).tolist() == [-1.0, 1.0, 0.0]
    worst = 0.0
    for path in FIXTURES:
        g = np.load(path)
        w = synth.synthetic_weights(int(g["seed_w"]), float(g["scale"]))
        prob, cls, _ = oracle_np.predict_windows_np(w, g["X"], activations='eigen')
        worst = max(worst, float(np.abs(prob - g["prob"]).max()))
        assert np.array_equal(cls, g["cls"])
    w = trained_like_weights()
    for name in ("trained_like_case.npz", "trained_like_tail_case.npz"):
        g = np.load(os.path.join(GOLDEN, name))
        prob, cls, _ = oracle_np.predict_windows_np(w, g["X"], activations='eigen')
        worst = max(worst, float(np.abs(prob - g["prob"]).max()))
        assert np.array_equal(cls, g["cls"])
    assert worst <= 5e-6, worst
    w = synth.synthetic_weights(26, 4.0)
    xw = synth.synthetic_windows(4000, seed=5)
    pe, ce, _ = oracle_np.predict_windows_np(w, xw, activations='eigen')
    pl, cl, _ = oracle_np.predict_windows_np(w, xw)
    assert np.abs(pe - pl).max() <= 1e-5
    flips = ce != cl
    assert (np.abs(pl[flips, 1] - 0.5) <= 1e-5).all()
    with pytest.raises(ValueError):
        oracle_np.predict_windows_np(w, xw[:2], dtype=np.float64, activations='eigen')


def test_torch_restatement_equals_c_oracle():
    """oracle/oracle_torch.py (bench.py's cpu_baseline.gemm leg) is the same graph as the C oracle."""
    from oracle import oracle_torch
    w = synth.synthetic_weights(26, 4.0)
    x = synth.synthetic_windows(300, seed=3)
    p = oracle_torch.TorchGraph(w, 2).predict(x)
    ref, _ = oracle_np.predict_windows_c(w, x)
    assert np.abs(p - ref).max() <= 1e-5


def test_oracle_thread_count_invariant():
    w = synth.synthetic_weights(3, 1.0)
    x = synth.synthetic_windows(97, seed=9)
    p1, c1 = oracle_np.predict_windows_c(w, x, nthreads=1)
    p4, c4 = oracle_np.predict_windows_c(w, x, nthreads=4)
    assert np.array_equal(p1, p4) and np.array_equal(c1, c4)


def test_oracle_empty():
    w = synth.synthetic_weights(3, 1.0)
    p, c = oracle_np.predict_windows_c(w, np.zeros((0, 21, 7), np.float32))
    assert p.shape == (0, 2) and c.shape == (0,)


def test_index_tables_match_synthetic_layout():
    """The real checkpoints' .index tables (recorded as data) agree with the layout the synthetic
    checkpoint writer reproduces."""
    tables = json.load(open(os.path.join(GOLDEN, "index_tables.json")))
    assert len(tables) == 5
    for model, t in tables.items():
        ent = t["entries"]
        for name, off in synth.REAL_LAYOUT.items():
            assert ent[name]["offset"] == off, (model, name)
        for name, shape in synth.variable_shapes():
            assert tuple(ent[name]["shape"]) == shape
        assert max(e["offset"] + e["size"] for e in ent.values()) == synth.REAL_DATA_SIZE


def test_bundle_roundtrip_real_layout(tmp_path):
    prefix = str(tmp_path / "mod_train_synth")
    w = synth.write_synthetic_checkpoint(prefix, seed=5, scale=1.0)
    assert os.path.getsize(tfbundle.data_path(prefix)) == synth.REAL_DATA_SIZE
    back = tfbundle.load_bundle(prefix, verify_crc=True)
    assert set(back) == set(w)
    for k in w:
        assert np.array_equal(back[k], w[k])
    assert tfbundle.latest_checkpoint(str(tmp_path)) == prefix


@pytest.mark.skipif(not os.path.isdir("/root/reference/train_deepmod"), reason="reference tree not present")
def test_reader_on_real_index_files():
    import glob as g
    paths = sorted(g.glob("/root/reference/train_deepmod/rnn_*/*.index"))
    assert len(paths) == 5
    tables = json.load(open(os.path.join(GOLDEN, "index_tables.json")))
    for p in paths:
        ent = tfbundle.read_index(p)
        rec = tables[os.path.basename(os.path.dirname(p))]["entries"]
        assert {n: (list(e.shape), e.offset, e.size, e.crc32c) for n, e in ent.items()} == \
               {n: (v["shape"], v["offset"], v["size"], v["crc32c"]) for n, v in rec.items()}
    # the one complete checkpoint in the tree (cluster MLP): CRCs verify
    t = tfbundle.load_bundle("/root/reference/train_deepmod/na12878_cluster_train_mod-keep_prob0.7-nb25-chr1/Cg.cov5.nb25",
                             verify_crc=True)
    assert t["W_1"].shape == (14, 100)
