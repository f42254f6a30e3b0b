"""Generate `trained_like_weights.npz` + `trained_like_case.npz`: weights with the statistics of a TRAINED BiLSTM.

Runs ONLY in the build container (torch CPU autograd for the training, /root/reference for the pin); nothing here ships with or is
imported by the product - like tools/graphdef_interp.py it is a fixture generator.  Why: all five BiLSTM `.data` shards of the
reference are absent (/root/reference/.MISSING_LARGE_BLOBS:1-5), so every other parity fixture uses U(-a, a) kernels.  Trained LSTMs
have saturated gates, outlier rows and learned forget biases - the statistics the split-f16 kernels' error bounds depend on.

What it does
  1. builds the exact architecture of /root/reference/bin/DeepMod_scripts/myMultiBiRNN.py:21-61 in torch (two independent 3 x 100
     unidirectional stacks, BasicLSTMCell gate order i, j, f, o, forget_bias 1.0 added at run time, zero state, head on the centre
     output = 11 live steps per direction), initialised like TF1 does (glorot-uniform kernels, zero biases, truncated-normal head,
     `:36-37`), trained like `:71-77` (softmax cross entropy, Adam 1e-3);
  2. trains it for a few minutes on synthetic windows with a planted per-5-mer modification signal (below);
  3. writes the variables under the names / shapes of the real `.index` files (tests/golden/index_tables.json) as a small npz;
  4. pins a 64-window case on them to the reference's own serialized graph through tools/graphdef_interp.py (as make_golden.py does);
  5. prints the weight-magnitude statistics next to the synthetic scales 1 / 4 / 16 (-> trained_like_stats.json).

    python tests/golden/make_trained_like.py [minutes=6]
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

NFEAT, HID, WIN, LIVE = 7, 100, 21, 11


def planted_windows(n, rng, tables):
    """Windows in the feature layout of myDetect.py:894-900 with a learnable modification signal: the event mean of a position follows
    its 5-mer (a fixed table), and a MODIFIED centre C shifts the means (and widens the spread) of the five positions whose 5-mer contains
    it by a per-5-mer amount.  Label 1 = centre is a modified C; centres other than C are never modified."""
    mu, delta = tables
    seq = rng.integers(0, 4, (n, WIN + 4))                      # two bases of context on both sides
    centre = 2 + WIN // 2
    is_c = seq[:, centre] == 1
    mod = is_c & (rng.random(n) < 0.5)
    kmer = np.zeros((n, WIN), np.int64)
    for j in range(5):
        kmer = kmer * 4 + seq[:, j:j + WIN]
    mean = mu[kmer] + rng.normal(0.0, 0.3, (n, WIN))
    stdv = np.abs(rng.normal(0.25, 0.15, (n, WIN)))
    lo, hi = WIN // 2 - 2, WIN // 2 + 3
    mean[mod, lo:hi] += delta[kmer[mod, lo:hi]]
    stdv[mod, lo:hi] *= 1.3
    x = np.zeros((n, WIN, NFEAT), np.float32)
    none = rng.random((n, WIN)) < 0.04                          # rows without a reference base (insertions), as in synth.synthetic_windows
    none[:, WIN // 2] = False
    for b in range(4):
        x[:, :, b] = (seq[:, 2:2 + WIN] == b) & ~none
    x[:, :, 4] = np.round(np.clip(mean, -5, 5), 3)
    x[:, :, 5] = np.round(stdv, 3)
    x[:, :, 6] = rng.geometric(0.12, (n, WIN))
    return x, mod.astype(np.int64)


def main(minutes: float):
    import torch
    torch.manual_seed(20260928)
    torch.set_num_threads(os.cpu_count() or 8)
    rng = np.random.default_rng(20260928)
    tables = (rng.normal(0.0, 1.2, 4 ** 5), rng.normal(0.0, 0.8, 4 ** 5))

    def glorot(shape):
        a = float(np.sqrt(6.0 / (shape[0] + shape[1])))
        return torch.nn.Parameter(torch.empty(shape).uniform_(-a, a))

    P = {}
    for d in ("fw", "bw"):
        for l in range(3):
            kin = NFEAT if l == 0 else HID
            P["%s%d_k" % (d, l)] = glorot((kin + HID, 4 * HID))
            P["%s%d_b" % (d, l)] = torch.nn.Parameter(torch.zeros(4 * HID))
    P["head_w"] = torch.nn.Parameter(torch.nn.init.trunc_normal_(torch.empty(2 * HID, 2), std=1.0, a=-2.0, b=2.0))
    P["head_b"] = torch.nn.Parameter(torch.nn.init.trunc_normal_(torch.empty(2), std=1.0, a=-2.0, b=2.0))

    def forward(x):
        finals = []
        for d in ("fw", "bw"):
            h = [x.new_zeros(x.shape[0], HID) for _ in range(3)]
            c = [x.new_zeros(x.shape[0], HID) for _ in range(3)]
            for s in range(LIVE):
                inp = x[:, s if d == "fw" else WIN - 1 - s, :]
                for l in range(3):
                    g = torch.cat([inp, h[l]], 1) @ P["%s%d_k" % (d, l)] + P["%s%d_b" % (d, l)]
                    gi, gj, gf, go = g.split(HID, 1)
                    c[l] = c[l] * torch.sigmoid(gf + 1.0) + torch.sigmoid(gi) * torch.tanh(gj)
                    h[l] = torch.tanh(c[l]) * torch.sigmoid(go)
                    inp = h[l]
            finals.append(h[2])
        return torch.cat(finals, 1) @ P["head_w"] + P["head_b"]

    opt = torch.optim.Adam(P.values(), lr=1e-3)
    t0, it = time.time(), 0
    while time.time() - t0 < minutes * 60:
        x, y = planted_windows(512, rng, tables)
        loss = torch.nn.functional.cross_entropy(forward(torch.from_numpy(x)), torch.from_numpy(y))
        opt.zero_grad()
        loss.backward()
        opt.step()
        it += 1
        if it % 50 == 0:
            print("iter %d  %.0f s  loss %.4f" % (it, time.time() - t0, float(loss)), flush=True)

    from deepmod_amd import synth
    from oracle import oracle_np
    W = {synth.HEAD_W: P["head_w"].detach().numpy().astype(np.float32), synth.HEAD_B: P["head_b"].detach().numpy().astype(np.float32)}
    for d in ("fw", "bw"):
        for l in range(3):
            W[synth.cell_name(d, l, "kernel")] = P["%s%d_k" % (d, l)].detach().numpy().astype(np.float32)
            W[synth.cell_name(d, l, "bias")] = P["%s%d_b" % (d, l)].detach().numpy().astype(np.float32)
    tab = json.load(open(os.path.join(HERE, "index_tables.json")))["rnn_conmodC_P100wd21_f7ne1u0_4"]["entries"]
    for name, arr in W.items():
        assert list(arr.shape) == tab[name]["shape"], (name, arr.shape, tab[name]["shape"])
    assert set(W) == set(n for n in tab if "Adam" not in n and "power" not in n), sorted(set(tab) ^ set(W))

    # held-out quality of the planted task (the point is the weight statistics, not the task - but the net should have learnt it)
    xt, yt = planted_windows(20000, rng, tables)
    pt, ct = oracle_np.predict_windows_c(W, xt)
    from sklearn.metrics import roc_auc_score
    cm = xt[:, WIN // 2, 1] == 1
    auc = float(roc_auc_score(yt[cm], pt[cm, 1]))
    acc = float((ct == yt).mean())
    # the torch graph is the oracle's graph
    with torch.no_grad():
        pt_torch = torch.softmax(forward(torch.from_numpy(xt[:512])), 1).numpy()
    assert np.abs(pt_torch - pt[:512]).max() < 2e-5, np.abs(pt_torch - pt[:512]).max()

    np.savez_compressed(os.path.join(HERE, "trained_like_weights.npz"), **{k.replace("/", "|"): v for k, v in W.items()})
    # pin: the reference's serialized graph on these weights (64 planted windows + 64 config-2 windows)
    from graphdef_interp import load_meta, GraphRunner
    meta = "/root/reference/train_deepmod/rnn_conmodC_P100wd21_f7ne1u0_4/mod_train_conmodC_P100wd21_f3ne1u0.meta"
    if not os.path.exists(meta):
        import glob
        meta = sorted(glob.glob("/root/reference/train_deepmod/rnn_conmodC_P100wd21_f7ne1u0_4/*.meta"))[0]
    nodes, ver = load_meta(meta)
    X = np.concatenate([xt[:64], synth.synthetic_windows(64, seed=4242)])
    gr = GraphRunner(nodes, W)
    prob, cls = gr.run(["Softmax:0", "ArgMax:0"], {"Placeholder": X})
    assert gr.op_counts.get("MatMul") == 67
    np.savez_compressed(os.path.join(HERE, "trained_like_case.npz"), X=X, prob=prob, cls=cls, tf_version=ver, iters=it, auc=auc, acc=acc)

    def stats(w):
        ks = np.concatenate([np.abs(v).ravel() for n, v in w.items() if n.endswith("kernel")])
        fb = np.concatenate([v[2 * HID:3 * HID] for n, v in w.items() if n.endswith("bias")])
        rows = np.concatenate([np.abs(v).sum(0) for n, v in w.items() if n.endswith("kernel")])         # l1 norm of a gate column = worst-case pre-activation
        return {"kernel_abs_max": float(ks.max()), "kernel_abs_p999": float(np.quantile(ks, 0.999)), "kernel_abs_median": float(np.median(ks)),
                "kernel_rms": float(np.sqrt((ks ** 2).mean())), "gate_column_l1_max": float(rows.max()), "gate_column_l1_median": float(np.median(rows)),
                "forget_bias_mean": float(fb.mean()), "forget_bias_max": float(fb.max()), "bias_abs_max": float(max(np.abs(v).max() for n, v in w.items() if n.endswith("bias")))}
    # gate saturation on config-2 windows: fraction of sigmoid gates within 1e-3 of 0 or 1 over all cells of the graph
    out = {"iters": it, "minutes": minutes, "held_out_auc_on_C_centres": auc, "held_out_accuracy": acc, "trained_like": stats(W)}
    for sc in (1.0, 4.0, 16.0):
        out["synthetic_scale_%g" % sc] = stats(synth.synthetic_weights(17, sc))
    with open(os.path.join(HERE, "trained_like_stats.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 6.0)
