"""ORACLE — TEST INFRASTRUCTURE ONLY.  Loop-level restatement of the CpG-cluster second stage,
/root/reference/DeepMod_tools/hm_cluster_predict.py (features :128-154, MLP graph :94-103/:161,
output line :170), pinned by tests/golden/cluster_case.{json,npz}, which were produced by running
that script itself with the REAL checkpoint weights (tests/golden/make_golden_cluster.py)."""
from __future__ import annotations

import numpy as np

NBSIZE = 25


def mlp_np(w, x):
    """numpy fp32 restatement of the graph: relu(X W_1 + b_1) -> relu(. W_2 + b_2) -> sigmoid(. W_O + b_O)."""
    x = np.asarray(x, np.float32)
    h1 = np.maximum((x @ w["W_1"]).astype(np.float32) + w["b_1"], 0).astype(np.float32)
    h2 = np.maximum((h1 @ w["W_2"]).astype(np.float32) + w["b_2"], 0).astype(np.float32)
    o = (h2 @ w["W_O"]).astype(np.float32) + w["b_O"]
    return (np.float32(1) / (np.float32(1) + np.exp(-o))).astype(np.float32).ravel()


def features_loop(motif_txt: str, pred_txt: str, chrom: str):
    cg = {}
    for line in motif_txt.splitlines():
        lsp = line.split()
        cg[(lsp[0], lsp[2], int(lsp[1]))] = True                                  # :123
    pred = {}
    for line in pred_txt.splitlines():
        line = line.strip()
        if not line:
            continue
        lsp = line.split()
        key = (lsp[0], lsp[5], int(lsp[1]))
        if key not in cg or lsp[0] != chrom or int(lsp[9]) == 0:                   # :56-63
            continue
        pred[key] = [int(lsp[9]), round(int(lsp[10]) / 100.0, 3), int(lsp[11]), line]
    keys = sorted(pred)                                                            # :129
    rows = []
    for k in keys:
        partner = (k[0], '-' if k[1] == '+' else '+', k[2] + 1 if k[1] == '+' else k[2] - 1)
        x = [pred[k][1], pred[partner][1] if partner in pred else 0] + [0] * 12    # :135-138
        for rpos in range(k[2] - NBSIZE, k[2] + NBSIZE + 1):                       # :140
            if rpos in [k[2], partner[2]]:
                continue
            for s in '+-':                                                         # :143-150 (if / elif)
                kk = (k[0], s, rpos)
                if kk in cg and kk in pred:
                    x[int(pred[kk][1] / 0.1 + 0.5) + 3] += 1
                    x[2] += 1
                    break
        for i in range(3, 14):
            if x[2] > 0:
                x[i] = round(x[i] / float(x[2]), 3)                                # :151-152
        rows.append(x)
    return np.array(rows, dtype=np.float64), [pred[k][3] for k in keys]
