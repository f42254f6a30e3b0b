"""CpG-cluster second stage: counterpart of
/root/reference/DeepMod_tools/hm_cluster_predict.py.

For every CpG-motif C with coverage in a DeepMod BED: 14 features
  [own methylation fraction, partner-strand fraction (0 if absent), #neighbours,
   11-bin histogram of the neighbours' fractions (0.1 bins) normalised by #neighbours]
over the CpG sites within +-25 bp that are present in the BED (hm_cluster_predict.py:128-154), then the
MLP 14->100->20->1 on the GPU (dm_cluster_predict) and the original BED line with int(p*100) appended (:170).
Feature extraction is vectorised numpy on the host (searchsorted and per-bin prefix counts instead of dict probes).
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np

from . import _lib, tfbundle

NBSIZE = 25          # hm_cluster_predict.py:83
BATCH_SIZE = 4096    # hm_cluster_predict.py:16
WEIGHT_ORDER = ("W_1", "b_1", "W_2", "b_2", "W_O", "b_O")
CHRKEYS = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY", "chrM"]   # :86-91


def flatten_cluster_weights(tensors: Dict[str, np.ndarray]) -> np.ndarray:
    shapes = {"W_1": (14, 100), "b_1": (100,), "W_2": (100, 20), "b_2": (20,), "W_O": (20, 1), "b_O": (1,)}
    parts = []
    for name in WEIGHT_ORDER:
        a = np.asarray(tensors[name], np.float32)
        if a.shape != shapes[name]:
            raise ValueError("cluster tensor %s has shape %s, expected %s" % (name, a.shape, shapes[name]))
        parts.append(a.ravel())
    return np.ascontiguousarray(np.concatenate(parts))


class ClusterModel:
    def __init__(self, tensors: Dict[str, np.ndarray], device: int = 0):
        self._lib = _lib.load()
        flat = flatten_cluster_weights(tensors)
        self._h = self._lib.dm_cluster_create(device, flat.ctypes.data, flat.size)
        if not self._h:
            raise _lib.DeepModHipError("dm_cluster_create: " + _lib.last_error())

    @classmethod
    def from_checkpoint(cls, prefix: str, device: int = 0) -> "ClusterModel":
        return cls(tfbundle.load_bundle(prefix, names=WEIGHT_ORDER), device)

    def predict(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)     # the placeholder casts the float64 feed to fp32
        if x.ndim != 2 or x.shape[1] != 14:
            raise ValueError("expected [n,14] features, got %s" % (x.shape,))
        out = np.empty(x.shape[0], np.float32)
        _lib.check(self._lib.dm_cluster_predict(self._h, x.ctypes.data, x.shape[0], out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dm_cluster_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_motif(path: str) -> Dict[str, np.ndarray]:
    """motif_<chr>_C.bed rows `chr pos strand ...` -> sorted position arrays per strand (:117-123)."""
    pos = {"+": [], "-": []}
    with open(path) as fh:
        for line in fh:
            lsp = line.split()
            if len(lsp) >= 3:
                pos[lsp[2]].append(int(lsp[1]))
    return {s: np.unique(np.array(v, dtype=np.int64)) for s, v in pos.items()}


def read_pred(path: str, chrom: str, motif: Dict[str, np.ndarray]):
    """readpredmod (:43-72): keep rows of `chrom` that are CpG-motif sites with coverage > 0.
    Returns per strand: sorted positions, fraction round(pct/100, 3), and the stripped lines."""
    rows = {"+": {}, "-": {}}
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            lsp = line.split()
            c, p, s = lsp[0], int(lsp[1]), lsp[5]
            if s not in motif or c != chrom:
                continue
            i = np.searchsorted(motif[s], p)
            if i >= len(motif[s]) or motif[s][i] != p:
                continue
            cov, pct, mc = int(lsp[9]), int(lsp[10]), int(lsp[11])
            if cov == 0:
                continue
            if p not in rows[s]:
                rows[s][p] = [cov, round(pct / 100.0, 3), mc, line]
            else:                                           # duplicate row: counts add (:67-72)
                r = rows[s][p]
                r[0] += cov
                r[2] += mc
                if r[0] > 0:
                    r[1] = round(r[2] / float(r[0]), 3)
    out = {}
    for s in "+-":
        keys = sorted(rows[s])
        out[s] = (np.array(keys, dtype=np.int64), np.array([rows[s][k][1] for k in keys], dtype=np.float64),
                  [rows[s][k][3] for k in keys])
    return out


def cluster_features(pred) -> Tuple[np.ndarray, List[str]]:
    """14 features per site in the reference's key order (all '+' sites ascending, then all '-')."""
    pos_all = np.concatenate([pred["+"][0], pred["-"][0]])
    frac_all = np.concatenate([pred["+"][1], pred["-"][1]])
    order = np.argsort(pos_all, kind="stable")
    spos, sfrac = pos_all[order], frac_all[order]           # a position is a CpG C on at most one strand
    bins = (sfrac / 0.1 + 0.5).astype(np.int64)             # int(frac/0.1+0.5)  (:144)
    cum = np.zeros((11, len(spos) + 1), np.int64)
    for b in range(11):
        cum[b, 1:] = np.cumsum(bins == b)
    feats, lines = [], []
    for s in "+-":
        pos, frac, ln = pred[s]
        n = len(pos)
        x = np.zeros((n, 14))
        x[:, 0] = frac
        o = "-" if s == "+" else "+"
        ppos = pos + (1 if s == "+" else -1)
        opos, ofrac, _ = pred[o]
        if len(opos):
            j = np.clip(np.searchsorted(opos, ppos), 0, len(opos) - 1)
            hit = opos[j] == ppos
            x[hit, 1] = ofrac[j[hit]]
        # neighbourhood histogram without a per-site loop: per-bin prefix counts over the position-sorted sites give the
        # counts of any [pos - 25, pos + 25] range as a difference; the site itself and its partner C are taken out again
        lo = np.searchsorted(spos, pos - NBSIZE, side="left")
        hi = np.searchsorted(spos, pos + NBSIZE, side="right")
        cnt = cum[:, hi] - cum[:, lo]                                         # [11, n]
        for v in (pos, ppos):
            l, r = np.searchsorted(spos, v, side="left"), np.searchsorted(spos, v, side="right")
            cnt -= cum[:, r] - cum[:, l]
        nkeep = cnt.sum(axis=0)
        has = nkeep > 0
        x[:, 2] = nkeep
        x[has, 3:] = np.round(cnt[:, has].T / nkeep[has, None].astype(np.float64), 3)
        feats.append(x)
        lines.extend(ln)
    return np.concatenate(feats) if feats else np.zeros((0, 14)), lines


def hm_cluster_predict(pred_prefix: str, motif_folder: str, model_prefix: str, chrkeys=None, device: int = 0) -> List[str]:
    """Same file conventions as the reference script: reads `<pred_prefix>.<chr>.C.bed` and
    `<motif_folder>/motif_<chr>_C.bed`, writes `<pred_prefix>_clusterCpG.<chr>.C.bed`."""
    model = ClusterModel.from_checkpoint(model_prefix, device)
    written = []
    for chrom in (chrkeys or CHRKEYS):
        motif_path = "%s/motif_%s_C.bed" % (motif_folder, chrom)
        pred_path = "%s.%s.C.bed" % (pred_prefix, chrom)
        if not os.path.isfile(motif_path):
            print("Warning_motif!!! no file {}".format(motif_path))
            continue
        if not os.path.isfile(pred_path):
            print("Warning_pred!!! no file {}".format(pred_path))
            continue
        pred = read_pred(pred_path, chrom, read_motif(motif_path))
        x, lines = cluster_features(pred)
        if len(lines) == 0:
            continue
        p = model.predict(x)
        new_pct = (p * np.float32(100)).astype(np.int64)       # int(float32 p * 100)   (:170)
        out_path = "%s_clusterCpG.%s.C.bed" % (pred_prefix, chrom)
        with open(out_path, "w") as fh:
            for ln, v in zip(lines, new_pct.tolist()):
                fh.write("{} {}\n".format(ln, v))
        written.append(out_path)
    model.close()
    return written
