"""Site-level detection performance from BED files: the second half of the headline metric.

The reference evaluates a run per genomic site: every BED line carries coverage and methylation percentage, a site has a truth
label (fully methylated / unmethylated control sample), and `roc_curve(label, pct)` is drawn over the sites with
`Coverage >= k` for k in (1, 5) (DeepMod_tools/cal_EcoliDetPerf.py:255-276: `cov_plot_thr = [1, 5]`, score =
`Methylation_Percentage`).  This module computes those numbers from the BED bytes this package writes; no plotting.
"""
from __future__ import annotations

from typing import Dict, Iterable, Tuple

import numpy as np


def bed_sites(bed: bytes) -> Dict[str, np.ndarray]:
    """Columns of a mod_pos.*.bed file (myDetect.py:1112-1120): position (col 2), coverage (col 10), percentage (col 11),
    methylated reads (col 12)."""
    rows = [ln.split() for ln in bed.decode("ascii").splitlines() if ln.strip()]
    col = lambda i: np.array([int(r[i]) for r in rows], np.int64)
    if not rows:
        z = np.zeros(0, np.int64)
        return {"pos": z, "cov": z.copy(), "pct": z.copy(), "mod": z.copy()}
    return {"pos": col(1), "cov": col(9), "pct": col(10), "mod": col(11)}


def roc_auc(label: np.ndarray, score: np.ndarray) -> float:
    """Area under the ROC curve = auc(roc_curve(label, score)) (the Mann-Whitney statistic with midranks); nan if one class is empty."""
    label = np.asarray(label).astype(bool)
    score = np.asarray(score, np.float64)
    npos, nneg = int(label.sum()), int((~label).sum())
    if npos == 0 or nneg == 0:
        return float("nan")
    order = np.argsort(score, kind="mergesort")
    s = score[order]
    lo = np.searchsorted(s, s, "left")
    hi = np.searchsorted(s, s, "right")
    ranks = np.empty(len(s))
    ranks[order] = 0.5 * (lo + hi + 1)
    return float((ranks[label].sum() - npos * (npos + 1) / 2.0) / (npos * nneg))


def site_level_auc(bed: bytes, methylated_positions: Iterable[int], cov_thresholds: Tuple[int, ...] = (1, 5)) -> Dict[int, float]:
    """{k: AUC of the per-site methylation percentage against the truth label over the sites with coverage >= k}.
    methylated_positions: the positions (same strand as the BED) whose truth label is 1; every other site of the BED is 0."""
    sites = bed_sites(bed)
    truth = np.isin(sites["pos"], np.fromiter(methylated_positions, np.int64))
    out = {}
    for k in cov_thresholds:
        sel = sites["cov"] >= k
        out[int(k)] = roc_auc(truth[sel], sites["pct"][sel])
    return out
