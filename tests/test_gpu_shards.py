"""GPU (-m gpu): the single-GPU shards of BASELINE.json configs[3] and configs[4] - what one rank of the 8-GPU runs holds
and does (the 8-GPU runs themselves are the driver's): human-chromosome-sized counters with a reduce over the persistent
communicator, and the CpG-cluster second stage at >= 1e6 sites."""
import numpy as np
import pytest

from deepmod_amd import cluster, comm, summary
from oracle import cluster_oracle, oracle_np

pytestmark = pytest.mark.gpu

CHR1 = 248_956_422          # GRCh38 chr1


def test_config4_chr1_sized_counters_one_rank(gpu_device, tmp_path):
    """3 x 248,956,422 int32 counters (2.99 GB) for one contig x strand, 1e6 classified bases of ~8 kb reads scattered
    over the chromosome, the communicator reduce, fetch and BED - against the oracle's accumulation."""
    rng = np.random.default_rng(11)
    s = summary.PositionSummary(CHR1, gpu_device)
    starts = rng.integers(0, CHR1 - 9000, 125)
    pos = np.concatenate([st + np.arange(8000) for st in starts]).astype(np.int64)          # 1,000,000 rows
    pos[::97] = pos[1::97][:len(pos[::97])]                                                  # a few repeated positions inside a wave
    flags = (rng.random(len(pos)) < 0.25).astype(np.uint8) | ((rng.random(len(pos)) < 0.97).astype(np.uint8) << 1)
    cls = (rng.random(len(pos)) < 0.5).astype(np.uint8)
    for lo in range(0, len(pos), 65536):                                                     # classifier-batch sized adds
        s.add_classified(pos[lo:lo + 65536], flags[lo:lo + 65536], cls[lo:lo + 65536])
    rdv = comm.FileRendezvous(str(tmp_path / 'rdv'), 0, 1)
    c = comm.Communicator.from_rendezvous(gpu_device, rdv)
    s.reduce(c, 0)
    assert c.stats()["bytes"] == 12 * CHR1
    c.close()
    touch, cov, mod = s.fetch()
    s.close()
    # oracle on the compacted position set (the counters are zero everywhere else)
    upos, inv = np.unique(pos, return_inverse=True)
    want = [np.zeros(len(upos), np.int32) for _ in range(3)]
    oracle_np.summary_add_c(want[0], want[1], want[2], inv.astype(np.int64), (flags | (cls << 2)).astype(np.uint8))
    for got, w in zip((touch, cov, mod), want):
        assert np.array_equal(got[upos], w)
        assert int(got.sum(dtype=np.int64)) == int(w.sum(dtype=np.int64))
    bed = summary.bed_lines("chr1", "+", "C", touch, cov, mod)
    assert bed.count(b"\n") == int((want[0] > 0).sum()) > 200000
    first = int(upos[want[0] > 0][0])
    assert bed.startswith(("chr1 %d %d C " % (first, first + 1)).encode())


def _synthetic_cpg_sites(n_sites, seed):
    """CpG positions (C on '+', its partner C on '-' one base further) with coverage-derived fractions."""
    rng = np.random.default_rng(seed)
    gaps = rng.geometric(0.08, n_sites).astype(np.int64) + 1          # mean spacing ~13 bp: several neighbours within +-25
    plus = np.cumsum(gaps)
    minus = plus + 1
    keep_p, keep_m = rng.random(n_sites) < 0.93, rng.random(n_sites) < 0.93
    frac = lambda k: np.round(rng.integers(0, 101, k) / 100.0, 3)
    return {"+": (plus[keep_p], frac(int(keep_p.sum())), None), "-": (minus[keep_m], frac(int(keep_m.sum())), None)}


def test_config5_cluster_stage_million_sites(gpu_device):
    """hm_cluster_predict's feature construction and MLP on 1.1e6 CpG sites: features against the loop-level oracle on a
    window of the chromosome, the GPU MLP against the numpy graph on every row."""
    pred = _synthetic_cpg_sites(600_000, seed=7)
    n_rows = len(pred["+"][0]) + len(pred["-"][0])
    assert n_rows >= 1_000_000
    x, _ = cluster.cluster_features({s: (pred[s][0], pred[s][1], [""] * len(pred[s][0])) for s in "+-"})
    assert x.shape == (n_rows, 14) and (x[:, 2] > 0).mean() > 0.9
    # oracle features for the sites of a 40 kb window (neighbourhoods of +-25 bp stay inside a 100 bp margin)
    lo, hi = 2_000_000, 2_040_000
    motif_txt, pred_txt = [], []
    for s in "+-":
        p, f, _ = pred[s]
        sel = (p >= lo - 100) & (p < hi + 100)
        for q, fr in zip(p[sel].tolist(), f[sel].tolist()):
            motif_txt.append("chr1 %d %s" % (q, s))
            pct = int(round(fr * 100))
            pred_txt.append("chr1 %d %d C 10 %s %d %d 0,0,0 10 %d %d" % (q, q + 1, s, q, q + 1, pct, pct // 10))
    ox, olines = cluster_oracle.features_loop("\n".join(motif_txt), "\n".join(pred_txt), "chr1")
    opos = np.array([int(l.split()[1]) for l in olines])
    ostrand = np.array([l.split()[5] for l in olines])
    off = 0
    checked = 0
    for s in "+-":
        p = pred[s][0]
        inside = np.flatnonzero((p >= lo) & (p < hi))
        sel = (ostrand == s) & (opos >= lo) & (opos < hi)
        assert np.array_equal(opos[sel], p[inside])
        assert np.array_equal(ox[sel], x[off + inside])
        checked += len(inside)
        off += len(p)
    assert checked > 5000
    w = {"W_1": np.random.default_rng(5).normal(0, 0.5, (14, 100)).astype(np.float32),
         "b_1": np.random.default_rng(6).normal(0, 0.3, 100).astype(np.float32),
         "W_2": np.random.default_rng(7).normal(0, 0.3, (100, 20)).astype(np.float32),
         "b_2": np.random.default_rng(8).normal(0, 0.3, 20).astype(np.float32),
         "W_O": np.random.default_rng(9).normal(0, 0.5, (20, 1)).astype(np.float32),
         "b_O": np.random.default_rng(10).normal(0, 0.3, 1).astype(np.float32)}
    m = cluster.ClusterModel(w, gpu_device)
    got = m.predict(x)
    m.close()
    assert np.abs(got - cluster_oracle.mlp_np(w, x)).max() <= 2e-6
