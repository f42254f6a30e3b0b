"""Golden vectors for the host-side logic, produced by running the REFERENCE's own Python
(/root/reference/bin/DeepMod_scripts/myDetect.py) in the build container with stub
`tensorflow` / `h5py` modules (neither is installed) and the numpy aliases it still uses.

Outputs (plain data):
  host_mpredict1.npz  inputs + outputs of myDetect.mPredict1 (:787-834) for several read sizes:
                      batch shapes seen by the session, pred_mod_num, final mod_pred column
  host_getfeature.npz inputs + outputs of myDetect.get_Feature (:839-903) for '+' and '-' reads
  host_sum_handler.json  per-read prediction tables -> exact BED bytes written by sum_handler (:1028-1120)

Run only here (needs /root/reference):  python tests/golden/make_golden.py host
"""
from __future__ import annotations

import json
import os
import queue
import sys
import tempfile
import types
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    np.int = int      # removed numpy aliases the reference still uses (myDetect.py:660, :752, :1022)
    np.float = float
    tf = types.ModuleType("tensorflow")
    tf.constant = lambda *a, **k: None
    contrib = types.ModuleType("tensorflow.contrib")
    rnn = types.ModuleType("tensorflow.contrib.rnn")
    contrib.rnn = rnn
    tf.contrib = contrib
    sys.modules.update({"tensorflow": tf, "tensorflow.contrib": contrib, "tensorflow.contrib.rnn": rnn,
                        "h5py": types.ModuleType("h5py")})
    sys.path.insert(0, "/root/reference/bin")
    from DeepMod_scripts import myDetect  # noqa
    return myDetect


EVENT_DTYPE = [("mean", "<f4"), ("stdv", "<f4"), ("start", np.uint64), ("length", np.uint64), ("model_state", "U5")]
BMI_DTYPE = [("refbase", "U1"), ("readbase", "U1"), ("refbasei", np.uint64), ("readbasei", np.uint64), ("mod_pred", int)]


def fake_rule(x):
    """Deterministic stand-in classifier: class 1 iff the centre row is a 'C' row with mean > 0."""
    x = np.asarray(x)
    return ((x[:, 10, 1] > 0.5) & (x[:, 10, 4] > 0.0)).astype(np.int64)


class FakeSession:
    def __init__(self):
        self.batches = []

    def run(self, fetches, feed_dict=None):
        if feed_dict is None:
            return None
        x = feed_dict["X"]
        assert feed_dict["Y"].shape == (len(x), 2)
        self.batches.append(tuple(np.asarray(x).shape))
        return [fake_rule(x)]


def synth_read(rng, n_aligned, start_clip, end_clip, strand="+", p_ins=0.03, p_del=0.03, ref_start=1000):
    """A synthetic aligned read in the reference's in-memory form (sequencing orientation)."""
    n_events = start_clip + n_aligned + end_clip
    bases = rng.choice(list("ACGT"), n_events)
    ev = np.zeros(n_events, dtype=EVENT_DTYPE)
    ev["mean"] = np.round(np.clip(rng.normal(0, 1.2, n_events), -5, 5), 3)
    ev["stdv"] = np.round(np.abs(rng.normal(0.25, 0.15, n_events)), 3)
    ev["length"] = rng.geometric(0.12, n_events)
    ev["start"] = np.cumsum(np.r_[0, ev["length"][:-1]])
    ev["model_state"] = ["NN" + b + "NN" for b in bases]
    rows = []
    refpos = ref_start
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for k in range(n_aligned):
        rb = bases[start_clip + k]
        interior = 0 < k < n_aligned - 1
        u = rng.random()
        if interior and u < p_ins:
            rows.append(("-", rb, refpos, k, 0))
            continue
        if interior and u < p_ins + p_del:
            rows.append((rng.choice(list("ACGT")), "-", refpos, k, 0))
            refpos += 1
        refb = rb if rng.random() < 0.92 else rng.choice(list("ACGT"))
        rows.append((refb, rb, refpos, k, 0))
        refpos += 1
    bmi = np.array(rows, dtype=BMI_DTYPE)
    if strand == "-":
        # reference orientation positions descend along the read (myDetect.py:661-666 flips the table)
        span = int(bmi["refbasei"].max())
        bmi["refbasei"] = (span + ref_start - bmi["refbasei"]).astype(np.uint64)
    return ev, bmi


def make_mpredict1(md):
    rng = np.random.default_rng(42)
    out = {}
    cases = [(100, 5, 7), (614, 0, 3), (615, 2, 0), (1023, 4, 4), (1024, 1, 9), (1535, 6, 2), (3000, 11, 13)]
    for ci, (n_al, sc, ec) in enumerate(cases):
        ev, bmi = synth_read(rng, n_al, sc, ec)
        nrow = len(ev) - ec + 100 - (sc - 100)
        mfeat = np.zeros((nrow, 10))
        body = slice(100 - sc, 100 - sc + len(ev))  # rows of real events
        cat = rng.choice(5, len(ev), p=[.24, .24, .24, .24, .04])
        for b in range(4):
            mfeat[body, 3 + b] = (cat == b)
        mfeat[body, 7] = ev["mean"]
        mfeat[body, 8] = ev["stdv"]
        mfeat[body, 9] = ev["length"]
        mfeat[:, 0] = np.arange(nrow)
        sess = FakeSession()
        sp_options = {"rnn": (sess, "X", "Y", "init_l", "mfpred")}
        sp_param = {"f5data": {"r": (None, ev, None, "f.fast5")}}
        bmi_in = bmi.copy()
        pred = md.mPredict1({"windowsize": 21}, sp_options, sp_param, mfeat.copy(), bmi, "r", sc, ec)
        out["c%d_mfeatures" % ci] = mfeat
        out["c%d_model_state" % ci] = ev["model_state"].astype("U5")
        out["c%d_n_events" % ci] = len(ev)
        for f in ("refbase", "readbase"):
            out["c%d_bmi_%s" % (ci, f)] = bmi_in[f].astype("U1")
        out["c%d_bmi_refbasei" % ci] = bmi_in["refbasei"].astype(np.int64)
        out["c%d_bmi_readbasei" % ci] = bmi_in["readbasei"].astype(np.int64)
        out["c%d_clips" % ci] = np.array([sc, ec])
        out["c%d_batches" % ci] = np.array([b[0] for b in sess.batches])
        out["c%d_pred_mod_num" % ci] = pred
        out["c%d_mod_pred" % ci] = bmi["mod_pred"].astype(np.int64)
        print("mPredict1 case", ci, "n_aligned", n_al, "batches", [b[0] for b in sess.batches], "pred_mod_num", pred)
    out["n_cases"] = len(cases)
    np.savez_compressed(os.path.join(HERE, "host_mpredict1.npz"), **out)


def make_getfeature(md):
    rng = np.random.default_rng(43)
    out = {}
    k = 0
    for strand in "+-":
        for (n_al, sc, ec) in [(300, 5, 7), (180, 0, 0)]:
            ev, bmi = synth_read(rng, n_al, sc, ec, strand=strand)
            nins = int((bmi["refbase"] == "-").sum())
            ndel = int((bmi["readbase"] == "-").sum())
            mapped_start = int(bmi["refbasei"].min())
            sp_options = {"Error": defaultdict(list)}
            sp_param = {"f5data": {"r": (None, ev, None, "f.fast5")}, "f5status": ""}
            mfeat, isdif = md.get_Feature({"fnum": 7}, sp_options, sp_param, None, sp_param["f5data"], "r", sc, ec, bmi,
                                          strand, "chrS", mapped_start, nins, ndel)
            assert not isdif and sp_param["f5status"] == ""
            out["g%d_strand" % k] = strand
            out["g%d_clips" % k] = np.array([sc, ec])
            out["g%d_mapped_start" % k] = mapped_start
            out["g%d_indels" % k] = np.array([nins, ndel])
            for f in ("mean", "stdv"):
                out["g%d_ev_%s" % (k, f)] = ev[f]
            out["g%d_ev_length" % k] = ev["length"].astype(np.int64)
            out["g%d_model_state" % k] = ev["model_state"].astype("U5")
            for f in ("refbase", "readbase"):
                out["g%d_bmi_%s" % (k, f)] = bmi[f].astype("U1")
            out["g%d_bmi_refbasei" % k] = bmi["refbasei"].astype(np.int64)
            out["g%d_mfeatures" % k] = mfeat
            print("get_Feature case", k, strand, mfeat.shape)
            k += 1
    out["n_cases"] = k
    np.savez_compressed(os.path.join(HERE, "host_getfeature.npz"), **out)


def make_sum_handler(md):
    rng = np.random.default_rng(44)
    cases = []
    for strand in "+-":
        reads = []
        for r in range(6):
            _, bmi = synth_read(rng, int(rng.integers(60, 140)), 0, 0, strand=strand, p_ins=0.05, p_del=0.08,
                                ref_start=int(rng.integers(100, 160)))
            bmi["mod_pred"] = ((bmi["refbase"] == "C") & (bmi["readbase"] != "-") & (rng.random(len(bmi)) < 0.4)).astype(int)
            reads.append(bmi)
        # a position with coverage > 1000 (column 5 caps at 1000, column 10 does not)
        deep = np.zeros(1203, dtype=BMI_DTYPE)
        deep["refbase"] = "C"; deep["readbase"] = "C"; deep["refbasei"] = 50; deep["mod_pred"] = (np.arange(1203) % 3 == 0)
        reads.append(deep)
        # a deletion-only position (key created, cov stays 0) and an 'N' reference base
        odd = np.array([("C", "-", 40, 0, 0), ("N", "A", 41, 0, 0), ("C", "T", 42, 1, 1)], dtype=BMI_DTYPE)
        reads.append(odd)
        outdir = tempfile.mkdtemp()
        md.read_file_list = lambda cif, c, s, spo, _n=len(reads): spo.__setitem__("handlingList", list(range(_n)))
        md.read_pred_detail = lambda mo, spo, hl, _r=reads, _s=strand: (_r[hl].copy(), "chrS", _s)
        q = queue.Queue()
        q.put(("unused.ind", "chrS", strand))
        md.sum_handler({"Base": "C", "mod_cluster": 0, "outFolder": outdir}, q)
        bed = open(os.path.join(outdir, "mod_pos.chrS%s.C.bed" % strand), "rb").read()
        cases.append({"chr": "chrS", "strand": strand, "Base": "C",
                      "reads": [{"refbase": "".join(r["refbase"]), "readbase": "".join(r["readbase"]),
                                 "refbasei": [int(v) for v in r["refbasei"]], "mod_pred": [int(v) for v in r["mod_pred"]]}
                                for r in reads],
                      "bed": bed.decode("ascii")})
        print("sum_handler", strand, len(bed), "bytes;", bed.decode().splitlines()[0])
    json.dump(cases, open(os.path.join(HERE, "host_sum_handler.json"), "w"))


def make_host():
    md = import_reference()
    make_mpredict1(md)
    make_getfeature(md)
    make_sum_handler(md)


if __name__ == "__main__":
    make_host()
