"""Paper experiment: a per-(step, layer) PRECISION SCHEDULE for the split-f16 classifier.

The product kernel issues three 16-bit MFMA products per fp32 product everywhere (a_hi*w_hi + a_lo*w_hi + a_hi*w_lo: 3.09
issued units per algorithmic unit).  The forget gates attenuate what an early step contributes to the one output that is
read (h of layer 2 at step 10), so an early stage may get away with cheaper cross terms.  This script emulates, on the
oracle's graph (oracle/oracle_np.py; float64 accumulation of exactly-representable products), a schedule that assigns every
stage (step s, layer l) one of

    x3    the product kernel's arithmetic                                                  3 issued units per k-slot
    i8    hi*hi in f16 + BOTH cross terms as one int8 product with int32 accumulation      2 units
          ([a_lo8 | a_hi8] . [w_hi8 ; w_lo8]; v_mfma_i32_32x32x32_i8 has twice the K of the f16 MFMA at the same cost)
    x2w   a_hi*w_hi + a_lo*w_hi   (weights rounded to f16)                                   2 units
    x2a   a_hi*w_hi + a_hi*w_lo   (activations rounded to f16)                               2 units
    x1    a_hi*w_hi                                                                          1 unit

and reports max |dp| against the fp32 restatement (the contract: 1e-4; bar for a change of arithmetic: 3e-5), against a
float64 evaluation of the same graph, and the classes that flip away from near ties.  Dev / test infrastructure only.

    python tests/experiments/precision_schedule_experiment.py table   [n_windows]     # schedules "last k steps full"
    python tests/experiments/precision_schedule_experiment.py greedy  [n_windows]     # per-stage sensitivities + greedy schedule
    python tests/experiments/precision_schedule_experiment.py golden                  # the 10 golden fixtures
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepmod_amd import synth          # noqa: E402
from oracle import oracle_np as onp    # noqa: E402

LIVE, HID = onp.LIVE, onp.HID
UNITS = {"x3": 3.0, "i8": 2.0, "i8o": 2.0, "x2w": 2.0, "x2a": 2.0, "x1": 1.0}
KSTEPS = (7, 13, 13)            # k16-steps of a tile of layer 0 / 1 / 2 in the product kernel (lstm_f16s.hip.inc)


def f16(v):
    return np.asarray(v, np.float32).astype(np.float16).astype(np.float64)


class SplitWeights:
    """The per-matrix constants of every mode, computed once."""

    def __init__(self, kern, nx):
        w = np.asarray(kern, np.float64)
        self.nx = nx                          # layer 0: rows [0, nx) multiply the raw features (always x3)
        self.w = w
        self.hi = f16(w)
        self.lo = f16(w - self.hi)
        hh = self.hi[nx:]
        # int8 forms, per gate column.  One int32 accumulator takes both cross terms, so the two slot scales must agree:
        #   s(a_lo) s(w_hi) = s(a_hi) s(w_lo);   s(a_hi) = 1/127 (|h| < 1), s(a_lo) = 2^-12/127 (|h_lo| <= 2^-12)
        #   => s(w_lo) = 2^-12 s(w_hi), and s(w_hi) must cover both max|w_hi| and 2^12 max|w_lo| of the column
        ll = self.lo[nx:]
        sw = np.maximum(np.abs(hh).max(axis=0, keepdims=True), 4096.0 * np.abs(ll).max(axis=0, keepdims=True))
        self.sw = np.where(sw > 0, sw, 1.0)
        q = lambda t, sc: np.clip(np.rint(t / sc * 127.0), -127, 127)
        self.q_hi = q(hh, self.sw)
        self.q_lo = q(ll, self.sw / 4096.0)
        # round-2 scaling (a_lo against 2^-11, w_lo against 2^-11 max|w_hi|), kept for comparison
        swo = np.abs(hh).max(axis=0, keepdims=True)
        self.swo = np.where(swo > 0, swo, 1.0)
        self.qo_hi = q(hh, self.swo)
        self.qo_lo = q(ll, self.swo / 2048.0)


def matmul(a, sw: SplitWeights, mode):
    a = np.asarray(a, np.float64)
    if mode == "fp32":
        return (a.astype(np.float32) @ sw.w.astype(np.float32)).astype(np.float64)
    if mode == "f64":
        return a @ sw.w
    a_hi = f16(a)
    a_lo = f16(a - a_hi)
    out = a_hi @ sw.hi
    if mode == "x1":
        return out
    if mode == "x3":
        return out + a_lo @ sw.hi + a_hi @ sw.lo
    if mode == "x2w":
        return out + a_lo @ sw.hi
    if mode == "x2a":
        return out + a_hi @ sw.lo
    nx = sw.nx
    q = lambda t, sc: np.clip(np.rint(t / sc * 127.0), -127, 127)
    xpart = a_lo[:, :nx] @ sw.hi[:nx] + a_hi[:, :nx] @ sw.lo[:nx] if nx else 0.0
    if mode == "i8":
        cross = q(a_lo[:, nx:], 2.0 ** -12) @ sw.q_hi + q(a_hi[:, nx:], 1.0) @ sw.q_lo
        return out + xpart + cross * (sw.sw * 2.0 ** -12 / (127.0 * 127.0))
    if mode == "i8o":
        cross = q(a_lo[:, nx:], 2.0 ** -11) @ sw.qo_hi + q(a_hi[:, nx:], 1.0) @ sw.qo_lo
        return out + xpart + cross * (sw.swo * 2.0 ** -11 / (127.0 * 127.0))
    raise ValueError(mode)


class Graph:
    def __init__(self, weights):
        self.w = weights
        self.sw = {}
        for d in ("fw", "bw"):
            for l in range(3):
                self.sw[d, l] = SplitWeights(weights[onp.cell_name(d, l, "kernel")], onp.NFEAT if l == 0 else 0)

    def predict(self, x, schedule, state_dtype=np.float32):
        """schedule: mode name, or a function (step, layer) -> mode name.  state_dtype float64 + mode 'f64' = the exact graph."""
        sched = schedule if callable(schedule) else (lambda s, l: schedule)
        F = state_dtype
        x = np.asarray(x, np.float32)
        n = x.shape[0]
        sig = lambda t: (F(1) / (F(1) + np.exp(-t.astype(F)))).astype(F)
        finals = []
        for d, direction in enumerate(("fw", "bw")):
            h = [np.zeros((n, HID), F) for _ in range(3)]
            c = [np.zeros((n, HID), F) for _ in range(3)]
            for s in range(LIVE):
                row = s if d == 0 else onp.WIN - 1 - s
                inp = x[:, row, :].astype(F)
                for l in range(3):
                    bias = self.w[onp.cell_name(direction, l, "bias")].astype(F)
                    g = matmul(np.concatenate([inp, h[l]], axis=1), self.sw[direction, l], sched(s, l)).astype(F) + bias
                    gi, gj, gf, go = np.split(g, 4, axis=1)
                    c[l] = (c[l] * sig(gf + F(1.0)) + sig(gi) * np.tanh(gj)).astype(F)
                    h[l] = (np.tanh(c[l]) * sig(go)).astype(F)
                    inp = h[l]
            finals.append(h[2])
        hcat = np.concatenate(finals, axis=1)
        logits = (hcat @ self.w[onp.HEAD_W].astype(F)).astype(F) + self.w[onp.HEAD_B].astype(F)
        e = np.exp(logits - logits.max(axis=1, keepdims=True))
        return (e / e.sum(axis=1, keepdims=True)).astype(F)


def issued_units(schedule):
    """issued MFMA units per k-slot of the whole graph, weighted with the k16-steps of the product kernel's tiles; the zero-state
    k16-steps of step 0 are not issued at all (x3 everywhere = 3 * 345 / 363; times the K 208/201 and N 104/100 padding that is the
    kernel's 3.09 issued per algorithmic unit)."""
    sched = schedule if callable(schedule) else (lambda s, l: schedule)
    tot = iss = 0.0
    for s in range(LIVE):
        for l in range(3):
            tot += KSTEPS[l]
            iss += (KSTEPS[l] - (6 if s == 0 else 0)) * UNITS[sched(s, l)]
    return iss / tot


def last_k_full(k, cheap):
    return lambda s, l: "x3" if s >= LIVE - k else cheap


def compare(p, ref):
    p = np.asarray(p, np.float64)
    ref = np.asarray(ref, np.float64)
    dp = np.abs(p - ref).max()
    near = np.abs(ref[:, 1] - 0.5) < 1e-4
    flips = int((((p[:, 1] > p[:, 0]) != (ref[:, 1] > ref[:, 0])) & ~near).sum())
    return float(dp), flips


CASES = [(1.0, 7), (1.0, 26), (4.0, 7), (4.0, 26)]


def cmd_table(n):
    x = synth.synthetic_windows(n, seed=3)
    rows = []
    print("max |dp| vs the fp32 restatement / vs float64, %d windows; contract 1e-4, bar for new arithmetic 3e-5" % n)
    hdr = "%-26s %6s" % ("schedule", "issued") + "".join("%22s" % ("scale %g seed %d" % c) for c in CASES)
    print(hdr, flush=True)
    graphs = [Graph(synth.synthetic_weights(seed, scale)) for scale, seed in CASES]
    refs = [g.predict(x, "fp32") for g in graphs]
    exact = [g.predict(x, "f64", np.float64) for g in graphs]
    line = "%-26s %6s" % ("fp32 restatement vs f64", "-")
    for r, e in zip(refs, exact):
        line += "%22s" % ("- / %.2e" % compare(r, e)[0])
    print(line, flush=True)
    scheds = [("x3 everywhere", "x3")]
    for cheap in ("i8", "i8o", "x2w", "x2a", "x1"):
        for k in (0, 2, 3, 4, 5, 6, 7, 8):
            if cheap in ("x2w", "x2a", "x1", "i8o") and k not in (0, 3, 5, 7):
                continue
            scheds.append(("%s, last %d steps x3" % (cheap, k), last_k_full(k, cheap)))
    for name, sc in scheds:
        iss = issued_units(sc) * 208.0 / 201.0 * 104.0 / 100.0
        line = "%-26s %6.2f" % (name, iss)
        rec = {"schedule": name, "issued": iss, "cases": []}
        for g, r, e, c in zip(graphs, refs, exact, CASES):
            p = g.predict(x, sc)
            d32, fl = compare(p, r)
            d64, _ = compare(p, e)
            line += "%22s" % ("%.2e / %.2e%s" % (d32, d64, (" F%d" % fl) if fl else ""))
            rec["cases"].append({"scale": c[0], "seed": c[1], "dp_fp32": d32, "dp_f64": d64, "flips": fl})
        rows.append(rec)
        print(line, flush=True)
    return rows


def cmd_greedy(n, cheap="i8", bar=3e-5):
    """sensitivity of every stage (that stage alone in the cheap mode), then a greedy schedule: stages are made cheap in the
    order of their sensitivity as long as the worst case stays under the bar."""
    x = synth.synthetic_windows(n, seed=3)
    graphs = [Graph(synth.synthetic_weights(seed, scale)) for scale, seed in CASES[2:]]      # scale 4: the binding cases
    refs = [g.predict(x, "fp32") for g in graphs]
    worst = lambda sc: max(compare(g.predict(x, sc), r)[0] for g, r in zip(graphs, refs))
    base = worst("x3")
    print("x3 everywhere: %.2e" % base)
    sens = {}
    for s in range(LIVE):
        for l in range(3):
            sens[s, l] = worst(lambda ss, ll, s=s, l=l: cheap if (ss, ll) == (s, l) else "x3")
        print("step %2d: " % s + "  ".join("l%d %.2e" % (l, sens[s, l]) for l in range(3)), flush=True)
    order = sorted(sens, key=lambda k: sens[k])
    chosen = set()
    for st in order:
        trial = chosen | {st}
        w = worst(lambda s, l: cheap if (s, l) in trial else "x3")
        if w <= bar:
            chosen = trial
        print("try %s -> %.2e %s" % (st, w, "kept" if st in chosen else "rejected"), flush=True)
    sc = lambda s, l: cheap if (s, l) in chosen else "x3"
    print("greedy schedule: %d of 33 stages in %s, issued %.2f, worst %.2e" % (len(chosen), cheap, issued_units(sc) * 208 / 201 * 1.04, worst(sc)))
    print(sorted(chosen))


def cmd_golden():
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "bilstm_*.npz"))):
        z = np.load(path)
        g = Graph(synth.synthetic_weights(int(z["seed_w"]), float(z["scale"])))
        line = "%-52s" % os.path.basename(path)
        for name, sc in (("x3", "x3"), ("i8", "i8"), ("i8 last5 x3", last_k_full(5, "i8"))):
            d, fl = compare(g.predict(z["X"], sc), z["prob"])
            line += "  %s %.2e%s" % (name, d, (" F%d" % fl) if fl else "")
        print(line, flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "table"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    if cmd == "table":
        rows = cmd_table(n)
        if len(sys.argv) > 3:
            json.dump(rows, open(sys.argv[3], "w"), indent=1)
    elif cmd == "greedy":
        cmd_greedy(n, *(sys.argv[3:4]))
    elif cmd == "golden":
        cmd_golden()


def cmd_sources(n):
    """which of the four int8 quantisations of the i8 mode carries the error: each made exact in turn (scale 4 weights)"""
    x = synth.synthetic_windows(n, seed=3)
    q = lambda t, sc: np.clip(np.rint(t / sc * 127.0), -127, 127)
    ident = lambda t, sc: t / sc * 127.0

    def make_mm(qa_lo, qw_hi, qa_hi, qw_lo):
        def mm(a, sw):
            a = np.asarray(a, np.float64)
            a_hi = f16(a); a_lo = f16(a - a_hi)
            nx = sw.nx
            out = a_hi @ sw.hi
            xpart = a_lo[:, :nx] @ sw.hi[:nx] + a_hi[:, :nx] @ sw.lo[:nx] if nx else 0.0
            cross = qa_lo(a_lo[:, nx:], 2.0 ** -12) @ qw_hi(sw.hi[nx:], sw.sw) + qa_hi(a_hi[:, nx:], 1.0) @ qw_lo(sw.lo[nx:], sw.sw / 4096.0)
            return out + xpart + cross * (sw.sw * 2.0 ** -12 / (127.0 * 127.0))
        return mm

    variants = {"all four int8": (q, q, q, q), "a_lo exact": (ident, q, q, q), "w_hi exact": (q, ident, q, q), "a_hi exact": (q, q, ident, q),
                "w_lo exact": (q, q, q, ident), "slot 1 exact (a_lo . w_hi)": (ident, ident, q, q), "slot 2 exact (a_hi . w_lo)": (q, q, ident, ident),
                "all exact (= x3)": (ident, ident, ident, ident)}
    global matmul
    keep = matmul
    for scale, seed in CASES[2:]:
        g = Graph(synth.synthetic_weights(seed, scale))
        ref = g.predict(x, "fp32")
        for name, fs in variants.items():
            mm = make_mm(*fs)
            matmul = lambda a, sw, mode, mm=mm: mm(a, sw)
            d, fl = compare(g.predict(x, "custom"), ref)
            print("scale %g seed %d  %-30s %.2e" % (scale, seed, name, d), flush=True)
    matmul = keep


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sources":
    cmd_sources(int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
