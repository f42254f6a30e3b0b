"""Paper experiment: can the two cross terms of the split-f16 product
        a*b ~= a_hi*b_hi + a_lo*b_hi + a_hi*b_lo
be computed with narrower operands (fp8 e4m3 / e5m2 with MX block-32 scales, bf16) without leaving the
1e-4 probability tolerance of the path?  Emulated in numpy on the oracle's graph (oracle/oracle_np.py), float64
accumulation, synthetic weights at scales 1, 4, 16.  Dev/test infrastructure only (lives under tests/).

    python tests/experiments/split_precision_experiment.py [n_windows]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepmod_amd import synth          # noqa: E402
from oracle import oracle_np as onp    # noqa: E402


def q_float(v, mbits, emin, vmax):
    """Round to a binary float with `mbits` explicit mantissa bits, minimum normal exponent emin (subnormals below),
    saturating at vmax.  Round to nearest even."""
    v = np.asarray(v, np.float64)
    a = np.abs(v)
    e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, emin)
    quantum = np.exp2(e - mbits)
    q = np.rint(a / quantum) * quantum
    return np.sign(v) * np.minimum(q, vmax)


def f16(v):
    return np.asarray(v, np.float32).astype(np.float16).astype(np.float64)


def bf16(v):
    return q_float(v, 7, -126, 3.3895e38)


def mx_quant(v, axis, mbits, emin, emax, vmax):
    """MX block format: blocks of 32 along `axis` share a power-of-two scale 2^(floor(log2 max|block|) - emax)."""
    v = np.moveaxis(np.asarray(v, np.float64), axis, -1)
    k = v.shape[-1]
    pad = (-k) % 32
    vp = np.pad(v, [(0, 0)] * (v.ndim - 1) + [(0, pad)])
    blk = vp.reshape(vp.shape[:-1] + (-1, 32))
    m = np.abs(blk).max(axis=-1, keepdims=True)
    scale = np.exp2(np.floor(np.log2(np.where(m > 0, m, 1.0))) - emax)
    q = q_float(blk / scale, mbits, emin, vmax) * scale
    q = q.reshape(vp.shape)[..., :k]
    return np.moveaxis(q, -1, axis)


E4M3 = dict(mbits=3, emin=-6, emax=8, vmax=448.0)
E5M2 = dict(mbits=2, emin=-14, emax=15, vmax=57344.0)


def make_matmul(mode):
    def mm(a, w):
        a = np.asarray(a, np.float64)
        w = np.asarray(w, np.float64)
        if mode == "fp32":
            return (a.astype(np.float32) @ w.astype(np.float32)).astype(np.float64)
        a_hi = f16(a); a_lo = f16(a - a_hi)
        w_hi = f16(w); w_lo = f16(w - w_hi)
        out = a_hi @ w_hi
        if mode == "f16x3":
            return out + a_lo @ w_hi + a_hi @ w_lo
        if mode == "f16x1":
            return out
        if mode == "bf16cross":
            return out + bf16(a_lo) @ bf16(w_hi) + bf16(a_hi) @ bf16(w_lo)
        if mode in ("e4m3cross", "e5m2cross"):
            F = E4M3 if mode == "e4m3cross" else E5M2
            qa = lambda t: mx_quant(t, 1, **F)     # blocks along K of A [n, K]
            qw = lambda t: mx_quant(t, 0, **F)     # blocks along K of W [K, N]
            return out + qa(a_lo) @ qw(w_hi) + qa(a_hi) @ qw(w_lo)
        if mode == "i8cross":
            # both cross terms as ONE int8 product with int32 accumulation (v_mfma_i32_16x16x64_i8 runs at twice the f16 rate):
            #   [a_lo | a_hi] . [w_hi ; w_lo] in fixed point - activations against a global scale (|h| < 1, |h_lo| <= 2^-12; other
            #   inputs against their row maximum), weights against their column maximum; lo scales = 2^-12 x hi scales so that
            #   the two products share one scale and one accumulator
            nx = 7 if a.shape[1] == 107 else 0                                               # layer 0: the 7 raw features keep the f16 cross terms
            sw = np.abs(w_hi[nx:]).max(axis=0, keepdims=True)                                # per gate column
            sw = np.where(sw > 0, sw, 1.0)
            q = lambda t, sc: np.clip(np.rint(t / sc * 127.0), -127, 127)
            cross = q(a_lo[:, nx:], 2.0 ** -11) @ q(w_hi[nx:], sw) + q(a_hi[:, nx:], 1.0) @ q(w_lo[nx:], sw * 2.0 ** -11)
            return out + cross * (sw * 2.0 ** -11 / (127.0 * 127.0)) + a_lo[:, :nx] @ w_hi[:nx] + a_hi[:, :nx] @ w_lo[:nx]
        if mode in ("i8cross_alo", "i8cross_ahi"):     # only one of the two cross terms in int8, the other in f16
            nx = 7 if a.shape[1] == 107 else 0
            sw = np.abs(w_hi[nx:]).max(axis=0, keepdims=True)
            sw = np.where(sw > 0, sw, 1.0)
            q = lambda t, sc: np.clip(np.rint(t / sc * 127.0), -127, 127)
            xpart = a_lo[:, :nx] @ w_hi[:nx] + a_hi[:, :nx] @ w_lo[:nx]
            if mode == "i8cross_alo":
                return out + xpart + (q(a_lo[:, nx:], 2.0 ** -11) @ q(w_hi[nx:], sw)) * (sw * 2.0 ** -11 / (127.0 * 127.0)) + a_hi[:, nx:] @ w_lo[nx:]
            return out + xpart + a_lo[:, nx:] @ w_hi[nx:] + (q(a_hi[:, nx:], 1.0) @ q(w_lo[nx:], sw * 2.0 ** -11)) * (sw * 2.0 ** -11 / (127.0 * 127.0))
        if mode == "e4m3cross_full":               # cross terms from the FULL operand: a_lo*w + a*w_lo - drops nothing extra
            qa = lambda t: mx_quant(t, 1, **E4M3)
            qw = lambda t: mx_quant(t, 0, **E4M3)
            return out + qa(a_lo) @ qw(w) + qa(a) @ qw(w_lo)
        raise ValueError(mode)
    return mm


def predict(weights, x, mm):
    x = np.asarray(x, np.float32)
    n = x.shape[0]
    sig = lambda t: (np.float32(1) / (np.float32(1) + np.exp(-t, dtype=np.float32))).astype(np.float32)
    finals = []
    for d, direction in enumerate(("fw", "bw")):
        h = [np.zeros((n, onp.HID), np.float32) for _ in range(3)]
        c = [np.zeros((n, onp.HID), np.float32) for _ in range(3)]
        for s in range(onp.LIVE):
            row = s if d == 0 else onp.WIN - 1 - s
            inp = x[:, row, :]
            for l in range(3):
                kern = weights[onp.cell_name(direction, l, "kernel")]
                bias = weights[onp.cell_name(direction, l, "bias")].astype(np.float32)
                g = mm(np.concatenate([inp, h[l]], axis=1), kern).astype(np.float32) + bias
                gi, gj, gf, go = np.split(g, 4, axis=1)
                c[l] = (c[l] * sig(gf + np.float32(1.0)) + sig(gi) * np.tanh(gj)).astype(np.float32)
                h[l] = (np.tanh(c[l]) * sig(go)).astype(np.float32)
                inp = h[l]
        finals.append(h[2])
    hcat = np.concatenate(finals, axis=1)
    logits = (hcat @ weights[onp.HEAD_W].astype(np.float32)).astype(np.float32) + weights[onp.HEAD_B].astype(np.float32)
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    x = synth.synthetic_windows(n, seed=3)
    modes = ["f16x3", "bf16cross", "i8cross", "i8cross_alo", "i8cross_ahi", "e4m3cross", "e5m2cross", "f16x1"]
    print("max |dp| vs the fp32 restatement, %d windows (tolerance of the path: 1e-4)" % n)
    print("%-8s" % "scale" + "".join("%16s" % m for m in modes))
    for scale in (1.0, 4.0) + ((16.0,) if os.environ.get('DM_EXP_SCALE16') else ()):
        for seed in (7, 26):
            w = synth.synthetic_weights(seed, scale)
            ref = predict(w, x, make_matmul("fp32"))
            row = []
            for m in modes:
                p = predict(w, x, make_matmul(m))
                row.append(float(np.abs(p - ref).max()))
            print("%-8s" % ("%g/s%d" % (scale, seed)) + "".join("%16.3g" % v for v in row), flush=True)


if __name__ == "__main__":
    main()
