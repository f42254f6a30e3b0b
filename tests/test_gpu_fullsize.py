"""GPU (-m gpu): BASELINE.json config 2 at full size (10^6 windows, batches of 65,536) through
size-independent properties, plus an oracle spot check.

 * windows are independent work items (a read of L bases is L independent windows):
   the output of window i must not depend on its neighbours, its position in the batch, the batch
   split, or the workgroup/tile it lands in  ->  permutation equivariance + split invariance,
 * determinism (bit-identical reruns), rows of prob sum to 1, cls == (p1 > p0),
 * summary: counting the 10^6 classes per position in one call == in 16 calls == numpy bincount."""
import numpy as np
import pytest

from deepmod_amd import model, summary, synth
from oracle import oracle_np

pytestmark = pytest.mark.gpu
N = 1_000_000
BATCH = 65536


@pytest.fixture(scope="module")
def full_run(gpu_device):
    w = synth.synthetic_weights(17, 4.0)
    m = model.BiLSTMModel(w, gpu_device)
    x = synth.synthetic_windows(N, seed=20260928)
    prob = np.empty((N, 2), np.float32)
    cls = np.empty(N, np.uint8)
    for off in range(0, N, BATCH):          # 15 x 65,536 + 16,960: the config-2 batching
        p, c = m.predict_windows(x[off:off + BATCH])
        prob[off:off + BATCH] = p
        cls[off:off + BATCH] = c
    yield w, m, x, prob, cls
    m.close()


def test_outputs_are_well_formed(full_run):
    _, _, _, prob, cls = full_run
    assert np.isfinite(prob).all()
    assert np.abs(prob.sum(axis=1) - 1.0).max() <= 1e-6
    assert np.array_equal(cls.astype(bool), prob[:, 1] > prob[:, 0])
    assert 0 < cls.sum() < N               # both classes occur with the saturated synthetic weights


def test_single_call_equals_batched_calls(full_run):
    """one 10^6-window call (library-side staging in 65,536 batches, ragged tail tile) is bit-identical"""
    _, m, x, prob, cls = full_run
    p1, c1 = m.predict_windows(x)
    assert np.array_equal(p1, prob) and np.array_equal(c1, cls)


def test_permutation_equivariance_and_split_invariance(full_run):
    _, m, x, prob, cls = full_run
    rng = np.random.default_rng(5)
    idx = rng.permutation(N)[:200_000]
    p, c = m.predict_windows(x[idx])                       # different neighbours, tiles, waves, lanes
    assert np.array_equal(p, prob[idx]) and np.array_equal(c, cls[idx])
    cuts = [0, 1, 17, 130, 4099, 65537, 200_000]           # odd split points
    for a, b in zip(cuts[:-1], cuts[1:]):
        pp, cc = m.predict_windows(x[idx[a:b]])
        assert np.array_equal(pp, prob[idx[a:b]]) and np.array_equal(cc, cls[idx[a:b]])


def test_error_tail_of_the_default_and_the_fp32_kernel_against_the_oracle(full_run):
    """The TAIL of the probability error, not a spot check: a stratified quarter of the 10^6 windows - every fourth
    window plus the 1,000 windows nearest to p1 = 0.5, where a class can flip - against the fp32 C oracle, for the default (split-f16) and
    the fp32 kernel.  Worst window of the full 10^6 (tools/i8_tail.py, profiles/r04/i8_tail.txt): 9.1e-6 / 8.5e-6; the bound asserted here,
    3e-5, is what a change of the default's arithmetic would have to stay under."""
    w, m, x, prob, cls = full_run
    near_half = np.argsort(np.abs(prob[:, 1] - 0.5))[:1000]
    idx = np.union1d(np.arange(0, N, 4), near_half)
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x[idx])
    d = np.abs(prob[idx] - ref_prob).max(axis=1)
    assert d.max() <= 3e-5, float(d.max())
    near = np.abs(ref_prob[:, 1] - 0.5) < 1e-4
    assert np.array_equal(cls[idx][~near], ref_cls[~near])
    m.set_precision("f32")
    try:
        p32 = np.concatenate([m.predict_windows(x[idx[o:o + BATCH]])[0] for o in range(0, len(idx), BATCH)])
    finally:
        m.set_precision("f16x3")
    d32 = np.abs(p32 - ref_prob).max(axis=1)
    assert d32.max() <= 3e-5, float(d32.max())
    print("error tail on %d windows (every 4th of 10^6 + the 1,000 nearest to a tie): f16x3 max %.3g p99.99 %.3g | f32 max %.3g p99.99 %.3g; %d near ties"
          % (len(idx), float(d.max()), float(np.quantile(d, 0.9999)), float(d32.max()), float(np.quantile(d32, 0.9999)), int(near.sum())))


def test_summary_of_a_million_bases(full_run, gpu_device):
    _, _, _, _, cls = full_run
    rng = np.random.default_rng(7)
    length = 4_641_652
    starts = rng.integers(0, length - 9000, N // 8000 + 1)
    pos = np.concatenate([s + np.arange(8000) for s in starts])[:N].astype(np.int64)   # 8 kb reads
    flags = ((rng.random(N) < 0.25).astype(np.uint8) | ((rng.random(N) < 0.97).astype(np.uint8) << 1))
    s1 = summary.PositionSummary(length, gpu_device)
    s1.add_classified(pos, flags, cls)
    s16 = summary.PositionSummary(length, gpu_device)
    for off in range(0, N, BATCH):
        s16.add_classified(pos[off:off + BATCH], flags[off:off + BATCH], cls[off:off + BATCH])
    t1, c1, m1 = s1.fetch()
    t16, c16, m16 = s16.fetch()
    is_base = (flags & 1) != 0
    covd = is_base & ((flags & 2) != 0)
    assert np.array_equal(t1, np.bincount(pos[is_base], minlength=length).astype(np.int32))
    assert np.array_equal(c1, np.bincount(pos[covd], minlength=length).astype(np.int32))
    assert np.array_equal(m1, np.bincount(pos[covd & (cls == 1)], minlength=length).astype(np.int32))
    assert np.array_equal(t1, t16) and np.array_equal(c1, c16) and np.array_equal(m1, m16)


def test_opt_in_int8_precision_on_the_million_windows(full_run):
    """DM_PREC_F16I8 at full size, against the default kernel on all 10^6 windows.  The tail is what the small parity tests cannot
    see (profiles/r03/i8_tail.txt: worst window 1.1e-4 from the oracle, 2 of 10^6 above 1e-4, 99.99 % below 6.5e-5; the default:
    9e-6): the assertions are its documented bound - NOT the path's 1e-4 - plus classes equal wherever the default is not within that
    bound of a tie, bit-identical reruns and permutation / split invariance."""
    _, m, x, prob, cls = full_run
    m.set_precision("f16i8")
    try:
        p8 = np.empty((N, 2), np.float32)
        c8 = np.empty(N, np.uint8)
        for off in range(0, N, BATCH):
            p, c = m.predict_windows(x[off:off + BATCH])
            p8[off:off + BATCH] = p
            c8[off:off + BATCH] = c
        d = np.abs(p8 - prob).max(axis=1)
        assert d.max() <= 2e-4, float(d.max())
        assert np.quantile(d, 0.9999) <= 1e-4 and int((d > 1e-4).sum()) <= 20, (float(np.quantile(d, 0.9999)), int((d > 1e-4).sum()))
        clear = np.abs(prob[:, 1] - 0.5) > 2e-4
        assert np.array_equal(c8[clear], cls[clear])
        idx = np.random.default_rng(9).permutation(N)[:100_000]
        p, c = m.predict_windows(x[idx])
        assert np.array_equal(p, p8[idx]) and np.array_equal(c, c8[idx])
        print("f16i8 vs f16x3 on %d windows: max |dp| %.3g, p99.99 %.3g, %d windows above 1e-4, %d classes differ (all within 2e-4 of a tie)"
              % (N, float(d.max()), float(np.quantile(d, 0.9999)), int((d > 1e-4).sum()), int((c8 != cls).sum())))
    finally:
        m.set_precision("f16x3")
