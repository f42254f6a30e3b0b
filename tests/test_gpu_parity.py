"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the golden
fixtures.  Tolerance from BASELINE.json north_star: per-base probabilities within 1e-4 (fp32),
classes exact away from near ties (|p1-0.5| < 1e-4)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, trained_like_weights
from deepmod_amd import model, synth
from oracle import oracle_np

pytestmark = pytest.mark.gpu
TOL = 1e-4            # the path's tolerance (BASELINE.json north_star): what the default and the fp32 kernel are held to
TOL_I8 = 2e-4         # the documented bound of the OPT-IN DM_PREC_F16I8: on 10^6 windows at weight scale 4 its worst window is 1.1e-4 from
                      # the oracle and 2 windows exceed 1e-4 (profiles/r03/i8_tail.txt; the default: 9e-6) - it is NOT inside the path's
                      # tolerance in the tail, which is why it is never selected by default


def _check(prob, cls, ref_prob, ref_cls, tol=TOL):
    assert prob.shape == ref_prob.shape
    err = float(np.abs(prob - ref_prob).max()) if len(prob) else 0.0
    assert err <= tol, "max|dp| = %g" % err
    near = np.abs(ref_prob[:, 1] - 0.5) < tol
    bad = (cls.astype(np.int64) != ref_cls) & ~near
    assert not bad.any(), "%d class flips away from ties" % int(bad.sum())
    assert np.allclose(prob.sum(axis=1), 1.0, atol=1e-6)
    return err


@pytest.fixture(scope="module", params=["f32", "f16x3", "f16i8"])
def models(gpu_device, request):
    """Every parity test runs for every precision mode of the library: exact-fp32 MFMA, the split-f16 path (3 f16 products
    per fp32 product: the default) and the opt-in int8-cross-term variant of it (DM_PREC_F16I8).  Same tolerance for all."""
    from deepmod_amd import _lib
    prec = {"f32": _lib.DM_PREC_F32, "f16x3": _lib.DM_PREC_F16X3, "f16i8": _lib.DM_PREC_F16I8}[request.param]
    cache = {}
    get_name = request.param

    def get(seed, scale):
        key = (seed, scale)
        if key not in cache:
            w = trained_like_weights() if seed == "trained" else synth.synthetic_weights(seed, scale)
            m = model.BiLSTMModel(w, device=gpu_device)
            m.set_option(_lib.DM_OPT_PRECISION, prec)
            cache[key] = (w, m)
        return cache[key]
    get.precision = get_name
    get.tol = TOL_I8 if get_name == "f16i8" else TOL
    yield get
    for _, m in cache.values():
        m.close()


FIX = sorted(glob.glob(os.path.join(GOLDEN, "bilstm_*.npz")))


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[7:-4] for p in FIX])
def test_golden_reference_graph(path, models):
    g = np.load(path)
    w, m = models(int(g["seed_w"]), float(g["scale"]))
    prob, cls = m.predict_windows(g["X"])
    _check(prob, cls, g["prob"], g["cls"], models.tol)


def test_trained_like_weights_reference_graph_and_oracle(models):
    """Weights with TRAINED statistics (tests/golden/make_trained_like.py; kernel entries up to 1.7 beside a median of 0.05, saturated
    gates, learnt biases - the reference's own .data shards are absent): the interpreted reference graph on the committed case, then the
    oracle on 20,000 config-2 windows and on ragged sizes."""
    w, m = models("trained", 0.0)
    g = np.load(os.path.join(GOLDEN, "trained_like_case.npz"))
    prob, cls = m.predict_windows(g["X"])
    _check(prob, cls, g["prob"], g["cls"], models.tol)
    for n, seed in ((20000, 31), (129, 32), (1, 33)):
        x = synth.synthetic_windows(n, seed=seed)
        prob, cls = m.predict_windows(x)
        ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
        err = _check(prob, cls, ref_prob, ref_cls, models.tol)
        if n == 20000:
            # where the kernels stand on trained statistics (profiles/r04/i8_tail.txt: worst window of 10^6): default / fp32 an order of
            # magnitude inside the tolerance
            assert err <= (3e-5 if models.precision != "f16i8" else models.tol), err
            assert 0.02 < ref_cls.mean() < 0.98


def test_read_shaped_windows_reference_graph(models):
    """The reference's serialized graph on READ-SHAPED windows (tests/golden/trained_like_tail_case.npz, make_golden_tail.py: tail events with
    means on the +-5 clip and lengths up to 27,000 samples, trained-like weights): every kernel against the graph's own output.  The default and
    the fp32 kernel within 1e-5 - before the event-length cut of round 5 the split-f16 kernels were 2e-5 off on such inputs."""
    w, m = models("trained", 0.0)
    g = np.load(os.path.join(GOLDEN, "trained_like_tail_case.npz"))
    prob, cls = m.predict_windows(g["X"])
    err = _check(prob, cls, g["prob"], g["cls"], models.tol)
    assert err <= (1e-5 if models.precision != "f16i8" else 1e-4), err


@pytest.mark.parametrize("n", [1, 15, 16, 17, 127, 128, 129, 255, 1000, 4097])
@pytest.mark.parametrize("scale", [1.0, 4.0, "trained"])
def test_vs_oracle_ragged_sizes(n, scale, models):
    w, m = models("trained" if scale == "trained" else 21, 0.0 if scale == "trained" else scale)
    x = synth.synthetic_windows(n, seed=100 + n)
    prob, cls = m.predict_windows(x)
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
    _check(prob, cls, ref_prob, ref_cls, models.tol)


def test_auc_against_oracle_classes_and_run_to_run_determinism(models):
    """AUC of the build's p1 against the oracle's class on the same windows (1.0 when no window is a near
    tie), and bit-identical outputs when the same batch is classified twice."""
    from sklearn.metrics import roc_auc_score
    w, m = models(26, 4.0)                       # this seed gives a ~50/50 class mix on synthetic windows
    x = synth.synthetic_windows(6000, seed=77)
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
    prob, cls = m.predict_windows(x)
    prob2, cls2 = m.predict_windows(x)
    assert np.array_equal(prob.view(np.uint32), prob2.view(np.uint32)) and np.array_equal(cls, cls2)
    assert 0.05 < ref_cls.mean() < 0.95
    clear = np.abs(ref_prob[:, 1] - 0.5) > models.tol
    assert roc_auc_score(ref_cls[clear], prob[clear, 1]) == 1.0


def test_empty_batch(models):
    _, m = models(21, 1.0)
    prob, cls = m.predict_windows(np.zeros((0, 21, 7), np.float32))
    assert prob.shape == (0, 2) and cls.shape == (0,)


def test_float64_feed_is_cast_like_the_placeholder(models):
    """mPredict1 feeds float64 windows (myDetect.py:802); the placeholder casts to fp32."""
    w, m = models(21, 1.0)
    x = synth.synthetic_windows(200, seed=4).astype(np.float64)
    prob, cls = m.predict_windows(x)
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x.astype(np.float32))
    _check(prob, cls, ref_prob, ref_cls, models.tol)


def test_extreme_inputs_saturate_cleanly(models):
    """Signal clipped to +-5 MAD, raw length up to thousands of samples, all-zero padding rows."""
    w, m = models(22, 4.0)
    x = synth.synthetic_windows(256, seed=8)
    x[:64, :, 6] = 5000.0
    x[64:128, :, 4] = -5.0
    x[128:192] = 0.0
    prob, cls = m.predict_windows(x)
    assert np.isfinite(prob).all()
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
    _check(prob, cls, ref_prob, ref_cls, models.tol)


@pytest.mark.parametrize("scale", [4.0, 16.0])
def test_event_lengths_beyond_the_f16_range_are_exact(models, scale):
    """`length` is a raw sample count (myDetect.py:894-900): a stalled event can exceed 65,504 samples, the largest f16.
    The split-f16 kernel feeds such a value as x * 2^-k against a weight row stored x 2^k (exact) - nothing is clamped."""
    w, m = models(22, scale)
    x = synth.synthetic_windows(384, seed=9)
    rng = np.random.default_rng(1)
    for lo, val in ((0, 65505.0), (64, 1.0e6), (128, 6.0e7), (192, 70000.0)):
        sel = rng.random((64, 21)) < 0.3                     # some rows of the window, not all
        x[lo:lo + 64, :, 6] = np.where(sel, np.float32(val), x[lo:lo + 64, :, 6])
    x[256:320, 10, 6] = 65504.0                               # the last value of the ordinary slot
    prob, cls = m.predict_windows(x)
    assert np.isfinite(prob).all()
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
    if scale <= 4.0:
        _check(prob, cls, ref_prob, ref_cls, models.tol)
    else:
        # scale 16 exercises the weight fold (|w| x 2.886 x 2^k must stay an f16), but it is outside the regime in which
        # 1e-4 against an fp32 evaluation means anything: the recurrence amplifies fp32 round-off itself - the fp32 MFMA kernel
        # and the C oracle (both exact fp32 products, different summation order) already differ by 3.5e-4 on a few windows.
        # The yardstick there is the EXACT value of the graph (float64 on the same fp32 weights and inputs): a kernel may be
        # off by a small multiple of what the fp32 oracle itself is off by.  Measured ratios (tools/i8_check.py, 20,000 windows
        # x 3 weight seeds): fp32 kernel <= 2.6, split-f16 <= 1.6, int8 cross terms <= 25 (its operands carry 19 bits, not 22).
        p64 = oracle_np.predict_windows_np(w, x, np.float64)[0]
        err_oracle = float(np.abs(ref_prob - p64).max())
        err = np.abs(prob - p64).max(axis=1)
        allowed = {"f32": 4.0, "f16x3": 4.0, "f16i8": 40.0}[models.precision]
        assert err.max() <= allowed * err_oracle, (models.precision, float(err.max()), err_oracle)
        assert np.median(err) <= (2e-4 if models.precision == "f16i8" else 1e-5), float(np.median(err))
        far = np.abs(p64[:, 1] - 0.5) > 2.0 * allowed * err_oracle
        assert np.array_equal(cls[far].astype(np.int64), np.argmax(p64, axis=1)[far])


def test_unrepresentable_inputs_are_refused_not_clamped(gpu_device):
    """DM_PREC_F16X3 range contract (include/deepmod_hip.h): features 0-5 beyond +-65504, a length beyond 65504 * 2^k
    or a NaN fail the call with DM_ERANGE; the fp32 kernel takes the same input; the model object can fall back."""
    from deepmod_amd import _lib
    w = synth.synthetic_weights(22, 4.0)
    m = model.BiLSTMModel(w, device=gpu_device, precision="f16x3")
    k = m.get_info(_lib.DM_INFO_F16_LENGTH_SHIFT)
    assert m.get_info(_lib.DM_INFO_F16_REPRESENTABLE) == 1 and 5 <= k <= 10
    good = synth.synthetic_windows(300, seed=2)
    for col, val in ((4, 1.0e5), (6, 65504.0 * 2.0 ** k * 1.01), (5, np.nan)):
        x = good.copy()
        x[137, 3, col] = val
        with pytest.raises(_lib.DeepModRangeError):
            m.predict_windows(x)
        p_ok, c_ok = m.predict_windows(good)                 # the flag does not stick
        assert np.isfinite(p_ok).all()
        if not np.isnan(val):
            m.set_precision("f32")
            p32, c32 = m.predict_windows(x)
            ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
            _check(p32, c32, ref_prob, ref_cls)
            m.set_precision("f16x3")
    # asynchronous launches report at the next sync
    m.set_option(_lib.DM_OPT_ASYNC, 1)
    x = good.copy()
    x[5, 0, 0] = -1.0e6
    dx = model.DeviceArray.from_host(x, gpu_device)
    dc = model.DeviceArray((len(x),), np.uint8, gpu_device)
    m.predict_windows(dx, cls=dc, want_prob=False)
    with pytest.raises(_lib.DeepModRangeError):
        m.sync()
    m.sync()
    m.close()


def test_weights_outside_the_f16_range_select_the_fp32_kernel(gpu_device):
    from deepmod_amd import _lib
    w = synth.synthetic_weights(23, 1.0)
    name = synth.cell_name("fw", 1, "kernel")
    w[name] = w[name].copy()
    w[name][17, 230] = 4.0e4                                  # x 1.4427 = 5.8e4 fits; a forget-gate column
    m = model.BiLSTMModel(w, device=gpu_device)
    assert m.get_info(_lib.DM_INFO_F16_REPRESENTABLE) == 1
    m.close()
    w[name][17, 130] = 4.0e4                                  # j column: x 2.885 = 1.15e5 does not
    m = model.BiLSTMModel(w, device=gpu_device)
    assert m.get_info(_lib.DM_INFO_F16_REPRESENTABLE) == 0 and m.get_info(_lib.DM_INFO_PRECISION) == _lib.DM_PREC_F32
    with pytest.raises(_lib.DeepModHipError):
        m.set_precision("f16x3")
    x = synth.synthetic_windows(200, seed=6)
    prob, cls = m.predict_windows(x)                          # runs on the fp32 kernel
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
    _check(prob, cls, ref_prob, ref_cls)
    m.close()


def test_exact_tie_is_class_zero(models):
    """tf.argmax returns the first maximum (myMultiBiRNN.py:61): p0 == p1 -> class 0.  Head columns and biases made
    identical give exactly equal logits for every window."""
    w, _ = models(24, 1.0)
    w = dict(w)
    w[synth.HEAD_W] = np.repeat(w[synth.HEAD_W][:, :1], 2, axis=1).copy()
    w[synth.HEAD_B] = np.array([0.25, 0.25], np.float32)
    from deepmod_amd import _lib
    _, m0 = models(24, 1.0)
    m = model.BiLSTMModel(w, device=m0.device)
    m.set_option(_lib.DM_OPT_PRECISION, m0.get_info(_lib.DM_INFO_PRECISION))
    x = synth.synthetic_windows(500, seed=12)
    prob, cls = m.predict_windows(x)
    assert np.array_equal(prob, np.full((500, 2), 0.5, np.float32))
    assert not cls.any()
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
    assert np.array_equal(ref_prob, prob) and not ref_cls.any()
    m.close()


def test_device_resident_and_host_paths_agree(models, gpu_device):
    w, m = models(21, 4.0)
    x = synth.synthetic_windows(70000, seed=77)  # > one 65,536-window staging batch
    prob_h, cls_h = m.predict_windows(x)
    dx = model.DeviceArray.from_host(x, gpu_device)
    dp = model.DeviceArray((x.shape[0], 2), np.float32, gpu_device)
    dc = model.DeviceArray((x.shape[0],), np.uint8, gpu_device)
    m.predict_windows(dx, prob=dp, cls=dc)
    assert np.array_equal(dp.to_host(), prob_h)
    assert np.array_equal(dc.to_host(), cls_h)
    # determinism: same input, same bits
    prob2, cls2 = m.predict_windows(x)
    assert np.array_equal(prob2, prob_h) and np.array_equal(cls2, cls_h)
    # sample against the oracle
    idx = np.random.default_rng(0).choice(x.shape[0], 2000, replace=False)
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x[idx])
    _check(prob_h[idx], cls_h[idx], ref_prob, ref_cls, models.tol)


def test_predict_read_equals_materialised_windows(models):
    """On-device window assembly (dm_predict_read) == materialised tx[mind-10:mind+11] windows
    (reference myDetect.py:794-803)."""
    w, m = models(21, 4.0)
    rng = np.random.default_rng(5)
    nev = 700
    rows = np.zeros((nev + 200, 7), np.float32)
    rows[100:-100] = synth.synthetic_windows(nev, seed=3)[:, 0, :]
    first, count = 100, nev
    prob_r, cls_r = m.predict_read(rows, first, count)
    xw = np.stack([rows[first + i - 10:first + i + 11] for i in range(count)])
    prob_w, cls_w = m.predict_windows(xw)
    assert np.array_equal(prob_r, prob_w) and np.array_equal(cls_r, cls_w)
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, xw)
    _check(prob_r, cls_r, ref_prob, ref_cls, models.tol)


def test_bad_arguments_raise(models, hip_lib):
    _, m = models(21, 1.0)
    with pytest.raises(ValueError):
        m.predict_windows(np.zeros((4, 20, 7), np.float32))
    from deepmod_amd import _lib
    with pytest.raises(_lib.DeepModHipError):
        m.predict_read(np.zeros((50, 7), np.float32), 5, 10)  # first - 10 < 0


def test_session_adapter_matches_reference_call_forms(models, tmp_path, gpu_device):
    """sess.run(init_l); sess.run([mfpred], feed_dict={X: x, Y: y})[0] -> int64[n]
    (reference myDetect.py:805, :816-820), restoring by variable name from a TF bundle."""
    prefix = str(tmp_path / "mod_train_synth")
    w = synth.write_synthetic_checkpoint(prefix, seed=33, scale=4.0)
    _, init_l, _, _, _, X, Y, _, _, _, _, mfpred = model.mCreateSession(7, 100, 21, {"outputlayer": ""})
    sess = model.new_session(gpu_device)
    saver = model.import_meta_graph(prefix + ".meta")
    saver.restore(sess, model.latest_checkpoint(str(tmp_path)))
    assert sess.run(init_l) is None
    x = synth.synthetic_windows(600, seed=12).astype(np.float64)
    y = np.zeros((600, 2), int)
    out = sess.run([mfpred], feed_dict={X: x, Y: y})[0]
    assert out.dtype == np.int64 and out.shape == (600,)
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x.astype(np.float32))
    near = np.abs(ref_prob[:, 1] - 0.5) < TOL
    assert np.array_equal(out[~near], ref_cls[~near])
    sess.close()


def test_windows_picked_by_centre_row_equal_the_contiguous_call(models, gpu_device):
    """dm_predict_read_at (the streaming worker's call: only the windows centred on a base of interest) gives, for the picked
    centres, bit for bit what dm_predict_read gives for all rows - host arrays and device arrays, ragged counts."""
    w, m = models(26, 4.0)
    rng = np.random.default_rng(8)
    rows = np.concatenate([np.zeros((100, 7), np.float32), synth.synthetic_windows(400, seed=3).reshape(-1, 7)[:3000], np.zeros((100, 7), np.float32)])
    M = len(rows)
    prob_all, cls_all = m.predict_read(rows, 10, M - 20)
    for count in (1, 127, 128, 129, 777):
        centres = np.sort(rng.choice(np.arange(10, M - 10), count, replace=False)).astype(np.int32)
        prob, cls = m.predict_read_at(rows, centres)
        assert np.array_equal(prob.view(np.uint32), prob_all[centres - 10].view(np.uint32)) and np.array_equal(cls, cls_all[centres - 10])
    d_rows = model.DeviceArray.from_host(rows, gpu_device)
    centres = np.arange(10, M - 10, 4, dtype=np.int32)
    d_c = model.DeviceArray.from_host(centres, gpu_device)
    d_cls = model.DeviceArray((len(centres),), np.uint8, gpu_device)
    m.predict_read_at(d_rows, d_c, cls=d_cls, want_prob=False)
    m.sync()
    assert np.array_equal(d_cls.to_host(), cls_all[centres - 10])
    from deepmod_amd import _lib
    with pytest.raises(_lib.DeepModHipError):
        m.predict_read_at(rows, np.array([5], np.int32))            # centre - 10 < 0
    for d in (d_rows, d_c, d_cls):
        d.free()


def test_wave_pair_experiment_kernel_is_refused_or_bit_identical(gpu_device):
    """DM_PREC_F16X3_ROLES (tools/experiments/f16r, round 4) is not part of the product build: the library refuses it unless it was
    built with -DDM_WITH_F16X3_ROLES, in which case its output equals the default kernel's bit for bit (same arithmetic, same order)."""
    from deepmod_amd import _lib
    w = synth.synthetic_weights(26, 4.0)
    m = model.BiLSTMModel(w, device=gpu_device)
    if not m.get_info(_lib.DM_INFO_HAS_F16X3_ROLES):
        with pytest.raises(_lib.DeepModHipError):
            m.set_option(_lib.DM_OPT_PRECISION, _lib.DM_PREC_F16X3_ROLES)
        assert m.get_info(_lib.DM_INFO_PRECISION) == _lib.DM_PREC_F16X3
    else:
        m.set_option(_lib.DM_OPT_F16X3_SHAPE, 32)          # the experiment kernel is the 32x32x16 kernel's arithmetic, bit for bit
        for n in (1, 129, 4097, 70000):
            x = synth.synthetic_windows(n, seed=12 + n)
            m.set_option(_lib.DM_OPT_PRECISION, _lib.DM_PREC_F16X3)
            prob, cls = m.predict_windows(x)
            m.set_option(_lib.DM_OPT_PRECISION, _lib.DM_PREC_F16X3_ROLES)
            prob_r, cls_r = m.predict_windows(x)
            assert np.array_equal(prob.view(np.uint32), prob_r.view(np.uint32)) and np.array_equal(cls, cls_r), n
    m.close()


def test_bench_distributed_path_single_rank(gpu_device):
    """The N > 1 code path of bench.py exercised with one rank (DM_BENCH_FORCE_DIST=1 under torch.distributed.run, which is
    only the launcher): file rendezvous, ONE persistent RCCL communicator created through the C ABI (dm_comm_create), the
    per-position counters merged with dm_summary_reduce_scatter inside the timed region, barrier and max-over-ranks of the
    elapsed time through the same communicator."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, DM_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    # round 6: ONE JSON line on stdout and nothing else - RCCL's version banner (C stdio, printed when the first communicator of a process is made) goes to stderr
    assert [l for l in res.stdout.splitlines() if l.strip()] == [line], res.stdout[:600]
    assert "librccl" in out["multi_gpu"]["collective_library"] and "ncclGetVersion" in out["multi_gpu"]["collective_library"]
    assert out["n_gpus"] == 1 and out["config"]["forced_dist_dry_run"] is True
    assert out["value"] > 1e6 and out["summary_check"]["touch"] > 0
    mg = out["multi_gpu"]
    assert mg["rccl_nranks"] == 1
    assert mg["collectives"] == 2                      # one untimed warm-up merge of the same size + the timed one
    assert mg["bytes"] == 2 * 12 * 4_641_652           # touch | cov | mod, int32, both times
    assert "ncclReduceScatter" in mg["collective"] or "ncclReduce" in mg["collective"]


def test_bench_keeps_its_line_when_rccl_cannot_be_set_up(gpu_device):
    """RCCL never ran on more than one rank here: if it cannot be set up on the scaling box, the ranks agree to skip the final merge,
    barrier through the rendezvous files and still print the throughput line, with the failure named in it."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, DM_BENCH_FORCE_DIST="1", DM_BENCH_BREAK_RCCL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["value"] > 1e6 and out["summary_check"]["touch"] > 0
    mg = out["multi_gpu"]
    assert mg["collective"].startswith("NOT RUN") and "DM_BENCH_BREAK_RCCL" in mg["rccl_error"]
    assert len(mg["per_rank"]) == 1 and mg["per_rank"][0]["slice_sums"][0] == out["summary_check"]["touch"]


def test_bench_eight_processes_on_one_gpu(gpu_device):
    """The driver's N = 8 launch with eight REAL processes on this one-GPU box (DM_BENCH_ONE_DEVICE=1 puts every rank on device 0):
    torch.distributed.run as the launcher, the file rendezvous between eight processes, the RCCL id from rank 0 to the others.  RCCL refuses
    several ranks on one device (ncclCommInitRank: invalid usage) - on every rank; they then agree to go on without it: barriers and the
    max over ranks through the rendezvous files, one JSON line from rank 0 with all ranks' rates.  (The environment RCCL needs between
    processes, HSA_ENABLE_IPC_MODE_LEGACY=0, is set by the product - deepmod_amd/_lib.py, bench.py - not by this test.)"""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, DM_BENCH_ONE_DEVICE="1")
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    world = 8
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", "29535", os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["config"]["all_ranks_on_device_0_test_hook"] is True and out["scaling"] == "weak"
    mg = out["multi_gpu"]
    assert mg["collective"].startswith("NOT RUN") and "ncclCommInitRank" in mg["rccl_error"]
    assert [r["rank"] for r in mg["per_rank"]] == list(range(world)) and all(r["windows_per_s"] > 1e5 for r in mg["per_rank"])
    assert out["config"]["windows_total"] == world * 2 * 65536
    assert out["summary_check"]["touch"] == sum(r["slice_sums"][0] for r in mg["per_rank"]) > 0
    assert mg["measured_on_hardware_with_more_than_one_rank"] is False


def test_batched_reads_equal_per_read_calls(models, tmp_path, gpu_device):
    """mPredict_batch (one device call for many reads, concatenated feature matrices) must give exactly
    what the reference-granularity mPredict1 gives read by read."""
    import copy
    from deepmod_amd import detect, predstore, synth_reads
    prefix = str(tmp_path / "mod_train_synth")
    synth.write_synthetic_checkpoint(prefix, seed=9, scale=4.0)
    _, init_l, _, _, _, X, Y, _, _, _, _, mfpred = model.mCreateSession(7, 100, 21, {"outputlayer": ""})
    sess = model.new_session(gpu_device)
    sess.restore(prefix)
    files = synth_reads.write_synthetic_run(str(tmp_path / "reads"), n_reads=7, reads_per_file=7, genome_len=20000, seed=4,
                                            chrom="chrS", min_len=200, max_len=900)
    reads = predstore.load_feature_container(files[0])
    sp_options = {"rnn": (sess, X, Y, init_l, mfpred)}
    batch_reads = copy.deepcopy(reads)
    nums_b = detect.mPredict_batch({"windowsize": 21}, sp_options, batch_reads)
    for rd, rb, nb in zip(reads, batch_reads, nums_b):
        n1 = detect.mPredict1({"windowsize": 21}, sp_options, {"f5data": {rd["readk"]: (None, rd["events"], None, "f")}},
                              rd["mfeatures"], rd["base_map_info"], rd["readk"], rd["start_clip"], rd["end_clip"])
        assert n1 == nb and n1 > 0
        assert np.array_equal(rd["base_map_info"]["mod_pred"], rb["base_map_info"]["mod_pred"])
    sess.close()


def test_int8_mode_is_selected_per_model_by_the_calibration_gate(gpu_device):
    """dm_model_calibrate_i8 (round 4): the int8 cross-term mode becomes a model's precision only when the model's OWN calibration run
    (2^18 synthetic windows through the fp32 and the int8 kernel) agrees to 4e-5.  Weights with trained statistics pass (1.2e-5 in the
    tail of 10^6 windows, profiles/r04/i8_tail.txt) and the selected mode is inside the path's 1e-4 against the oracle; U(-a, a) kernels at
    scale 4 - where the mode's tail reaches 7e-5 .. 1.1e-4 - are refused and keep the three-product kernel."""
    from deepmod_amd import _lib
    x = synth.synthetic_windows(30000, seed=91)
    m = model.BiLSTMModel(trained_like_weights(), device=gpu_device, precision="auto")
    assert m.calibration["selected_f16i8"] and m.calibration["max_abs_dp"] <= 4e-5, m.calibration
    assert m.get_info(_lib.DM_INFO_PRECISION) == _lib.DM_PREC_F16I8
    err1 = m.calibration["max_abs_dp"]
    prob, cls = m.predict_windows(x)
    _check(prob, cls, *oracle_np.predict_windows_c(trained_like_weights(), x), tol=TOL)
    m.set_precision("f16x3")
    assert m.calibrate_i8()[0] == err1                       # the calibration windows are the same on every call and every box
    m.close()
    for seed in (17, 26):
        m = model.BiLSTMModel(synth.synthetic_weights(seed, 4.0), device=gpu_device, precision="auto")
        assert not m.calibration["selected_f16i8"] and m.calibration["max_abs_dp"] > 4e-5, m.calibration
        assert m.get_info(_lib.DM_INFO_PRECISION) == _lib.DM_PREC_F16X3
        m.close()
    m = model.BiLSTMModel(synth.synthetic_weights(17, 1.0), device=gpu_device, precision="auto")
    assert m.calibration["selected_f16i8"], m.calibration
    m.close()
    # an explicit precision is never overridden; a model outside the f16 range has nothing to calibrate
    m = model.BiLSTMModel(trained_like_weights(), device=gpu_device, precision="f32")
    assert m.calibrate_i8()[1] is False and m.get_info(_lib.DM_INFO_PRECISION) == _lib.DM_PREC_F32
    m.close()


def _shapes(m):
    """MFMA shapes this library can run the split-f16 modes on: 16 (the product) and, in an experiment build (DM_WITH_F16S=1), 32."""
    from deepmod_amd import _lib
    return (16, 32) if m.get_info(_lib.DM_INFO_HAS_F16S) else (16,)


def test_mfma_shape_option_of_the_split_f16_modes(gpu_device):
    """Round 6: the product runs DM_PREC_F16X3 / DM_PREC_F16I8 on the 16x16x32 / 16x16x64 MFMAs only (lstm16q::bilstm_f16q_kernel<0 | 1>); the
    32x32x16 kernels of rounds 2-3 moved to tools/experiments/f16s and a product library refuses DM_OPT_F16X3_SHAPE = 32.  In an experiment build
    (DM_WITH_F16S=1) both shapes are held to the oracle and to each other as in rounds 4-5: the same arithmetic in a different summation order -
    inside the path's tolerance with an order of magnitude to spare (int8 mode: inside its documented bound) on ragged sizes, three weight sets
    and out-of-range event lengths."""
    from deepmod_amd import _lib
    m = model.BiLSTMModel(synth.synthetic_weights(21, 1.0), device=gpu_device)
    shapes = _shapes(m)
    with pytest.raises(_lib.DeepModHipError):
        m.set_option(_lib.DM_OPT_F16X3_SHAPE, 8)
    m.set_option(_lib.DM_OPT_F16X3_SHAPE, 16)
    if shapes == (16,):
        with pytest.raises(_lib.DeepModHipError) as exc:
            m.set_option(_lib.DM_OPT_F16X3_SHAPE, 32)
        assert 'not part of this build' in str(exc.value)
        with pytest.raises(_lib.DeepModHipError):
            m.set_option(_lib.DM_OPT_PRECISION, _lib.DM_PREC_F16X3_ROLES)
    m.close()
    for prec, cases in (("f16x3", ((synth.synthetic_weights(21, 1.0), 3e-5, TOL), (synth.synthetic_weights(26, 4.0), 3e-5, TOL), (trained_like_weights(), 3e-5, TOL))),
                        ("f16i8", ((synth.synthetic_weights(21, 1.0), 5e-5, TOL_I8), (synth.synthetic_weights(26, 4.0), TOL_I8, TOL_I8), (trained_like_weights(), 5e-5, TOL_I8)))):
        for w, bound, tol in cases:
            ms = []
            for shape in shapes:
                mm = model.BiLSTMModel(w, device=gpu_device, precision=prec)
                mm.set_option(_lib.DM_OPT_F16X3_SHAPE, shape)
                ms.append(mm)
            for n in (1, 15, 16, 17, 31, 32, 33, 127, 129, 4097, 20000):
                x = synth.synthetic_windows(n, seed=300 + n)
                if n == 4097:
                    x[::7, :, 6] = 3.0e6                      # event lengths beyond the f16 range: the rescaled slot of lane group 3
                ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
                got = [mm.predict_windows(x) for mm in ms]
                for p, c in got:
                    assert _check(p, c, ref_prob, ref_cls, tol) <= bound
                if len(got) == 2:
                    assert np.abs(got[0][0] - got[1][0]).max() <= (3e-5 if prec == "f16x3" else TOL_I8)
            for mm in ms:
                mm.close()


def test_calibrated_int8_selection_does_not_survive_a_change_of_the_mfma_shape(gpu_device):
    """dm_model_calibrate_i8 gates the int8 kernel of the shape of that moment; DM_OPT_F16X3_SHAPE afterwards would run the OTHER shape's
    int8 kernel (its own pack and quantisation) uncalibrated - the selection now falls back to DM_PREC_F16X3.  (Only an experiment build has a second shape.)"""
    from deepmod_amd import _lib
    m = model.BiLSTMModel(trained_like_weights(), device=gpu_device)
    err, selected = m.calibrate_i8()
    assert selected and m.get_info(_lib.DM_INFO_PRECISION) == _lib.DM_PREC_F16I8
    m.set_option(_lib.DM_OPT_F16X3_SHAPE, 16)                 # the same shape: nothing changes
    assert m.get_info(_lib.DM_INFO_PRECISION) == _lib.DM_PREC_F16I8
    if m.get_info(_lib.DM_INFO_HAS_F16S):
        m.set_option(_lib.DM_OPT_F16X3_SHAPE, 32)
        assert m.get_info(_lib.DM_INFO_PRECISION) == _lib.DM_PREC_F16X3
    m.close()



def test_selected_mode_on_read_shaped_rows(gpu_device):
    """the mode DEEPMOD_PRECISION=auto selects for the trained-like model (the load-time gate -> int8 cross terms) on
    READ-SHAPED rows - per-read feature matrices of synthetic reads with 5 % tail events (normalised means over the whole clip range,
    a third exactly +-5; event lengths up to 30,000 samples), windows assembled on the device (dm_predict_read) - against the C oracle at
    the PATH's tolerance 1e-4, not the mode's own 2e-4; and the default mode on the same rows (fp32-class).  Reference: myDetect.py:787-834."""
    from conftest import trained_like_weights
    from deepmod_amd import synth_reads
    w = trained_like_weights()
    m_auto = model.BiLSTMModel(w, device=gpu_device, precision="auto")
    assert m_auto.calibration["selected_f16i8"], m_auto.calibration
    m_dflt = model.BiLSTMModel(w, device=gpu_device)
    m_f32 = model.BiLSTMModel(w, device=gpu_device, precision="f32")
    genome = synth_reads.synthetic_genome(30000, 5)
    rng = np.random.default_rng(77)
    worst = {"auto": 0.0, "default": 0.0, "f32": 0.0}
    n_tail = 0
    for i in range(12):
        rd = synth_reads.synthetic_read(rng, genome, 'chrT', 'r%d' % i, min_len=600, max_len=2500, p_tail=0.05)
        rows = np.ascontiguousarray(rd['mfeatures'][:, 3:], np.float32)
        n = rows.shape[0] - 200
        n_tail += int((np.abs(rows[100:-100, 4]) == 5.0).sum() + (rows[100:-100, 6] > 1000).sum())
        xw = np.stack([rows[100 + j - 10:100 + j + 11] for j in range(n)])
        ref_prob, ref_cls = oracle_np.predict_windows_c(w, xw)
        near = np.abs(ref_prob[:, 1] - 0.5) < 1e-4
        for name, mm in (("auto", m_auto), ("default", m_dflt), ("f32", m_f32)):
            prob, cls = mm.predict_read(rows, 100, n)
            worst[name] = max(worst[name], float(np.abs(prob - ref_prob).max()))
            assert not ((cls.astype(np.int64) != ref_cls) & ~near).any(), (name, i)
    assert n_tail > 300                                          # the tail events are really there
    assert worst["auto"] <= 1e-4, worst                          # the path's tolerance, with the int8 cross terms
    # the default mode is fp32-class: on these rows the fp32 kernel itself is ~2e-5 from the oracle (event lengths of 10^4 samples make
    # pre-activations of 10^2..10^3, whose fp32 round-off depends on the summation order), and the default stays within a few 1e-6 of that
    assert worst["f32"] <= 5e-5 and worst["default"] <= worst["f32"] + 1e-5, worst
    m_auto.close(); m_dflt.close(); m_f32.close()


def test_one_call_past_two_to_the_31_input_floats(gpu_device):
    """Maximum sizes: ONE dm_predict_windows call over 15,000,000 windows = 2.2x10^9 input floats (8.8 GB resident in HBM; a 32-bit element
    index wraps at 14.6 M windows).  The input is 15 copies of one block of 10^6 windows, so the size-independent property is periodicity:
    every block's probabilities and classes equal block 0's bit for bit (a wrapped index would land 608,732 windows into a block), and
    block 0 is checked against the oracle on a sample.  All three precisions."""
    from deepmod_amd import _lib
    lib = _lib.load()
    nb, reps = 1_000_000, 15
    x = synth.synthetic_windows(nb, seed=20260928)
    w = synth.synthetic_weights(26, 4.0)
    big = model.DeviceArray((reps * nb, 21, 7), np.float32, gpu_device)
    for r in range(reps):
        _lib.check(lib.dm_memcpy_h2d(gpu_device, big.ptr + r * x.nbytes, x.ctypes.data, x.nbytes))
    assert big.nbytes // 4 > 2 ** 31
    d_prob = model.DeviceArray((reps * nb, 2), np.float32, gpu_device)
    d_cls = model.DeviceArray((reps * nb,), np.uint8, gpu_device)
    pick = np.random.default_rng(3).choice(nb, 2048, replace=False)
    ref_p, ref_c = oracle_np.predict_windows_c(w, x[pick])
    for prec, tol in (("f16x3", TOL), ("f32", TOL), ("f16i8", TOL_I8)):
        m = model.BiLSTMModel(w, device=gpu_device, precision=prec)
        m.predict_windows(big, prob=d_prob, cls=d_cls)
        p = d_prob.to_host().reshape(reps, nb, 2)
        c = d_cls.to_host().reshape(reps, nb)
        for r in range(1, reps):
            assert np.array_equal(p[r].view(np.uint32), p[0].view(np.uint32)) and np.array_equal(c[r], c[0]), (prec, r)
        _check(p[0][pick], c[0][pick], ref_p, ref_c, tol)
        m.close()
    for a in (big, d_prob, d_cls):
        a.free()
