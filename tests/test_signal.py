"""Raw-signal normalisation + per-event statistics (reference myDetect.py:266-282, :332-343).

CPU: the numpy oracle against the golden vectors produced by the reference's own functions.
GPU: the HIP path (through the C ABI) against the goldens and against the oracle on larger seeded inputs —
bit-exact: the outputs are float32 roundings of float64 values computed in numpy's own operation order."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import signal_oracle

CASES = ['typical', 'even_slice', 'long_events', 'mid_events', 'outliers', 'clamped_tail', 'empty_late', 'empty_early']
EVENT_DTYPE = [("mean", "<f4"), ("stdv", "<f4"), ("start", np.uint64), ("length", np.uint64), ("model_state", "U5")]


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "host_signal.npz"))


def _events(g, name):
    ev = np.zeros(len(g[name + '.start']), dtype=EVENT_DTYPE)
    ev['start'] = g[name + '.start']
    ev['length'] = g[name + '.length']
    return ev


def _apply_reference_rule(ev, mean, stdv, first_empty):
    """what the reference's loop leaves behind (myDetect.py:332-343)"""
    ev = ev.copy()
    ev['mean'][:first_empty] = mean[:first_empty]
    ev['stdv'][:first_empty] = stdv[:first_empty]
    if first_empty < len(ev) and first_empty > 500:
        ev = ev[:first_empty - 1]
    return ev


def _same_f32(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(golden, name):
    raw = golden[name + '.raw']
    ev = _events(golden, name)
    sig, norm = signal_oracle.mnormalized(raw, ev)
    mean, stdv, first_empty = signal_oracle.event_stats(sig, ev)
    out = _apply_reference_rule(ev, mean, stdv, first_empty)
    assert len(out) == int(golden[name + '.n_kept'])
    assert _same_f32(out['mean'], golden[name + '.mean'])
    assert _same_f32(out['stdv'], golden[name + '.stdv'])
    if name + '.signal' in golden:
        assert np.array_equal(sig, golden[name + '.signal'])


def _random_case(seed, n_raw, mean_len, loc=480.0, scale=70.0):
    rng = np.random.default_rng(seed)
    raw = np.clip(np.round(rng.normal(loc, scale, n_raw)), -32768, 32767).astype(np.int16)
    raw[rng.integers(0, n_raw, n_raw // 500)] = rng.integers(-32768, 32767, n_raw // 500)
    lens = rng.geometric(1.0 / mean_len, int(n_raw / mean_len * 1.2)).astype(np.uint64)
    first = int(rng.integers(0, 200))
    start = (first + np.concatenate([[0], np.cumsum(lens[:-1])])).astype(np.uint64)
    keep = (start + lens) <= n_raw - 3
    return raw, start[keep], lens[keep]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_golden(golden, name):
    from deepmod_amd import signal
    sp = {'raw_signals': golden[name + '.raw'], 'm_event': _events(golden, name), 'mfile_path': name}
    signal.mnormalized_event_stats({}, sp, want_signal=True)
    ev = sp['m_event']
    assert len(ev) == int(golden[name + '.n_kept'])
    assert _same_f32(ev['mean'], golden[name + '.mean'])
    assert _same_f32(ev['stdv'], golden[name + '.stdv'])
    if name + '.signal' in golden:
        assert np.array_equal(sp['raw_signals'], golden[name + '.signal'])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_raw,mean_len", [(1, 200_000, 9.0), (2, 1_000_003, 10.0), (3, 400_000, 300.0),
                                                 (4, 65_536, 2.0), (5, 300_000, 20000.0)])
def test_hip_matches_oracle_bit_exact(seed, n_raw, mean_len):
    from deepmod_amd import signal
    raw, start, length = _random_case(seed, n_raw, mean_len)
    ev = np.zeros(len(start), dtype=EVENT_DTYPE)
    ev['start'], ev['length'] = start, length
    sig, norm = signal_oracle.mnormalized(raw, ev)
    mean, stdv, first_empty = signal_oracle.event_stats(sig, ev)
    nz = signal.SignalNormalizer(0)
    hmean, hstdv, hnorm, hfirst, hsig = nz.event_stats(raw, start, length, want_signal=True)
    nz.close()
    assert hfirst == first_empty == len(ev)
    for k in norm:
        assert hnorm[k] == norm[k], (k, hnorm[k], norm[k])
    assert np.array_equal(hsig, sig)
    assert _same_f32(hmean, mean)
    assert _same_f32(hstdv, stdv)


@pytest.mark.gpu
def test_hip_device_resident_signal_and_errors():
    from deepmod_amd import _lib, model, signal
    raw, start, length = _random_case(9, 50_000, 8.0)
    nz = signal.SignalNormalizer(0)
    m0, s0, n0, f0, _ = nz.event_stats(raw, start, length)
    d_raw = model.DeviceArray.from_host(raw, 0)                       # same call on a device-resident signal
    mean = np.empty(len(start), np.float32)
    stdv = np.empty(len(start), np.float32)
    lib = _lib.load()
    _lib.check(lib.dm_signal_event_stats(nz._h, d_raw.ptr, len(raw), start.ctypes.data, length.ctypes.data, len(start),
                                         mean.ctypes.data, stdv.ctypes.data, None, None, None))
    assert _same_f32(mean, m0) and _same_f32(stdv, s0)
    with pytest.raises(_lib.DeepModHipError):                          # events that cover no signal
        nz.event_stats(raw, start + np.uint64(10 ** 7), length)
    with pytest.raises(ValueError):
        nz.event_stats(raw.astype(np.float32), start, length)
    # a first start of 2^63 and more (a damaged table) is a large index - the slice is empty, as numpy cuts it - not a negative offset into the samples
    for huge in (2 ** 63 + 7, 2 ** 64 - 5):
        bad = start.copy()
        bad[0] = huge
        with pytest.raises(_lib.DeepModHipError, match='cover no signal'):
            nz.event_stats(raw, bad, length)
        with pytest.raises(_lib.DeepModHipError, match='cover no signal'):
            nz.event_stats_batch([(raw, start, length), (raw, bad, length)])
    d_raw.free()
    nz.close()


def test_plan_of_a_batch_treats_event_starts_as_unsigned():
    """dm_signal_plan_batch (host arithmetic, no device): the slice a read's events cover is raw[start_0 : start_last + length_last] in numpy's uint64
    arithmetic (myDetect.py:272) - a start of 2^63 and more clamps to the end of the read (an empty slice, the read is refused), a sum that wraps is a
    small index; inside a read such an event is an empty event and the first of them is `first_empty`."""
    from deepmod_amd import _lib
    lib = _lib.load()
    raw_off, ev_off = np.array([0, 1000], np.int64), np.array([0, 4], np.int64)
    length = np.array([10, 10, 10, 10], np.uint64)
    fe = np.zeros(1, np.int64)
    plan = lambda st, ln=length: lib.dm_signal_plan_batch(1, raw_off.ctypes.data, ev_off.ctypes.data, st.ctypes.data, ln.ctypes.data, fe.ctypes.data)
    assert plan(np.array([0, 10, 20, 30], np.uint64)) == 0 and fe[0] == 4
    for huge in (2 ** 63 + 7, 2 ** 64 - 5):
        assert plan(np.array([huge, 10, 20, 30], np.uint64)) != 0 and 'cover no signal' in _lib.last_error()
        assert plan(np.array([0, huge, 20, 30], np.uint64)) == 0 and fe[0] == 1            # an empty event inside the read
    assert plan(np.array([8, 10, 20, 30], np.uint64), np.array([2, 10, 10, 2 ** 64 - 25], np.uint64)) != 0      # 30 + (2^64 - 25) wraps to 5: the end lies before the start
    assert plan(np.array([0, 10, 20, 30], np.uint64), np.array([10, 2 ** 64 - 5, 10, 10], np.uint64)) == 0 and fe[0] == 1


@pytest.mark.gpu
def test_batched_call_is_bit_identical_to_per_read_calls():
    """dm_signal_event_stats_batch (all reads of a worker batch in one device round trip) against n per-read calls:
    every mean / stdv / median / limit bit for bit, the first-empty index of a read whose last events run off its
    signal, and the per-read fallback when one read of the list cannot be processed."""
    from deepmod_amd import _lib, signal
    cases = [_random_case(20 + i, n, ml) for i, (n, ml) in enumerate([(120_000, 9.0), (65_536, 2.0), (300_000, 300.0), (90_001, 11.0),
                                                                      (1_000_003, 10.0), (7_000, 5.0)])]
    raw, start, length = cases[3]
    start, length = start.copy(), length.copy()
    length[-3:] = 50                                                   # the last events end past the signal / start at its end
    start[-1] = np.uint64(len(raw))
    cases[3] = (raw, start, length)
    nz = signal.SignalNormalizer(0)
    single = [nz.event_stats(r, s, l) for r, s, l in cases]
    batch = nz.event_stats_batch(cases)
    assert len(batch) == len(cases)
    for (m1, s1, n1, f1, _), (m2, s2, n2, f2) in zip(single, batch):
        assert f1 == f2
        assert n1 == n2
        assert _same_f32(m1, m2) and _same_f32(s1, s2)
    assert batch[3][3] == len(cases[3][1]) - 1 and np.isnan(batch[3][0][-1])
    # the sp_param-level entry: same event tables as the per-read entry, including the error isolation
    def params():
        out = []
        for i, (r, s, l) in enumerate(cases):
            ev = np.zeros(len(s), dtype=EVENT_DTYPE)
            ev['start'], ev['length'] = s, l
            out.append({'raw_signals': r, 'm_event': ev, 'mfile_path': 'case%d' % i})
        return out
    a, b = params(), params()
    for sp in a:
        signal.mnormalized_event_stats({}, sp, nz)
    errs = signal.mnormalized_event_stats_batch({}, b, nz)
    assert errs == [None] * len(cases)
    for x, y in zip(a, b):
        assert len(x['m_event']) == len(y['m_event'])
        assert _same_f32(x['m_event']['mean'], y['m_event']['mean']) and _same_f32(x['m_event']['stdv'], y['m_event']['stdv'])
        assert x['norm'] == y['norm']
    c = params()
    c[2]['m_event']['start'] += np.uint64(10 ** 9)                     # this read's events cover no signal
    errs = signal.mnormalized_event_stats_batch({}, c, nz)
    assert isinstance(errs[2], _lib.DeepModHipError) and all(e is None for i, e in enumerate(errs) if i != 2)
    assert _same_f32(c[0]['m_event']['mean'], a[0]['m_event']['mean']) and _same_f32(c[5]['m_event']['stdv'], a[5]['m_event']['stdv'])
    nz.close()


@pytest.mark.gpu
def test_resident_statistics_block_equals_the_batched_call():
    """Round 6 (dm_signal_plan_batch + dm_signal_event_stats_device): the statistics that stay on the device are, bit for bit, what the batched call
    returns to the host with the reference's rule applied (events before a read's first empty event take the signal's statistics, the ones behind keep
    the basecaller's, myDetect.py:334-340), next to float(length) - the three values get_Feature copies into a feature row (:892-900).  first_empty
    from the host-only planner equals the batched call's; a read with an empty event and no fall-back values is refused; a length beyond the
    split-f16 kernels' range raises the range flag; a read that makes the device order statistics step aside (constant signal) takes the same path
    as in the batched call."""
    from deepmod_amd import _lib, signal
    from deepmod_amd.model import DeviceArray
    cases = [_random_case(60 + i, n, ml) for i, (n, ml) in enumerate([(120_000, 9.0), (65_536, 2.0), (90_001, 11.0), (300_000, 300.0), (7_000, 5.0)])]
    raw, start, length = cases[2]
    start, length = start.copy(), length.copy()
    start[700] = np.uint64(len(raw) + 5)                                 # an empty slice in the middle (> 500: the reference cuts the table there)
    cases[2] = (raw, start, length)
    raw, start, length = cases[4]
    start, length = start.copy(), length.copy()
    start[40] = np.uint64(len(raw))                                      # ... and one at index 40 (<= 500: the table keeps its length)
    cases[4] = (raw, start, length)
    nz = signal.SignalNormalizer(0)
    batch = nz.event_stats_batch(cases)
    raw_off = np.concatenate([[0], np.cumsum([len(c[0]) for c in cases])]).astype(np.int64)
    ev_off = np.concatenate([[0], np.cumsum([len(c[1]) for c in cases])]).astype(np.int64)
    raw_all = np.concatenate([c[0] for c in cases])
    st, ln = np.concatenate([c[1] for c in cases]), np.concatenate([c[2] for c in cases])
    n_ev = int(ev_off[-1])
    rng = np.random.default_rng(5)
    fb_mean, fb_stdv = rng.normal(0, 1, n_ev).astype(np.float32), rng.random(n_ev).astype(np.float32)
    blk = DeviceArray((n_ev, 3), np.float32, 0)
    try:
        nz.event_stats_device(raw_all, raw_off, st, ln, ev_off, blk.ptr)
        raise AssertionError('a read with an empty event needs fall-back values')
    except _lib.DeepModHipError as exc:
        assert 'fall-back' in str(exc)
    fe, flag = nz.event_stats_device(raw_all, raw_off, st, ln, ev_off, blk.ptr, fb_mean, fb_stdv)
    got = blk.to_host()
    assert flag == 0
    assert fe.tolist() == [b[3] for b in batch] and fe[2] == 700 and fe[4] == 40
    for r, (mean, stdv, _, f) in enumerate(batch):
        e0, e1 = int(ev_off[r]), int(ev_off[r + 1])
        want_mean, want_stdv = fb_mean[e0:e1].copy(), fb_stdv[e0:e1].copy()
        want_mean[:f], want_stdv[:f] = mean[:f], stdv[:f]
        assert _same_f32(got[e0:e1, 0], want_mean) and _same_f32(got[e0:e1, 1], want_stdv)
        assert _same_f32(got[e0:e1, 2], ln[e0:e1].astype(np.float64).astype(np.float32))
    # a stalled event of 70,000 samples: representable by the fp32 kernel only
    ln2 = ln.copy()
    ln2[10] = 70_000
    _, flag = nz.event_stats_device(raw_all, raw_off, st, ln2, ev_off, blk.ptr, fb_mean, fb_stdv)
    assert flag == 1
    # the host order statistics (constant signal: division by zero like numpy) behind the same call
    const = np.full(30_000, 612, np.int16)
    cl = np.full(3000, 10, np.uint64)
    cs = (np.arange(3000) * 10).astype(np.uint64)
    with np.errstate(all='ignore'):
        ref = nz.event_stats_batch([cases[0], (const, cs, cl)])
        ro2 = np.array([0, len(cases[0][0]), len(cases[0][0]) + len(const)], np.int64)
        eo2 = np.array([0, len(cases[0][1]), len(cases[0][1]) + len(cs)], np.int64)
        blk2 = DeviceArray((int(eo2[-1]), 3), np.float32, 0)
        _, flag = nz.event_stats_device(np.concatenate([cases[0][0], const]), ro2, np.concatenate([cases[0][1], cs]), np.concatenate([cases[0][2], cl]), eo2, blk2.ptr)
        got2 = blk2.to_host()
    assert flag == 1                                                       # NaN statistics: the fp32 kernel reproduces what the reference would feed its graph
    assert _same_f32(got2[:eo2[1], 0], ref[0][0]) and _same_f32(got2[eo2[1]:, 0], ref[1][0]) and _same_f32(got2[eo2[1]:, 1], ref[1][1])
    blk.free()
    blk2.free()
    nz.close()


@pytest.mark.gpu
def test_device_order_statistics_equal_host_order_statistics(monkeypatch):
    """The batched call computes the four medians of every read on the device (signal_norm_batch_kernel); with
    DEEPMOD_SIGNAL_HOST_NORM=1 it takes the host path (norm_from_hist).  Every median, limit, mean and stdv must agree bit for
    bit on odd and even sample counts, narrow and wide value ranges, and on the reads that make the device path step aside
    (a constant signal; more distinct values than the kernel compacts)."""
    from deepmod_amd import signal
    rng = np.random.default_rng(77)
    cases = [_random_case(40 + i, n, ml) for i, (n, ml) in enumerate([(100_001, 9.0), (100_000, 9.0), (7_001, 4.0), (400_000, 30.0)])]
    def events(n, mean_len):
        length = np.maximum(1, rng.poisson(mean_len, max(2, int(n / mean_len) - 2))).astype(np.uint64)
        start = np.concatenate([[3], 3 + np.cumsum(length)[:-1]]).astype(np.uint64)
        keep = start + length <= n
        return start[keep], length[keep]
    narrow = rng.integers(500, 504, 50_000).astype(np.int16)                       # four distinct values
    wide = rng.integers(-20_000, 20_000, 300_000).astype(np.int16)                 # ~40,000 distinct values: host path
    constant = np.full(30_000, 612, np.int16)                                      # zero scale: host path (division by zero as numpy)
    two = np.where(rng.random(40_001) < 0.5, 600, 601).astype(np.int16)            # medians between two bins
    for arr in (narrow, wide, two):
        cases.append((arr,) + events(len(arr), 8.0))
    lists = {'plain': cases, 'with_constant': cases[:2] + [(constant,) + events(len(constant), 8.0)]}
    for name, reads in lists.items():
        monkeypatch.delenv('DEEPMOD_SIGNAL_HOST_NORM', raising=False)
        dev = signal.SignalNormalizer(0)
        monkeypatch.setenv('DEEPMOD_SIGNAL_HOST_NORM', '1')
        host = signal.SignalNormalizer(0)
        with np.errstate(all='ignore'):
            a = dev.event_stats_batch(reads)
            b = host.event_stats_batch(reads)
        for i, ((m1, s1, n1, f1), (m2, s2, n2, f2)) in enumerate(zip(a, b)):
            assert f1 == f2, (name, i)
            for k in n1:
                assert n1[k] == n2[k] or (np.isnan(n1[k]) and np.isnan(n2[k])), (name, i, k, n1[k], n2[k])
            assert _same_f32(m1, m2) and _same_f32(s1, s2), (name, i)
        dev.close()
        host.close()
