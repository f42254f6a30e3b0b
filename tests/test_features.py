"""CPU: get_Feature counterpart vs outputs recorded from the reference's own get_Feature
(tests/golden/make_golden_host.py; '+' and '-' reads with clips and indels)."""
import os
from collections import defaultdict

import numpy as np
import pytest

from conftest import GOLDEN
from deepmod_amd import features, predstore

G = np.load(os.path.join(GOLDEN, 'host_getfeature.npz'))


@pytest.mark.parametrize('k', range(int(G['n_cases'])))
def test_get_feature_matches_reference(k):
    ev = predstore.events_from_bases([s[2] for s in G['g%d_model_state' % k]], G['g%d_ev_mean' % k],
                                     G['g%d_ev_stdv' % k], G['g%d_ev_length' % k])
    bmi = predstore.make_base_map_info(G['g%d_bmi_refbase' % k], G['g%d_bmi_readbase' % k], G['g%d_bmi_refbasei' % k])
    sc, ec = [int(v) for v in G['g%d_clips' % k]]
    nins, ndel = [int(v) for v in G['g%d_indels' % k]]
    sp_param = {'f5data': {'r': (None, ev, None, 'f')}, 'f5status': ''}
    mf, isdif = features.get_Feature({'fnum': 7}, {'Error': defaultdict(list)}, sp_param, None, sp_param['f5data'], 'r',
                                     sc, ec, bmi, str(G['g%d_strand' % k]), 'chrS', int(G['g%d_mapped_start' % k]),
                                     nins, ndel)
    assert not isdif
    want = G['g%d_mfeatures' % k]
    assert mf.shape == want.shape
    assert np.array_equal(mf, want)


@pytest.mark.parametrize('k', range(int(G['n_cases'])))
def test_get_feature_oracle_matches_reference(k):
    from oracle import detect_oracle
    sc, ec = [int(v) for v in G['g%d_clips' % k]]
    nins, ndel = [int(v) for v in G['g%d_indels' % k]]
    mf, isdif = detect_oracle.get_feature_oracle(
        G['g%d_ev_mean' % k], G['g%d_ev_stdv' % k], G['g%d_ev_length' % k], [s[2] for s in G['g%d_model_state' % k]],
        list(G['g%d_bmi_refbase' % k]), list(G['g%d_bmi_readbase' % k]), G['g%d_bmi_refbasei' % k], sc, ec,
        str(G['g%d_strand' % k]), int(G['g%d_mapped_start' % k]), nins)
    assert not isdif
    assert np.array_equal(mf, G['g%d_mfeatures' % k])


def test_compiled_feature_rows_match_reference_get_feature():
    """The feature rows dm_rows_emit builds for reads that enter with an alignment table and an event table (dm_rows_add_mapped; the
    same code writes the rows of dm_rows_add_raw) against the reference's own get_Feature output on the four golden reads: columns
    3..9 of mfeatures (one-hot of the strand-corrected reference base, mean, stdv, length) as fp32, positions of the aligned rows."""
    import ctypes
    from deepmod_amd import _lib
    lib = _lib.load()
    n = int(G['n_cases'])
    S1 = lambda a: np.array([str(v).encode('ascii') for v in a], 'S1')
    cat = lambda parts, dt: np.ascontiguousarray(np.concatenate(parts), dt)
    off = lambda parts: np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    refb = [S1(G['g%d_bmi_refbase' % k]) for k in range(n)]
    readb = [S1(G['g%d_bmi_readbase' % k]) for k in range(n)]
    refi = [G['g%d_bmi_refbasei' % k].astype(np.int64) for k in range(n)]
    evb = [S1([s[2] for s in G['g%d_model_state' % k]]) for k in range(n)]
    arrs = dict(bmi_off=off(refb), refb=cat(refb, 'S1'), readb=cat(readb, 'S1'), refi=cat(refi, np.int64),
                sc=np.array([G['g%d_clips' % k][0] for k in range(n)], np.int64), ec=np.array([G['g%d_clips' % k][1] for k in range(n)], np.int64),
                contig=np.zeros(n, np.int32), strand=np.array([0 if str(G['g%d_strand' % k]) == '+' else 1 for k in range(n)], np.int32),
                mev_off=off(evb), mean=cat([G['g%d_ev_mean' % k] for k in range(n)], np.float32),
                stdv=cat([G['g%d_ev_stdv' % k] for k in range(n)], np.float32),
                length=cat([G['g%d_ev_length' % k] for k in range(n)], np.uint64), base=cat(evb, 'S1'))
    h = lib.dm_rows_create(b'C')
    try:
        order = ('bmi_off', 'refb', 'readb', 'refi', 'sc', 'ec', 'contig', 'strand', 'mev_off', 'mean', 'stdv', 'length', 'base')
        _lib.check(lib.dm_rows_add_mapped(h, n, len(arrs['refb']), len(arrs['mean']), 1, *[arrs[k].ctypes.data for k in order], None, None, None))
        R, T, S = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        info = np.zeros((n, 8), np.int64)
        assert lib.dm_rows_info(h, ctypes.byref(R), ctypes.byref(T), ctypes.byref(S), info.ctypes.data, None, 0, None) == n
        assert (info[:, 0] == 0).all(), info[:, 0]
        rows = np.empty((R.value, 7), np.float32)
        pos, flags = np.empty(T.value, np.int64), np.empty(T.value, np.uint8)
        groups = np.zeros((4, 8), np.int64)
        ng = lib.dm_rows_emit(h, None, rows.ctypes.data, None, pos.ctypes.data, flags.ctypes.data, groups.ctypes.data, 4, None, 0, None)
        assert ng == 2                                    # one contig, both strands: '+' reads first, '-' reads after them
    finally:
        lib.dm_rows_destroy(h)
    # emit order: stable by (contig, strand): the '+' cases in file order, then the '-' cases
    order_k = [k for k in range(n) if str(G['g%d_strand' % k]) == '+'] + [k for k in range(n) if str(G['g%d_strand' % k]) == '-']
    r0 = 0
    for k in order_k:
        want = G['g%d_mfeatures' % k]
        got = rows[r0:r0 + len(want)]
        assert np.array_equal(got, want[:, 3:].astype(np.float32)), k
        nal = len(G['g%d_model_state' % k]) - int(G['g%d_clips' % k][0]) - int(G['g%d_clips' % k][1])
        assert np.array_equal(pos[r0 + 100:r0 + 100 + nal], want[100:100 + nal, 0].astype(np.int64)), k     # column 0: the reference position
        r0 += len(want)
    assert r0 == R.value


def test_events_merge_rounds_like_numpy_bit_for_bit():
    """getEvent stores round(mean, 3) / round(stdv, 3) (numpy.round: rint(x * 1000) / 1000 in float64) into '<f4' fields (myDetect.py:241-249).
    dm_events_merge's rint goes through the 1.5 * 2^52 constant; ADVICE r05: (v + M) - M is +0.0 where numpy gives -0.0 for inputs in (-0.0005, 0) -
    the sign is put back, so the float32 BITS equal numpy's, also on exact ties (round half to even) and on values beyond the constant's range."""
    import numpy as np
    from deepmod_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    special = np.array([-0.0004, -0.00049999, -0.0005, -0.0, 0.0, 0.0004, 0.0005, 0.0015, 0.0025, -0.0015, -0.0025, 1.0005, 2.0015, 123.4565, -123.4565,
                        1e-300, -1e-300, 4.5e12, -4.5e12, 3e15, -3e15, 0.5 / 1000, 1.5 / 1000, 2.5 / 1000, -2.5 / 1000], np.float64)
    mean = np.concatenate([special, rng.normal(0, 50, 4000), rng.normal(0, 1e-3, 2000), (rng.integers(-4000, 4000, 2000) + 0.5) / 1000.0])
    stdv = np.abs(mean[::-1]).copy()
    n = len(mean)
    ev_off = np.array([0, n], np.int64)
    start = np.arange(n, dtype=np.uint64) * 10
    length = np.full(n, 10, np.uint64)
    move = np.ones(n, np.int64)
    ms = np.zeros((n, 5), np.uint32)
    ms[:, 2] = ord('A')
    mev_off = np.empty(2, np.int64)
    m_mean, m_stdv = np.empty(n, np.float32), np.empty(n, np.float32)
    m_start, m_len, m_base = np.empty(n, np.uint64), np.empty(n, np.uint64), np.empty(n, 'S1')
    got = lib.dm_events_merge(1, n, ev_off.ctypes.data, mean.ctypes.data, stdv.ctypes.data, start.ctypes.data, length.ctypes.data, ms.ctypes.data, 5,
                              move.ctypes.data, mev_off.ctypes.data, m_mean.ctypes.data, m_stdv.ctypes.data, m_start.ctypes.data, m_len.ctypes.data, m_base.ctypes.data)
    assert got == n
    want_mean, want_stdv = np.round(mean, 3).astype(np.float32), np.round(stdv, 3).astype(np.float32)
    assert np.array_equal(m_mean.view(np.uint32), want_mean.view(np.uint32)), np.flatnonzero(m_mean.view(np.uint32) != want_mean.view(np.uint32))[:5]
    assert np.array_equal(m_stdv.view(np.uint32), want_stdv.view(np.uint32))
    assert np.signbit(m_mean[0]) and m_mean[0] == 0.0                 # -0.0004 -> -0.0, as numpy
    # the values are optional: without them the call produces the same tables (start, length, base) and never reads mean / stdv
    s2, l2, b2 = np.empty(n, np.uint64), np.empty(n, np.uint64), np.empty(n, 'S1')
    assert lib.dm_events_merge(1, n, ev_off.ctypes.data, None, None, start.ctypes.data, length.ctypes.data, ms.ctypes.data, 5, move.ctypes.data,
                               mev_off.ctypes.data, None, None, s2.ctypes.data, l2.ctypes.data, b2.ctypes.data) == n
    assert np.array_equal(s2, m_start) and np.array_equal(l2, m_len) and np.array_equal(b2, m_base)
