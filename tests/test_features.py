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
