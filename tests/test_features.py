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
