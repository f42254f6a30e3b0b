"""ORACLE — TEST INFRASTRUCTURE ONLY.  Pure-Python (loop-level) restatements of the reference's
host-side logic on the hot path, pinned to tests/golden/host_*.{npz,json} (which were produced by
running the reference's own functions, tests/golden/make_golden_host.py).

  mpredict1_oracle   /root/reference/bin/DeepMod_scripts/myDetect.py:787-834
  sum_handler_oracle /root/reference/bin/DeepMod_scripts/myDetect.py:1089-1120
"""
from __future__ import annotations

import numpy as np


def mpredict1_oracle(mfeatures, readbase, ev_bases, start_clip, end_clip, classify, windowsize=21, batch=512):
    """Returns (batch sizes, pred_mod_num, mod_pred column).  `classify(x[k,21,7]) -> int[k]`."""
    n_events = len(ev_bases)
    tx = mfeatures[:, 3:]
    m_data = []
    for ie in range(start_clip - 100, n_events - end_clip + 100):        # :794
        mind = ie - (start_clip - 100)                                     # :795
        if ie >= start_clip and ie < n_events - end_clip:                 # :796
            m_data.append(tx[mind - windowsize // 2: mind + windowsize // 2 + 1])   # :799
    test_feature = np.reshape(m_data, (len(m_data), windowsize, tx.shape[1]))      # :802
    if len(test_feature) > batch * 1.2:                                   # :808
        groups = np.array_split(test_feature, int(len(test_feature) / batch))     # :809
    else:
        groups = [test_feature]
    out = np.concatenate([classify(g) for g in groups])                    # :814-820
    mod_pred = np.zeros(len(readbase), dtype=np.int64)
    aligni = 0
    pred_mod_num = 0
    for ie in range(start_clip, n_events - end_clip):                     # :825
        while readbase[aligni] == '-':                                    # :826
            aligni += 1
        assert readbase[aligni] == ev_bases[ie]                           # :827
        if out[ie - start_clip] == 1:                                     # :829
            mod_pred[aligni] = 1
            pred_mod_num += 1
        aligni += 1
    return [len(g) for g in groups], pred_mod_num, mod_pred


def sum_handler_oracle(chrom, strand, base, reads):
    """reads: list of dicts with refbase/readbase strings, refbasei and mod_pred lists -> BED bytes."""
    table = {}
    for rd in reads:
        for mi in range(len(rd['refbase'])):
            rb = rd['refbase'][mi]
            if rb != base:                                                # :1091
                continue
            if rb in ['-', 'N', 'n']:                                     # :1092
                continue
            key = (chrom, strand, int(rd['refbasei'][mi]))
            if key not in table:                                          # :1093-1094
                table[key] = [0, 0, rb]
            if rd['readbase'][mi] != '-':                                 # :1097
                table[key][0] += 1
                if -0.1 < rd['mod_pred'][mi] - 1 < 0.1:                   # :1099
                    table[key][1] += 1
    out = []
    for pk in sorted(table):                                              # :1112-1120
        cov, mod, na = table[pk]
        out.append(' '.join([pk[0], str(pk[2]), str(pk[2] + 1), na, str(1000 if cov > 1000 else cov), pk[1],
                             str(pk[2]), str(pk[2] + 1), '0,0,0', str(cov),
                             ('%d' % (100 * mod / (cov if cov > 0 else 1))), str(mod), '\n']))
    return ''.join(out).encode('ascii')


def get_feature_oracle(ev_mean, ev_stdv, ev_length, ev_bases, refbase, readbase, refbasei, start_clip, end_clip, strand,
                       mapped_start_pos, num_insertions):
    """Loop-level restatement of the reference's get_Feature for fnum = 7
    (/root/reference/bin/DeepMod_scripts/myDetect.py:839-903).  -> (mfeatures float64[rows, 10], mismatch flag).
    PARITY PIN: tests/golden/host_getfeature.npz (tests/test_features.py)."""
    import numpy as np
    nev = len(ev_bases)
    mfeatures = np.zeros((nev - end_clip + 100 - (start_clip - 100), 10))                 # :850-851
    align_ref_pos = mapped_start_pos if strand == '+' else mapped_start_pos + len(refbase) - num_insertions - 1   # :843-846
    aligni = 0
    isdif = False
    for ie in range(start_clip - 100, nev - end_clip + 100):                              # :855
        row = ie - (start_clip - 100)
        cur_base = ''
        if start_clip <= ie < nev - end_clip:                                             # :857
            while readbase[aligni] == '-':                                                # :861-867
                if refbase[aligni] != '-':
                    align_ref_pos += 1 if strand == '+' else -1
                aligni += 1
            if readbase[aligni] != ev_bases[ie]:                                          # :868-874
                if aligni > 50:
                    break
                isdif = True
            mfeatures[row][0] = align_ref_pos                                             # :875
            cur_base = refbase[aligni]
            if refbase[aligni] != '-':                                                    # :879-881
                align_ref_pos += 1 if strand == '+' else -1
            aligni += 1
        if 0 <= ie < nev:                                                                 # :892-900
            if cur_base in 'ACGT' and cur_base != '':
                mfeatures[row][3 + 'ACGT'.index(cur_base)] = 1
            mfeatures[row][7] = ev_mean[ie]
            mfeatures[row][8] = ev_stdv[ie]
            mfeatures[row][9] = ev_length[ie]
    return mfeatures, isdif
