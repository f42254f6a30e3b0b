"""Per-read feature matrix, counterpart of the reference's get_Feature
(bin/DeepMod_scripts/myDetect.py:839-903) for fnum = 7, vectorised.

mfeatures[row] = [ref position, label0, label1, A, C, G, T one-hot of the (strand-corrected)
reference base, mean, stdv, length];  rows cover events ie in [start_clip-100, len-end_clip+100):
100 zero rows of padding each side, clipped-but-present events carry signal features without a
one-hot, aligned events carry both ("Feature semantics").
"""
from __future__ import annotations

import numpy as np

g_ACGT = ['A', 'C', 'G', 'T']   # myCom.py g_ACGT


def _event_bases(model_state) -> np.ndarray:
    """model_state[2] of every event as a 'U1' array (no Python loop over the events)."""
    ms = np.ascontiguousarray(model_state)
    if len(ms) == 0:
        return np.zeros(0, 'U1')
    return ms.view('U1').reshape(len(ms), ms.dtype.itemsize // 4)[:, 2]


def get_Feature(moptions, sp_options, sp_param, f5align, f5data, readk, start_clip, end_clip, base_map_info,
                forward_reverse, rname, mapped_start_pos, num_insertions, num_deletions):
    if moptions['fnum'] != 7:
        raise ValueError('only fnum=7 is built (the 57-feature histogram variant is unused by the shipped models)')
    modevents = sp_param['f5data'][readk][1]
    nev = len(modevents)
    n = nev - end_clip - start_clip                       # aligned events
    nrow = nev - end_clip + 100 - (start_clip - 100)
    mfeatures = np.zeros((nrow, 10))
    refb = base_map_info['refbase']
    readb = base_map_info['readbase']
    step = 1 if forward_reverse == '+' else -1
    start = mapped_start_pos if forward_reverse == '+' else mapped_start_pos + len(base_map_info) - num_insertions - 1
    has_ref = refb != '-'
    # reference position seen at entry e = start +- (#entries before e that consume a reference base)
    refpos = start + step * (np.cumsum(has_ref) - has_ref)
    aligned = np.flatnonzero(readb != '-')[:n]
    isdif = False
    ev_base = _event_bases(modevents['model_state'][start_clip:start_clip + n])
    bad = np.flatnonzero(readb[aligned] != ev_base)
    if len(bad):                                          # myDetect.py:868-874
        print('Error Does not match', readb[aligned[bad[0]]], ev_base[bad[0]], aligned[bad[0]], bad[0] + start_clip)
        sp_param['f5status'] = "Error Does not match"
        if f5data[readk][3] not in sp_options["Error"]['Error Does not match']:
            sp_options["Error"]['Error Does not match'].append(f5data[readk][3])
        isdif = True
    rows = 100 + np.arange(n)                             # cur_row_num of ie = start_clip + k
    mfeatures[rows, 0] = refpos[aligned]
    cur_base = refb[aligned]
    for bi, b in enumerate(g_ACGT):
        mfeatures[rows[cur_base == b], 3 + bi] = 1
    # signal features for every event that exists (0 <= ie < nev)
    ie = np.arange(start_clip - 100, nev - end_clip + 100)
    ok = (ie >= 0) & (ie < nev)
    mfeatures[ok, 7] = modevents['mean'][ie[ok]]
    mfeatures[ok, 8] = modevents['stdv'][ie[ok]]
    mfeatures[ok, 9] = modevents['length'][ie[ok]]
    return (mfeatures, isdif)
