"""TEST INFRASTRUCTURE: the oracle chain for feature containers - C classifier (oracle/deepmod_oracle.c) + the loop-level
restatements of mPredict1 and sum_handler (oracle/detect_oracle.py) - as one call.  Never imported by the product."""
import numpy as np

from deepmod_amd import predstore
from oracle import detect_oracle, oracle_np


def oracle_beds(files, weights, base):
    """-> ({(chr, strand): BED bytes}, smallest |p1 - 0.5| over all windows, number of windows)"""
    classify = lambda x: oracle_np.predict_windows_c(weights, np.asarray(x, np.float32))[1]
    by, margin, nwin = {}, 1.0, 0
    for f in files:
        for rd in predstore.load_feature_container(f):
            bmi = rd['base_map_info']
            ev_bases = [s[2] for s in rd['events']['model_state']]
            n = len(ev_bases) - rd['start_clip'] - rd['end_clip']
            if n < 50:
                continue
            tx = np.asarray(rd['mfeatures'][:, 3:], np.float32)
            win = np.lib.stride_tricks.sliding_window_view(tx, (21, 7))[:, 0][90:90 + n]
            prob = oracle_np.predict_windows_c(weights, np.ascontiguousarray(win))[0]
            margin = min(margin, float(np.abs(prob[:, 1] - 0.5).min()))
            nwin += n
            _, _, mod_pred = detect_oracle.mpredict1_oracle(rd['mfeatures'], list(bmi['readbase']), ev_bases, rd['start_clip'],
                                                            rd['end_clip'], classify)
            by.setdefault((rd['chr'], rd['strand']), []).append(
                {'refbase': ''.join(bmi['refbase']), 'readbase': ''.join(bmi['readbase']),
                 'refbasei': [int(v) for v in bmi['refbasei']], 'mod_pred': mod_pred.tolist()})
    return {k: detect_oracle.sum_handler_oracle(k[0], k[1], base, v) for k, v in by.items()}, margin, nwin
