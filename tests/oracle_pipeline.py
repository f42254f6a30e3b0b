"""TEST INFRASTRUCTURE: the oracle chain for feature containers - C classifier (oracle/deepmod_oracle.c) + the loop-level
restatements of mPredict1 and sum_handler (oracle/detect_oracle.py) - as one call.  Never imported by the product."""
import numpy as np

from deepmod_amd import predstore
from oracle import detect_oracle, oracle_np


def oracle_beds(files, weights, base):
    """-> ({(chr, strand): BED bytes}, smallest |p1 - 0.5| over all windows, number of windows)"""
    state = {'margin': 1.0}

    def classify(x):          # mPredict1's session stand-in; the smallest |p1 - 0.5| of the run is taken from the same oracle pass
        prob, cls = oracle_np.predict_windows_c(weights, np.asarray(x, np.float32))
        if len(prob):
            state['margin'] = min(state['margin'], float(np.abs(prob[:, 1] - 0.5).min()))
        return cls
    by, nwin = {}, 0
    for f in files:
        for rd in predstore.load_feature_container(f):
            bmi = rd['base_map_info']
            ev_bases = [s[2] for s in rd['events']['model_state']]
            n = len(ev_bases) - rd['start_clip'] - rd['end_clip']
            if n < 50:
                continue
            nwin += n
            _, _, mod_pred = detect_oracle.mpredict1_oracle(rd['mfeatures'], list(bmi['readbase']), ev_bases, rd['start_clip'],
                                                            rd['end_clip'], classify)
            by.setdefault((rd['chr'], rd['strand']), []).append(
                {'refbase': ''.join(bmi['refbase']), 'readbase': ''.join(bmi['readbase']),
                 'refbasei': [int(v) for v in bmi['refbasei']], 'mod_pred': mod_pred.tolist()})
    return {k: detect_oracle.sum_handler_oracle(k[0], k[1], base, v) for k, v in by.items()}, state['margin'], nwin
