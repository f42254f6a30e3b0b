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


def oracle_raw_container(path, genome, weights, tie=1e-4, nthreads=0):
    """The oracle chain for ONE raw container (signal samples + basecaller events + side-car SAM): getEvent restated as the reference's loop
    (myDetect.py:237-251), numpy signal oracle (mnormalized :266-282 + the per-event statistics loop :332-343), Python alignment-walk restatement
    (handle_record :515-714), loop-level get_Feature (:839-903), C classifier, loop-level mPredict1 (:787-834).
    -> ({strand: [read dicts for sum_handler_oracle]}, reads, smallest |p1 - 0.5|, {strand: {position: windows within `tie` of 0.5}})"""
    from deepmod_amd import rawreads
    from oracle import readmap_oracle, signal_oracle
    sam = {ln.split('\t')[0]: ln.rstrip('\n').split('\t') for ln in open(path[:-len(rawreads.RAW_SUFFIX)] + '.sam') if not ln.startswith('@')}
    by_strand, ties = {'+': [], '-': []}, {'+': {}, '-': {}}
    n_reads, min_margin = 0, 1.0
    for rd in rawreads.load_raw_container(path):
        ed = rd['events_data']
        rows, pre_i, pre_len = [], 0, int(ed['length'][0])
        for cur_i in range(1, len(ed)):
            if ed['move'][cur_i] > 0:
                rows.append((int(ed['start'][pre_i]), pre_len, ed['model_state'][pre_i]))
                pre_i, pre_len = cur_i, int(ed['length'][cur_i])
            else:
                pre_len += int(ed['length'][cur_i])
        rows.append((int(ed['start'][pre_i]), pre_len, ed['model_state'][pre_i]))
        ev = np.zeros(len(rows), dtype=rawreads.EVENT_DTYPE)
        ev['start'] = [r[0] for r in rows]
        ev['length'] = [r[1] for r in rows]
        ev['model_state'] = [r[2] for r in rows]
        sig, _ = signal_oracle.mnormalized(rd['raw'], ev)
        mean, stdv, first_empty = signal_oracle.event_stats(sig, ev)
        assert first_empty == len(ev)
        s = sam[rd['read_id']]
        o = readmap_oracle.map_read(int(s[1]), int(s[3]), s[5], s[9], genome, len(ev))
        assert o['status'] == 'ok' and o['n_ev'] >= 50
        refb = [r[0] for r in o['rows']]
        readb = [r[1] for r in o['rows']]
        ev_bases = [ms[2] for ms in ev['model_state']]
        mf, isdif = detect_oracle.get_feature_oracle(mean, stdv, ev['length'], ev_bases, refb, readb, None, o['leftclip'], o['rightclip'], o['strand'],
                                                     o['first_match_pos'], o['num_insertions'])
        assert not isdif
        n = len(ev) - o['leftclip'] - o['rightclip']
        win = np.stack([mf[100 + i - 10:100 + i + 11, 3:] for i in range(n)]).astype(np.float32)
        prob, cls_all = oracle_np.predict_windows_c(weights, win, nthreads=nthreads)
        margin = np.abs(prob[:, 1] - 0.5)
        min_margin = min(min_margin, float(margin.min()))
        taken = [0]

        def classify(x):        # mPredict1's session stand-in: the classes of its windows, in the order it asks for them (the same windows as `win`, cut into its batches)
            x = np.asarray(x, np.float32)
            assert np.array_equal(x, win[taken[0]:taken[0] + len(x)])
            out = cls_all[taken[0]:taken[0] + len(x)]
            taken[0] += len(x)
            return out
        _, _, mod_pred = detect_oracle.mpredict1_oracle(mf, readb, ev_bases, o['leftclip'], o['rightclip'], classify)
        assert taken[0] == n
        refi = [int(r[2]) for r in o['rows']]
        aligned = [j for j, b in enumerate(readb) if b != '-'][:n]
        for k in np.flatnonzero(margin < tie):
            p = refi[aligned[int(k)]]
            ties[o['strand']][p] = ties[o['strand']].get(p, 0) + 1
        by_strand[o['strand']].append({'refbase': ''.join(refb), 'readbase': ''.join(readb), 'refbasei': refi, 'mod_pred': mod_pred.tolist()})
        n_reads += 1
    return by_strand, n_reads, min_margin, ties
