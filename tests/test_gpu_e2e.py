"""GPU (-m gpu): `bin/DeepMod.py detect` end to end on synthetic reads (BASELINE config 1 shape,
scaled down): worker processes, per-read prediction store, index merge, GPU summary, BED files and the
.done marker - compared with the oracle pipeline (C classifier + loop-level mPredict1 / sum_handler
restatements): per-read classes exact away from near ties, BED bytes identical."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from deepmod_amd import predstore, synth, synth_reads
from oracle import detect_oracle, oracle_np

pytestmark = pytest.mark.gpu


def test_detect_cli_matches_oracle_pipeline(tmp_path, gpu_device):
    wrk = tmp_path / 'reads'
    files = synth_reads.write_synthetic_run(str(wrk), n_reads=24, reads_per_file=4, genome_len=20000, seed=3,
                                            chrom='chrS', min_len=300, max_len=1500)
    prefix = str(tmp_path / 'model' / 'mod_train_synth')
    os.makedirs(os.path.dirname(prefix))
    w = synth.write_synthetic_checkpoint(prefix, seed=9, scale=4.0)   # min|p1-0.5| = 2.2e-4 on this read set
    out = str(tmp_path / 'out')
    cmd = [sys.executable, os.path.join(ROOT, 'bin', 'DeepMod.py'), 'detect', '--wrkBase', str(wrk), '--modfile', prefix,
           '--outFolder', out, '--FileID', 'run1', '--threads', '2', '--files_per_thread', '2', '--Base', 'C', '--gpus', '1']
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert os.path.exists(out + '/run1.done')

    # oracle pipeline
    classify = lambda x: oracle_np.predict_windows_c(w, np.asarray(x, np.float32))[1]
    by_strand = {'+': [], '-': []}
    near_tie = False
    for f in files:
        for rd in predstore.load_feature_container(f):
            bmi = rd['base_map_info']
            ev_bases = [s[2] for s in rd['events']['model_state']]
            n = len(ev_bases) - rd['start_clip'] - rd['end_clip']
            win = np.stack([rd['mfeatures'][100 + i - 10:100 + i + 11, 3:] for i in range(n)]).astype(np.float32)
            prob = oracle_np.predict_windows_c(w, win)[0]
            near_tie |= bool((np.abs(prob[:, 1] - 0.5) < 1e-4).any())
            _, _, mod_pred = detect_oracle.mpredict1_oracle(rd['mfeatures'], list(bmi['readbase']), ev_bases,
                                                            rd['start_clip'], rd['end_clip'], classify)
            by_strand[rd['strand']].append({'refbase': ''.join(bmi['refbase']), 'readbase': ''.join(bmi['readbase']),
                                            'refbasei': [int(v) for v in bmi['refbasei']], 'mod_pred': mod_pred.tolist()})
    assert not near_tie, 'synthetic set has a near-tie window; pick another seed'
    for strand, reads in by_strand.items():
        want = detect_oracle.sum_handler_oracle('chrS', strand, 'C', reads)
        got = open('%s/run1/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read()
        assert got == want
        assert len(got) > 1000
    # per-read store: every read is there, with the reference's attribute set
    stores = glob.glob(out + '/run1/*/rnn.pred.detail.npz.*')
    assert len(stores) == 3      # 6 containers / 2 per batch
    assert os.path.exists(out + '/run1/rnn.pred.ind.chrS')
    lines = [l for l in open(out + '/run1/rnn.pred.ind.chrS') if not l.startswith('#')]
    assert len(lines) == 24 and lines == sorted(lines, key=lambda l: (l.split()[1], int(l.split()[2])))

    # --predDet 0 resume: summary only, from the stored predictions, same BED
    for f in glob.glob(out + '/run1/mod_pos.*'):
        os.remove(f)
    os.remove(out + '/run1.done')
    cmd2 = [sys.executable, os.path.join(ROOT, 'bin', 'DeepMod.py'), 'detect', '--predDet', '0', '--predpath', out + '/run1',
            '--threads', '2', '--Base', 'C', '--gpus', '1']
    res = subprocess.run(cmd2, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    for strand, reads in by_strand.items():
        assert open('%s/run1/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read() == \
            detect_oracle.sum_handler_oracle('chrS', strand, 'C', reads)
