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
           '--outFolder', out, '--FileID', 'run1', '--threads', '2', '--files_per_thread', '2', '--Base', 'C', '--gpus', '1',
           '--storePred', '1']
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert os.path.exists(out + '/run1.done')
    # the default (streaming) mode on the same inputs: same BED bytes, no per-read files
    cmd_s = cmd[:-2]
    cmd_s[cmd_s.index('run1')] = 'stream1'
    res_s = subprocess.run(cmd_s, capture_output=True, text=True, timeout=600)
    assert res_s.returncode == 0, res_s.stdout[-1500:] + res_s.stderr[-3000:]
    assert os.path.exists(out + '/stream1.done') and 'Streaming detect: 24 reads' in res_s.stdout
    assert not glob.glob(out + '/stream1/*/rnn.pred.detail.npz.*') and not glob.glob(out + '/stream1/rnn.pred.ind.*')
    for strand in '+-':
        assert open('%s/stream1/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read() == \
            open('%s/run1/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read()

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


def test_detect_cli_on_raw_containers_matches_oracle_pipeline(tmp_path, gpu_device):
    """The whole path from DAC samples: GPU signal normalisation + event statistics, SAM records through
    dm_map_read, features, BiLSTM, summary — against the oracle chain (numpy signal oracle, Python alignment-walk
    restatement, loop-level get_Feature / mPredict1 / sum_handler restatements, C classifier)."""
    from deepmod_amd import readmap
    wrk = tmp_path / 'raw'
    files, fasta = synth_reads.write_synthetic_raw_run(str(wrk), n_reads=18, reads_per_file=4, genome_len=20000, seed=5,
                                                       chrom='chrS')
    prefix = str(tmp_path / 'model' / 'mod_train_synth')
    os.makedirs(os.path.dirname(prefix))
    w = synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)   # min|p1-0.5| = 2.2e-4, 51 % class 1 on this read set
    out = str(tmp_path / 'out')
    cmd = [sys.executable, os.path.join(ROOT, 'bin', 'DeepMod.py'), 'detect', '--wrkBase', str(wrk), '--modfile', prefix,
           '--Ref', fasta, '--outFolder', out, '--FileID', 'raw1', '--threads', '2', '--files_per_thread', '2', '--Base', 'C',
           '--gpus', '1', '--alignStr', 'minimap2', '--storePred', '1']
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert os.path.exists(out + '/raw1.done')
    cmd_s = cmd[:-2]                                       # default = streaming detect, from DAC samples to BED
    cmd_s[cmd_s.index('raw1')] = 'rawstream1'
    res_s = subprocess.run(cmd_s, capture_output=True, text=True, timeout=600)
    assert res_s.returncode == 0, res_s.stdout[-1500:] + res_s.stderr[-3000:]
    assert 'Streaming detect: 18 reads' in res_s.stdout
    for strand in '+-':
        assert open('%s/rawstream1/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read() == \
            open('%s/raw1/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read()
    # round 6: that streaming run kept its event statistics on the device (resident form: dm_signal_event_stats_device -> dm_rows_assemble, nothing per
    # event through the feeders); with DEEPMOD_STATS_ON_DEVICE=0 the statistics take round 5's way through the host - the same BED bytes
    import re
    m = re.search(r'event statistics resident on the device for (\d+) of (\d+) rows', res_s.stdout)
    assert m and int(m.group(1)) == int(m.group(2)) > 0, res_s.stdout[-1500:]
    cmd_h = list(cmd_s)
    cmd_h[cmd_h.index('rawstream1')] = 'rawstream_host_stats'
    res_h = subprocess.run(cmd_h, capture_output=True, text=True, timeout=600, env=dict(os.environ, DEEPMOD_STATS_ON_DEVICE='0'))
    assert res_h.returncode == 0, res_h.stdout[-1500:] + res_h.stderr[-3000:]
    m = re.search(r'event statistics resident on the device for (\d+) of (\d+) rows', res_h.stdout)
    assert m and int(m.group(1)) == 0
    for strand in '+-':
        assert open('%s/rawstream_host_stats/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read() == \
            open('%s/raw1/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read()

    genome = readmap.read_fasta(fasta)['chrS']
    from oracle_pipeline import oracle_raw_container
    by_strand = {'+': [], '-': []}
    n_reads = 0
    min_margin = 1.0
    for f in files:
        got_reads, n, margin, _ties = oracle_raw_container(f, genome, w)
        for strand in '+-':
            by_strand[strand].extend(got_reads[strand])
        n_reads += n
        min_margin = min(min_margin, margin)
    assert n_reads == 18
    assert min_margin > 1e-4, 'synthetic set has a near-tie window (%.2e); pick another seed' % min_margin
    for strand, reads in by_strand.items():
        want = detect_oracle.sum_handler_oracle('chrS', strand, 'C', reads)
        got = open('%s/raw1/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read()
        assert got == want
        assert len(got) > 500


def test_raw_reads_with_an_empty_event_and_a_stalled_event_through_the_resident_form(tmp_path, gpu_device):
    """Round 6: the corners of the resident form through the whole command.  One read gets an event whose slice is EMPTY (index 60 <= 500: the reference's loop
    stops there and the events from there on keep the basecaller's values, myDetect.py:334-340 - here the fall-back values are merged ON THE DEVICE), another a
    stalled event of 70,000 samples (beyond the split-f16 kernels' range: its batch takes the fp32 kernel).  The streaming command with the statistics resident
    on the device, the same command with DEEPMOD_STATS_ON_DEVICE=0 (statistics through the host, round 5's form) and the stored path (per-read prediction
    files, the reference's shape) must write the same BED bytes."""
    import re
    from deepmod_amd import npzmap
    wrk = tmp_path / 'raw'
    files, fasta = synth_reads.write_synthetic_raw_run(str(wrk), n_reads=16, reads_per_file=4, genome_len=20000, seed=8, chrom='chrS')
    z = {k: np.array(v) for k, v in npzmap.load(files[0]).items()}
    eo, ro = z['ev_off'], z['raw_off']
    st, ln = np.array(z['ev_start']), np.array(z['ev_length'])
    st[eo[1] + 60] = int(ro[2] - ro[1]) + 1000              # read 1 of the first container: an empty slice at event 60
    z['ev_start'] = st
    npzmap.savez_aligned(files[0], **z)
    z = {k: np.array(v) for k, v in npzmap.load(files[1]).items()}
    ln = np.array(z['ev_length'])
    ln[z['ev_off'][2] + 300] = 70000                          # read 2 of the second container: a stalled event (the slice is clamped at the signal's end)
    z['ev_length'] = ln
    npzmap.savez_aligned(files[1], **z)
    # a damaged table in the third container: the FIRST event of read 0 starts at 2^63 + 11 (numpy's uint64 slice: no signal covered - the batched signal
    # call refuses the batch, it is built read by read and that read becomes a line of the error ledger), an event of read 3 starts at 2^64 - 3 (an empty event)
    z = {k: np.array(v) for k, v in npzmap.load(files[2]).items()}
    st = np.array(z['ev_start'])
    st[0] = 2 ** 63 + 11
    st[z['ev_off'][3] + 40] = 2 ** 64 - 3
    z['ev_start'] = st
    npzmap.savez_aligned(files[2], **z)
    prefix = str(tmp_path / 'model' / 'mod_train_synth')
    os.makedirs(os.path.dirname(prefix))
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    out = str(tmp_path / 'out')
    base = [sys.executable, os.path.join(ROOT, 'bin', 'DeepMod.py'), 'detect', '--wrkBase', str(wrk), '--modfile', prefix, '--Ref', fasta, '--outFolder', out,
            '--threads', '2', '--files_per_thread', '2', '--Base', 'C', '--gpus', '1', '--alignStr', 'minimap2']
    runs = {}
    for name, extra, env in (('resident', [], {}), ('hoststats', [], {'DEEPMOD_STATS_ON_DEVICE': '0'}), ('stored', ['--storePred', '1'], {})):
        res = subprocess.run(base + ['--FileID', name] + extra, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
        runs[name] = res.stdout
    m = re.search(r'event statistics resident on the device for (\d+) of (\d+) rows', runs['resident'])
    assert m and 0 < int(m.group(1)) < int(m.group(2))              # (the batch with the damaged table went through the host)
    assert 'Streaming detect: 15 reads' in runs['resident'] and 'Streaming detect: 15 reads' in runs['hoststats']
    assert all('cover no signal' in so for so in runs.values())
    for strand in '+-':
        a = open('%s/resident/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read()
        assert len(a) > 300
        assert a == open('%s/hoststats/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read()
        assert a == open('%s/stored/mod_pos.chrS%s.C.bed' % (out, strand), 'rb').read()


def test_two_rank_run_that_cannot_build_its_communicator_fails_fast_and_clean(tmp_path, gpu_device):
    """`--gpus 2` with both GPU processes on device 0 (DEEPMOD_ONE_DEVICE=1, a test hook for one-GPU boxes): two real ranks and their
    feeders start, meet at the file rendezvous, and RCCL refuses the communicator (two ranks on one device).  The product has no merge
    without RCCL: the run must end within seconds with a non-zero exit code and the RCCL error in its output - not hang in a
    collective, not write a BED or the .done marker, not leave hand-over files in /dev/shm."""
    import time
    wrk = tmp_path / 'reads'
    synth_reads.write_synthetic_run(str(wrk), n_reads=24, reads_per_file=3, genome_len=20000, seed=3, chrom='chrA', min_len=300, max_len=900)
    prefix = str(tmp_path / 'model' / 'm')
    os.makedirs(os.path.dirname(prefix))
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    out = str(tmp_path / 'out')
    before = set(os.listdir('/dev/shm')) if os.path.isdir('/dev/shm') else set()
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'DeepMod.py'), 'detect', '--wrkBase', str(wrk), '--modfile', prefix, '--outFolder', out,
                          '--FileID', 'two', '--threads', '4', '--Base', 'C', '--gpus', '2'], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, DEEPMOD_ONE_DEVICE='1', DEEPMOD_COMM_TIMEOUT='60'))
    took = time.time() - t0
    # RCCL refuses within seconds as a rule; on a box where its bootstrap never comes back (seen once: > 2 min) the communicator watchdog ends the rank
    # (comm.Communicator, DEEPMOD_COMM_TIMEOUT) - either way the command ends by itself, with the reason on stderr
    assert res.returncode != 0 and took < 150, (res.returncode, took, res.stderr[-2000:])
    assert ('ncclCommInitRank' in res.stderr or 'has not returned after 60 s' in res.stderr) and 'a streaming detect worker died' in res.stderr, (took, res.stderr[-3000:])
    assert not os.path.exists(out + '/two.done') and not glob.glob(out + '/two/*.bed')
    if os.path.isdir('/dev/shm'):
        assert not [f for f in set(os.listdir('/dev/shm')) - before if f.startswith('deepmod')]
