"""SAM record -> base_map_info -> features -> prediction table against the golden output of the
reference's own handle_record (tests/golden/make_golden_record.py).  CPU only: dm_map_read is host code in the C-ABI
library, the classifier is the same deterministic stand-in session the golden run used."""
import json
import os
from collections import defaultdict

import numpy as np
import pytest

from conftest import GOLDEN
from deepmod_amd import _lib, detect, predstore, rawreads, readmap

EVENT_DTYPE = rawreads.EVENT_DTYPE


def fake_rule(x):
    x = np.asarray(x)
    return ((x[:, 10, 1] > 0.5) & (x[:, 10, 4] > 0.0)).astype(np.int64)


class FakeSession:
    model = None

    def run(self, fetches, feed_dict=None):
        if feed_dict is None:
            return None
        return [fake_rule(feed_dict["X"])]


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(GOLDEN, "host_record.json")))


def _inputs(golden):
    f5data, f5align = {}, {}
    for rd in golden['reads']:
        n = len(rd['basecall'])
        ev = np.zeros(n, dtype=EVENT_DTYPE)
        ev['mean'] = np.array(rd['ev_mean'], np.float32)
        ev['stdv'] = np.array(rd['ev_stdv'], np.float32)
        ev['length'] = rd['ev_length']
        ev['start'] = np.cumsum(np.r_[0, ev['length'][:-1]])
        ev['model_state'] = ['NN' + b + 'NN' for b in rd['basecall']]
        f5data[rd['name']] = (rd['basecall'], ev, None, '/wrk/' + rd['name'] + '.fast5', (0, 0))
        f5align[rd['name']] = (60, rd['flag'], rd['rname'], rd['pos'], rd['cigar'], rd['seq'])
    return f5data, f5align


def test_map_read_matches_reference_tables(golden, hip_lib):
    f5data, f5align = _inputs(golden)
    for name, res in golden['results'].items():
        mapq, flag, rname, pos, cigar, seq = f5align[name]
        mp = readmap.map_read(flag, pos, cigar, seq, golden['genome'], len(f5data[name][1]))
        assert mp['status'] == _lib.DM_MAP_OK, name
        bmi = mp['base_map_info']
        assert ''.join(bmi['refbase']) == res['refbase'], name
        assert ''.join(bmi['readbase']) == res['readbase'], name
        assert bmi['refbasei'].tolist() == res['refbasei'], name
        assert bmi['readbasei'].tolist() == res['readbasei'], name
        a = res['attrs']
        assert mp['strand'] == a['mapped_strand']
        # :729-734: clipped_bases_start = leftclip for '+', rightclip for '-' (after the swap of :667)
        start, end = (mp['leftclip'], mp['rightclip']) if mp['strand'] == '+' else (mp['rightclip'], mp['leftclip'])
        assert (start, end) == (a['clipped_bases_start'], a['clipped_bases_end']), name
        assert (mp['num_insertions'], mp['num_deletions'], mp['num_mismatches']) == \
               (a['num_insertions'], a['num_deletions'], a['num_mismatches']), name
        assert len(bmi) - mp['num_mismatches'] - mp['num_insertions'] - mp['num_deletions'] == a['num_matches']


def test_map_read_status_and_errors(golden, hip_lib):
    f5data, f5align = _inputs(golden)
    _, flag, _, pos, cigar, seq = f5align['no_match']
    assert readmap.map_read(flag, pos, cigar, seq, golden['genome'], len(seq))['status'] == _lib.DM_MAP_NO_MATCH
    with pytest.raises(_lib.DeepModHipError):
        readmap.map_read(0, 1, '10S', 'ACGTACGTAC', golden['genome'], 10)           # nothing aligned
    with pytest.raises(_lib.DeepModHipError):
        readmap.map_read(0, len(golden['genome']) - 3, '10M', 'ACGTACGTAC', golden['genome'], 10)   # runs past the reference
    with pytest.raises(_lib.DeepModHipError):
        readmap.map_read(0, 1, '10M', 'ACGT', golden['genome'], 4)                  # runs past the read
    with pytest.raises(_lib.DeepModHipError):
        readmap.map_read(0, 1, '10M5', 'ACGTACGTAC', golden['genome'], 10)          # malformed (count without an operation)


def test_records_to_prediction_tables_match_reference(golden, hip_lib, tmp_path):
    """map_records + mPredict1 + PredWriter + index lines == what handle_record stored."""
    f5data, f5align = _inputs(golden)
    mo = {'ConUnk': False, 'region': [[None, None, None]], 'outLevel': 3, 'fnum': 7, 'windowsize': 21, 'wrkBase': '/wrk',
          'outFolder': str(tmp_path) + '/', 'FileID': 'mod'}
    ct = str(tmp_path / 'mod' / '0')
    os.makedirs(ct)
    sp_options = defaultdict()
    sp_options.update({'ctfolder': ct, 'batchid': 3, 'Mod': [], 'Error': defaultdict(list),
                       'rnn': (FakeSession(), 'X', 'Y', None, 'mfpred')})
    sp_param = defaultdict()
    sp_param.update({'f5data': f5data, 'ref_info': {'chrS': golden['genome']}, 'f5status': '', 'line': ''})
    reads = readmap.map_records(mo, sp_options, sp_param, f5align, f5data)
    assert sorted(rd['readk'] for rd in reads) == sorted(golden['results'])
    store = predstore.PredWriter(ct, 3)
    detect._predict_and_store(mo, sp_options, store, reads, '/wrk/x')
    store.close()
    assert {k: sorted(v) for k, v in sp_options['Error'].items()} == {k: sorted(v) for k, v in golden['errors'].items()}
    by_name = {rd['readk']: rd for rd in reads}
    for name, res in golden['results'].items():
        rd = by_name[name]
        assert rd['base_map_info']['mod_pred'].tolist() == res['mod_pred'], name
        assert int((rd['base_map_info']['mod_pred'] == 1).sum()) == res['attrs']['pred_mod_num']
    # stored attributes
    z = np.load(store.path, allow_pickle=False)
    attrs = json.loads(str(z['attrs']))
    got = {a['readk']: a for a in attrs.values()}
    for name, res in golden['results'].items():
        for k, v in res['attrs'].items():
            assert got[name][k] == v, (name, k, got[name][k], v)
    # index entries: chr, strand, 0-based SAM pos, key, source path relative to wrkBase (the key numbering differs:
    # the reference numbers keys by alignment-record index, PredWriter by stored table)
    ref_idx = sorted((m[0], m[1], m[2], m[4]) for m in golden['mod_index'])
    assert sorted((m[0], m[1], m[2], m[4]) for m in sp_options['Mod']) == ref_idx


def test_getEvent_merges_stay_events():
    ed = np.zeros(9, dtype=rawreads.EVENTS_DATA_DTYPE)
    ed['move'] = [1, 0, 0, 1, 2, 0, 1, 0, 0]
    ed['length'] = [3, 4, 5, 6, 7, 8, 9, 10, 11]
    ed['start'] = np.cumsum(np.r_[100, ed['length'][:-1]])
    ed['mean'] = np.arange(9) + 0.12349
    ed['stdv'] = 0.5
    ed['model_state'] = ['AAAAA', 'AAAAA', 'AAAAA', 'AACAA', 'AAGAA', 'AAGAA', 'AATAA', 'AATAA', 'AATAA']
    sp = {'events_data': ed, 'f5status': ''}
    rawreads.getEvent({'SignalGroup': 'simple'}, sp)
    ev = sp['m_event']
    assert ev['length'].tolist() == [12, 6, 15, 30]
    assert ev['start'].tolist() == [100, 112, 118, 133]
    assert sp['m_event_basecall'] == 'ACGT'
    assert np.allclose(ev['mean'], [0.123, 3.123, 4.123, 6.123])


def test_oracle_restatement_matches_reference_tables(golden):
    from oracle import readmap_oracle
    f5data, f5align = _inputs(golden)
    for name, res in golden['results'].items():
        _, flag, _, pos, cigar, seq = f5align[name]
        o = readmap_oracle.map_read(flag, pos, cigar, seq, golden['genome'], len(f5data[name][1]))
        assert o['status'] == 'ok'
        assert ''.join(r[0] for r in o['rows']) == res['refbase'], name
        assert ''.join(r[1] for r in o['rows']) == res['readbase'], name
        assert [r[2] for r in o['rows']] == res['refbasei'] and [r[3] for r in o['rows']] == res['readbasei']
    _, flag, _, pos, cigar, seq = f5align['no_match']
    assert readmap_oracle.map_read(flag, pos, cigar, seq, golden['genome'], len(seq))['status'] == 'no match'


def _random_record(rng, genome):
    """A random (mostly sane, sometimes odd) SAM record over `genome`."""
    start = int(rng.integers(0, len(genome) - 900))
    pos = start
    seq, cig = [], []

    def push(op, n):
        if cig and cig[-1][0] == op:
            cig[-1][1] += n
        else:
            cig.append([op, n])
    lead = rng.choice(['', 'S', 'H', 'SI', 'SD', 'I', 'D', 'X'], p=[.3, .3, .1, .05, .05, .07, .07, .06])
    tail = rng.choice(['', 'S', 'H', 'IS', 'DS', 'I', 'D', 'X'], p=[.3, .3, .1, .05, .05, .07, .07, .06])
    def clip_ops(spec):
        nonlocal pos
        for op in spec:
            n = int(rng.integers(1, 6))
            if op in 'SIX':
                seq.extend(rng.choice(list('ACGT'), n))
            if op in 'DX':
                pos += n
            push(op, n)
    clip_ops(lead)
    n_ops = int(rng.integers(3, 40))
    for k in range(n_ops):
        u = rng.random()
        n = int(rng.integers(1, 15))
        if u < 0.55:
            for _ in range(n):
                seq.append(genome[pos] if rng.random() > 0.1 else rng.choice(list('ACGT'))); pos += 1
            push('M', n)
        elif u < 0.65:
            seq.extend(genome[pos:pos + n]); pos += n; push('=', n)
        elif u < 0.72:
            seq.extend(rng.choice(list('ACGT'), n)); pos += n; push('X', n)
        elif u < 0.84:
            seq.extend(rng.choice(list('ACGT'), n)); push('I', n)
        elif u < 0.96:
            pos += n; push('D', n)
        else:
            pos += n; push('N', n)
    clip_ops(tail)
    cigar = ''.join('%d%s' % (n, op) for op, n in cig)
    readseq = ''.join(seq)
    hard = sum(n for op, n in cig if op == 'H')
    n_events = len(readseq) + hard + int(rng.integers(-2, 3)) * (rng.random() < 0.1)
    return int(rng.choice([0, 16])), start + 1, cigar, readseq, max(n_events, 1)


def test_map_read_matches_oracle_on_random_records(hip_lib):
    from oracle import readmap_oracle
    rng = np.random.default_rng(123)
    genome = ''.join(rng.choice(list('ACGT'), 5000))
    genome = genome[:2000] + 'CGGCCGCGGGCCCG' * 20 + genome[2280:]       # CpG-rich stretch for the gap-swap rule
    n_ok = n_nomatch = 0
    for it in range(600):
        flag, pos1, cigar, readseq, n_events = _random_record(rng, genome)
        try:
            o = readmap_oracle.map_read(flag, pos1, cigar, readseq, genome, n_events)
        except IndexError:
            with pytest.raises(_lib.DeepModHipError):
                readmap.map_read(flag, pos1, cigar, readseq, genome, n_events)
            continue
        mp = readmap.map_read(flag, pos1, cigar, readseq, genome, n_events)
        for k in ('strand', 'pos_after_clip', 'events_after_clip', 'num_insertions', 'num_deletions', 'num_mismatches'):
            assert mp[k] == o[k], (it, k, cigar)
        if o['status'] == 'no match':
            assert mp['status'] == _lib.DM_MAP_NO_MATCH
            n_nomatch += 1
            continue
        n_ok += 1
        bmi = mp['base_map_info']
        assert ''.join(bmi['refbase']) == ''.join(r[0] for r in o['rows']), (it, cigar)
        assert ''.join(bmi['readbase']) == ''.join(r[1] for r in o['rows']), (it, cigar)
        assert bmi['refbasei'].tolist() == [r[2] for r in o['rows']]
        assert bmi['readbasei'].tolist() == [r[3] for r in o['rows']]
        assert (mp['leftclip'], mp['rightclip']) == (o['leftclip'], o['rightclip']), (it, cigar)
        assert mp['ev_hi'] - mp['ev_lo'] == o['n_ev'], (it, cigar)
        if o['n_ev']:
            assert (mp['ev_lo'], mp['ev_hi']) == (o['ev_lo'], o['ev_hi']), (it, cigar)
        assert (mp['first_match_pos'], mp['last_match_pos']) == (o['first_match_pos'], o['last_match_pos'])
    assert n_ok > 400


def test_region_and_contig_filters(golden, hip_lib):
    """--region / --ConUnk semantics of handle_record (myDetect.py:501-511, :544-553)."""
    f5data, f5align = _inputs(golden)
    base = {'ConUnk': True, 'outLevel': 3, 'fnum': 7, 'windowsize': 21, 'wrkBase': '/wrk', 'outFolder': '/tmp/', 'FileID': 'mod'}

    def names(mo):
        sp_options = defaultdict()
        sp_options.update({'Mod': [], 'Error': defaultdict(list)})
        sp_param = defaultdict()
        sp_param.update({'f5data': f5data, 'ref_info': {'chrS': golden['genome'], 'chrU_random': golden['genome']}, 'f5status': '', 'line': ''})
        return sorted(rd['readk'] for rd in readmap.map_records(dict(base, **mo), sp_options, sp_param, f5align, f5data))

    everything = names({'region': [[None, None, None]]})
    assert 'unknown_contig' in everything                                   # ConUnk=True keeps names with '_'
    assert 'unknown_contig' not in names({'region': [[None, None, None]], 'ConUnk': False})
    assert names({'region': [['chrX', None, None]]}) == []                  # other chromosome
    only = names({'region': [['chrS', 1900, 2700]]})                        # pos > 1900 and pos + len(events) < 2700
    assert only == ['hard_clips']
    assert set(names({'region': [['chrS', None, 1500]]})) <= {'plain_fwd', 'plain_rev', 'cpg_swap_fwd', 'cpg_swap_rev'}


def test_no_match_diagnostics_print_at_the_default_warning_level(golden, hip_lib, capsys):
    """myCom.OUTPUT_WARNING is 2 = the default --outLevel: the reference prints 'Errorfast5' / 'match-Error!!!' for a read without a
    first / last match at that level (myDetect.py:617-622) and stays silent at outLevel 3."""
    f5data, f5align = _inputs(golden)
    only = {'no_match': f5align['no_match']}
    for level, expect in ((2, True), (3, False)):
        sp_options = defaultdict()
        sp_options.update({'Mod': [], 'Error': defaultdict(list)})
        sp_param = defaultdict()
        sp_param.update({'f5data': f5data, 'ref_info': {'chrS': golden['genome']}, 'f5status': '', 'line': ''})
        mo = {'ConUnk': True, 'outLevel': level, 'fnum': 7, 'windowsize': 21, 'region': [[None, None, None]]}
        assert readmap.map_records(mo, sp_options, sp_param, only, f5data) == []
        out = capsys.readouterr().out
        assert ('Errorfast5' in out and 'match-Error!!! no first and/or last match' in out) == expect


def test_raw_batch_without_alignment_goes_to_error_channel(tmp_path, hip_lib, monkeypatch):
    """No aligner on PATH and no side-car SAM -> every read of the batch is reported under the reference's key."""
    from deepmod_amd import synth_reads
    files, fasta = synth_reads.write_synthetic_raw_run(str(tmp_path / 'raw'), n_reads=3, reads_per_file=3, genome_len=5000, seed=2,
                                                       chrom='chrS', min_len=200, max_len=300)
    os.remove(files[0][:-len(rawreads.RAW_SUFFIX)] + '.sam')
    f5data = {'r%d' % i: ('ACGT', None, None, files[0], (0, 0)) for i in range(3)}
    monkeypatch.setattr(rawreads, 'get_Event_Signals', lambda mo, so, fl, normalizer=None: f5data)
    sp_options = defaultdict()
    sp_options.update({'Mod': [], 'Error': defaultdict(list), 'ctfolder': str(tmp_path), 'batchid': 0})
    detect.mDetect1_raw({'alignStr': 'no-such-aligner', 'Ref': fasta}, sp_options, None, files)
    assert sp_options['Error'] == {'Cannot running aligment': [files[0]] * 3}
