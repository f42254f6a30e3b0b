"""Golden vectors for the SAM-record -> base_map_info -> features -> prediction-table stage, produced by
running the REFERENCE's own `handle_record` (/root/reference/bin/DeepMod_scripts/myDetect.py:491-782) in the build
container with stub `tensorflow` / `h5py` modules: an in-memory stand-in for `h5py.File` captures what it would
store, `getRefSeq` (samtools) is replaced by an in-memory genome, the TF session by a deterministic rule.

Output (plain data): host_record.json — genome, per read: SAM fields, event bases / mean / stdv / length, and the
reference's results: the stored `predetail` table, the group attributes, the index entry; plus the error channel.

Run only here (needs /root/reference):  python tests/golden/make_golden_record.py
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_host import EVENT_DTYPE, FakeSession, import_reference  # noqa: E402

COMP = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A'}


class FakeGroup(dict):
    def __init__(self):
        super().__init__()
        self.attrs = {}
        self.datasets = {}

    def create_group(self, name):
        g = FakeGroup()
        self[name] = g
        return g

    def create_dataset(self, name, data=None, compression=None):
        self.datasets[name] = np.array(data)


class FakeFile(FakeGroup):
    store = {}

    def __init__(self, path, mode):
        super().__init__()
        self.path = path

    def __enter__(self):
        return FakeFile.store.setdefault(self.path, self)

    def __exit__(self, *a):
        return False

    def flush(self):
        pass

    def close(self):
        pass


def revcomp(s):
    return ''.join(COMP[c] for c in reversed(s))


def make_read(rng, genome, name, strand, ops, start, lead_clip=('S', 0), tail_clip=('S', 0)):
    """ops: list of (op, n) in reference orientation for the aligned part.  Returns SAM fields + the basecall."""
    pos = start
    seq = []
    for op, n in ops:
        for _ in range(n):
            if op in 'M=':
                seq.append(genome[pos]); pos += 1
            elif op == 'X':
                seq.append(rng.choice([b for b in 'ACGT' if b != genome[pos]])); pos += 1
            elif op == 'm':          # an 'M' position that is a mismatch
                seq.append(rng.choice([b for b in 'ACGT' if b != genome[pos]])); pos += 1
            elif op == 'I':
                seq.append(rng.choice(list('ACGT')))
            elif op in 'DN':
                pos += 1
    cig = []
    for op, n in ops:
        op = 'M' if op == 'm' else op
        if cig and cig[-1][0] == op:
            cig[-1][1] += n
        else:
            cig.append([op, n])
    lead = ''.join(rng.choice(list('ACGT'), lead_clip[1])) if lead_clip[0] == 'S' else ''
    tail = ''.join(rng.choice(list('ACGT'), tail_clip[1])) if tail_clip[0] == 'S' else ''
    samseq = lead + ''.join(seq) + tail
    cigar = ''
    if lead_clip[1]:
        cigar += '%d%s' % (lead_clip[1], lead_clip[0])
    cigar += ''.join('%d%s' % (n, op) for op, n in cig)
    if tail_clip[1]:
        cigar += '%d%s' % (tail_clip[1], tail_clip[0])
    # events: one per basecalled base, in sequencing orientation; hard-clipped bases are still events
    hl = ''.join(rng.choice(list('ACGT'), lead_clip[1])) if lead_clip[0] == 'H' else ''
    ht = ''.join(rng.choice(list('ACGT'), tail_clip[1])) if tail_clip[0] == 'H' else ''
    full = hl + samseq + ht
    basecall = full if strand == '+' else revcomp(full)
    return {'name': name, 'flag': 0 if strand == '+' else 16, 'pos': start + 1, 'cigar': cigar, 'seq': samseq, 'basecall': basecall}


def random_ops(rng, span, p_mis=0.06, p_ins=0.03, p_del=0.03):
    ops = [('M', 3)]
    used = 3
    while used < span - 3:
        u = rng.random()
        if u < p_ins:
            ops.append(('I', int(rng.integers(1, 4))))
        elif u < p_ins + p_del:
            n = int(rng.integers(1, 4)); ops.append(('D', n)); used += n
        elif u < p_ins + p_del + p_mis:
            ops.append(('m', 1)); used += 1
        else:
            n = int(rng.integers(1, 12)); ops.append(('M', n)); used += n
    ops.append(('M', 3))
    return ops


def main():
    myDetect = import_reference()
    myDetect.h5py.File = FakeFile
    rng = np.random.default_rng(77)
    genome = ''.join(rng.choice(list('ACGT'), 6000))
    # plant CpG special-case material: reference "CGG" / "CCG"
    genome = genome[:1500] + 'ACGGGTTACCGTA' + genome[1513:]
    myDetect.getRefSeq = lambda mo, sp, rname: sp['ref_info'].__setitem__(rname, genome)

    reads = []
    reads.append(make_read(rng, genome, 'plain_fwd', '+', random_ops(rng, 400), 100, ('S', 7), ('S', 4)))
    reads.append(make_read(rng, genome, 'plain_rev', '-', random_ops(rng, 500), 700, ('S', 5), ('S', 9)))
    reads.append(make_read(rng, genome, 'hard_clips', '+', random_ops(rng, 300), 2000, ('H', 6), ('H', 3)))
    reads.append(make_read(rng, genome, 'mismatch_ends_fwd', '+', [('m', 2)] + random_ops(rng, 300) + [('m', 3)], 2500, ('S', 2), ('S', 0)))
    reads.append(make_read(rng, genome, 'mismatch_ends_rev', '-', [('m', 1)] + random_ops(rng, 300) + [('m', 2)], 3000, ('S', 0), ('S', 6)))
    reads.append(make_read(rng, genome, 'edge_indels', '+', [('I', 2), ('D', 3)] + random_ops(rng, 300) + [('D', 2), ('I', 3)], 3500))
    reads.append(make_read(rng, genome, 'eq_x_ops', '-', [('=', 20), ('X', 2), ('=', 30), ('I', 1), ('=', 40), ('D', 2), ('=', 60), ('N', 3), ('=', 25)], 4000, ('S', 3), ('S', 3)))
    # CpG special case: read "C - G" against reference "C G G" (deletion of the first G) and mirror image
    reads.append(make_read(rng, genome, 'cpg_swap_fwd', '+', [('M', 60), ('M', 2), ('D', 1), ('M', 1), ('D', 2), ('M', 4), ('M', 1), ('D', 1), ('M', 80)], 1440))
    reads.append(make_read(rng, genome, 'cpg_swap_rev', '-', [('M', 60), ('M', 2), ('D', 1), ('M', 1), ('D', 2), ('M', 4), ('M', 1), ('D', 1), ('M', 80)], 1440))
    reads.append(make_read(rng, genome, 'too_short', '+', [('M', 40)], 5000))
    reads.append(make_read(rng, genome, 'no_match', '+', [('m', 60)], 5200))
    reads.append(make_read(rng, genome, 'unknown_contig', '+', random_ops(rng, 100), 5300))

    f5data, f5align = {}, {}
    for rd in reads:
        n = len(rd['basecall'])
        ev = np.zeros(n, dtype=EVENT_DTYPE)
        ev['mean'] = np.round(np.clip(rng.normal(0, 1.2, n), -5, 5), 3)
        ev['stdv'] = np.round(np.abs(rng.normal(0.25, 0.15, n)), 3)
        ev['length'] = rng.geometric(0.12, n)
        ev['start'] = np.cumsum(np.r_[0, ev['length'][:-1]])
        ev['model_state'] = ['NN' + b + 'NN' for b in rd['basecall']]
        rd['ev_mean'] = [float(v) for v in ev['mean']]
        rd['ev_stdv'] = [float(v) for v in ev['stdv']]
        rd['ev_length'] = [int(v) for v in ev['length']]
        rd['rname'] = 'chrU_random' if rd['name'] == 'unknown_contig' else 'chrS'
        f5data[rd['name']] = (rd['basecall'], ev, None, '/wrk/' + rd['name'] + '.fast5', (0, 0))
        f5align[rd['name']] = (60, rd['flag'], rd['rname'], rd['pos'], rd['cigar'], rd['seq'])

    tmp = tempfile.mkdtemp()
    mo = {'ConUnk': False, 'region': [[None, None, None]], 'outLevel': 2, 'fnum': 7, 'windowsize': 21, 'wrkBase': '/wrk',
          'outFolder': tmp + '/', 'FileID': 'mod'}
    os.makedirs(tmp + '/mod/0')
    sess = FakeSession()
    sp_options = defaultdict()
    sp_options.update({'ctfolder': tmp + '/mod/0', 'batchid': 3, 'Mod': [], 'Error': defaultdict(list),
                       'rnn': (sess, 'X', 'Y', None, 'mfpred')})
    sp_param = defaultdict()
    sp_param.update({'f5data': f5data, 'ref_info': defaultdict(), 'f5status': '', 'line': ''})
    myDetect.handle_record(mo, sp_options, sp_param, f5align, f5data)

    out = {'genome': genome, 'reads': reads, 'results': {}, 'errors': {k: list(v) for k, v in sp_options['Error'].items()},
           'mod_index': sp_options['Mod'], 'index_files': {}}
    for fn in sorted(os.listdir(tmp + '/mod/0')):
        out['index_files'][fn] = open(os.path.join(tmp, 'mod/0', fn)).read()
    (store,) = FakeFile.store.values()
    keys = list(f5align.keys())
    for key, grp in store['pred'].items():
        tab = grp.datasets['predetail']
        name = keys[int(key.split('_')[1])]
        out['results'][name] = {
            'key': key,
            'attrs': {k: (v if isinstance(v, str) else int(v)) for k, v in grp.attrs.items()},
            'refbase': ''.join(b.decode() for b in tab['refbase']), 'readbase': ''.join(b.decode() for b in tab['readbase']),
            'refbasei': [int(v) for v in tab['refbasei']], 'readbasei': [int(v) for v in tab['readbasei']],
            'mod_pred': [int(v) for v in tab['mod_pred']]}
        print(name, key, len(tab), 'rows', {k: out['results'][name]['attrs'][k] for k in ('mapped_strand', 'clipped_bases_start', 'clipped_bases_end', 'num_insertions', 'num_deletions', 'num_mismatches', 'pred_mod_num')})
    print('errors', out['errors'])
    json.dump(out, open(os.path.join(HERE, 'host_record.json'), 'w'))


if __name__ == '__main__':
    main()
