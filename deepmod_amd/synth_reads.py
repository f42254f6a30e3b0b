"""Synthetic aligned reads in the in-memory form the reference builds just before mPredict1
(SURVEY.md 8d config 1): no FAST5, no aligner - a truth alignment with substitutions, insertions,
deletions, soft clips and both strands, events one per read base.  Used by tests, the CLI demo and
bench-adjacent tooling; written as feature containers (deepmod_amd/predstore.py)."""
from __future__ import annotations

import os
from collections import defaultdict
from typing import Dict, List

import numpy as np

from . import features, predstore

_COMP = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', '-': '-', 'N': 'N'}
_MU = {'A': -0.9, 'C': -0.2, 'G': 0.4, 'T': 1.0}


def synthetic_genome(length: int, seed: int = 1) -> str:
    rng = np.random.default_rng(seed)
    return ''.join(rng.choice(list('ACGT'), length))


def synthetic_read(rng, genome: str, chrom: str, readk: str, min_len=2000, max_len=10000,
                   p_sub=0.06, p_ins=0.02, p_del=0.02, max_clip=20) -> Dict:
    strand = '+' if rng.random() < 0.5 else '-'
    span = int(rng.integers(min_len, max_len + 1))
    start = int(rng.integers(0, len(genome) - span - 1))
    rows = []        # reference-orientation alignment rows (refbase, readbase, refpos)
    pos = start
    while pos < start + span:
        interior = start < pos < start + span - 1
        u = rng.random()
        if interior and u < p_ins:
            rows.append(('-', str(rng.choice(list('ACGT'))), pos))
            continue
        if interior and u < p_ins + p_del:
            rows.append((genome[pos], '-', pos))
            pos += 1
            continue
        rb = genome[pos]
        if interior and rng.random() < p_sub:
            rb = str(rng.choice([b for b in 'ACGT' if b != genome[pos]]))
        rows.append((genome[pos], rb, pos))
        pos += 1
    refb = np.array([r[0] for r in rows], dtype='U1')
    readb = np.array([r[1] for r in rows], dtype='U1')
    refi = np.array([r[2] for r in rows], dtype=np.uint64)
    if strand == '-':   # handle_record flips the table and complements both bases (myDetect.py:661-666)
        refb = np.array([_COMP[b] for b in refb[::-1]], dtype='U1')
        readb = np.array([_COMP[b] for b in readb[::-1]], dtype='U1')
        refi = refi[::-1].copy()
    readi = np.cumsum(readb != '-') - (readb != '-')
    bmi = predstore.make_base_map_info(refb, readb, refi, readi.astype(np.uint64))
    start_clip, end_clip = int(rng.integers(0, max_clip + 1)), int(rng.integers(0, max_clip + 1))
    aligned_bases = readb[readb != '-']
    bases = np.concatenate([rng.choice(list('ACGT'), start_clip), aligned_bases, rng.choice(list('ACGT'), end_clip)])
    n_ev = len(bases)
    mu = np.array([_MU[b] for b in bases])
    mean = np.round(np.clip(rng.normal(mu, 0.3), -5, 5), 3)
    stdv = np.round(np.abs(rng.normal(0.25, 0.15, n_ev)), 3)
    length = rng.geometric(0.12, n_ev)
    events = predstore.events_from_bases(bases, mean, stdv, length)
    nins = int((refb == '-').sum())
    ndel = int((readb == '-').sum())
    mapped_start = int(refi.min())
    sp_param = {'f5data': {readk: (None, events, None, readk)}, 'f5status': ''}
    mfeat, isdif = features.get_Feature({'fnum': 7}, {'Error': defaultdict(list)}, sp_param, None, sp_param['f5data'],
                                        readk, start_clip, end_clip, bmi, strand, chrom, mapped_start, nins, ndel)
    assert not isdif
    return {'readk': readk, 'chr': chrom, 'strand': strand, 'mapped_start': mapped_start, 'start_clip': start_clip,
            'end_clip': end_clip, 'events': events, 'base_map_info': bmi, 'mfeatures': mfeat}


def write_synthetic_run(out_dir: str, n_reads: int = 100, reads_per_file: int = 5, genome_len: int = 100000,
                        seed: int = 1, chrom: str = 'NC_000913.3', **read_kw) -> List[str]:
    os.makedirs(out_dir, exist_ok=True)
    genome = synthetic_genome(genome_len, seed)
    rng = np.random.default_rng(seed + 1)
    files, batch = [], []
    for i in range(n_reads):
        batch.append(synthetic_read(rng, genome, chrom, 'read_%05d' % i, **read_kw))
        if len(batch) == reads_per_file or i == n_reads - 1:
            path = os.path.join(out_dir, 'reads_%04d%s' % (len(files), predstore.CONTAINER_SUFFIX))
            predstore.save_feature_container(path, batch)
            files.append(path)
            batch = []
    return files


# ---------------------------------------------------------------------------------------------
# raw reads: DAC samples + basecaller event table + truth SAM + FASTA (inputs of the whole path)
# ---------------------------------------------------------------------------------------------
def _revcomp(s: str) -> str:
    return ''.join(_COMP[c] for c in reversed(s))


def synthetic_raw_read(rng, genome: str, chrom: str, read_id: str, min_len=400, max_len=1500, p_sub=0.06, p_ins=0.02,
                       p_del=0.02, max_clip=12, p_stay=0.25) -> Dict:
    """-> {'read_id', 'raw' int16, 'events_data', 'sam' line}: a truth alignment with substitutions / indels / soft
    clips on either strand; one basecaller event (plus `stay` continuation events, move == 0) per read base; the raw
    signal is 520 + 75 * (level of the base + noise) DAC counts with ~8 samples per event."""
    strand = '+' if rng.random() < 0.5 else '-'
    span = int(rng.integers(min_len, max_len + 1))
    start = int(rng.integers(0, len(genome) - span - 1))
    seq, cig = [], []

    def push(op, n=1):
        if cig and cig[-1][0] == op:
            cig[-1][1] += n
        else:
            cig.append([op, n])
    pos = start
    while pos < start + span:
        interior = start + 2 < pos < start + span - 3
        u = rng.random()
        if interior and u < p_ins:
            seq.append(str(rng.choice(list('ACGT'))))
            push('I')
            continue
        if interior and u < p_ins + p_del:
            pos += 1
            push('D')
            continue
        b = genome[pos]
        if interior and rng.random() < p_sub:
            b = str(rng.choice([x for x in 'ACGT' if x != genome[pos]]))
        seq.append(b)
        push('M')
        pos += 1
    lead, tail = int(rng.integers(0, max_clip + 1)), int(rng.integers(0, max_clip + 1))
    samseq = ''.join(rng.choice(list('ACGT'), lead)) + ''.join(seq) + ''.join(rng.choice(list('ACGT'), tail))
    cigar = ('%dS' % lead if lead else '') + ''.join('%d%s' % (n, op) for op, n in cig) + ('%dS' % tail if tail else '')
    basecall = samseq if strand == '+' else _revcomp(samseq)
    # events and raw signal in sequencing orientation
    from . import rawreads
    ev_rows, chunks = [], []
    cursor = int(rng.integers(20, 200))
    chunks.append(np.round(520 + 75 * rng.normal(0, 1.0, cursor)))
    for b in basecall:
        n_sub = 1 + int(rng.random() < p_stay) + int(rng.random() < p_stay * 0.3)
        for k in range(n_sub):
            ln = int(rng.geometric(0.2)) + 1
            lvl = rng.normal(_MU[b], 0.3, ln)
            chunks.append(np.round(520 + 75 * lvl))
            ev_rows.append((float(lvl.mean()), float(lvl.std()), cursor, ln, 'NN' + b + 'NN', 1 if k == 0 else 0))
            cursor += ln
    chunks.append(np.round(520 + 75 * rng.normal(0, 1.0, int(rng.integers(5, 60)))))
    raw = np.clip(np.concatenate(chunks), -32768, 32767).astype(np.int16)
    events_data = np.array(ev_rows, dtype=rawreads.EVENTS_DATA_DTYPE)
    sam = '\t'.join([read_id, '0' if strand == '+' else '16', chrom, str(start + 1), '60', cigar, '*', '0', '0', samseq, '*'])
    return {'read_id': read_id, 'raw': raw, 'events_data': events_data, 'sam': sam}


def write_synthetic_raw_run(out_dir: str, n_reads: int = 40, reads_per_file: int = 5, genome_len: int = 30000, seed: int = 1,
                            chrom: str = 'NC_000913.3', **read_kw):
    """Raw containers + side-car SAM files + genome FASTA.  -> (container paths, fasta path)"""
    from . import rawreads
    os.makedirs(out_dir, exist_ok=True)
    genome = synthetic_genome(genome_len, seed)
    fasta = os.path.join(out_dir, 'genome.fa')
    with open(fasta, 'w') as fh:
        fh.write('>%s synthetic\n' % chrom)
        for i in range(0, len(genome), 60):
            fh.write(genome[i:i + 60].lower() if (i // 60) % 7 == 3 else genome[i:i + 60])   # soft-masked stretches: upper-cased on load
            fh.write('\n')
    rng = np.random.default_rng(seed + 11)
    files, batch = [], []
    for i in range(n_reads):
        batch.append(synthetic_raw_read(rng, genome, chrom, 'rawread_%05d' % i, **read_kw))
        if len(batch) == reads_per_file or i == n_reads - 1:
            stem = os.path.join(out_dir, 'raw_%04d' % len(files))
            rawreads.save_raw_container(stem + rawreads.RAW_SUFFIX, batch)
            with open(stem + '.sam', 'w') as fh:
                fh.write('@SQ\tSN:%s\tLN:%d\n' % (chrom, len(genome)))
                for rd in batch:
                    fh.write(rd['sam'] + '\n')
            files.append(stem + rawreads.RAW_SUFFIX)
            batch = []
    return files, fasta
