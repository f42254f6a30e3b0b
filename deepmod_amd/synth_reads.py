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
                   p_sub=0.06, p_ins=0.02, p_del=0.02, max_clip=20, p_tail=0.0) -> Dict:
    """p_tail > 0: that fraction of the events gets READ-SHAPED extremes a nominal draw hardly produces - a normalised mean anywhere in
    the clip range with a third of them exactly on the clip (+-5, myDetect.py's clipped MAD normalisation) and a length log-uniform
    between 50 and 30,000 samples (stalled events).  p_tail = 0 draws exactly what earlier rounds drew (same random stream)."""
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
    if p_tail > 0:
        tail = rng.random(n_ev) < p_tail
        kind = rng.random(n_ev)
        wide = np.round(rng.uniform(-5, 5, n_ev), 3)
        mean = np.where(tail, np.where(kind < 1 / 6, -5.0, np.where(kind < 1 / 3, 5.0, wide)), mean)
        length = np.where(tail, np.exp(rng.uniform(np.log(50.0), np.log(30000.0), n_ev)).astype(np.int64), length)
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
                            chrom: str = 'NC_000913.3', part: int = 0, **read_kw):
    """Raw containers + side-car SAM files + genome FASTA.  -> (container paths, fasta path).
    `part` > 0 writes another slice of the same run (same genome, its own reads and file names; the FASTA is part 0's),
    so that a large run can be generated by several processes."""
    from . import rawreads
    os.makedirs(out_dir, exist_ok=True)
    genome = synthetic_genome(genome_len, seed)
    fasta = os.path.join(out_dir, 'genome.fa')
    if part == 0:
        with open(fasta, 'w') as fh:
            fh.write('>%s synthetic\n' % chrom)
            for i in range(0, len(genome), 60):
                fh.write(genome[i:i + 60].lower() if (i // 60) % 7 == 3 else genome[i:i + 60])   # soft-masked stretches: upper-cased on load
                fh.write('\n')
    rng = np.random.default_rng(seed + 11 + 1000003 * part)
    files, batch = [], []
    for i in range(n_reads):
        batch.append(synthetic_raw_read(rng, genome, chrom, 'rawread_%05d' % i if part == 0 else 'rawread_p%d_%05d' % (part, i), **read_kw))
        if len(batch) == reads_per_file or i == n_reads - 1:
            stem = os.path.join(out_dir, 'raw_%04d' % len(files) if part == 0 else 'raw_p%d_%04d' % (part, len(files)))
            rawreads.save_raw_container(stem + rawreads.RAW_SUFFIX, batch)
            with open(stem + '.sam', 'w') as fh:
                fh.write('@SQ\tSN:%s\tLN:%d\n' % (chrom, len(genome)))
                for rd in batch:
                    fh.write(rd['sam'] + '\n')
            files.append(stem + rawreads.RAW_SUFFIX)
            batch = []
    return files, fasta


# ---------------------------------------------------------------------------------------------
# full-size synthetic runs (BASELINE config 3: 4.64 Mb genome at 30x = ~1.4e8 base-positions): vectorised per read,
# written straight into packed feature containers (deepmod_amd/predstore.py, format 2)
# ---------------------------------------------------------------------------------------------
_ACGT = np.frombuffer(b'ACGT', np.uint8)
_COMP_LUT = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b'ACGT', b'TGCA'):
    _COMP_LUT[_a] = _b
_MU_LUT = np.zeros(256, np.float64)
for _b, _v in _MU.items():
    _MU_LUT[ord(_b)] = _v
_CODE_LUT = np.full(256, 4, np.int64)           # A, C, G, T -> 0..3, anything else 4
for _i, _b in enumerate(b'ACGT'):
    _CODE_LUT[_b] = _i


def synthetic_genome_codes(length: int, seed: int = 1) -> np.ndarray:
    """uint8 ASCII codes of a uniform ACGT genome."""
    return _ACGT[np.random.default_rng(seed).integers(0, 4, length)]


def synthetic_packed_read(rng, genome: np.ndarray, chrom: str, readk: str, min_len=2000, max_len=10000, p_sub=0.06,
                          p_ins=0.02, p_del=0.02, max_clip=20, keep_events=False) -> Dict:
    """One aligned read in the packed layout (tx rows, table columns, event bases): truth alignment with substitutions,
    single-base insertions and deletions away from the read ends, soft clips, either strand; one event per read base
    (SURVEY.md 8d generators).  The feature rows are what features.get_Feature builds for this read (tested)."""
    strand = '+' if rng.random() < 0.5 else '-'
    span = int(rng.integers(min_len, max_len + 1))
    start = int(rng.integers(0, len(genome) - span - 1))
    idx = np.arange(span)
    interior = (idx > 0) & (idx < span - 1)
    ins = (interior & (rng.random(span) < p_ins)).astype(np.int64)          # an inserted read base before this position
    dele = interior & (rng.random(span) < p_del)
    sub = interior & ~dele & (rng.random(span) < p_sub)
    g = genome[start:start + span]
    readb_pos = np.where(sub, _ACGT[(_CODE_LUT[g] + rng.integers(1, 4, span)) % 4], g)
    readb_pos = np.where(dele, np.uint8(ord('-')), readb_pos)
    first_row = np.cumsum(1 + ins) - (1 + ins)
    nrow = int(span + ins.sum())
    refb = np.full(nrow, ord('-'), np.uint8)
    readb = np.empty(nrow, np.uint8)
    refi = np.repeat(start + idx, 1 + ins).astype(np.int64)
    pos_rows = first_row + ins
    refb[pos_rows] = g
    readb[pos_rows] = readb_pos
    ins_rows = first_row[ins == 1]
    readb[ins_rows] = _ACGT[rng.integers(0, 4, len(ins_rows))]
    if strand == '-':        # handle_record flips the table and complements both bases (myDetect.py:661-666)
        refb, readb, refi = _COMP_LUT[refb[::-1]], _COMP_LUT[readb[::-1]], refi[::-1].copy()
    aligned = np.flatnonzero(readb != ord('-'))
    n = len(aligned)
    start_clip, end_clip = int(rng.integers(0, max_clip + 1)), int(rng.integers(0, max_clip + 1))
    bases = np.concatenate([_ACGT[rng.integers(0, 4, start_clip)], readb[aligned], _ACGT[rng.integers(0, 4, end_clip)]])
    nev = len(bases)
    mean = np.round(np.clip(rng.normal(_MU_LUT[bases], 0.3), -5, 5), 3).astype(np.float32)
    stdv = np.round(np.abs(rng.normal(0.25, 0.15, nev)), 3).astype(np.float32)
    length = rng.geometric(0.12, nev).astype(np.float32)
    tx = np.zeros((n + 200, 7), np.float32)
    code = _CODE_LUT[refb[aligned]]
    hit = np.flatnonzero(code < 4)
    tx[100 + hit, code[hit]] = 1.0
    ie = np.arange(start_clip - 100, nev - end_clip + 100)
    ok = (ie >= 0) & (ie < nev)
    tx[ok, 4] = mean[ie[ok]]
    tx[ok, 5] = stdv[ie[ok]]
    tx[ok, 6] = length[ie[ok]]
    rd = {'readk': readk, 'chr': chrom, 'strand': strand, 'mapped_start': int(refi.min()), 'start_clip': start_clip,
          'end_clip': end_clip, 'tx': tx, 'refbase': refb.view('S1'), 'readbase': readb.view('S1'), 'refbasei': refi,
          'evbase': bases.view('S1')}
    if keep_events:
        rd['_events'] = (mean, stdv, length)
    return rd


def write_synthetic_packed_run(out_dir: str, genome_len: int = 4_641_652, coverage: float = 30.0, reads_per_file: int = 100,
                               seed: int = 1, chrom: str = 'NC_000913.3', first_file: int = 0, n_files: int = None,
                               **read_kw) -> List[str]:
    """Packed feature containers covering `genome_len` at `coverage` (read lengths U[min_len, max_len]).  Containers are
    independent of each other (container i is seeded with seed + i), so a subset [first_file, first_file + n_files) can be
    regenerated alone - the oracle comparison of the full-size test uses the first few."""
    os.makedirs(out_dir, exist_ok=True)
    genome = synthetic_genome_codes(genome_len, seed)
    mean_len = 0.5 * (read_kw.get('min_len', 2000) + read_kw.get('max_len', 10000))
    total_files = max(1, int(np.ceil(coverage * genome_len / mean_len / reads_per_file)))
    last = total_files if n_files is None else min(total_files, first_file + n_files)
    files = []
    for fi in range(first_file, last):
        rng = np.random.default_rng([seed, fi])
        reads = [synthetic_packed_read(rng, genome, chrom, 'read_%05d_%03d' % (fi, j), **read_kw) for j in range(reads_per_file)]
        path = os.path.join(out_dir, 'reads_%05d%s' % (fi, predstore.CONTAINER_SUFFIX))
        predstore.save_packed_container(path, reads, {chrom: genome_len})
        files.append(path)
    return files
