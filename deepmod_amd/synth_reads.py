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
