"""Positions of a base and of a motif in a reference genome (counterpart of the reference's
/root/reference/DeepMod_tools/generate_motif_pos.py: read_genome :9-27, handle_motif_pos :30-72), the input of the
CpG-cluster stage (`motif_<chr>_<Base>.bed`) and of per-base evaluation (`na_<chr>_<Base>.bed`).

Per chromosome: `na` file = one line `chr\\tpos\\t+` for every position holding the base and `chr\\tpos\\t-` for every position
holding its complement; `motif` file = for every position where the motif matches with the base of interest at offset
`motif_pos`: `chr\\tpos\\t+` and `chr\\tpos+1\\t-` (the reference's hard-wired CpG partner rule, :62-63).
A byte scan of the sequence: vectorised comparisons over the ASCII array instead of a per-base Python loop.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import numpy as np

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "t", "c": "g", "g": "c", "t": "a", "N": "N", "n": "n"}


def read_genome(mfafile: str) -> Dict[str, str]:
    ref_genome, cur_chr, seqlist = {}, None, []
    with open(mfafile) as mr:
        for line in mr:
            line = line.strip()
            if len(line) == 0:
                continue
            if line[0] == '>':
                if cur_chr is not None:
                    ref_genome[cur_chr] = ''.join(seqlist)
                cur_chr, seqlist = line[1:].split()[0], []
            else:
                seqlist.append(line.upper())
    if cur_chr is not None:
        ref_genome[cur_chr] = ''.join(seqlist)
    print("Total chr: {}".format(len(ref_genome)), flush=True)
    return ref_genome


def motif_positions(seq: str, curna: str, motif: Optional[str], motif_pos: int):
    """-> (na_pos int64, na_is_plus bool, motif_pos int64): positions of the base / its complement, and of motif hits."""
    s = np.frombuffer(seq.encode('ascii'), dtype=np.uint8)
    is_plus = s == ord(curna)
    # na_bp[ref] == curna  <=>  ref is a base whose complement is curna (only keys of the table qualify)
    comp_srcs = [k for k, v in _COMP.items() if v == curna]
    is_minus = np.zeros(len(s), bool)
    for k in comp_srcs:
        is_minus |= s == ord(k)
    na = np.flatnonzero(is_plus | is_minus)
    hits = np.zeros(0, np.int64)
    if motif is not None and len(s) >= len(motif):
        m = np.frombuffer(motif.encode('ascii'), dtype=np.uint8)
        n = len(s) - len(m) + 1
        ok = np.ones(n, bool)
        for j, ch in enumerate(m):
            ok &= s[j:j + n] == ch
        starts = np.flatnonzero(ok)                       # window start = na_ind - motif_pos
        pos = starts + motif_pos
        hits = pos[(pos >= 0) & (pos < len(s)) & (s[np.clip(pos, 0, len(s) - 1)] == ord(curna))]
    return na, is_plus[na], hits


def generate_motif_pos(ref_fa: str, res_folder: str, curna: str, motif: str, motif_pos: int, chrkeys: Optional[Iterable[str]] = None):
    if not res_folder.endswith('/'):
        res_folder += '/'
    os.makedirs(res_folder, exist_ok=True)
    if chrkeys is None:
        chrkeys = ['chr%d' % i for i in range(1, 23)] + ['chrX', 'chrY', 'chrM']
    ref_genome = read_genome(ref_fa)
    out = []
    for rgkey in sorted(set(chrkeys)):
        seq = ref_genome[rgkey]
        print("get motif for {}={}".format(rgkey, len(seq)), flush=True)
        na, plus, hits = motif_positions(seq, curna, motif, motif_pos)
        nafile = '%sna_%s_%s.bed' % (res_folder, rgkey, curna)
        motiffile = '%smotif_%s_%s.bed' % (res_folder, rgkey, curna)
        with open(nafile, 'w') as mw:
            mw.write(''.join('%s\t%d\t%s\n' % (rgkey, p, '+' if pl else '-') for p, pl in zip(na.tolist(), plus.tolist())))
        with open(motiffile, 'w') as mw:
            mw.write(''.join('%s\t%d\t+\n%s\t%d\t-\n' % (rgkey, p, rgkey, p + 1) for p in hits.tolist()))
        out.extend([nafile, motiffile])
    return out
