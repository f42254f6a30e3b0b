"""Per-chromosome merge of DeepMod BED files from several runs (counterpart of the reference's
/root/reference/DeepMod_tools/sum_chr_mod.py: readbed2 :37-46, mergeMod :48-54, save_mod :56-66, file discovery :77-84).

For one chromosome: every `*.<chr>-.<Base>.bed` and `*.<chr>+.<Base>.bed` one to three directory levels below the
prediction folder is read, coverage (column 10) and modified count (column 12) are summed per (chr, position, strand),
positions whose summed modified count is 0 are dropped, and the rest is written sorted by (chr, position, strand)
in the tool's own dialect (two spaces after the strand column, percentage = int(mod * 100 / cov)).
Integer sums of a few 10^5 - 10^7 text rows: parse-bound, stays on the host (numpy, no per-row Python dict).
"""
from __future__ import annotations

import glob
import os
from typing import Iterable, List

import numpy as np


def find_bed_files(pred_folder: str, ck: str, base: str) -> List[str]:
    files: List[str] = []
    for strand in ('-', '+'):
        for depth in ('*/*/*/', '*/*/', '*/'):
            files.extend(glob.glob(os.path.join(pred_folder, '%s*.%s%s.%s.bed' % (depth, ck, strand, base))))
    return files


def readbed2(bedf: str):
    """-> (chrom str array, pos int64, strand str array, cov int64, mod int64)"""
    chrom, pos, strand, cov, mod = [], [], [], [], []
    with open(bedf) as mr:
        for line in mr:
            lsp = line.split()
            if not lsp:
                continue
            chrom.append(lsp[0])
            pos.append(int(lsp[1]))
            strand.append(lsp[5])
            cov.append(int(lsp[9]))
            mod.append(int(lsp[11]))
    return (np.array(chrom, dtype=object), np.array(pos, np.int64), np.array(strand, dtype=object),
            np.array(cov, np.int64), np.array(mod, np.int64))


def merge_beds(bed_files: Iterable[str]):
    parts = [readbed2(f) for f in bed_files]
    parts = [p for p in parts if len(p[1])]
    if not parts:
        e = np.zeros(0, np.int64)
        return np.zeros(0, object), e, np.zeros(0, object), e, e
    chrom = np.concatenate([p[0] for p in parts])
    pos = np.concatenate([p[1] for p in parts])
    strand = np.concatenate([p[2] for p in parts])
    cov = np.concatenate([p[3] for p in parts])
    mod = np.concatenate([p[4] for p in parts])
    # sort by the tuple (chr, pos, strand) the way Python sorts the reference's dict keys
    order = np.lexsort((strand.astype(str), pos, chrom.astype(str)))
    chrom, pos, strand, cov, mod = chrom[order], pos[order], strand[order], cov[order], mod[order]
    new = np.r_[True, (chrom[1:] != chrom[:-1]) | (pos[1:] != pos[:-1]) | (strand[1:] != strand[:-1])]
    heads = np.flatnonzero(new)
    return chrom[heads], pos[heads], strand[heads], np.add.reduceat(cov, heads), np.add.reduceat(mod, heads)


def save_mod(res_file: str, merged, base: str) -> int:
    chrom, pos, strand, cov, mod = merged
    keep = mod != 0
    with open(res_file, 'w') as mw:
        for c, p, s, cv, md in zip(chrom[keep], pos[keep].tolist(), strand[keep], cov[keep].tolist(), mod[keep].tolist()):
            mw.write('%s %d %d %s %d %s  %d %d 0,0,0 %d %d %d\n' % (c, p, p + 1, base, cv if cv < 1000 else 1000, s, p, p + 1, cv,
                                                                    int(md * 100 / cv) if cv > 0 else 0, md))
    return int(keep.sum())


def sum_chr_mod(pred_folder: str, base: str, sum_fileid: str, chrkeys=None, verbose: bool = True):
    """Merge every chromosome in chrkeys (default chr1..22, X, Y, M) -> {chr: output path}."""
    if chrkeys is None:
        chrkeys = ['chr%d' % i for i in range(1, 23)] + ['chrX', 'chrY', 'chrM']
    out = {}
    for ck in sorted(set(chrkeys)):
        files = find_bed_files(pred_folder, ck, base)
        if verbose:
            print("%s -+ %s: %d" % (ck, base, len(files)), flush=True)
        res_file = "%s/%s.%s.%s.bed" % (pred_folder, sum_fileid, ck, base)
        save_mod(res_file, merge_beds(files), base)
        out[ck] = res_file
    return out
