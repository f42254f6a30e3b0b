"""SAM record -> base_map_info -> features -> prediction for raw reads.

Host-side mirror of the reference's `handle_line` (myDetect.py:929-943), `getRefSeq` (:472-486) and `handle_record`
(:491-782): same names, arguments and side effects (`sp_options['Mod']`, `sp_options['Error']`).  The per-base CIGAR
walk runs in compiled code behind `dm_map_read` (deepmod_amd/csrc/readmap.inc); `samtools faidx` is replaced by a
plain FASTA reader (no external binaries in this image) and the HDF5 prediction file by deepmod_amd.predstore.
"""
from __future__ import annotations

import ctypes
from collections import defaultdict
from typing import Dict

import numpy as np

from . import _lib, features, predstore

OUTPUT_WARNING = 2   # myCom.py:7 (OUTPUT_DEBUG 0, OUTPUT_INFO 1, OUTPUT_WARNING 2, OUTPUT_ERROR 3)


# SAM fields of a record and the checks that make it unusable, in the order the reference tests them (the status strings are contract:
# the error ledger is keyed by them, myDetect.py:931-936).  An int() of a malformed field raises ValueError as it does there.
_SAM_FIELDS = ('qname', 'flag', 'rname', 'pos', 'mapq', 'cigar')
_UNUSABLE = (
    ("qname is *", lambda r: r['qname'] == '*'),
    ("mapq is 255", lambda r: int(r['mapq']) == 255),
    ("pos is 0", lambda r: int(r['pos']) == 0),
    ("cigar is *", lambda r: r['cigar'] == '*'),
    ("rname is *", lambda r: r['rname'] == '*'),
)


def handle_line(moptions, sp_param, f5align):
    """One SAM line (sp_param['line']) -> f5align[qname] = (mapq, flag, rname, pos, cigar, seq); of several records of a read the one
    with the best mapq is kept; a record that cannot be used sets sp_param['f5status'] to the first reason that applies and changes
    nothing else (behaviour of myDetect.py:929-943).  Returns qname."""
    cols = sp_param['line'].split('\t')
    if len(cols) < 11:
        raise ValueError('a SAM record has 11 mandatory fields, this line has %d' % len(cols))
    rec = dict(zip(_SAM_FIELDS, cols[:6]))
    rec['seq'] = cols[9]
    reason = next((status for status, applies in _UNUSABLE if applies(rec)), None)
    if reason is not None:
        sp_param['f5status'] = reason
    if sp_param['f5status'] != "":          # (also a status an earlier step of this read left behind)
        return rec['qname']
    mapq = int(rec['mapq'])
    best = f5align.get(rec['qname'])
    if best is None or best[0] < mapq:
        f5align[rec['qname']] = (mapq, int(rec['flag']), rec['rname'], int(rec['pos']), rec['cigar'], rec['seq'])
    return rec['qname']


_fasta_cache: Dict[str, Dict[str, str]] = {}


def read_fasta(path: str) -> Dict[str, str]:
    """{name: upper-cased sequence}; name = first word of the header (what `samtools faidx ref name` resolves)."""
    if path not in _fasta_cache:
        seqs, name, parts = {}, None, []
        with open(path) as fh:
            for line in fh:
                line = line.strip()
                if line.startswith('>'):
                    if name is not None:
                        seqs[name] = ''.join(parts).upper()
                    name, parts = line[1:].split()[0], []
                elif line:
                    parts.append(line)
        if name is not None:
            seqs[name] = ''.join(parts).upper()
        _fasta_cache[path] = seqs
    return _fasta_cache[path]


def getRefSeq(moptions, sp_param, rname):
    """sp_param['ref_info'][rname] = upper-cased chromosome sequence (myDetect.py:472-486)."""
    seqs = read_fasta(moptions['Ref'])
    if rname not in seqs:
        print('Fatal Error!!! cannot find the chrosome sequence %s' % rname)
    else:
        sp_param['ref_info'][rname] = seqs[rname]


def map_read(flag: int, pos1: int, cigar: str, readseq: str, refseq, n_events: int):
    """dm_map_read wrapper -> dict(status, base_map_info, leftclip, rightclip, ev_lo, ev_hi, counts, ...).
    `refseq` may be a str or an ASCII bytes object (pass bytes to avoid re-encoding a chromosome per read)."""
    lib = _lib.load()
    ref_b = refseq if isinstance(refseq, (bytes, bytearray)) else refseq.encode('ascii')
    seq_b = readseq.encode('ascii')
    info = (ctypes.c_int64 * _lib.DM_MAP_INFO_LEN)()
    cap = len(seq_b) + 64
    while True:
        refb = np.empty(cap, 'S1')
        readb = np.empty(cap, 'S1')
        refi = np.empty(cap, np.uint64)
        readi = np.empty(cap, np.uint64)
        _lib.check(lib.dm_map_read(int(flag), int(pos1), cigar.encode('ascii'), seq_b, len(seq_b), ref_b, len(ref_b),
                                   int(n_events), refb.ctypes.data, readb.ctypes.data, refi.ctypes.data, readi.ctypes.data,
                                   cap, info))
        if info[_lib.DM_MAP_STATUS] != _lib.DM_MAP_NEED_ROWS:
            break
        cap = int(info[_lib.DM_MAP_N_ROWS])
    out = {'status': int(info[_lib.DM_MAP_STATUS]), 'strand': '-' if info[_lib.DM_MAP_STRAND] else '+',
           'pos_after_clip': int(info[_lib.DM_MAP_POS_AFTER_CLIP]), 'events_after_clip': int(info[_lib.DM_MAP_EVENTS_AFTER_CLIP]),
           'num_insertions': int(info[_lib.DM_MAP_NUM_INSERT]), 'num_deletions': int(info[_lib.DM_MAP_NUM_DELETE]),
           'num_mismatches': int(info[_lib.DM_MAP_NUM_MISMATCH])}
    if out['status'] == _lib.DM_MAP_OK:
        n = int(info[_lib.DM_MAP_N_ROWS])
        out['base_map_info'] = predstore.make_base_map_info(predstore.s1_to_u1(refb[:n]), predstore.s1_to_u1(readb[:n]), refi[:n], readi[:n])
        out['table_s1'] = (refb[:n], readb[:n], refi[:n].astype(np.int64))      # the same columns as the C walk wrote them (streaming path)
        out.update(leftclip=int(info[_lib.DM_MAP_LEFTCLIP]), rightclip=int(info[_lib.DM_MAP_RIGHTCLIP]),
                   ev_lo=int(info[_lib.DM_MAP_EV_LO]), ev_hi=int(info[_lib.DM_MAP_EV_HI]),
                   first_match_pos=int(info[_lib.DM_MAP_FIRST_MATCH_POS]), last_match_pos=int(info[_lib.DM_MAP_LAST_MATCH_POS]))
    return out


def _in_region(moptions, rname, pos=None, n_events=None):
    for cur_mr in moptions.get('region', [[None, None, None]]):
        if cur_mr[0] in ['', None, rname]:
            if pos is None:
                return True
            if (cur_mr[1] in ['', None] or pos > cur_mr[1]) and (cur_mr[2] in ['', None] or pos + n_events < cur_mr[2]):
                return True
    return False


def map_records(moptions, sp_options, sp_param, f5align, f5data):
    """First half of handle_record (myDetect.py:491-713): alignment table, clips and feature matrix per aligned read.
    -> list of read dicts in the form mPredict1 / mPredict_batch / PredWriter consume."""
    reads = []
    ref_bytes = sp_param.setdefault('ref_bytes', {})
    for readk_ind, readk in enumerate(list(f5align.keys())):
        sp_param['f5status'] = ""
        sp_param['mfile_path'] = f5data[readk][3]
        mapq, flag, rname, pos, cigar, readseq = f5align[readk]
        if (not moptions.get('ConUnk', True)) and any(ch in rname for ch in '_-/:'):
            continue
        if not _in_region(moptions, rname):
            continue
        if rname not in sp_param['ref_info']:
            getRefSeq(moptions, sp_param, rname)
        if rname not in sp_param['ref_info']:
            sp_options["Error"]["No reference sequence"].append(f5data[readk][3])
            continue
        if rname not in ref_bytes:
            ref_bytes[rname] = sp_param['ref_info'][rname].encode('ascii')
        events = f5data[readk][1]
        try:
            mp = map_read(flag, pos, cigar, readseq, ref_bytes[rname], len(events))
        except _lib.DeepModHipError as exc:
            sp_options["Error"]["CIGAR-Error: %s" % exc].append(f5data[readk][3])
            continue
        if not _in_region(moptions, rname, mp['pos_after_clip'], mp['events_after_clip']):
            continue
        if mp['status'] == _lib.DM_MAP_NO_MATCH:
            if moptions.get('outLevel', OUTPUT_WARNING) <= OUTPUT_WARNING:
                print("Errorfast5 " + f5data[readk][3])
                print('match-Error!!! no first and/or last match', f5data[readk][3], str(flag), rname, str(pos))
            continue
        if mp['ev_hi'] - mp['ev_lo'] < 50:                                            # :702-705
            sp_param['f5status'] = "Less Event"
            sp_options["Error"]["Less Event"].append(f5data[readk][3])
            continue
        bmi = mp['base_map_info']
        sp_param['f5data'] = f5data
        mfeatures, isdif = features.get_Feature(moptions, sp_options, sp_param, f5align, f5data, readk, mp['leftclip'],
                                                mp['rightclip'], bmi, mp['strand'], rname, mp['first_match_pos'],
                                                mp['num_insertions'], mp['num_deletions'])
        if not sp_param['f5status'] == "":
            continue
        reads.append({'readk': readk, 'readk_ind': readk_ind, 'chr': rname, 'strand': mp['strand'],
                      'mapped_start': f5align[readk][3] - 1, 'start_clip': mp['leftclip'], 'end_clip': mp['rightclip'],
                      'base_map_info': bmi, 'table_s1': mp['table_s1'], 'mfeatures': mfeatures, 'events': events, 'src': f5data[readk][3],
                      'num_insertions': mp['num_insertions'], 'num_deletions': mp['num_deletions'],
                      'num_mismatches': mp['num_mismatches']})
    return reads


def parse_sam(moptions, sp_options, sp_param, align_info, f5data):
    """SAM text lines -> f5align; reads without a usable record go to the error channel (myDetect.py:436-457)."""
    f5align = defaultdict()
    f5keydict = {}
    for line in align_info:
        line = line.strip()
        if len(line) == 0 or line[0] == '@':
            continue
        sp_param['f5status'] = ""
        sp_param['line'] = line
        qname = handle_line(moptions, sp_param, f5align)
        if sp_param['f5status'] == "":
            f5keydict[qname] = True
    for f5k in sorted(f5data.keys()):
        if f5k not in f5keydict:
            sp_options["Error"]["Not in alignment sam"].append(f5data[f5k][3])
    for qname in list(f5align.keys()):
        if qname not in f5data:
            del f5align[qname]
    return f5align
