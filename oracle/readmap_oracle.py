"""ORACLE — TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).  Never imported by deepmod_amd/.

Pure-Python restatement of the alignment walk of the reference's handle_record
(/root/reference/bin/DeepMod_scripts/myDetect.py:515-714), used to check dm_map_read on randomised records beyond
the golden set.  PARITY PIN: tests/golden/host_record.json (output of the reference's own handle_record, see
tests/golden/make_golden_record.py); tests/test_readmap.py checks this restatement against it.
"""
from __future__ import annotations

import re

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "t", "c": "g", "g": "c", "t": "a", "N": "N", "n": "n"}   # myCom.py:14-25
_CLIPPABLE = "IDNSHPX"


def map_read(flag, pos1, cigar, readseq, refseq, n_events):
    """-> None (no matching base, :617-622) or dict(rows, leftclip, rightclip, ev_lo, ev_hi, counts, strand).
    rows = list of [refbase, readbase, refpos, readpos] in the orientation the reference stores them."""
    counts = [int(x) for x in re.findall(r"\d+", cigar)]                     # :497, :519
    ops = re.findall(r"[MIDNSHPX=]{1}", cigar)                               # :498, :518
    pos = pos1 - 1                                                           # :515
    strand = "-" if flag & 0x10 else "+"                                     # :516
    left = right = 0
    while ops[0] in _CLIPPABLE:                                              # :524-530
        if ops[0] in "ISX":
            left += counts[0]
            readseq = readseq[counts[0]:]
        if ops[0] == "H":
            left += counts[0]
        if ops[0] in "DNX":
            pos += counts[0]
        counts, ops = counts[1:], ops[1:]
    while ops[-1] in _CLIPPABLE:                                             # :532-536
        if ops[-1] in "ISX":
            right += counts[-1]
            readseq = readseq[:-counts[-1]]
        if ops[-1] == "H":
            right += counts[-1]
        counts, ops = counts[:-1], ops[:-1]
    ev = list(range(n_events))                                               # stand-in for the event table: indices
    if strand == "+":                                                        # :537-542
        ev = ev[left:-right] if right > 0 else ev[left:]
    else:
        ev = ev[right:-left] if left > 0 else ev[right:]
    pos_after_clip, events_after_clip = pos, len(ev)
    rows = []
    firstmatch = lastmatch = first_al = last_al = first_pos = last_pos = None
    nmis = nins = ndel = 0
    ri = 0
    for cnt, op in zip(counts, ops):                                         # :566-616
        for _ in range(cnt):
            if op in "M=X":
                rows.append([refseq[pos], readseq[ri], pos, ri])
                if op == "=" or (op == "M" and refseq[pos] == readseq[ri]):
                    if firstmatch is None:
                        firstmatch, first_al, first_pos = ri, len(rows) - 1, pos
                    if lastmatch is None or lastmatch < ri:
                        lastmatch = ri
                    last_al = len(rows) - 1
                    if last_pos is None or last_pos < pos:
                        last_pos = pos
                else:
                    nmis += 1
                pos += 1
                ri += 1
            elif op == "I":
                rows.append(["-", readseq[ri], pos, ri])
                ri += 1
                nins += 1
            elif op in "DN":
                rows.append([refseq[pos], "-", pos, ri])
                pos += 1
                ndel += op == "D"
            elif op == "S":
                ri += 1
    out = dict(strand=strand, pos_after_clip=pos_after_clip, events_after_clip=events_after_clip,
               num_insertions=nins, num_deletions=ndel, num_mismatches=nmis)
    if firstmatch is None:
        return dict(out, status="no match")
    n_ev = len(ev)
    if strand == "+":                                                        # :625-630
        left += firstmatch
        if n_ev - lastmatch > 1:
            right += n_ev - lastmatch - 1
    else:
        right += firstmatch
        if n_ev - lastmatch > 1:
            left += n_ev - lastmatch - 1
    if strand == "+":                                                        # :632-640
        if n_ev - lastmatch > 1:
            ev = ev[firstmatch:(lastmatch + 1 - n_ev)]
        elif firstmatch > 0:
            ev = ev[firstmatch:]
    else:
        if firstmatch > 0:
            ev = ev[(n_ev - 1 - lastmatch):-firstmatch]
        elif n_ev - lastmatch > 1:
            ev = ev[(n_ev - 1 - lastmatch):]
    if firstmatch > 0 or len(rows) - last_al > 1:                            # :642-659
        if len(rows) - last_al > 1:
            rows = rows[first_al:(last_al + 1 - len(rows))]
        elif first_al > 0:
            rows = rows[first_al:]
    if strand == "-":                                                        # :661-667
        rows = [[_COMP.get(r[0], r[0]), _COMP.get(r[1], r[1]), r[2], r[3]] for r in rows[::-1]]
        left, right = right, left
    n = len(rows)
    for a in range(n):                                                       # :684-704
        if rows[a][0] == "C" and rows[a][1] == "C":
            if a + 1 < n and rows[a + 1][1] == "-" and rows[a + 1][0] == "G":
                add = 2
                while a + add < n and rows[a + add][1] == "-" and rows[a + add][0] == "G":
                    add += 1
                if a + add < n and rows[a + add][1] == "G" and rows[a + add][0] == "G":
                    rows[a + 1][1], rows[a + add][1] = rows[a + add][1], rows[a + 1][1]
        if rows[a][0] == "G" and rows[a][1] == "G":
            if a - 1 > -1 and rows[a - 1][1] == "-" and rows[a - 1][0] == "C":
                add = 2
                while a - add > -1 and rows[a - add][1] == "-" and rows[a - add][0] == "C":
                    add += 1
                if a - add > -1 and rows[a - add][1] == "C" and rows[a - add][0] == "C":
                    rows[a - 1][1], rows[a - add][1] = rows[a - add][1], rows[a - 1][1]
    return dict(out, status="ok", rows=rows, leftclip=left, rightclip=right,
                ev_lo=ev[0] if ev else 0, ev_hi=(ev[-1] + 1) if ev else 0, n_ev=len(ev),
                first_match_pos=first_pos, last_match_pos=last_pos)
