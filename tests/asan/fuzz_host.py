"""TEST INFRASTRUCTURE (run by tests/test_asan_host.py in a subprocess, with libasan preloaded): the host-only part of the C ABI
built with -fsanitize=address,undefined (tests/asan/host_shim.cpp) driven with valid batches (compiled path == Python restatement, as
tests/test_stream_feeders.py does against the product library) and with DAMAGED container tables: every call must come back with
a result or an error code - a sanitizer report aborts this process.

    LD_PRELOAD=$(gcc -print-file-name=libasan.so) python tests/asan/fuzz_host.py <shim.so> <scratch dir> [iterations]
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from deepmod_amd import _lib, predstore, stream, synth_reads                      # noqa: E402


def load_shim(path):
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    have = 0
    for name, restype, argtypes in _lib.SIGNATURES:
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
            have += 1
    assert have >= 11, have
    _lib._LIB = lib              # everything in this process that asks for the library gets the sanitized host part
    return lib


def p(a):
    return a.ctypes.data


def drain(lib, h, n_contigs, compact):
    """dm_rows_info + dm_rows_emit into exactly sized buffers (the sanitizer guards their ends)."""
    R, T, S, nm = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    n = lib.dm_rows_info(h, ctypes.byref(R), ctypes.byref(T), ctypes.byref(S), None, None, 0, ctypes.byref(nm))
    if n < 0:
        return n
    info = np.zeros((max(n, 1), 8), np.int64)
    mism = np.zeros((3, 4), np.int64)
    lib.dm_rows_info(h, ctypes.byref(R), ctypes.byref(T), ctypes.byref(S), p(info), p(mism), 3, ctypes.byref(nm))
    R, T, S = R.value, T.value, S.value
    n_pf = (S + (T - R)) if compact else T
    rows = np.empty((R, 7), np.float32)
    pos, flags, sel = np.empty(n_pf, np.int64), np.empty(n_pf, np.uint8), np.empty(max(S, 1), np.int32)
    rank = np.arange(max(n_contigs, 1), dtype=np.int32)
    clen = np.zeros(max(n_contigs, 1), np.int64)
    groups = np.zeros((2 * max(n_contigs, 1), 8), np.int64)
    in_range = ctypes.c_int32(1)
    ng = lib.dm_rows_emit(h, p(rank), p(rows) if R else None, p(sel) if compact else None, p(pos) if n_pf else None, p(flags) if n_pf else None,
                          p(groups), len(groups), p(clen), len(clen), ctypes.byref(in_range))
    if ng >= 0 and R:
        assert np.isfinite(rows).all() or not in_range.value
    return ng


def drain_resident(lib, h, n_contigs, n_events):
    """dm_rows_info + dm_rows_device_info + dm_rows_emit_resident (round 6: no per-event values, descriptors that index the signal stage's device block) into
    exactly sized buffers; every descriptor must stay inside the batch's merged event tables."""
    R, T, S, nm = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    n = lib.dm_rows_info(h, ctypes.byref(R), ctypes.byref(T), ctypes.byref(S), None, None, 0, ctypes.byref(nm))
    if n < 0:
        return n
    R, T, S = R.value, T.value, S.value
    ne, nr = ctypes.c_int64(), ctypes.c_int64()
    ok = lib.dm_rows_device_info(h, ctypes.byref(ne), ctypes.byref(nr))
    if ok != 1:
        return 0                                   # no emitted read: nothing to hand over
    NR = nr.value
    n_pf = S + (T - R)
    code, rdesc = np.empty(R, np.uint8), np.empty((NR, 4), np.int64)
    pos, flags, sel = np.empty(n_pf, np.int64), np.empty(n_pf, np.uint8), np.empty(max(S, 1), np.int32)
    rank = np.arange(max(n_contigs, 1), dtype=np.int32)
    clen = np.zeros(max(n_contigs, 1), np.int64)
    groups = np.zeros((2 * max(n_contigs, 1), 8), np.int64)
    in_range = ctypes.c_int32(1)
    ng = lib.dm_rows_emit_resident(h, p(rank), p(code), p(rdesc), p(sel), p(pos) if n_pf else None, p(flags) if n_pf else None, p(groups), len(groups),
                                   p(clen), len(clen), ctypes.byref(in_range))
    if ng >= 0:
        assert (np.diff(rdesc[:, 0]) > 0).all() and rdesc[0, 0] == 0
        assert (rdesc[:, 2] >= 0).all() and (rdesc[:, 3] >= rdesc[:, 2]).all() and (rdesc[:, 3] <= n_events).all(), (rdesc, n_events)
        assert ((code <= 3) | (code == 255)).all() and (sel[:S] >= 0).all() and (sel[:S] < R).all()
        # the host-statistics emit of a resident batch is refused, not misread
        assert lib.dm_rows_emit(h, p(rank), p(np.empty((R, 7), np.float32)), p(sel), p(pos) if n_pf else None, p(flags) if n_pf else None, p(groups), len(groups),
                                p(clen), len(clen), ctypes.byref(in_range)) < 0
    return ng


def fuzz_packed(lib, files, rng, iters):
    counts = {'refused': 0, 'accepted': 0}
    pk = predstore.load_packed(files[0])
    meta = pk['reads']
    n = len(meta)
    good = dict(row_off=np.array(pk['row_off'], np.int64), bmi_off=np.array(pk['bmi_off'], np.int64), ev_off=np.array(pk['ev_off'], np.int64),
                tx=np.array(pk['tx'], np.float32), refbase=np.array(pk['refbase'], 'S1'), readbase=np.array(pk['readbase'], 'S1'),
                refbasei=np.array(pk['refbasei'], np.int64), evbase=np.array(pk['evbase'], 'S1'),
                start_clip=np.array([m['start_clip'] for m in meta], np.int64), end_clip=np.array([m['end_clip'] for m in meta], np.int64),
                contig=np.zeros(n, np.int32), strand=np.array([0 if m['strand'] == '+' else 1 for m in meta], np.int32))
    order = ('row_off', 'bmi_off', 'ev_off', 'tx', 'refbase', 'readbase', 'refbasei', 'evbase', 'start_clip', 'end_clip', 'contig', 'strand')
    for it in range(iters):
        a = {k: v.copy() for k, v in good.items()}
        n_contigs = 1
        for _ in range(int(rng.integers(0, 4))):           # 0 (the undamaged container) to 3 kinds of damage at once
            kind = int(rng.integers(0, 9))
            if kind == 0:
                k = ('row_off', 'bmi_off', 'ev_off')[int(rng.integers(0, 3))]
                a[k][int(rng.integers(0, n + 1))] = int(rng.integers(-50, int(good[k][-1]) + 5000))
            elif kind == 1:
                k = ('row_off', 'bmi_off', 'ev_off')[int(rng.integers(0, 3))]
                a[k] = a[k] + int(rng.integers(-3, 2000))
            elif kind == 2:                                  # a truncated column
                k = ('tx', 'refbase', 'readbase', 'refbasei', 'evbase')[int(rng.integers(0, 5))]
                a[k] = a[k][:int(rng.integers(0, len(a[k])))].copy()
            elif kind == 3:
                k = ('start_clip', 'end_clip')[int(rng.integers(0, 2))]
                a[k][int(rng.integers(0, n))] = int(rng.choice([-1, -10 ** 9, 10 ** 9, 2 ** 62, 0, 7, 2 ** 63 - 1, -2 ** 63]))
                if rng.integers(0, 4) == 0:                  # both clips at the ends of int64: n_events - start - end wraps into a small number
                    i = int(rng.integers(0, n))
                    a['start_clip'][i] = a['end_clip'][i] = int(rng.choice([2 ** 63 - 1, -2 ** 63, 2 ** 62]))
            elif kind == 4:
                a['contig'][int(rng.integers(0, n))] = int(rng.choice([-1, 1, 5, 2 ** 30]))
            elif kind == 5:
                a['strand'][int(rng.integers(0, n))] = int(rng.choice([-1, 2, 77]))
            elif kind == 6:
                a['readbase'][rng.integers(0, len(a['readbase']), 50)] = b'-'
            elif kind == 7:
                a['refbasei'][rng.integers(0, len(a['refbasei']), 20)] = rng.choice([-5, 2 ** 62, 0], 20)
            else:
                a['tx'][rng.integers(0, len(a['tx']), 5)] = rng.choice([np.nan, np.inf, 1e30], (5, 1))
        h = lib.dm_rows_create(b'C')
        n_tab = min(len(a['refbase']), len(a['readbase']), len(a['refbasei']))
        rc = lib.dm_rows_add_packed(h, n, len(a['tx']), n_tab, len(a['evbase']), n_contigs, *[p(a[k]) for k in order])
        if rc == 0:
            ng = drain(lib, h, n_contigs, compact=bool(it & 1))
            assert ng >= 0, _lib.last_error()
            counts['accepted'] += 1
        else:
            counts['refused'] += 1
        lib.dm_rows_destroy(h)
    return counts


def fuzz_map_read(lib, rng, iters):
    ref = bytes(rng.choice(list(b'ACGT'), 3000).astype(np.uint8))
    ops = 'MIDNSHPX=Z'
    counts = {'ok': 0, 'error': 0}
    for _ in range(iters):
        nops = int(rng.integers(1, 9))
        cig = ''.join('%d%s' % (int(rng.choice([0, 1, 3, 17, 120, 900, 10 ** 6, 10 ** 11, 10 ** 15])) if rng.random() < 0.3 else int(rng.integers(1, 200)),
                                ops[int(rng.integers(0, len(ops)))]) for _ in range(nops))
        if rng.random() < 0.1:
            cig = cig.replace('M', '', 1) + str(int(rng.integers(0, 99)))
        rlen = int(rng.integers(0, 900))
        seq = bytes(rng.choice(list(b'ACGT'), rlen).astype(np.uint8))
        cap = int(rng.choice([0, 5, 100, 5000]))
        rb, qb = np.empty(cap, 'S1'), np.empty(cap, 'S1')
        ri, qi = np.empty(cap, np.uint64), np.empty(cap, np.uint64)
        info = np.zeros(32, np.int64)
        rc = lib.dm_map_read(int(rng.choice([0, 16, 4, 2048])), int(rng.choice([1, 0, -5, 2500, 2990, 3001, 10 ** 12, int(rng.integers(1, 2900))])),
                             cig.encode(), ctypes.c_char_p(seq), rlen, ctypes.c_char_p(ref), len(ref), int(rng.choice([0, 10, rlen, 5000])),
                             p(rb) if cap else None, p(qb) if cap else None, p(ri) if cap else None, p(qi) if cap else None, cap, p(info))
        counts['ok' if rc == 0 else 'error'] += 1
    return counts


def fuzz_events_and_raw(lib, rng, iters):
    counts = {'merge_refused': 0, 'merge_ok': 0, 'raw_refused': 0, 'raw_ok': 0, 'resident_refused': 0, 'resident_ok': 0}
    ref = bytes(rng.choice(list(b'ACGT'), 4000).astype(np.uint8))
    for _ in range(iters):
        n = int(rng.integers(1, 5))
        per = rng.integers(60, 400, n)
        ne = int(per.sum())
        ev_off = np.concatenate([[0], np.cumsum(per)]).astype(np.int64)
        mean, stdv = rng.normal(0, 1, ne), np.abs(rng.normal(0.3, 0.1, ne))
        length = rng.integers(1, 20, ne).astype(np.uint64)
        start = np.cumsum(length).astype(np.uint64)
        bases = rng.choice(list('ACGT'), ne)
        ms = np.array(['AA%sAA' % b for b in bases], 'U5')
        move = rng.integers(0, 3, ne).astype(np.int64)
        damaged = rng.random() < 0.5
        n_events = ne
        if damaged:
            kind = int(rng.integers(0, 3))
            if kind == 0:
                ev_off[int(rng.integers(0, n + 1))] = int(rng.integers(-20, ne + 500))
            elif kind == 1:
                n_events = int(rng.integers(0, ne))
                mean, stdv, length, start, ms, move = mean[:n_events].copy(), stdv[:n_events].copy(), length[:n_events].copy(), start[:n_events].copy(), ms[:n_events].copy(), move[:n_events].copy()
            else:
                ev_off = ev_off[::-1].copy()
        cap_out = max(int(ev_off[-1]), 0)
        mev_off = np.empty(n + 1, np.int64)
        m_mean, m_stdv = np.empty(cap_out, np.float32), np.empty(cap_out, np.float32)
        m_start, m_len, m_base = np.empty(cap_out, np.uint64), np.empty(cap_out, np.uint64), np.empty(cap_out, 'S1')
        got = lib.dm_events_merge(n, n_events, p(ev_off), p(mean), p(stdv), p(start), p(length), p(ms), 5, p(move), p(mev_off), p(m_mean), p(m_stdv),
                                  p(m_start), p(m_len), p(m_base))
        if got < 0:
            counts['merge_refused'] += 1
            continue
        counts['merge_ok'] += 1
        # ---- the merged tables into dm_rows_add_raw, with good or damaged alignment records
        nrec = n
        flag = rng.choice([0, 16], nrec).astype(np.int32)
        seqs, cigs = [], []
        for r in range(nrec):
            k = int(mev_off[r + 1] - mev_off[r])
            seqs.append(bytes(m_base[mev_off[r]:mev_off[r + 1]].view(np.uint8)))
            cigs.append(('%dM' % k) if rng.random() < 0.6 else '%dS%dM%dI%dM%dD%dM' % tuple(int(v) for v in rng.integers(0, 90, 6)))
            if rng.random() < 0.15:
                cigs[-1] = cigs[-1] + str(10 ** int(rng.integers(3, 17))) + 'M'
        pos1 = rng.integers(1, 3000, nrec).astype(np.int64)
        rlen = np.array([len(s) for s in seqs], np.int64)
        cidx = np.zeros(nrec, np.int32)
        ev_read = np.arange(nrec, dtype=np.int32)
        skip = np.zeros(nrec, np.uint8)
        mo = mev_off.copy()
        n_ev_total = int(got)
        fe = np.full(nrec, -1, np.int64)
        s_mean, s_stdv = m_mean[:got].copy(), m_stdv[:got].copy()
        kind = int(rng.integers(0, 8))
        if kind == 0:
            ev_read[int(rng.integers(0, nrec))] = int(rng.choice([-1, nrec, 2 ** 30]))
        elif kind == 1:
            mo[int(rng.integers(0, nrec + 1))] = int(rng.integers(-30, got + 800))
        elif kind == 2:
            n_ev_total = int(rng.integers(0, max(got, 1)))
        elif kind == 3:
            cidx[int(rng.integers(0, nrec))] = int(rng.choice([-1, 1, 9]))
        elif kind == 4:
            fe[:] = rng.choice([-7, 0, 3, 499, 501, 10 ** 9], nrec)
        elif kind == 5:
            pos1[int(rng.integers(0, nrec))] = int(rng.choice([0, -9, 3990, 4001, 10 ** 14]))
        ref_ptr = (ctypes.c_char_p * 1)(ref)
        ref_len = np.array([len(ref)], np.int64)
        cig_ptr = (ctypes.c_char_p * nrec)(*[c.encode() for c in cigs])
        seq_ptr = (ctypes.c_char_p * nrec)(*seqs)
        rg_c, rg_lo, rg_hi = np.array([0, -1], np.int32), np.array([100, -1], np.int64), np.array([3500, 50], np.int64)
        h = lib.dm_rows_create(b'C')
        rc = lib.dm_rows_add_raw(h, nrec, p(flag), p(pos1), cig_ptr, seq_ptr, p(rlen), p(cidx), p(ev_read), p(skip), 1, ref_ptr, p(ref_len), nrec, n_ev_total,
                                 p(mo), p(m_mean), p(m_stdv), p(m_len), p(m_base), p(s_mean), p(s_stdv), p(fe), int(rng.integers(0, 3)), p(rg_c), p(rg_lo), p(rg_hi))
        if rc == 0:
            assert drain(lib, h, 1, compact=bool(kind & 1)) >= 0, _lib.last_error()
            counts['raw_ok'] += 1
        else:
            counts['raw_refused'] += 1
        lib.dm_rows_destroy(h)
        # the same records in the RESIDENT form (statistics on the device: no s_mean / s_stdv; the basecaller's values optional - without them a read
        # that needs them is refused)
        if fe.min() < 0:
            fe = np.where(fe < 0, 10 ** 9, fe)
        with_fb = bool(rng.integers(0, 2))
        h = lib.dm_rows_create(b'C')
        rc = lib.dm_rows_add_raw(h, nrec, p(flag), p(pos1), cig_ptr, seq_ptr, p(rlen), p(cidx), p(ev_read), p(skip), 1, ref_ptr, p(ref_len), nrec, n_ev_total,
                                 p(mo), p(m_mean) if with_fb else None, p(m_stdv) if with_fb else None, p(m_len), p(m_base), None, None, p(fe),
                                 int(rng.integers(0, 3)), p(rg_c), p(rg_lo), p(rg_hi))
        if rc == 0:
            assert drain_resident(lib, h, 1, n_ev_total) >= 0, _lib.last_error()
            counts['resident_ok'] += 1
        else:
            counts['resident_refused'] += 1
        lib.dm_rows_destroy(h)
    return counts


def fuzz_mapped(lib, rng, iters):
    """dm_rows_add_mapped: the caller's own alignment tables and event tables, good and damaged."""
    counts = {'refused': 0, 'accepted': 0}
    for it in range(iters):
        n = int(rng.integers(1, 5))
        ntab = rng.integers(80, 500, n)
        bmi_off = np.concatenate([[0], np.cumsum(ntab)]).astype(np.int64)
        T = int(bmi_off[-1])
        refb = rng.choice(list(b'ACGT-N'), T).astype(np.uint8).view('S1')
        readb = rng.choice(list(b'ACGT-'), T, p=[.24, .24, .24, .24, .04]).astype(np.uint8).view('S1')
        refi = np.sort(rng.integers(0, 100000, T)).astype(np.int64)
        nev = np.array([int((readb[bmi_off[r]:bmi_off[r + 1]] != b'-').sum()) + int(rng.integers(0, 30)) for r in range(n)])
        mev_off = np.concatenate([[0], np.cumsum(nev)]).astype(np.int64)
        E = int(mev_off[-1])
        m_mean, m_stdv = rng.normal(0, 1, E).astype(np.float32), np.abs(rng.normal(0.3, 0.1, E)).astype(np.float32)
        m_len = rng.integers(1, 30, E).astype(np.uint64)
        m_base = rng.choice(list(b'ACGT'), E).astype(np.uint8).view('S1')
        sc, ec = rng.integers(0, 10, n).astype(np.int64), rng.integers(0, 10, n).astype(np.int64)
        contig, strand = np.zeros(n, np.int32), rng.integers(0, 2, n).astype(np.int32)
        fe = np.full(n, -1, np.int64)
        n_tab, n_ev, n_contigs = T, E, 1
        kind = int(rng.integers(0, 9))
        if kind == 0:
            bmi_off[int(rng.integers(0, n + 1))] = int(rng.integers(-20, T + 900))
        elif kind == 1:
            mev_off[int(rng.integers(0, n + 1))] = int(rng.integers(-20, E + 900))
        elif kind == 2:
            n_tab = int(rng.integers(0, T))
            refb, readb, refi = refb[:n_tab].copy(), readb[:n_tab].copy(), refi[:n_tab].copy()
        elif kind == 3:
            n_ev = int(rng.integers(0, E))
            m_mean, m_stdv, m_len, m_base = m_mean[:n_ev].copy(), m_stdv[:n_ev].copy(), m_len[:n_ev].copy(), m_base[:n_ev].copy()
        elif kind == 4:
            i = int(rng.integers(0, n))
            sc[i] = int(rng.choice([-3, 10 ** 12, 2 ** 62, 2 ** 63 - 1, -2 ** 63]))
            if rng.integers(0, 2) == 0:
                ec[i] = sc[i]                                # INT64_MAX twice: the subtraction wraps to n_events + 2
        elif kind == 5:
            contig[int(rng.integers(0, n))] = int(rng.choice([-1, 1, 2 ** 30]))
        elif kind == 6:
            strand[int(rng.integers(0, n))] = int(rng.choice([-1, 2]))
        elif kind == 7:
            fe[:] = rng.choice([-7, 0, 3, 499, 501, 10 ** 9], n)
        h = lib.dm_rows_create(b'C')
        with_stats = bool(it & 1)
        rc = lib.dm_rows_add_mapped(h, n, n_tab, n_ev, n_contigs, p(bmi_off), p(refb), p(readb), p(refi), p(sc), p(ec), p(contig), p(strand), p(mev_off), p(m_mean),
                                    p(m_stdv), p(m_len), p(m_base), p(m_mean) if with_stats else None, p(m_stdv) if with_stats else None, p(fe) if with_stats else None)
        if rc == 0:
            assert drain(lib, h, n_contigs, compact=bool(it & 2)) >= 0, _lib.last_error()
            counts['accepted'] += 1
        else:
            counts['refused'] += 1
        lib.dm_rows_destroy(h)
    return counts


def fuzz_bed(lib, rng, iters):
    for _ in range(iters):
        n = int(rng.integers(0, 400))
        touch = (rng.random(n) < 0.3).astype(np.int32) * rng.integers(1, 5, n).astype(np.int32)
        cov = rng.integers(0, 3000, n).astype(np.int32)
        mod = np.minimum(cov, rng.integers(0, 3000, n)).astype(np.int32)
        first = int(rng.choice([0, 7, 10 ** 9, 2 ** 40]))
        need = lib.dm_bed_format_at(b'chrWithAVeryLongName_12345', b'+', b'C', first, p(touch), p(cov), p(mod), n, None, 0)
        assert need >= 0
        for cap in (0, 1, max(need - 1, 0), need):
            buf = np.zeros(max(cap, 1), np.uint8)            # (exactly sized: the sanitizer guards its end)
            got = lib.dm_bed_format_at(b'chrWithAVeryLongName_12345', b'+', b'C', first, p(touch), p(cov), p(mod), n, p(buf) if cap else None, cap)
            assert (got == need) if cap < need else (0 <= got <= need), (got, need, cap)      # the safe bound, or the bytes written


def valid_batches(lib, scratch):
    """The equalities of tests/test_stream_feeders.py, through the sanitized library."""
    import test_stream_feeders as T
    files = synth_reads.write_synthetic_run(os.path.join(scratch, 'pk'), n_reads=12, reads_per_file=4, genome_len=20000, seed=5, chrom='chrS', min_len=300, max_len=1500)
    mo = {'Base': 'C', 'outFolder': scratch, 'fnum': 7, 'hidden': 100, 'windowsize': 21, 'select_base': False}
    T._same_batch(stream._prepare_batch_c(dict(mo), files), stream._prepare_batch_py(dict(mo), files))
    stream._prepare_batch_c(dict(mo, select_base=True), files)
    raw, fasta = synth_reads.write_synthetic_raw_run(os.path.join(scratch, 'raw'), n_reads=12, reads_per_file=4, genome_len=30000, seed=4, chrom='chrS',
                                                     min_len=300, max_len=1200)[:2]
    mo = dict(mo, Ref=fasta, alignStr='minimap2', region=[[None, None, None]], ConUnk=True, SignalGroup='simple', outLevel=3)
    norm = T._OracleNormalizer()
    T._same_batch(stream._prepare_batch_c(dict(mo), raw, lambda: norm), stream._prepare_batch_py(dict(mo), raw, lambda: norm))
    stream._prepare_batch_c(dict(mo, select_base=True, region=[['chrS', None, 15000]]), raw, lambda: norm)
    return files


if __name__ == '__main__':
    shim, scratch = sys.argv[1], sys.argv[2]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    lib = load_shim(shim)
    rng = np.random.default_rng(20260928)
    files = valid_batches(lib, scratch)
    print('valid batches: compiled == Python through the sanitized library')
    print('damaged feature containers:', fuzz_packed(lib, files, rng, iters))
    print('alignment walk:', fuzz_map_read(lib, rng, 4 * iters))
    print('event tables and alignment records:', fuzz_events_and_raw(lib, rng, iters))
    print('caller-supplied alignment tables:', fuzz_mapped(lib, rng, iters))
    fuzz_bed(lib, rng, iters // 3)
    print('BED formatter: ok')
    print('FUZZ-OK')
