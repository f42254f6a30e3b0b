"""Host-side counterparts of the reference's detect path (bin/DeepMod_scripts/myDetect.py), same
function names, argument meaning and side effects, driving the HIP library instead of TensorFlow.

    mPredict1        myDetect.py:787-834   windows + batched inference + class -> base scatter
    detect_handler   myDetect.py:948-984   worker: one model per process, pulls batches from a queue
    sum_handler      myDetect.py:1028-1120 per (chr, strand) coverage / mod-count summary -> BED
    mDetect_manager  myDetect.py:1124-1263 sharding, worker processes, index merge, .done marker

Two run modes behind `mDetect_manager` (deepmod_amd/stream.py has the first):
  * streaming (default): one process per GPU keeps the per-position counters on the device, classifier output goes
    straight into them, one RCCL reduce per contig x strand at the end, rank 0 writes the BED files; no per-read files;
  * stored (`--storePred 1`, and the `--predDet 0` resume): the reference's file shape - per-read prediction tables and
    per-chromosome index files written by the detect workers, read back by the summary workers.

Inputs are containers instead of .fast5 files (no h5py / libhdf5 in this image): *raw containers*
(deepmod_amd/rawreads.py: DAC samples + basecaller events; normalised on the GPU, aligned with the SAM
records, mapped by dm_map_read) or *feature containers* (deepmod_amd/predstore.py: per-read `mfeatures`,
`base_map_info`, clips, event bases - exactly the arguments the reference hands to mPredict1 at
myDetect.py:715).
"""
from __future__ import annotations

import glob
import multiprocessing
import multiprocessing.connection
import os
import time
from collections import defaultdict

import numpy as np

from . import _lib, predstore, rawreads, readmap

rnn_pred_batch_size = 512   # myDetect.py:30
pre_base_str = 'rnn.pred.ind'  # myDetect.py:33-ish: index-file stem used by the manager and workers

OUTPUT_DEBUG, OUTPUT_INFO, OUTPUT_WARNING, OUTPUT_ERROR = 0, 1, 2, 3  # myCom.py:5-8


def _windows_view(tx: np.ndarray, first: int, count: int, half: int) -> np.ndarray:
    """[count, 2*half+1, nfeat] strided view: window i = tx[first+i-half : first+i+half+1]."""
    from numpy.lib.stride_tricks import sliding_window_view
    v = sliding_window_view(tx, (2 * half + 1, tx.shape[1]))[:, 0]
    return v[first - half:first - half + count]


def mPredict1(moptions, sp_options, sp_param, mfeatures, base_map_info, readk, start_clip, end_clip):
    """Same contract as the reference: returns pred_mod_num and sets base_map_info['mod_pred']=1 in
    place for every aligned read base whose window is classified 1.

    sp_options['rnn'] = (sess, X, Y, init_l, mfpred).  With this build's Session the per-read
    feature rows are shipped once and windows are assembled on the GPU (dm_predict_read); with any
    other session-like object the reference's exact call sequence is reproduced (float64 windows,
    int label matrix, sess.run(init_l), the ~512 batch split of myDetect.py:808-812)."""
    modevents = sp_param['f5data'][readk][1]
    half = int(moptions['windowsize'] / 2)
    n = len(modevents) - end_clip - start_clip
    if n <= 0:
        return 0
    tx = mfeatures[:, 3:]            # np.split(mfeatures, [1,3], axis=1)[2]   (myDetect.py:791)
    first = 100                      # mind of ie == start_clip (myDetect.py:795)
    sess, X, Y, init_l, mfpred = sp_options['rnn']

    if hasattr(sess, 'model') and getattr(sess, 'model') is not None and hasattr(sess.model, 'predict_read'):
        sess.run(init_l)
        _, cls = sess.model.predict_read(np.ascontiguousarray(tx, dtype=np.float32), first, n, want_prob=False)
        mfpred_output = cls
    else:
        test_feature = np.array(_windows_view(tx, first, n, half))                  # float64 [n,21,7]
        test_label = np.zeros((n, 2), dtype=int)                                    # labels are all zero (:878)
        sess.run(init_l)
        if len(test_feature) > rnn_pred_batch_size * 1.2:
            k = int(len(test_feature) / rnn_pred_batch_size)
            x_sub_group = np.array_split(test_feature, k)
            y_sub_group = np.array_split(test_label, k)
        else:
            x_sub_group = [test_feature]
            y_sub_group = [test_label]
        outs = [sess.run([mfpred], feed_dict={X: xs, Y: ys})[0] for xs, ys in zip(x_sub_group, y_sub_group)]
        mfpred_output = np.concatenate(outs, axis=0)

    return scatter_predictions(modevents, base_map_info, start_clip, n, mfpred_output)


def scatter_predictions(modevents, base_map_info, start_clip, n, mfpred_output):
    """The class -> aligned-read-base association of myDetect.py:824-833 (vectorised): the k-th
    aligned event is the k-th row whose readbase is not '-'.  Returns pred_mod_num."""
    aligned = np.flatnonzero(base_map_info['readbase'] != '-')[:n]
    if len(aligned) < n:
        raise IndexError('base_map_info has %d aligned read bases but %d events are aligned' % (len(aligned), n))
    ev_base = rawreads.event_bases(modevents['model_state'][start_clip:start_clip + n])
    bad = np.flatnonzero(base_map_info['readbase'][aligned] != ev_base)
    for b in bad:
        print('Error Does not match', base_map_info['readbase'][aligned[b]], ev_base[b], aligned[b], b + start_clip)
    hit = aligned[np.asarray(mfpred_output) == 1]
    base_map_info['mod_pred'][hit] = 1
    return int(len(hit))


def predict_rows_any_range(model, rows, first, count):
    """dm_predict_read -> classes.  The split-f16 kernel refuses inputs outside its range (DM_ERANGE: an event length beyond
    65504 * 2^k samples, a feature beyond +-65504, NaN) instead of clamping them; the reference computes such a read in fp32
    like any other, so the call is repeated with the fp32 kernel."""
    try:
        return model.predict_read(rows, first, count, want_prob=False)[1]
    except _lib.DeepModRangeError:
        keep = model.get_info(_lib.DM_INFO_PRECISION)
        model.set_option(_lib.DM_OPT_PRECISION, _lib.DM_PREC_F32)
        try:
            return model.predict_read(rows, first, count, want_prob=False)[1]
        finally:
            model.set_option(_lib.DM_OPT_PRECISION, keep)


def mPredict_batch(moptions, sp_options, reads):
    """Classify several reads with ONE device call.  A read of n aligned bases is only n/128 tiles and
    every tile has a fixed latency, so per-read calls (the reference's granularity, mPredict1) leave most
    of the GPU idle; the reads' feature matrices already carry 100 all-zero rows on both sides
    (myDetect.py:850-851), so they are concatenated and every row in between is classified in one
    dm_predict_read; windows centred on padding rows are computed and dropped (~200/(n+200) of the work).
    Results are identical to calling mPredict1 read by read (windows are independent).
    reads: dicts with mfeatures, base_map_info, events, start_clip, end_clip.  Returns [pred_mod_num]."""
    sess, X, Y, init_l, mfpred = sp_options['rnn']
    half = int(moptions['windowsize'] / 2)
    mats, spans, off = [], [], 0
    for rd in reads:
        tx = np.ascontiguousarray(rd['mfeatures'][:, 3:], dtype=np.float32)
        n = len(rd['events']) - rd['end_clip'] - rd['start_clip']
        mats.append(tx)
        spans.append((off + 100, n))                  # first window centre (mind = 100) and count
        off += len(tx)
    rows = np.concatenate(mats)
    sess.run(init_l)
    cls = predict_rows_any_range(sess.model, rows, half, len(rows) - 2 * half)
    out = []
    for rd, (first, n) in zip(reads, spans):
        if n <= 0:
            out.append(0)
            continue
        out.append(scatter_predictions(rd['events'], rd['base_map_info'], rd['start_clip'], n,
                                       cls[first - half:first - half + n]))
    return out


# ---------------------------------------------------------------------------------------------
# worker
# ---------------------------------------------------------------------------------------------
_PREDICT_ERRORS = (IndexError, ValueError, KeyError)     # malformed read tables; anything else is a bug and propagates


def _predict_reads(moptions, sp_options, good, src):
    """-> [pred_mod_num or None (failed)] for a group of reads.  One device call for the whole group (mPredict_batch);
    if a read's tables are malformed the group is redone read by read so that only the offending read is skipped and
    recorded - the reference's granularity (mPredict1 per read inside handle_record's try, myDetect.py:715)."""
    sess = sp_options['rnn'][0]
    per_read = lambda rd: mPredict1(moptions, sp_options, {'f5data': {rd['readk']: (None, rd['events'], None, src)}},
                                    rd['mfeatures'], rd['base_map_info'], rd['readk'], rd['start_clip'], rd['end_clip'])
    if hasattr(sess, 'model') and getattr(sess, 'model') is not None:
        try:
            return mPredict_batch(moptions, sp_options, good)
        except _PREDICT_ERRORS:
            for rd in good:
                rd['base_map_info']['mod_pred'] = 0      # a partial scatter of the failed batch call must not survive
    out = []
    for rd in good:
        try:
            out.append(per_read(rd))
        except _PREDICT_ERRORS as exc:
            sp_options["Error"]["Prediction failed: %s" % type(exc).__name__].append(rd.get('src', src))
            out.append(None)
    return out


def _predict_and_store(moptions, sp_options, store, good, src):
    """Prediction, prediction tables and index entries of a group of reads (myDetect.py:715-718)."""
    if not good:
        return
    for rd, pred_mod_num in zip(good, _predict_reads(moptions, sp_options, good, src)):
        if pred_mod_num is None:
            continue
        rsrc = rd.get('src', src)
        key = store.add(rd, rd['base_map_info'], pred_mod_num, rsrc, moptions)
        sp_options['Mod'].append([rd['chr'], rd['strand'], rd['mapped_start'], key,
                                  os.path.relpath(rsrc, moptions['wrkBase']), store.relpath(moptions)])


def _alignment_lines(moptions, sp_options, raw_files, f5data):
    """SAM lines for the reads of this batch.  With the aligner binary on PATH: the reference's own call
    (myDetect.py:396-425: FASTA of the event basecalls -> `bwa mem -x ont2d` / `minimap2 -ax map-ont`); otherwise
    the side-car `<container>.sam` files written next to the raw containers."""
    import shutil
    import subprocess
    import tempfile
    align_str = moptions.get('alignStr', 'bwa')
    if shutil.which(align_str) and moptions.get('Ref'):
        with tempfile.NamedTemporaryFile(suffix='.fa', mode='w') as temp_fa, tempfile.NamedTemporaryFile() as temp_sam:
            for f5k in sorted(f5data.keys()):
                temp_fa.write(''.join(['>', f5k, '\n', f5data[f5k][0], '\n']))
            temp_fa.flush()
            if align_str == 'bwa':
                cmd_opt = ['mem', '-x', 'ont2d', '-v', '1', '-t', '1', moptions['Ref'], temp_fa.name]
            else:
                cmd_opt = ['-ax', 'map-ont', moptions['Ref'], temp_fa.name]
            if subprocess.call([align_str] + cmd_opt, stdout=temp_sam) != 0:
                return None
            temp_sam.seek(0)
            return [str(ln, 'utf-8').strip() for ln in temp_sam.readlines()]
    lines = []
    for rf in raw_files:
        sam = rf[:-len(rawreads.RAW_SUFFIX)] + '.sam'
        if not os.path.isfile(sam):
            return None
        with open(sam) as fh:
            lines.extend(ln.rstrip('\n') for ln in fh)
    return lines


def mDetect1_raw(moptions, sp_options, store, raw_files):
    """The reference's mDetect1 for raw reads (myDetect.py:392-465): signal -> events (GPU normalisation and event
    statistics) -> alignment records -> base_map_info + features -> prediction."""
    f5data = rawreads.get_Event_Signals(moptions, sp_options, raw_files, sp_options.get('normalizer'))
    if not f5data:
        return
    align_info = _alignment_lines(moptions, sp_options, raw_files, f5data)
    if align_info is None:
        for f5k in sorted(f5data.keys()):
            sp_options["Error"]["Cannot running aligment"].append(f5data[f5k][3])
        return
    sp_param = defaultdict()
    sp_param['f5data'] = f5data
    sp_param['ref_info'] = defaultdict()
    f5align = readmap.parse_sam(moptions, sp_options, sp_param, align_info, f5data)
    sp_param['f5status'] = ""
    sp_param['line'] = ""
    reads = readmap.map_records(moptions, sp_options, sp_param, f5align, f5data)
    _predict_and_store(moptions, sp_options, store, reads, raw_files[0])


def mDetect1(moptions, sp_options, container_files):
    """Per-batch body of the worker (counterpart of myDetect.mDetect1 :392-465): raw containers (`.dmraw.npz`:
    DAC samples + basecaller events, aligned here) go through the whole path, feature containers (`.dmfeat.npz`:
    the arguments the reference hands to mPredict1) enter at the prediction step.  Per-read prediction tables and
    the per-chromosome index lines are written as the reference does."""
    store = predstore.PredWriter(sp_options['ctfolder'], sp_options['batchid'])
    raw_files = [cf for cf in container_files if cf.endswith(rawreads.RAW_SUFFIX)]
    if raw_files:
        mDetect1_raw(moptions, sp_options, store, raw_files)
    for cf in container_files:
        if cf.endswith(rawreads.RAW_SUFFIX):
            continue
        try:
            reads = predstore.load_feature_container(cf)
        except Exception:
            sp_options["Error"]["Cannot open container"].append(cf)
            continue
        good = []
        for rd in reads:
            if len(rd['events']) - rd['start_clip'] - rd['end_clip'] < 50:           # myDetect.py:702-705
                sp_options["Error"]["Less Event"].append(cf)
            else:
                good.append(rd)
        _predict_and_store(moptions, sp_options, store, good, cf)
    store.close()
    # index files <ctfolder>/<chr>.rnn.pred.ind.<batchid>   (myDetect.py:762-782)
    by_chr = defaultdict(list)
    for mfi in sorted(sp_options['Mod']):
        by_chr[mfi[0]].append(mfi)
    for cur_chr, rows in by_chr.items():
        with open(os.path.join(sp_options['ctfolder'], '%s.%s.%s' % (cur_chr, pre_base_str, sp_options['batchid'])), 'w') as fh:
            for mfi in rows:
                fh.write(' '.join([str(v) for v in mfi] + ['\n']))


class _StoreWorker:
    """One detect worker of the stored mode: a model and a signal normaliser on its GPU, processes work items
    (files, sub-folder id, batch id) into `<out>/<sub-folder>/rnn.pred.detail.npz.<batch>` + per-chromosome index files."""

    def __init__(self, moptions, device):
        from . import model as dm, signal as dmsignal
        handles = dm.mCreateSession(moptions['fnum'], moptions['hidden'], moptions['windowsize'], moptions)
        self.tokens = (handles[5], handles[6], handles[1], handles[11])          # X, Y, init_l, mfpred
        self.sess = dm.new_session(device)
        dm.import_meta_graph(moptions['modfile'][0] + '.meta').restore(
            self.sess, dm.latest_checkpoint(moptions['modfile'][1]) or moptions['modfile'][0])
        self.normalizer = dmsignal.SignalNormalizer(device)       # signal stage on the worker's own GPU
        self.moptions = moptions

    def process(self, files, subfolder, batchid):
        mo = self.moptions
        X, Y, init_l, mfpred = self.tokens
        folder = '%s%s/%d' % (mo['outFolder'], mo['FileID'], subfolder)
        os.makedirs(folder, exist_ok=True)
        sp_options = {'ctfolderid': subfolder, 'ctfolder': folder, 'batchid': batchid, 'Mod': [],
                      'rnn': (self.sess, X, Y, init_l, mfpred), 'Error': defaultdict(list), 'normalizer': self.normalizer}
        mDetect1(mo, sp_options, files)
        return sp_options['Error']

    def close(self):
        self.normalizer.close()
        self.sess.close()


def _take(work_q):
    """Next item of a shared work queue, None when it is drained."""
    try:
        return work_q.get(block=False)
    except Exception:
        return None


def detect_handler(moptions, h5files_Q, failed_Q, file_map_info_q, device=0):
    """Detect worker process of the stored mode, the reference's entry point (myDetect.py:948-984): one model per
    process, work items taken from `h5files_Q` until it is drained, (reason, files) pairs posted to `failed_Q`."""
    worker = _StoreWorker(moptions, device)
    verbose = moptions.get('outLevel', OUTPUT_WARNING) <= OUTPUT_INFO
    item = _take(h5files_Q)
    while item is not None:
        t0 = time.time()
        files, subfolder, batchid = item
        for reason, where in worker.process(files, subfolder, batchid).items():
            failed_Q.put((reason, where))
        if verbose:
            print("Cur Prediction consuming time %d for %d %d" % (time.time() - t0, subfolder, batchid))
        item = _take(h5files_Q)
    worker.close()


# ---------------------------------------------------------------------------------------------
# summary
# ---------------------------------------------------------------------------------------------
def read_file_list(cur_cif, cur_chr, cur_strand, sp_options):
    """Index file `rnn.pred.ind.<chr>` -> sp_options['handlingList'] (records of `cur_strand`) and the two base folders
    of its header (format written by merge_index_files / the reference, myDetect.py:1194-1221; reader :989-1008)."""
    header_keys = {'#base_folder_fast5': 'base_folder_fast5', '#base_folder_output': 'base_folder_output'}
    records = []
    with open(cur_cif) as fh:
        for fields in (ln.split() for ln in fh):
            if not fields:
                continue
            if fields[0].startswith('#'):
                folder = fields[1] if fields[1].endswith(('/', '\\')) else fields[1] + '/'
                if fields[0] in header_keys:
                    sp_options[header_keys[fields[0]]] = folder
                continue
            if fields[0] != cur_chr:
                print('Warning!!! The chr should be %s but %s is found.' % (cur_chr, fields[0]))
            if fields[1] == cur_strand:
                records.append(fields)
    sp_options['handlingList'] = records


def read_pred_detail(moptions, sp_options, f5info):
    """-> (m_pred struct array, mapped_chr, mapped_strand); counterpart of myDetect.py:1013-1023."""
    return predstore.read_pred(os.path.join(sp_options['base_folder_output'], f5info[5]), f5info[3])


def base_flags(m_pred, base: str) -> np.ndarray:
    """Per-row flag byte of the summary kernel from a prediction table (myDetect.py:1091-1100):
    bit0 refbase == Base (never '-', 'N', 'n'), bit1 readbase != '-', bit2 mod_pred == 1."""
    refb = m_pred['refbase']
    is_base = (refb == base) & ~np.isin(refb, ['-', 'N', 'n'])
    not_gap = m_pred['readbase'] != '-'
    is_mod = np.abs(m_pred['mod_pred'].astype(np.float64) - 1) < 0.1
    return (is_base.astype(np.uint8) | (not_gap.astype(np.uint8) << 1) | (is_mod.astype(np.uint8) << 2))


def summarize_tables(tables, base: str, device: int = 0, length=None, chunk_rows: int = 4_000_000):
    """Accumulate an iterable of prediction tables on the GPU -> (touch, cov, mod) int32 arrays.  Rows are shipped in
    chunks of ~chunk_rows (a human chromosome at 30x is ~1e10 table rows: never held on the host at once); without a
    known length the counters grow as the tables reach further."""
    from . import summary
    summ = None
    pend_p, pend_f, pending = [], [], 0

    def flush():
        nonlocal summ, pend_p, pend_f, pending
        if not pending:
            return
        p, f = np.concatenate(pend_p), np.concatenate(pend_f)
        need = int(p.max()) + 1
        if summ is None:
            summ = summary.PositionSummary(length if length else need, device)
        elif need > summ.length:
            if length:
                raise IndexError('position %d outside the %d positions of the contig' % (need - 1, length))
            summ.grow(int(need * 1.5))
        summ.add(p, f)
        pend_p, pend_f, pending = [], [], 0

    for m_pred in tables:
        if isinstance(m_pred, tuple):           # (positions, flags) of rows on the base of interest: stored_chunks
            p, f = m_pred
            if len(p) == 0:
                continue
            pend_p.append(p)
            pend_f.append(f)
            pending += len(p)
        else:
            if len(m_pred) == 0:
                continue
            fl = base_flags(m_pred, base)
            keep = (fl & 1) != 0
            if not keep.any():
                continue
            pend_p.append(m_pred['refbasei'][keep].astype(np.int64))
            pend_f.append(fl[keep])
            pending += int(keep.sum())
        if pending >= chunk_rows:
            flush()
    flush()
    if summ is None:
        z = np.zeros(max(int(length or 0), 0), np.int32)
        return z, z.copy(), z.copy()
    touch, cov, mod = summ.fetch()
    summ.close()
    if not length:      # trim the growth slack: positions past the last touched one carry nothing
        last = int(np.flatnonzero(touch).max()) + 1 if touch.any() else 0
        touch, cov, mod = touch[:last], cov[:last], mod[:last]
    return touch, cov, mod


def stored_chunks(moptions, sp_options, cur_chr, cur_strand, base, readers: int = 4):
    """The records of sp_options['handlingList'] as (positions int64, flags uint8) chunks of the rows on `base` - what
    summarize_tables makes of the tables read_pred_detail returns, without building them: one chunk per prediction store (format 2:
    five members inflated per batch, the flags of all its reads in one pass over bytes).  Stores are inflated by `readers` threads
    ahead of the consumer (zlib releases the interpreter lock).  A format-1 store goes read by read through read_pred_detail."""
    from concurrent.futures import ThreadPoolExecutor
    by_store = {}
    for hl in sp_options['handlingList']:
        by_store.setdefault(hl[5], []).append(hl)
    base_code = ord(base) if base not in ('-', 'N', 'n') else -1

    def one(rel):
        st = predstore.load_pred_store(os.path.join(sp_options['base_folder_output'], rel))
        if st['format'] == 1:
            return [read_pred_detail(moptions, sp_options, hl) for hl in by_store[rel]]
        spans = []
        for hl in by_store[rel]:
            at = st['attrs'][hl[3]]
            if not (at['mapped_chr'] == cur_chr and at['mapped_strand'] == cur_strand):
                print("ERRoR not the same chr (real=%s vs expect=%s) and strand (real=%s VS expect=%s)" %
                      (at['mapped_chr'], cur_chr, at['mapped_strand'], cur_strand))
            spans.append(predstore.pred_rows(st, hl[3]))
        if not spans:
            return (np.zeros(0, np.int64), np.zeros(0, np.uint8))
        rows = np.concatenate([np.arange(lo, hi, dtype=np.int64) for lo, hi in spans])
        refb = st['refbase'].view(np.uint8)[rows]
        on_base = (refb == base_code) if base_code >= 0 else np.zeros(len(rows), bool)
        rows = rows[on_base]
        fl = (np.uint8(1) | ((st['readbase'].view(np.uint8)[rows] != 45).astype(np.uint8) << 1)
              | ((st['mod_pred'][rows] == 1).astype(np.uint8) << 2))
        return (st['refbasei'][rows].astype(np.int64), fl)

    stores = list(by_store)
    ahead = 2 * max(1, readers)             # stores in flight: the readers never run further ahead of the consumer than this
    with ThreadPoolExecutor(max(1, readers)) as pool:
        for g in range(0, len(stores), ahead):
            for out in pool.map(one, stores[g:g + ahead]):
                if isinstance(out, list):
                    for m_pred, mapped_chrom, mapped_strand in out:
                        if not (mapped_chrom == cur_chr and mapped_strand == cur_strand):
                            print("ERRoR not the same chr (real=%s vs expect=%s) and strand (real=%s VS expect=%s)" %
                                  (mapped_chrom, cur_chr, mapped_strand, cur_strand))
                        yield m_pred
                else:
                    yield out


def sum_handler(moptions, chr_strand_Q, device=0):
    """Per (chr, strand) summary worker, same queue protocol and output file as the reference
    (myDetect.py:1028-1120); the per-base accumulation runs on the GPU."""
    from . import summary
    while not chr_strand_Q.empty():
        try:
            cur_cif, cur_chr, cur_strand = chr_strand_Q.get(block=False)
        except Exception:
            break
        if moptions.get('mod_cluster'):
            raise NotImplementedError("mod_cluster is marked 'should not used now' in the reference and is not built")
        sp_options = {}
        read_file_list(cur_cif, cur_chr, cur_strand, sp_options)
        nak = moptions['Base']
        outfile = '%s/mod_pos.%s%s.%s.bed' % (moptions['outFolder'], cur_chr, cur_strand, nak)

        touch, cov, mod = summarize_tables(stored_chunks(moptions, sp_options, cur_chr, cur_strand, nak), nak, device)
        if moptions.get('outLevel', OUTPUT_WARNING) <= OUTPUT_INFO:
            print('====sum done! To save')
            print('\tSave %s' % outfile)
        bed = summary.bed_lines(cur_chr, cur_strand, nak, touch, cov, mod)
        if len(bed) > 0:                        # the reference writes no file for an empty table (:1109)
            with open(outfile, 'wb') as mw:
                mw.write(bed)


# ---------------------------------------------------------------------------------------------
# manager
# ---------------------------------------------------------------------------------------------
SUBFOLDER_BATCHES = 100     # a new output sub-folder every 100 batches (myDetect.py:1161, :1168-1169)


def discover_inputs(wrk_base, recursive, sizes=None):
    """Containers under the working folder, optionally up to three levels down, sorted (the names `glob` would match: no hidden
    entries, folders behind symbolic links followed).  One directory scan; `sizes` (a dict) receives path -> bytes for
    plan_batches_sized - a folder of 10^5 containers is listed in a fraction of a second."""
    suffixes = (predstore.CONTAINER_SUFFIX, rawreads.RAW_SUFFIX)
    found = []

    def scan(folder, depth):
        try:
            entries = list(os.scandir(folder))
        except OSError:
            return
        for e in entries:
            if e.name.startswith('.'):
                continue
            if e.name.endswith(suffixes):
                path = os.path.join(folder, e.name)
                found.append(path)
                if sizes is not None:
                    try:
                        sizes[path] = e.stat().st_size
                    except OSError:
                        pass                            # (the feeder reports the unreadable input)
            elif depth > 0:
                try:
                    if e.is_dir():
                        scan(os.path.join(folder, e.name), depth - 1)
                except OSError:
                    pass

    scan(wrk_base, 3 if recursive else 0)
    return sorted(found)


def plan_batches(files, per_batch):
    """[(files, sub-folder id, batch id)]: consecutive groups of `per_batch` inputs - the work items of the reference's
    h5files_Q (myDetect.py:1160-1172)."""
    return [(files[i:i + per_batch], (i // per_batch) // SUBFOLDER_BATCHES, i // per_batch)
            for i in range(0, len(files), per_batch)]


def plan_batches_sized(files, per_batch, max_bytes, sizes=None):
    """The work items of a streaming run: consecutive inputs, at most `per_batch` of them and at most `max_bytes` of input per batch
    (one input always fits) - a feeder hands a batch over in one shared-memory slot, and a batch that outgrows the slot takes the slow
    way through a file of its own (stream.StreamEngine.run_processes)."""
    items, cur, cur_bytes = [], [], 0
    for f in files:
        size = sizes.get(f) if sizes is not None else None
        if size is None:
            try:
                size = os.path.getsize(f)
            except OSError:
                size = 0                                # (the feeder reports the unreadable input)
        if cur and (len(cur) >= per_batch or cur_bytes + size > max_bytes):
            items.append((cur, len(items) // SUBFOLDER_BATCHES, len(items)))
            cur, cur_bytes = [], 0
        cur.append(f)
        cur_bytes += size
    if cur:
        items.append((cur, len(items) // SUBFOLDER_BATCHES, len(items)))
    return items


def _run_processes(ctx, target, argsets, what, poll=None):
    """Start one process per argument tuple, call `poll()` while they run, fail loudly if one dies."""
    procs = [ctx.Process(target=target, args=a) for a in argsets]
    for p in procs:
        p.start()
    while any(p.is_alive() for p in procs):
        # a worker that died takes the run down at once: the ranks of a streaming run sit in collectives (communicator
        # creation, the final reduce) that a dead rank never joins - waiting for the others to end would wait for ever
        if any(p.exitcode not in (None, 0) for p in procs):
            for p in procs:
                if p.is_alive():
                    p.terminate()
            break
        if poll is None or not poll():
            # returns the moment one of the RUNNING ones ends (the sentinel of a process that has exited stays ready: waiting on it
            # too would turn this loop into a busy spin for the tail of the run)
            multiprocessing.connection.wait([p.sentinel for p in procs if p.is_alive()] or [], timeout=0.02)
    for p in procs:
        p.join()
    if any(p.exitcode != 0 for p in procs):
        raise RuntimeError('a %s worker died: exit codes %s' % (what, [p.exitcode for p in procs]))


def _report_errors(ledger):
    if ledger:
        print('Error information for different fast5 files:')
        for reason, files in ledger.items():
            print('\t' + reason, len(files))


def merge_index_files(out_root, wrk_base):
    """Per-batch `<chr>.rnn.pred.ind.<batch>` files of all sub-folders -> one sorted `rnn.pred.ind.<chr>` per chromosome
    with the two `#base_folder_*` header lines (byte format of myDetect.py:1194-1221; every line ends with ' ' + newline)."""
    per_chr = defaultdict(list)
    for path in glob.glob(os.path.join(out_root, '*', '*.' + pre_base_str + '.*')):
        per_chr[os.path.basename(path).split('.' + pre_base_str)[0]].append(path)
    for chrom, paths in per_chr.items():
        records = []
        for path in paths:
            with open(path) as fh:
                for fields in (ln.split() for ln in fh):
                    if fields:
                        records.append(fields[:2] + [int(fields[2])] + fields[3:])
        records.sort()                     # (chr, strand, mapped start, key, ...)
        lines = [['#base_folder_fast5', wrk_base], ['#base_folder_output', os.path.abspath(out_root)]] + records
        with open(os.path.join(out_root, pre_base_str + '.' + chrom), 'w') as fh:
            fh.writelines(' '.join(str(v) for v in rec) + ' \n' for rec in lines)


def _run_stored_detect(moptions, ctx, pmanager, items, ngpu):
    work_q, failed_q, info_q = pmanager.Queue(), pmanager.Queue(), pmanager.Queue()
    for it in items:
        work_q.put(it)
    ledger = defaultdict(list)

    def poll():
        try:
            reason, files = failed_q.get(block=False)
        except Exception:
            return False
        ledger[reason].extend(files)
        return True

    _run_processes(ctx, _worker_entry, [(detect_handler, (moptions, work_q, failed_q, info_q), w % ngpu)
                                        for w in range(moptions['threads'])], 'detect', poll)
    while poll():
        pass
    merge_index_files(moptions['outFolder'] + moptions['FileID'], moptions['wrkBase'])
    return ledger


# what one feeder process prepares per second (profiles/r03/e2e_rate.txt) against what one GPU classifies: packed feature containers
# 6.4e7 rows/s per feeder - one or two feed a GPU; raw containers (signal statistics + alignment walk + rows) 1.7e7 - four to six do
RAW_FEEDERS_PER_GPU = 4


def feeder_budget(threads: int, world: int, usable_cpus: int, raw_input: bool):
    """How the host cores of a streaming run are shared out: (feeder threads per rank asked for, feeder PROCESSES per rank, warning or
    None).  `--threads` is the total over all ranks; a rank gets threads // world feeders, within the CPUs this job may use once every
    rank's own process (device queue, uploads, the signal server of raw input) has one: (usable_cpus - world) // world.  On a 16-CPU
    allowance with 8 GPUs that is ONE feeder per GPU - enough for packed feature containers, a third of what a GPU takes from raw
    containers: the run goes on, feeder-bound, and says so once."""
    world = max(1, int(world))
    feeders = max(1, int(threads) // world)
    procs = max(1, min(feeders, (int(usable_cpus) - world) // world))
    warning = None
    if raw_input and procs < RAW_FEEDERS_PER_GPU:
        warning = ("Warning: %d feeder process(es) per GPU (--threads %d over %d GPU(s), %d usable CPUs): raw input needs about %d per GPU to "
                   "keep it busy - this run will be bound by its feeders" % (procs, threads, world, usable_cpus, RAW_FEEDERS_PER_GPU))
    return feeders, procs, warning


def _run_streaming_detect(moptions, ctx, pmanager, items, ngpu):
    """One process per GPU (rank), work items pulled from one shared queue by all ranks, counters merged with RCCL
    (deepmod_amd/stream.py).  -> (error ledger, per-rank stats)"""
    from . import stream
    world = max(1, min(ngpu, len(items)))
    # (no manager process on this path: the items are known, a shared counter hands them out - stream.WorkList)
    work_q, result_q = stream.WorkList(items, ctx), ctx.Queue()
    run_opts = dict(moptions, outFolder=moptions['outFolder'] + moptions['FileID'])
    rdv = os.path.join(run_opts['outFolder'], '.rendezvous')
    if os.path.isdir(rdv):
        for f in os.listdir(rdv):
            os.remove(os.path.join(rdv, f))
    from . import rawreads
    raw_input = any(f.endswith(rawreads.RAW_SUFFIX) for it in items for f in it[0])
    feeders, feeder_procs, warning = feeder_budget(moptions['threads'], world, stream.usable_cpus(), raw_input)
    if not moptions.get('feeder_procs', 1):
        feeder_procs = 0
    elif warning:
        print(warning)
    # raw containers: the signal stage of all feeders runs in the GPU process (stream.signal_server).  Without it every
    # feeder owns a HIP context, and beyond ~8 such processes per GPU the hardware queues are oversubscribed (8 feeders
    # 9.1e6 base-positions/s, 11 feeders 6.3e6 with the device queue waiting 2.9 of 4.3 s - profiles/r02/README.md)
    if not run_opts.get('signal_server', True) and raw_input:
        feeder_procs = min(feeder_procs, int(moptions.get('feeder_procs_raw', 8)))
    ledger, stats = defaultdict(list), []

    def collect():          # while the ranks run: a rank's report may be larger than the pipe holds, and it ends only once it is read
        try:
            res = result_q.get(block=False)
        except Exception:
            return False
        for reason, files in res['errors'].items():
            ledger[reason].extend(files)
        stats.append(res['stats'])
        return True

    try:
        if world == 1 and not int(moptions.get('rank_process', os.environ.get('DEEPMOD_RANK_PROCESS', 0))):
            # one GPU: this process is the rank (no second interpreter to start and to import into: ~0.45 s of a 4.8 s command,
            # profiles/r06/raw_profile.txt "GPU process running"); its feeders are processes of their own as before
            res = stream.stream_rank_main(run_opts, 0, 1, 0, work_q, None, feeders, feeder_procs)
            for reason, files in res['errors'].items():
                ledger[reason].extend(files)
            stats.append(res['stats'])
        else:
            _run_processes(ctx, stream.stream_rank_main,
                           [(run_opts, r, world, 0 if moptions.get('one_device') else r, work_q, result_q, feeders, feeder_procs) for r in range(world)],
                           'streaming detect', collect)
    finally:
        work_q.close()
    while collect():
        pass
    if len(stats) != world:
        raise RuntimeError('streaming detect: %d of %d ranks reported' % (len(stats), world))
    return ledger, stats


_BASE_OF_RUN = ['?']


def _print_stream_stats(stats, wall, since_start=None, planning=None):
    tot = defaultdict(float)
    for st in stats:
        for k, v in st.items():
            tot[k] += v
    rows = tot['windows']
    print('Streaming detect: %d reads, %d base-positions classified on %d GPU(s) in %.1f s = %.3g base-positions/s'
          % (tot['reads'], rows, len(stats), wall, rows / max(wall, 1e-9)))
    if tot.get('submit_classified'):
        print('\twindows run through the classifier: %d of those %d base-positions (only a window centred on base %s can reach the BED, '
              'myDetect.py:1091; --storePred 1 classifies every base as the reference does) = %.3g windows/s'
              % (tot['submit_classified'], rows, _BASE_OF_RUN[0], tot['submit_classified'] / max(wall, 1e-9)))
    host = {k[5:]: v for k, v in tot.items() if k.startswith('prep_')}
    busy = sum(host.values()) + tot['submit']
    parts = ['%s %.0f%%' % (k, 100 * v / max(busy, 1e-9)) for k, v in sorted(host.items(), key=lambda kv: -kv[1])]
    print('\thost stages (share of feeder + submit time): ' + ', '.join(parts) + ', device submit %.0f%%' % (100 * tot['submit'] / max(busy, 1e-9))
          + '; %d batches (%d handed over in shared-memory slots); detect wall %.1f s, waiting for feeders %.1f s, merge + BED %.1f s per rank'
          % (tot['batches'], tot['slot_batches'], tot['detect_wall'] / len(stats), tot['wait_feed'] / len(stats), tot['merge+bed'] / len(stats))
          + '; device queue: waiting for the device %.1f s, staging copy %.1f s, launches %.1f s'
          % (tot['submit_wait_device'] / len(stats), tot['submit_stage'] / len(stats), tot['submit_launch'] / len(stats))
          + ('; signal server %.1f s (%.1f s copying requests into page-locked memory, %.1f s inside the signal call) for %d requests; event statistics resident on the device '
             'for %d of %d rows, the batch loop waited %.2f s for them'
             % (tot['signal_server'] / len(stats), tot['signal_server_copy'] / len(stats), tot['signal_server_call'] / len(stats), tot['signal_requests'],
                tot['submit_stats_on_device'], tot['rows'], tot['submit_wait_signal'] / len(stats)) if tot['signal_requests'] else ''))
    if tot['signal_requests']:
        print('\tsignal stage: %d samples, %d merged events in %d requests' % (tot['signal_samples'], tot['signal_events'], tot['signal_requests']))
    if 'at_rank_start' in tot and 'at_drained' in tot:
        n, r0 = len(stats), tot['at_rank_start'] / len(stats)
        print('\ttimeline, seconds after the command began its detect step (mean over ranks): %sGPU process running %.2f, feeder processes started %.2f, model on the device %.2f, '
              'first batch from a feeder %.2f, last batch %.2f, device drained %.2f, feeders gone %.2f, BED written %.2f, process done %.2f, processes joined %.2f'
              % ('' if planning is None else 'inputs listed %.2f, batches planned %.2f, ' % planning, r0, r0 + tot['at_feeders_started'] / n, r0 + tot['at_backend_ready'] / n, r0 + tot['at_first_batch'] / n, r0 + tot['at_last_batch'] / n, r0 + tot['at_drained'] / n,
                 r0 + tot['at_feeders_gone'] / n, tot['at_bed_written'] / n, tot['at_rank_end'] / n, wall if since_start is None else since_start))


def _run_summary_jobs(moptions, ctx, pmanager, ngpu):
    """One job per index file x strand, `threads` summary workers (myDetect.py:1232-1252)."""
    index_files = sorted(glob.glob(os.path.join(moptions['predpath'], pre_base_str + '.*')))
    print('Find: %s %d %s' % (moptions['predpath'], len(index_files), pre_base_str))
    job_q = pmanager.Queue()
    for path in index_files:
        chrom = os.path.basename(path)[len(pre_base_str) + 1:]
        for strand in '+-':
            job_q.put((path, chrom, strand))
    nworkers = min(moptions['threads'], 2 * len(index_files))
    _run_processes(ctx, _worker_entry, [(sum_handler, (moptions, job_q), w % ngpu) for w in range(nworkers)], 'summary')


def _worker_entry(target, args, device):
    target(*args, device=device)


class _LazyManager:
    """ctx.Manager() on first use: a streaming run never needs one."""

    def __init__(self, ctx):
        self._ctx, self._m = ctx, None

    def Queue(self):
        if self._m is None:
            self._m = self._ctx.Manager()
        return self._m.Queue()


def mDetect_manager(moptions):
    """The detect run (counterpart of myDetect.py:1124-1263): inputs -> worker batches -> detect -> per-position summary
    -> `<outFolder>.done`.  predDet == 1 runs the streaming mode unless `storePred` asks for the reference's per-read
    files; predDet == 0 summarises the stored predictions under `predpath` (bin/DeepMod.py:143-148)."""
    if moptions.get('mod_cluster'):
        # the reference marks this branch of sum_handler "should not used now" (myDetect.py:1054-1087); neither run mode builds it,
        # and writing plain mod_pos.* files for a run that asked for cluster_mod_pos.* would be a silent change of meaning
        raise NotImplementedError("--mod_cluster 1 is not built (the reference marks that branch 'should not used now', myDetect.py:1054)")
    moptions['_t_manager'] = time.time()
    ctx = multiprocessing.get_context('spawn')      # never fork a process that may hold a HIP context
    pmanager = _LazyManager(ctx)       # the stored-prediction paths share their queues through a manager like the reference (myDetect.py:1141)
    ngpu = max(1, int(moptions.get('gpus', 1)))
    moptions['threads'] = max(1, int(moptions['threads']))
    if moptions.get('wrkBase'):
        moptions['wrkBase'] = moptions['wrkBase'].rstrip('/\\') or moptions['wrkBase']
    streamed = False
    if moptions['predDet'] == 1:
        t0 = time.time()
        cut = moptions['modfile'].rfind('/')
        moptions['modfile'] = [moptions['modfile'], './' if cut == -1 else moptions['modfile'][:cut + 1]]
        sizes = {}
        files = discover_inputs(moptions['wrkBase'], moptions['recursive'] == 1, sizes)
        moptions['_t_listed'] = time.time() - moptions['_t_manager']
        print('Total files=%d' % len(files))
        out_root = moptions['outFolder'] + moptions['FileID']
        os.makedirs(out_root, exist_ok=True)
        streamed = not moptions.get('storePred', 0)
        per_batch = moptions['files_per_thread']
        if streamed:
            # a streaming run writes nothing per batch: the batch is only the unit the feeders take from the work queue, and the
            # counters are sums, so the BED does not depend on it. The reference's default (1000 single-read FAST5 files) would put
            # a whole run of multi-read containers into ONE batch and leave all feeders but one idle: keep >= 8 batches per feeder,
            # and cut them to fit the hand-over slots (feature rows are no larger than the containers they come from: a batch of
            # <= 0.8 slot of input fits its slot)
            per_batch = max(1, min(per_batch, -(-len(files) // (8 * moptions['threads']))))
            items = plan_batches_sized(files, per_batch, int(0.8 * (int(moptions.get('feeder_slot_mb', 128)) << 20)), sizes)
            _BASE_OF_RUN[0] = moptions['Base']
            moptions['_t_planned'] = time.time() - moptions['_t_manager']
            ledger, stats = _run_streaming_detect(moptions, ctx, pmanager, items, ngpu)
            _print_stream_stats(stats, time.time() - t0, time.time() - moptions['_t_manager'], (moptions['_t_listed'], moptions['_t_planned']))
        else:
            items = plan_batches(files, per_batch)
            ledger = _run_stored_detect(moptions, ctx, pmanager, items, ngpu)
            moptions['predpath'] = moptions['outFolder'] + '/' + moptions['FileID']
        _report_errors(ledger)
        moptions['outFolder'] = out_root
        print("Per-read Prediction consuming time %d" % (time.time() - t0))
    if not streamed:
        t0 = time.time()
        _run_summary_jobs(moptions, ctx, pmanager, ngpu)
        print("Genomic-position Detection consuming time %d" % (time.time() - t0))
    open(moptions['outFolder'] + '.done', 'a').close()
