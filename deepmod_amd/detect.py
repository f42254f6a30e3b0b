"""Host-side counterparts of the reference's detect path (bin/DeepMod_scripts/myDetect.py), same
function names, argument meaning and side effects, driving the HIP library instead of TensorFlow.

    mPredict1        myDetect.py:787-834   windows + batched inference + class -> base scatter
    detect_handler   myDetect.py:948-984   worker: one model per process, pulls batches from a queue
    sum_handler      myDetect.py:1028-1120 per (chr, strand) coverage / mod-count summary -> BED
    mDetect_manager  myDetect.py:1124-1263 sharding, worker processes, index merge, .done marker

Inputs are containers instead of .fast5 files (no h5py / libhdf5 in this image): *raw containers*
(deepmod_amd/rawreads.py: DAC samples + basecaller events; normalised on the GPU, aligned with the SAM
records, mapped by dm_map_read) or *feature containers* (deepmod_amd/predstore.py: per-read `mfeatures`,
`base_map_info`, clips, event bases - exactly the arguments the reference hands to mPredict1 at
myDetect.py:715).
"""
from __future__ import annotations

import glob
import multiprocessing
import os
import time
from collections import defaultdict

import numpy as np

from . import predstore, rawreads, readmap

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
    ev_base = np.array([s[2] for s in modevents['model_state'][start_clip:start_clip + n]], dtype='U1')
    bad = np.flatnonzero(base_map_info['readbase'][aligned] != ev_base)
    for b in bad:
        print('Error Does not match', base_map_info['readbase'][aligned[b]], ev_base[b], aligned[b], b + start_clip)
    hit = aligned[np.asarray(mfpred_output) == 1]
    base_map_info['mod_pred'][hit] = 1
    return int(len(hit))


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
    _, cls = sess.model.predict_read(rows, half, len(rows) - 2 * half, want_prob=False)
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
def _predict_and_store(moptions, sp_options, store, good, src):
    """mPredict1 for a group of reads (one device call), prediction tables and index entries (myDetect.py:715-718)."""
    if not good:
        return
    try:      # one device call for all reads of the group (see mPredict_batch)
        sess = sp_options['rnn'][0]
        if hasattr(sess, 'model') and getattr(sess, 'model') is not None:
            pred_nums = mPredict_batch(moptions, sp_options, good)
        else:
            pred_nums = [mPredict1(moptions, sp_options, {'f5data': {rd['readk']: (None, rd['events'], None, src)}},
                                   rd['mfeatures'], rd['base_map_info'], rd['readk'], rd['start_clip'], rd['end_clip'])
                         for rd in good]
    except Exception as exc:  # same (reason -> files) error channel as the reference
        sp_options["Error"]["Prediction failed: %s" % type(exc).__name__].append(src)
        return
    for rd, pred_mod_num in zip(good, pred_nums):
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
    f5data = rawreads.get_Event_Signals(moptions, sp_options, raw_files)
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


def detect_handler(moptions, h5files_Q, failed_Q, file_map_info_q, device=0):
    """Worker process: one model on one GPU, pulls (files, ctfolderid, batchid) until the queue is
    empty (myDetect.py:948-984)."""
    from . import model as dm
    _, init_l, _, _, _, X, Y, _, _, _, _, mfpred = dm.mCreateSession(moptions['fnum'], moptions['hidden'],
                                                                     moptions['windowsize'], moptions)
    sess = dm.new_session(device)
    new_saver = dm.import_meta_graph(moptions['modfile'][0] + '.meta')
    new_saver.restore(sess, dm.latest_checkpoint(moptions['modfile'][1]) or moptions['modfile'][0])

    while not h5files_Q.empty():
        cur_start_time = time.time()
        try:
            f5files, ctfolderid, batchid = h5files_Q.get(block=False)
        except Exception:
            break
        sp_options = defaultdict()
        sp_options['ctfolderid'] = ctfolderid
        sp_options['ctfolder'] = moptions['outFolder'] + moptions['FileID'] + '/' + str(ctfolderid)
        os.makedirs(sp_options['ctfolder'], exist_ok=True)
        sp_options['rnn'] = (sess, X, Y, init_l, mfpred)
        sp_options['batchid'] = batchid
        sp_options['Mod'] = []
        sp_options['Error'] = defaultdict(list)
        mDetect1(moptions, sp_options, f5files)
        for errtype, errfiles in sp_options["Error"].items():
            failed_Q.put((errtype, errfiles))
        if moptions.get('outLevel', OUTPUT_WARNING) <= OUTPUT_INFO:
            print("Cur Prediction consuming time %d for %d %d" % (time.time() - cur_start_time, ctfolderid, batchid))
    sess.close()


# ---------------------------------------------------------------------------------------------
# summary
# ---------------------------------------------------------------------------------------------
def read_file_list(cur_cif, cur_chr, cur_strand, sp_options):
    """Index-file reader, same format as the reference (myDetect.py:989-1008)."""
    cur_list = []
    with open(cur_cif, 'r') as mr:
        for line in mr:
            line = line.strip()
            if not line:
                continue
            lsp = line.split()
            if line[0] == '#':
                if lsp[1][-1] not in ['/', '\\']:
                    lsp[1] = lsp[1] + '/'
                if lsp[0] == '#base_folder_fast5':
                    sp_options['base_folder_fast5'] = lsp[1]
                elif lsp[0] == '#base_folder_output':
                    sp_options['base_folder_output'] = lsp[1]
            else:
                if lsp[1] == cur_strand:
                    cur_list.append(lsp)
                if not lsp[0] == cur_chr:
                    print('Warning!!! The chr should be %s but %s is found.' % (cur_chr, lsp[0]))
    sp_options['handlingList'] = cur_list


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


def summarize_tables(tables, base: str, device: int = 0, length=None):
    """Accumulate an iterable of prediction tables on the GPU -> (touch, cov, mod) int32 arrays."""
    from . import summary
    pos_parts, flag_parts = [], []
    maxpos = -1
    for m_pred in tables:
        if len(m_pred) == 0:
            continue
        fl = base_flags(m_pred, base)
        keep = (fl & 1) != 0
        if not keep.any():
            continue
        p = m_pred['refbasei'][keep].astype(np.int64)
        pos_parts.append(p)
        flag_parts.append(fl[keep])
        maxpos = max(maxpos, int(p.max()))
    if length is None:
        length = maxpos + 1
    if length <= 0:
        z = np.zeros(0, np.int32)
        return z, z.copy(), z.copy()
    summ = summary.PositionSummary(length, device)
    if pos_parts:
        summ.add(np.concatenate(pos_parts), np.concatenate(flag_parts))
    out = summ.fetch()
    summ.close()
    return out


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

        def tables():
            for hl in sp_options['handlingList']:
                m_pred, mapped_chrom, mapped_strand = read_pred_detail(moptions, sp_options, hl)
                if not (mapped_chrom == cur_chr and mapped_strand == cur_strand):
                    print("ERRoR not the same chr (real=%s vs expect=%s) and strand (real=%s VS expect=%s)" %
                          (mapped_chrom, cur_chr, mapped_strand, cur_strand))
                yield m_pred

        touch, cov, mod = summarize_tables(tables(), nak, device)
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
def _worker_entry(target, args, device):
    target(*args, device=device)


def mDetect_manager(moptions):
    """Same orchestration shape as the reference (myDetect.py:1124-1263): batches of
    `files_per_thread` inputs, `threads` worker processes (round-robin over the visible GPUs), per-chr
    index merge, one summary job per chr x strand, `<outFolder>.done` marker."""
    ctx = multiprocessing.get_context('spawn')   # never fork a process that may hold a HIP context
    pmanager = ctx.Manager()
    while moptions.get('wrkBase') and moptions['wrkBase'][-1] in ['/', '\\']:
        moptions['wrkBase'] = moptions['wrkBase'][:-1]
    ngpu = max(1, int(moptions.get('gpus', 1)))

    if moptions['predDet'] == 1:
        if moptions['modfile'].rfind('/') == -1:
            moptions['modfile'] = [moptions['modfile'], './']
        else:
            moptions['modfile'] = [moptions['modfile'], moptions['modfile'][:moptions['modfile'].rfind('/') + 1]]
        start_time = time.time()
        f5files = []
        for pat in ('*' + predstore.CONTAINER_SUFFIX, '*' + rawreads.RAW_SUFFIX):
            f5files.extend(glob.glob(os.path.join(moptions['wrkBase'], pat)))
            if moptions['recursive'] == 1:
                for depth in ('*/', '*/*/', '*/*/*/'):
                    f5files.extend(glob.glob(os.path.join(moptions['wrkBase'], depth + pat)))
        f5files = sorted(f5files)
        print('Total files=%d' % len(f5files))
        os.makedirs(moptions['outFolder'] + moptions['FileID'], exist_ok=True)

        h5files_Q = pmanager.Queue()
        file_map_info_q = pmanager.Queue()
        failed_Q = pmanager.Queue()
        h5_batch = []
        h5batchind = 0
        sub_folder_size = 100
        sub_folder_id = 0
        for f5f in f5files:
            h5_batch.append(f5f)
            if len(h5_batch) == moptions['files_per_thread']:
                h5files_Q.put((h5_batch, sub_folder_id, h5batchind))
                h5_batch = []
                h5batchind += 1
                if h5batchind % sub_folder_size == 0:
                    sub_folder_id += 1
        if len(h5_batch) > 0:
            h5files_Q.put((h5_batch, sub_folder_id, h5batchind))
            h5batchind += 1

        share_var = (moptions, h5files_Q, failed_Q, file_map_info_q)
        handlers = []
        for hid in range(moptions['threads']):
            p = ctx.Process(target=_worker_entry, args=(detect_handler, share_var, hid % ngpu))
            p.start()
            handlers.append(p)
        failed_files = defaultdict(list)
        while any(p.is_alive() for p in handlers):
            try:
                errk, fns = failed_Q.get(block=False)
                failed_files[errk].extend(fns)
            except Exception:
                time.sleep(0.05)
        while not failed_Q.empty():
            errk, fns = failed_Q.get(block=False)
            failed_files[errk].extend(fns)
        if any(p.exitcode != 0 for p in handlers):
            raise RuntimeError('a detect worker died: exit codes %s' % [p.exitcode for p in handlers])

        # merge per-batch index files -> rnn.pred.ind.<chr>, sorted (myDetect.py:1194-1221)
        moptions['predpath'] = moptions['outFolder'] + '/' + moptions['FileID']
        pred_ind_pref = moptions['outFolder'] + '/' + moptions['FileID'] + '/' + pre_base_str
        pred_chr_files = glob.glob(os.path.join(moptions['outFolder'] + moptions['FileID'], '*/*.' + pre_base_str + '.*'))
        chr_dict = defaultdict(list)
        for pcf in pred_chr_files:
            chr_dict[pcf.split('/')[-1].split('.' + pre_base_str)[0]].append(pcf)
        for ck in chr_dict:
            cur_list = [['#base_folder_fast5', moptions['wrkBase']],
                        ['#base_folder_output', os.path.abspath(moptions['outFolder'] + moptions['FileID'])]]
            for sub_c_f in chr_dict[ck]:
                with open(sub_c_f, 'r') as mr:
                    for line in mr:
                        line = line.strip()
                        if len(line) > 0:
                            lsp = line.split()
                            lsp[2] = int(lsp[2])
                            cur_list.append(lsp)
            cur_list = sorted(cur_list)   # '#...' header rows sort first, records by (chr, strand, start, key)
            with open(pred_ind_pref + '.' + ck, 'w') as indf_writer:
                for mfi in cur_list:
                    indf_writer.write(' '.join([str(v) for v in mfi] + ['\n']))
        if len(failed_files) > 0:
            print('Error information for different fast5 files:')
            for errtype, errfiles in failed_files.items():
                print('\t' + errtype, len(errfiles))
        moptions['outFolder'] = moptions['outFolder'] + moptions['FileID']
        print("Per-read Prediction consuming time %d" % (time.time() - start_time))

    # ---- summary phase (also the --predDet 0 --predpath resume path, DeepMod.py:143-148)
    start_time = time.time()
    all_chr_ind_files = sorted(glob.glob(os.path.join(moptions['predpath'], pre_base_str + '.*')))
    print('Find: %s %d %s' % (moptions['predpath'], len(all_chr_ind_files), pre_base_str))
    chr_strand_Q = pmanager.Queue()
    jobnum = 0
    for cur_cif in all_chr_ind_files:
        chrname = cur_cif.split(pre_base_str)[-1][1:]
        chr_strand_Q.put((cur_cif, chrname, '+'))
        chr_strand_Q.put((cur_cif, chrname, '-'))
        jobnum += 2
    handlers = []
    for hid in range(min(moptions['threads'], jobnum)):
        p = ctx.Process(target=_worker_entry, args=(sum_handler, (moptions, chr_strand_Q), hid % ngpu))
        p.start()
        handlers.append(p)
    for p in handlers:
        p.join()
    if any(p.exitcode != 0 for p in handlers):
        raise RuntimeError('a summary worker died: exit codes %s' % [p.exitcode for p in handlers])
    print("Genomic-position Detection consuming time %d" % (time.time() - start_time))
    open(moptions['outFolder'] + '.done', 'a').close()
