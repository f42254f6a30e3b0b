"""On-disk containers used on either side of the hot path.

* feature container  `<name>.dmfeat.npz`: for each read exactly what the reference hands to
  mPredict1 (myDetect.py:715): `mfeatures float64[N+200,10]` (get_Feature, :839-903),
  `base_map_info` (:660), the clips, and the basecalled base of every event.  Stands in for the
  FAST5 + aligner front end, which is out of scope this round (no h5py / minimap2 here).
* prediction store   `rnn.pred.detail.npz.<batchid>`: per-read `predetail` table + the attributes the
  reference stores in HDF5 (`pred/<key>/predetail`, myDetect.py:716-760).  h5py is not available in
  this image, so the same fields are kept in an .npz; the index-file lines are byte-compatible.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List

import numpy as np

CONTAINER_SUFFIX = '.dmfeat.npz'
EVENT_DTYPE = [('mean', '<f4'), ('stdv', '<f4'), ('start', np.uint64), ('length', np.uint64), ('model_state', 'U5')]
BMI_DTYPE = [('refbase', 'U1'), ('readbase', 'U1'), ('refbasei', np.uint64), ('readbasei', np.uint64), ('mod_pred', int)]


def s1_to_u1(a) -> np.ndarray:
    """'S1' -> 'U1' through the code points (numpy's own string cast converts element by element: 20x slower)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.dtype('U1'):
        return a
    return a.view(np.uint8).astype(np.uint32).view('U1')


def u1_to_s1(a) -> np.ndarray:
    """'U1' (ASCII) -> 'S1' through the code points."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.dtype('S1'):
        return a
    return a.view(np.uint32).astype(np.uint8).view('S1')


def make_base_map_info(refbase, readbase, refbasei, readbasei=None, mod_pred=None) -> np.ndarray:
    n = len(refbase)
    bmi = np.zeros(n, dtype=BMI_DTYPE)
    bmi['refbase'] = refbase
    bmi['readbase'] = readbase
    bmi['refbasei'] = refbasei
    bmi['readbasei'] = readbasei if readbasei is not None else 0
    bmi['mod_pred'] = mod_pred if mod_pred is not None else 0
    return bmi


def events_from_bases(bases, mean=None, stdv=None, length=None) -> np.ndarray:
    ev = np.zeros(len(bases), dtype=EVENT_DTYPE)
    ev['model_state'] = np.char.add(np.char.add('NN', np.asarray(bases, dtype='U1')), 'NN')
    if mean is not None:
        ev['mean'] = mean
    if stdv is not None:
        ev['stdv'] = stdv
    if length is not None:
        ev['length'] = length
    return ev


def save_feature_container(path: str, reads: List[Dict]) -> None:
    arrays = {}
    metas = []
    for i, rd in enumerate(reads):
        bmi = rd['base_map_info']
        arrays['r%d_mfeatures' % i] = np.asarray(rd['mfeatures'], dtype=np.float64)
        arrays['r%d_refbase' % i] = s1_to_u1(bmi['refbase'])
        arrays['r%d_readbase' % i] = s1_to_u1(bmi['readbase'])
        arrays['r%d_refbasei' % i] = bmi['refbasei'].astype(np.uint64)
        arrays['r%d_readbasei' % i] = bmi['readbasei'].astype(np.uint64)
        arrays['r%d_evbase' % i] = np.array([s[2] for s in rd['events']['model_state']], dtype='U1')
        metas.append({k: rd[k] for k in ('readk', 'chr', 'strand', 'mapped_start', 'start_clip', 'end_clip')})
    arrays['meta'] = np.array(json.dumps(metas))
    if not path.endswith(CONTAINER_SUFFIX):
        raise ValueError('feature containers must end with ' + CONTAINER_SUFFIX)
    with open(path, 'wb') as fh:
        np.savez_compressed(fh, **arrays)


def load_feature_container(path: str) -> List[Dict]:
    z = np.load(path, allow_pickle=False)
    if 'format' in z.files and int(z['format']) == 2:
        return _classic_reads(load_packed(path))
    metas = json.loads(str(z['meta']))
    reads = []
    for i, m in enumerate(metas):
        rd = dict(m)
        rd['mfeatures'] = z['r%d_mfeatures' % i]
        rd['base_map_info'] = make_base_map_info(z['r%d_refbase' % i], z['r%d_readbase' % i], z['r%d_refbasei' % i],
                                                 z['r%d_readbasei' % i])
        rd['events'] = events_from_bases(z['r%d_evbase' % i])
        reads.append(rd)
    return reads


# ---------------------------------------------------------------------------------------------
# packed feature containers (format 2): the same per-read arguments of mPredict1, concatenated over the reads of a
# container and stored uncompressed, in the dtypes the device consumes - a 30x E. coli run is 1.4e8 table rows, and
# per-read float64 matrices + U1 columns behind deflate cost 40 % of a worker's host time in round 1.
#   tx        float32 [R, 7]   mfeatures[:, 3:] of every read, 100 zero-padded rows each side (myDetect.py:850-851)
#   refbase / readbase  S1 [B] base_map_info columns (dtype :752 of the on-disk table), refbasei int64 [B]
#   evbase    S1 [E]           basecalled base of every event (model_state[2], :824-833)
#   row_off / bmi_off / ev_off  int64 [n_reads + 1]
#   meta      JSON: per read readk, chr, strand, mapped_start, start_clip, end_clip; 'contig_len' {chr: length}
# ---------------------------------------------------------------------------------------------
def save_packed_container(path: str, reads: List[Dict], contig_len: Dict[str, int] = None) -> None:
    """reads: classic read dicts (mfeatures, base_map_info, events, ...) or packed ones (tx, refbase, ..., evbase)."""
    if not path.endswith(CONTAINER_SUFFIX):
        raise ValueError('feature containers must end with ' + CONTAINER_SUFFIX)
    tx, refb, readb, refi, evb, metas = [], [], [], [], [], []
    for rd in reads:
        if 'tx' in rd:
            tx.append(np.asarray(rd['tx'], np.float32))
            refb.append(np.asarray(rd['refbase'], 'S1'))
            readb.append(np.asarray(rd['readbase'], 'S1'))
            refi.append(np.asarray(rd['refbasei'], np.int64))
            evb.append(np.asarray(rd['evbase'], 'S1'))
        else:
            bmi = rd['base_map_info']
            tx.append(np.asarray(rd['mfeatures'][:, 3:], np.float32))
            refb.append(u1_to_s1(bmi['refbase']))
            readb.append(u1_to_s1(bmi['readbase']))
            refi.append(bmi['refbasei'].astype(np.int64))
            evb.append(np.array([s[2] for s in rd['events']['model_state']], dtype='S1'))
        metas.append({k: rd[k] for k in ('readk', 'chr', 'strand', 'mapped_start', 'start_clip', 'end_clip')})
    off = lambda parts: np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    cat = lambda parts, dt, shape: np.concatenate(parts) if parts else np.zeros(shape, dt)
    arrays = {'format': np.array(2), 'tx': cat(tx, np.float32, (0, 7)), 'refbase': cat(refb, 'S1', 0), 'readbase': cat(readb, 'S1', 0),
              'refbasei': cat(refi, np.int64, 0), 'evbase': cat(evb, 'S1', 0), 'row_off': off(tx), 'bmi_off': off(refb),
              'ev_off': off(evb), 'meta': np.array(json.dumps({'reads': metas, 'contig_len': contig_len or {}}))}
    from . import npzmap
    with open(path, 'wb') as fh:
        npzmap.savez_aligned(fh, **arrays)          # members 64-byte aligned in the file: load_packed's views are aligned


def load_packed(path: str) -> Dict:
    """-> {'tx', 'refbase', 'readbase', 'refbasei', 'evbase', 'row_off', 'bmi_off', 'ev_off', 'reads' (meta dicts),
    'contig_len'} for either container format (format-1 files are converted on the fly).  The arrays of a format-2
    container are read-only views into a mapping of the file (deepmod_amd/npzmap.py)."""
    from . import npzmap
    z = npzmap.load(path)
    if 'format' in z and int(z['format']) == 2:
        meta = json.loads(str(z['meta']))
        out = {k: z[k] for k in ('tx', 'refbase', 'readbase', 'refbasei', 'evbase', 'row_off', 'bmi_off', 'ev_off')}
        out['reads'] = meta['reads']
        out['contig_len'] = meta.get('contig_len', {})
        return out
    metas = json.loads(str(z['meta']))
    tx = [z['r%d_mfeatures' % i][:, 3:].astype(np.float32) for i in range(len(metas))]
    refb = [u1_to_s1(z['r%d_refbase' % i]) for i in range(len(metas))]
    readb = [u1_to_s1(z['r%d_readbase' % i]) for i in range(len(metas))]
    refi = [z['r%d_refbasei' % i].astype(np.int64) for i in range(len(metas))]
    evb = [u1_to_s1(z['r%d_evbase' % i]) for i in range(len(metas))]
    off = lambda parts: np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    cat = lambda parts, dt, shape: np.concatenate(parts) if parts else np.zeros(shape, dt)
    return {'tx': cat(tx, np.float32, (0, 7)), 'refbase': cat(refb, 'S1', 0), 'readbase': cat(readb, 'S1', 0),
            'refbasei': cat(refi, np.int64, 0), 'evbase': cat(evb, 'S1', 0), 'row_off': off(tx), 'bmi_off': off(refb),
            'ev_off': off(evb), 'reads': metas, 'contig_len': {}}


def _classic_reads(pk: Dict) -> List[Dict]:
    """Packed arrays -> the classic per-read dicts (mfeatures float64 [N+200, 10] with the feature columns filled,
    base_map_info with the reference's dtype, event table carrying the basecalled bases)."""
    reads = []
    ro, bo, eo = pk['row_off'], pk['bmi_off'], pk['ev_off']
    for i, m in enumerate(pk['reads']):
        rd = dict(m)
        tx = pk['tx'][ro[i]:ro[i + 1]]
        mf = np.zeros((len(tx), 10))
        mf[:, 3:] = tx
        rd['mfeatures'] = mf
        sl = slice(bo[i], bo[i + 1])
        refi = pk['refbasei'][sl].astype(np.uint64)
        readb = s1_to_u1(pk['readbase'][sl])
        ng = readb != '-'
        rd['base_map_info'] = make_base_map_info(s1_to_u1(pk['refbase'][sl]), readb, refi,
                                                 (np.cumsum(ng) - ng).astype(np.uint64))
        rd['events'] = events_from_bases(s1_to_u1(pk['evbase'][eo[i]:eo[i + 1]]))
        reads.append(rd)
    return reads


def savez_fast(path: str, arrays: Dict) -> None:
    """np.savez_compressed with deflate level 1: the reference gzip-compresses its per-read tables too (myDetect.py:752),
    but zlib's default level made the store 60 % of a worker's host time; level 1 is ~4x faster for +15 % bytes.
    np.load reads the result like any .npz."""
    import zipfile
    with zipfile.ZipFile(path, 'w', zipfile.ZIP_DEFLATED, compresslevel=1) as zf:
        for name, arr in arrays.items():
            with zf.open(name + '.npy', 'w', force_zip64=True) as fh:
                np.lib.format.write_array(fh, np.asanyarray(arr), allow_pickle=False)


class PredWriter:
    """Collects the per-read prediction tables of one worker batch (file name and index-line fields
    follow myDetect.py:716-718).
    Store format 2: the tables of all reads of the batch one after the other in ONE member per column (`refbase`, `readbase` as
    bytes, `refbasei`, `readbasei` as uint64, `mod_pred` as int8) with `row_off[n + 1]`; read i has the key `pred_<i>`.  A summary
    worker inflates five members per batch instead of five per read (format 1: 65 of the 80 s of a 30x E. coli run went into opening
    2 x 23,300 x 5 zip members and converting them read by read)."""

    def __init__(self, ctfolder: str, batchid: int):
        self.path = os.path.join(ctfolder.rstrip('/\\'), 'rnn.pred.detail.npz.' + str(batchid))
        self.cols = {k: [] for k in ('refbase', 'readbase', 'refbasei', 'readbasei', 'mod_pred')}
        self.row_off = [0]
        self.attrs = {}
        self.n = 0

    def relpath(self, moptions) -> str:
        return os.path.relpath(self.path, moptions['outFolder'] + moptions['FileID'])

    def add(self, rd, bmi, pred_mod_num, src_file, moptions) -> str:
        key = 'pred_' + str(self.n)
        self.n += 1
        for f in ('refbase', 'readbase'):
            self.cols[f].append(u1_to_s1(bmi[f]))
        self.cols['refbasei'].append(bmi['refbasei'].astype(np.uint64))
        self.cols['readbasei'].append(bmi['readbasei'].astype(np.uint64))
        mp = bmi['mod_pred']
        if len(mp) and (mp.min() < -128 or mp.max() > 127):
            raise ValueError('mod_pred outside the int8 range of the prediction store')
        self.cols['mod_pred'].append(mp.astype(np.int8))
        self.row_off.append(self.row_off[-1] + len(bmi))
        fwd = rd['strand'] == '+'
        if 'num_insertions' in rd:      # counters of the CIGAR walk (myDetect.py:737-741), when the read came through it
            nins, ndel, nmis = int(rd['num_insertions']), int(rd['num_deletions']), int(rd['num_mismatches'])
        else:
            nins = int((bmi['refbase'] == '-').sum())
            ndel = int((bmi['readbase'] == '-').sum())
            nmis = int(((bmi['refbase'] != bmi['readbase']) & (bmi['refbase'] != '-') & (bmi['readbase'] != '-')).sum())
        self.attrs[key] = {
            'mapped_chr': rd['chr'], 'mapped_strand': rd['strand'],
            'mapped_start': int(bmi['refbasei'][0] if fwd else bmi['refbasei'][-1]),
            'mapped_end': int(bmi['refbasei'][-1] if fwd else bmi['refbasei'][0]),
            # myDetect.py:729-734: for '-' reads the two clips are stored the other way round
            'clipped_bases_start': int(rd['start_clip'] if fwd else rd['end_clip']),
            'clipped_bases_end': int(rd['end_clip'] if fwd else rd['start_clip']),
            'num_insertions': nins, 'num_deletions': ndel, 'num_mismatches': nmis,
            'num_matches': int(len(bmi) - nmis - nins - ndel),
            'pred_mod_num': int(pred_mod_num), 'f5file': src_file, 'readk': rd['readk']}
        return key

    def close(self):
        if self.n == 0:
            return
        arrays = {'format': np.array(2), 'row_off': np.array(self.row_off, np.int64)}
        for k, parts in self.cols.items():
            arrays[k] = np.concatenate(parts)
        arrays['attrs'] = np.array(json.dumps(self.attrs))
        os.makedirs(os.path.dirname(self.path), exist_ok=True)
        savez_fast(self.path, arrays)


_cache = {'path': None, 'store': None}


def load_pred_store(path: str) -> Dict:
    """One prediction store, inflated: {'format', 'attrs', and for format 2 'row_off' + the five columns; format 1 (per-read
    members `pred_<i>/<column>`): 'z', the open archive}."""
    z = np.load(path, allow_pickle=False)
    attrs = json.loads(str(z['attrs']))
    if 'format' not in z.files:
        return {'format': 1, 'attrs': attrs, 'z': z}
    return {'format': int(z['format']), 'attrs': attrs, 'row_off': z['row_off'], 'refbase': z['refbase'], 'readbase': z['readbase'],
            'refbasei': z['refbasei'], 'readbasei': z['readbasei'], 'mod_pred': z['mod_pred']}


def pred_rows(store: Dict, key: str):
    """(first, last + 1) of read `key` in the columns of a format-2 store."""
    i = int(key[5:])
    return int(store['row_off'][i]), int(store['row_off'][i + 1])


def read_pred(path: str, key: str):
    """-> (m_pred, mapped_chr, mapped_strand) with the dtype the reference builds at myDetect.py:1022."""
    if _cache['path'] != path:
        _cache.update(path=path, store=load_pred_store(path))
    st = _cache['store']
    attrs = st['attrs']
    if st['format'] == 1:
        z = st['z']
        m_pred = make_base_map_info(s1_to_u1(z[key + '/refbase']), s1_to_u1(z[key + '/readbase']),
                                    z[key + '/refbasei'], z[key + '/readbasei'], z[key + '/mod_pred'])
    else:
        lo, hi = pred_rows(st, key)
        m_pred = make_base_map_info(s1_to_u1(st['refbase'][lo:hi]), s1_to_u1(st['readbase'][lo:hi]), st['refbasei'][lo:hi],
                                    st['readbasei'][lo:hi], st['mod_pred'][lo:hi])
    return m_pred, attrs[key]['mapped_chr'], attrs[key]['mapped_strand']
