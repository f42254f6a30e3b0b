"""Raw-read containers and the signal -> event-table stage.

The reference reads one FAST5 (HDF5) per read: raw DAC samples `Raw/Reads/*/Signal` and the basecaller's event
table `Analyses/Basecall_1D_*/BaseCalled_template/Events`.  Neither h5py nor libhdf5 exists in this image, so the
same arrays travel in `.dmraw.npz` containers (several reads per file); everything downstream of the HDF5 read
follows the reference:

  getEvent            myDetect.py:237-251  (Albacore 2.x, SignalGroup 'simple': stay events (move == 0) are
                                           merged into the preceding event; mean / stdv rounded to 3 decimals)
  mnormalized + stats myDetect.py:266-282, :332-343  -> deepmod_amd.signal (GPU)
  get_Event_Signals   myDetect.py:348-386  -> f5data[read_id] = (basecall, m_event, raw, file, (0, 0))

Not built: the Albacore 1.x timing arithmetic (:163-232) and the `EventTable` re-segmentation (`SignalGroup != simple`).
"""
from __future__ import annotations

import json
from collections import defaultdict
from typing import Dict, List

import numpy as np

from . import signal as dm_signal

RAW_SUFFIX = '.dmraw.npz'
EVENT_DTYPE = [('mean', '<f4'), ('stdv', '<f4'), ('start', np.uint64), ('length', np.uint64), ('model_state', 'U5')]
EVENTS_DATA_DTYPE = [('mean', '<f8'), ('stdv', '<f8'), ('start', np.uint64), ('length', np.uint64),
                     ('model_state', 'U5'), ('move', np.int64)]


_EV_FIELDS = ('mean', 'stdv', 'start', 'length', 'model_state', 'move')


class EventColumns:
    """The basecaller's event table of one read as columns (views into the container's arrays): `ec['move']`, `len(ec)` -
    what getEvent needs of the reference's structured `events_data` array, without building one per read."""
    __slots__ = ('cols', 'n')

    def __init__(self, cols: Dict[str, np.ndarray]):
        self.cols = cols
        self.n = len(cols['start'])

    def __getitem__(self, name: str) -> np.ndarray:
        return self.cols[name]

    def __len__(self) -> int:
        return self.n


def save_raw_container(path: str, reads: List[Dict]) -> None:
    """reads: dicts with read_id, raw (int16), events_data (EVENTS_DATA_DTYPE or EventColumns).
    Layout (format 2): the samples / event columns of all reads concatenated + offsets - nine arrays per container instead of
    seven per read (the zip directory walk and per-member headers were 39 % of a feeder's time), uncompressed (inflating the
    samples was 40 % before that)."""
    if not path.endswith(RAW_SUFFIX):
        raise ValueError('raw containers must end with ' + RAW_SUFFIX)
    off = lambda parts: np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    raws = [np.asarray(rd['raw'], dtype=np.int16) for rd in reads]
    arrays = {'format': np.array(2), 'raw': np.concatenate(raws) if raws else np.zeros(0, np.int16), 'raw_off': off(raws),
              'ev_off': off([rd['events_data']['start'] for rd in reads]),
              'meta': np.array(json.dumps([{'read_id': rd['read_id']} for rd in reads]))}
    dtypes = dict(EVENTS_DATA_DTYPE)
    for f in _EV_FIELDS:
        cols = [np.asarray(rd['events_data'][f], dtype=dtypes[f]) for rd in reads]
        arrays['ev_' + f] = np.concatenate(cols) if cols else np.zeros(0, dtypes[f])
    from . import npzmap
    with open(path, 'wb') as fh:
        npzmap.savez_aligned(fh, **arrays)          # members 64-byte aligned in the file: load_raw_container's views are aligned


def load_raw_container(path: str) -> List[Dict]:
    """-> read dicts; the arrays are read-only views into a mapping of the file (deepmod_amd/npzmap.py)."""
    from . import npzmap
    z = npzmap.load(path)
    metas = json.loads(str(z['meta']))
    if 'format' not in z:                       # format 1: seven arrays per read
        reads = []
        for i, m in enumerate(metas):
            reads.append({'read_id': m['read_id'], 'raw': z['r%d_raw' % i],
                          'events_data': EventColumns({f: z['r%d_ev_%s' % (i, f)] for f in _EV_FIELDS})})
        return reads
    raw, ro, eo = z['raw'], z['raw_off'], z['ev_off']
    cols = {f: z['ev_' + f] for f in _EV_FIELDS}
    return [{'read_id': m['read_id'], 'raw': raw[ro[i]:ro[i + 1]],
             'events_data': EventColumns({f: c[eo[i]:eo[i + 1]] for f, c in cols.items()})} for i, m in enumerate(metas)]


def event_bases(model_state) -> np.ndarray:
    """model_state[2] of every event (the basecalled base of a 5-mer state) as a 'U1' array, without a Python loop."""
    ms = np.ascontiguousarray(model_state)
    if len(ms) == 0:
        return np.zeros(0, 'U1')
    width = ms.dtype.itemsize // 4
    return ms.view('U1').reshape(len(ms), width)[:, 2]


def getEvent(moptions, sp_param):
    """Albacore-2 'simple' branch of the reference's getEvent (myDetect.py:237-251)."""
    events_data = sp_param['events_data']
    if moptions.get('SignalGroup', 'simple') != 'simple':
        raise NotImplementedError("SignalGroup %r (EventTable re-segmentation) is not built" % moptions.get('SignalGroup'))
    n = len(events_data)
    if n == 0:
        sp_param['f5status'] = 'No events data'
        return
    move = np.asarray(events_data['move'])
    heads = np.flatnonzero(np.r_[True, move[1:] > 0])            # an event starts where move > 0 (and at index 0)
    seg_len = np.add.reduceat(events_data['length'].astype(np.uint64), heads)
    m_event = np.zeros(len(heads), dtype=EVENT_DTYPE)
    m_event['mean'] = np.round(events_data['mean'][heads], 3)
    m_event['stdv'] = np.round(events_data['stdv'][heads], 3)
    m_event['start'] = events_data['start'][heads]
    m_event['length'] = seg_len
    m_event['model_state'] = events_data['model_state'][heads]
    sp_param['m_event'] = m_event
    sp_param['m_event_basecall'] = ''.join(event_bases(m_event['model_state']).tolist())
    sp_param['left_right_skip'] = (0, 0)


def get_Event_Signals(moptions, sp_options, raw_files, normalizer=None):
    """-> f5data {read_id: (basecall, m_event, raw_signals, file, left_right_skip)}   (myDetect.py:348-386)"""
    f5data = {}
    if "Error" not in sp_options:
        sp_options["Error"] = defaultdict(list)
    # events of every read of every container of the batch first (host), then ONE device round trip for their signal
    # statistics (a feeder process shares the GPU with the classifier: few, larger launches)
    pending = []
    for f5f in raw_files:
        try:
            reads = load_raw_container(f5f)
        except Exception:
            sp_options["Error"]["Cannot open fast5 or other errors"].append(f5f)
            print("Cannot open fast5 or other errors: {}".format(f5f))
            continue
        for rd in reads:
            sp_param = {'mfile_path': f5f, 'f5status': '', 'raw_signals': rd['raw'], 'events_data': rd['events_data'],
                        'read_id': rd['read_id'].replace(" ", ":::").replace("\t", "|||")}
            try:
                getEvent(moptions, sp_param)
            except Exception as exc:
                sp_param['f5status'] = "Cannot open fast5 or other errors"
                print("Cannot open fast5 or other errors: {} ({})".format(f5f, exc))
            pending.append(sp_param)
    ok = [sp for sp in pending if sp['f5status'] == '']
    for sp, exc in zip(ok, dm_signal.mnormalized_event_stats_batch(moptions, ok, normalizer) if ok else []):
        if exc is not None:
            sp['f5status'] = "Cannot open fast5 or other errors"
            print("Cannot open fast5 or other errors: {} ({})".format(sp['mfile_path'], exc))
    for sp_param in pending:
        f5f = sp_param['mfile_path']
        if sp_param['f5status'] == '':
            if sp_param['read_id'] in f5data:
                print('Duplicate id', sp_param['read_id'], f5f)
            f5data[sp_param['read_id']] = (sp_param['m_event_basecall'], sp_param['m_event'], None, f5f,
                                           sp_param['left_right_skip'])
        else:
            sp_options["Error"][sp_param['f5status']].append(f5f)
    return f5data
