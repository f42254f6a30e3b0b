"""CPU: host-side logic (windowing, batching, class->base scatter, flags, BED bytes) against golden
vectors recorded from the reference's own Python, and against the loop-level oracle."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from deepmod_amd import detect, predstore, summary
from oracle import detect_oracle, oracle_np


def fake_rule(x):
    x = np.asarray(x)
    return ((x[:, 10, 1] > 0.5) & (x[:, 10, 4] > 0.0)).astype(np.int64)


class FakeSession:
    """Any session-like object: mPredict1 must drive it exactly as the reference does."""
    def __init__(self):
        self.batches, self.inits, self.dtypes = [], 0, []

    def run(self, fetches, feed_dict=None):
        if feed_dict is None:
            self.inits += 1
            return None
        assert fetches == ['mfpred']
        x, y = feed_dict['X'], feed_dict['Y']
        assert y.shape == (len(x), 2) and y.dtype.kind == 'i' and not y.any()
        self.batches.append(len(x))
        self.dtypes.append(x.dtype)
        return [fake_rule(x)]


G = np.load(os.path.join(GOLDEN, 'host_mpredict1.npz'))


def _case(ci):
    ev = predstore.events_from_bases([s[2] for s in G['c%d_model_state' % ci]])
    assert (ev['model_state'] == G['c%d_model_state' % ci]).all()
    bmi = predstore.make_base_map_info(G['c%d_bmi_refbase' % ci], G['c%d_bmi_readbase' % ci],
                                       G['c%d_bmi_refbasei' % ci], G['c%d_bmi_readbasei' % ci])
    sc, ec = [int(v) for v in G['c%d_clips' % ci]]
    return ev, bmi, sc, ec


@pytest.mark.parametrize('ci', range(int(G['n_cases'])))
def test_mpredict1_matches_reference_run(ci):
    ev, bmi, sc, ec = _case(ci)
    sess = FakeSession()
    sp_options = {'rnn': (sess, 'X', 'Y', 'init_l', 'mfpred')}
    sp_param = {'f5data': {'r': (None, ev, None, 'f')}}
    pred = detect.mPredict1({'windowsize': 21}, sp_options, sp_param, G['c%d_mfeatures' % ci].copy(), bmi, 'r', sc, ec)
    assert sess.inits == 1
    assert sess.batches == G['c%d_batches' % ci].tolist()           # the ~512 split rule (myDetect.py:808-812)
    assert all(dt == np.float64 for dt in sess.dtypes)                # float64 feed, as the reference
    assert pred == int(G['c%d_pred_mod_num' % ci])
    assert np.array_equal(bmi['mod_pred'].astype(np.int64), G['c%d_mod_pred' % ci])


@pytest.mark.parametrize('ci', [0, 4, 6])
def test_loop_oracle_matches_reference_run(ci):
    ev, bmi, sc, ec = _case(ci)
    ev_bases = [s[2] for s in G['c%d_model_state' % ci]]
    batches, pred, mod_pred = detect_oracle.mpredict1_oracle(G['c%d_mfeatures' % ci], list(bmi['readbase']), ev_bases,
                                                            sc, ec, fake_rule)
    assert batches == G['c%d_batches' % ci].tolist()
    assert pred == int(G['c%d_pred_mod_num' % ci])
    assert np.array_equal(mod_pred, G['c%d_mod_pred' % ci])


def test_mpredict1_zero_aligned_events_returns_zero():
    ev = predstore.events_from_bases(list('ACGT'))
    bmi = predstore.make_base_map_info(['A'], ['A'], [1])
    assert detect.mPredict1({'windowsize': 21}, {'rnn': (FakeSession(), 'X', 'Y', 'i', 'mfpred')},
                            {'f5data': {'r': (None, ev, None, 'f')}}, np.zeros((200, 10)), bmi, 'r', 2, 2) == 0


SUMS = json.load(open(os.path.join(GOLDEN, 'host_sum_handler.json')))


def _tables(case):
    return [predstore.make_base_map_info(list(r['refbase']), list(r['readbase']), r['refbasei'], None, r['mod_pred'])
            for r in case['reads']]


@pytest.mark.parametrize('case', SUMS, ids=[c['strand'] for c in SUMS])
def test_bed_bytes_match_reference_sum_handler(case):
    """flags -> counters (oracle C restatement of the accumulation) -> bed_lines == the reference's bytes."""
    length = 1 + max(max(r['refbasei']) for r in case['reads'])
    touch = np.zeros(length, np.int32); cov = np.zeros(length, np.int32); mod = np.zeros(length, np.int32)
    for m_pred in _tables(case):
        fl = detect.base_flags(m_pred, case['Base'])
        oracle_np.summary_add_c(touch, cov, mod, m_pred['refbasei'].astype(np.int64), fl)
    bed = summary.bed_lines(case['chr'], case['strand'], case['Base'], touch, cov, mod)
    assert bed == case['bed'].encode('ascii')
    assert b' 1000 ' in bed and b' 1203 ' in bed          # column-5 cap vs real coverage
    assert bed.startswith(b'chrS 40 41 C 0 ')             # deletion-only position is emitted with cov 0
    # loop-level oracle agrees as well
    assert detect_oracle.sum_handler_oracle(case['chr'], case['strand'], case['Base'], case['reads']) == bed


def test_feature_container_roundtrip(tmp_path):
    ev, bmi, sc, ec = _case(0)
    rd = {'readk': 'read0', 'chr': 'chrS', 'strand': '+', 'mapped_start': 1000, 'start_clip': sc, 'end_clip': ec,
          'mfeatures': G['c0_mfeatures'], 'base_map_info': bmi, 'events': ev}
    path = str(tmp_path / ('a' + predstore.CONTAINER_SUFFIX))
    predstore.save_feature_container(path, [rd, rd])
    back = predstore.load_feature_container(path)
    assert len(back) == 2
    assert np.array_equal(back[1]['mfeatures'], rd['mfeatures'])
    for f in ('refbase', 'readbase', 'refbasei', 'readbasei'):
        assert np.array_equal(back[0]['base_map_info'][f], bmi[f])
    assert (back[0]['events']['model_state'] == ev['model_state']).all()


def test_pred_store_roundtrip(tmp_path):
    ev, bmi, sc, ec = _case(1)
    bmi['mod_pred'][::7] = 1
    rd = {'readk': 'read0', 'chr': 'chrS', 'strand': '-', 'mapped_start': 1000, 'start_clip': sc, 'end_clip': ec}
    mo = {'outFolder': str(tmp_path) + '/', 'FileID': 'mod'}
    w = predstore.PredWriter(str(tmp_path / 'mod' / '0'), 3)
    key = w.add(rd, bmi, 17, 'x.dmfeat.npz', mo)
    w.close()
    assert w.relpath(mo) == '0/rnn.pred.detail.npz.3'
    m_pred, c, s = predstore.read_pred(w.path, key)
    assert (c, s) == ('chrS', '-')
    for f in ('refbase', 'readbase', 'refbasei', 'mod_pred'):
        assert np.array_equal(m_pred[f], bmi[f])


def test_sum_chr_mod_matches_reference_tool(tmp_path):
    """BED merge across runs == output of the reference's DeepMod_tools/sum_chr_mod.py (tests/golden/make_golden_merge.py)."""
    import json
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    g = json.load(open(os.path.join(GOLDEN, "merge_case.json")))
    for rel, text in g['inputs'].items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "DeepMod_tools", "sum_chr_mod.py"), str(tmp_path)] + g['argv'],
                          stdout=subprocess.DEVNULL)
    for fn, text in g['outputs'].items():
        assert (tmp_path / fn).read_text() == text, fn


def test_generate_motif_pos_matches_reference_tool(tmp_path):
    """na_/motif_ position files == output of the reference's DeepMod_tools/generate_motif_pos.py."""
    import json
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    g = json.load(open(os.path.join(GOLDEN, "motif_case.json")))
    (tmp_path / "g.fa").write_text(g['fasta'])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "DeepMod_tools", "generate_motif_pos.py"), str(tmp_path / "g.fa"),
                           str(tmp_path / "out")] + g['argv'], stdout=subprocess.DEVNULL)
    assert sorted(os.listdir(tmp_path / "out")) == sorted(g['outputs'])
    for fn, text in g['outputs'].items():
        assert (tmp_path / "out" / fn).read_text() == text, fn


def test_c_bed_writer_equals_line_by_line_restatement(hip_lib):
    """dm_bed_format (host C) against the Python restatement of the reference's writer (myDetect.py:1112-1120), and both
    against the bytes the reference's own sum_handler produced (host_sum_handler.json)."""
    import json
    from deepmod_amd import summary
    rng = np.random.default_rng(7)
    n = 200_000
    touch = ((rng.random(n) < 0.3) * rng.integers(1, 2000, n)).astype(np.int32)
    cov = np.where(touch > 0, rng.integers(0, 2500, n), 0).astype(np.int32)
    cov[rng.random(n) < 0.02] = 0                                   # deletion-only positions: cov = 0 lines
    mod = np.minimum(cov, rng.integers(0, 2500, n)).astype(np.int32)
    for chrom in ("chr1", "NC_000913.3", "x"):
        assert summary.bed_lines(chrom, "+", "C", touch, cov, mod) == summary.bed_lines_py(chrom, "+", "C", touch, cov, mod)
    z = np.zeros(10, np.int32)
    assert summary.bed_lines("c", "-", "A", z, z, z) == b""
    one = np.array([0, 3, 0], np.int32)
    assert summary.bed_lines("c", "-", "A", one, np.array([0, 1001, 0], np.int32), np.array([0, 7, 0], np.int32)) == \
        b"c 1 2 A 1000 - 1 2 0,0,0 1001 0 7 \n"
    case = json.load(open(os.path.join(GOLDEN, "host_sum_handler.json")))[0]
    length = 1 + max(max(r["refbasei"]) for r in case["reads"])
    counts = [np.zeros(length, np.int32) for _ in range(3)]
    for r in case["reads"]:
        for rb, qb, pos, mp in zip(r["refbase"], r["readbase"], r["refbasei"], r["mod_pred"]):
            if rb == case["Base"]:
                counts[0][pos] += 1
                if qb != '-':
                    counts[1][pos] += 1
                    counts[2][pos] += int(mp == 1)
    assert summary.bed_lines(case["chr"], case["strand"], case["Base"], *counts) == case["bed"].encode("ascii")


def test_npzmap_views_equal_numpy_load(tmp_path):
    """deepmod_amd/npzmap.py: zero-copy views of stored .npz members == numpy.load, compressed members fall back."""
    from deepmod_amd import npzmap
    rng = np.random.default_rng(3)
    arrays = {'f32': rng.normal(size=(1000, 7)).astype(np.float32), 'i64': rng.integers(0, 1 << 40, 333), 'empty': np.zeros(0, np.int16),
              'u5': np.array(['ACGTA', 'TTTTT', 'GGCAA']), 's1': np.frombuffer(b'ACGT-', 'S1'), 'scalar': np.array(2),
              'meta': np.array('{"reads": [1, 2]}'), 'fortran': np.asfortranarray(rng.normal(size=(5, 4)))}
    for name, saver in (('aligned.npz', npzmap.savez_aligned), ('stored.npz', np.savez), ('deflated.npz', np.savez_compressed)):
        path = str(tmp_path / name)
        saver(path, **arrays)
        got = npzmap.load(path)
        assert sorted(got) == sorted(arrays)
        for k, v in arrays.items():
            assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), (name, k)
            assert got[k].ctypes.data % max(v.dtype.alignment, 1) == 0, (name, k)      # never a misaligned pointer for the C ABI
        assert str(got['meta']) == str(arrays['meta']) and int(got['scalar']) == 2
    # the package's own writer aligns every member to 64 bytes: zero-copy views into a read-only mapping (numpy.savez members
    # start wherever the zip header ends: those are copied when misaligned); numpy.load reads the aligned file as well
    aligned = npzmap.load(str(tmp_path / 'aligned.npz'))
    for k in ('f32', 'i64', 'fortran'):
        assert not aligned[k].flags.writeable and aligned[k].ctypes.data % 64 == 0, k
    z = np.load(str(tmp_path / 'aligned.npz'))
    assert all(np.array_equal(z[k], v) for k, v in arrays.items())
    # lazy members: the same views, their pages left to the first touch; a kernel without MADV_POPULATE_READ maps whole files as before
    for name in ('aligned.npz', 'stored.npz', 'deflated.npz'):
        lazy = npzmap.load(str(tmp_path / name), lazy=('f32', 'fortran', 'not a member'))
        assert sorted(lazy) == sorted(arrays) and all(np.array_equal(lazy[k], v) for k, v in arrays.items())
    import mmap
    assert npzmap._populate(mmap.mmap(-1, 1 << 16), [(100, 5000), (4000, 9000), (70000, 60000), (60000, 1 << 16)]) in (True, False)    # overlapping, empty, up to the end
    ok = npzmap._populate_ranges_ok[0]
    try:
        npzmap._populate_ranges_ok[0] = False
        whole = npzmap.load(str(tmp_path / 'aligned.npz'), lazy=('f32',))
        assert all(np.array_equal(whole[k], v) for k, v in arrays.items())
    finally:
        npzmap._populate_ranges_ok[0] = ok


def test_stored_chunks_equal_the_per_read_tables(tmp_path):
    """detect.stored_chunks (five members per prediction store, flags of all its reads in one pass over bytes) hands summarize_tables
    the rows base_flags finds in the tables read_pred_detail builds read by read - for format-2 stores and for the per-read
    members of format 1."""
    from deepmod_amd import detect
    rng = np.random.default_rng(3)
    mo = {'outFolder': str(tmp_path) + '/', 'FileID': 'mod', 'Base': 'C'}
    records, tables = [], {}
    for batch in range(3):
        w = predstore.PredWriter(str(tmp_path / 'mod' / '0'), batch)
        old = {}
        for r in range(5):
            n = int(rng.integers(50, 400))
            strand = '+-'[int(rng.integers(0, 2))]
            bmi = predstore.make_base_map_info(rng.choice(list('ACGT-N'), n, p=[.23, .23, .23, .23, .04, .04]), rng.choice(list('ACGT-'), n),
                                               np.sort(rng.integers(0, 5000, n)).astype(np.uint64), np.arange(n, dtype=np.uint64),
                                               (rng.random(n) < 0.3).astype(int))
            rd = {'readk': 'read%d_%d' % (batch, r), 'chr': 'chrS', 'strand': strand, 'mapped_start': 0, 'start_clip': 1, 'end_clip': 2}
            key = w.add(rd, bmi, int(bmi['mod_pred'].sum()), 'x.dmfeat.npz', mo)
            records.append(['chrS', strand, '0', key, 'x.dmfeat.npz', w.relpath(mo)])
            tables[(w.relpath(mo), key)] = bmi
            for f in ('refbase', 'readbase'):
                old[key + '/' + f] = predstore.u1_to_s1(bmi[f])
            old[key + '/refbasei'], old[key + '/readbasei'], old[key + '/mod_pred'] = bmi['refbasei'], bmi['readbasei'], bmi['mod_pred'].astype(np.int64)
        w.close()
        if batch == 1:              # the middle store in the per-read layout of format 1
            old['attrs'] = np.array(json.dumps(w.attrs))
            predstore.savez_fast(w.path, old)
            assert predstore.load_pred_store(w.path)['format'] == 1
        else:
            assert predstore.load_pred_store(w.path)['format'] == 2
    for strand in '+-':
        sp = {'base_folder_output': str(tmp_path / 'mod') + '/', 'handlingList': [r for r in records if r[1] == strand]}
        want = np.zeros((3, 5000), np.int64)
        for r in sp['handlingList']:
            bmi = tables[(r[5], r[3])]
            fl = detect.base_flags(bmi, 'C')
            for k in range(3):
                np.add.at(want[k], bmi['refbasei'][(fl & 1) != 0].astype(np.int64), ((fl[(fl & 1) != 0] >> k) & 1))
        got = np.zeros((3, 5000), np.int64)
        for chunk in detect.stored_chunks(mo, sp, 'chrS', strand, 'C', readers=2):
            if isinstance(chunk, tuple):
                p, fl = chunk
            else:
                fl = detect.base_flags(chunk, 'C')
                p, fl = chunk['refbasei'][(fl & 1) != 0].astype(np.int64), fl[(fl & 1) != 0]
            assert (fl & 1).all()
            for k in range(3):
                np.add.at(got[k], p, (fl >> k) & 1)
        assert np.array_equal(got, want) and want[0].sum() > 0


def test_bed_parts_join_to_bed_lines():
    """summary.bed_parts (slices of the table formatted by several threads, in position order) joined = summary.bed_lines = the
    line-by-line restatement of the reference's writer - also for a slice that starts at a later position (a rank's share)."""
    from deepmod_amd import _lib, summary
    try:
        _lib.load()
    except _lib.DeepModHipError:
        pytest.skip("library not built")
    rng = np.random.default_rng(8)
    n = 50000
    touch = ((rng.random(n) < 0.3) * rng.integers(1, 4, n)).astype(np.int32)
    touch[12000:30000] = 0                                   # slices without a line
    cov = rng.integers(0, 1500, n).astype(np.int32)
    mod = np.minimum(cov, rng.integers(0, 1500, n)).astype(np.int32)
    whole = summary.bed_lines('chrQ', '-', 'C', touch, cov, mod)
    assert whole == summary.bed_lines_py('chrQ', '-', 'C', touch, cov, mod)
    for sl, th in ((4096, 8), (7777, 3), (1 << 22, 8), (50000, 1)):
        assert b''.join(p.tobytes() for p in summary.bed_parts('chrQ', '-', 'C', touch, cov, mod, slice_positions=sl, threads=th)) == whole
    shifted = summary.bed_lines('chrQ', '-', 'C', touch, cov, mod, first_pos=10 ** 9)
    assert b''.join(p.tobytes() for p in summary.bed_parts('chrQ', '-', 'C', touch, cov, mod, first_pos=10 ** 9, slice_positions=999, threads=4)) == shifted
    assert list(summary.bed_parts('chrQ', '+', 'C', np.zeros(100, np.int32), cov[:100], mod[:100])) == []


def test_feeder_budget_of_a_streaming_run():
    """detect.feeder_budget: --threads is the total over the ranks; a rank's feeder processes fit into the CPUs the
    job may use after every rank's own process has one.  The driver's box (16 CPUs, 8 GPUs) leaves ONE feeder per GPU: enough for packed
    containers, a warning for raw ones."""
    from deepmod_amd.detect import feeder_budget
    assert feeder_budget(threads=32, world=8, usable_cpus=16, raw_input=False) == (4, 1, None)
    f, p, warn = feeder_budget(threads=32, world=8, usable_cpus=16, raw_input=True)
    assert (f, p) == (4, 1) and warn and "1 feeder process(es) per GPU" in warn and "16 usable CPUs" in warn
    assert feeder_budget(32, 8, 32, False) == (4, 3, None)
    assert feeder_budget(32, 8, 32, True)[1] == 3 and feeder_budget(32, 8, 32, True)[2]
    assert feeder_budget(64, 8, 256, True) == (8, 8, None)           # a full node: asked-for feeders fit, nothing to warn about
    assert feeder_budget(16, 8, 256, False) == (2, 2, None)
    assert feeder_budget(4, 1, 16, True) == (4, 4, None)
    assert feeder_budget(1, 1, 2, False) == (1, 1, None)
    assert feeder_budget(2, 8, 16, False) == (1, 1, None)            # fewer threads than ranks: still one per rank
    assert feeder_budget(8, 1, 1, False) == (8, 1, None)             # never below one


def test_gpu_count_from_the_kernel_drivers_topology_files(tmp_path, monkeypatch):
    """bin/DeepMod.py kfd_gpu_count: with --gpus N the command asks the KFD topology, not a HIP runtime of its own (0.14 s of a 1.8 s run):
    gfx950 nodes that can be read count, CPU nodes and nodes hidden by the device cgroup do not, a *_VISIBLE_DEVICES filter hands the question to the runtime,
    an unreadable topology is None (the caller then asks the runtime)."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location('dmcli', os.path.join(ROOT, 'bin', 'DeepMod.py'))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'GPU_DEVICE_ORDINAL'):
        monkeypatch.delenv(var, raising=False)
    assert cli.kfd_gpu_count(str(tmp_path / 'missing')) is None
    base = tmp_path / 'nodes'
    for i, props in enumerate(("simd_count 0\ngfx_target_version 0\n", "simd_count 1024\ngfx_target_version 90500\n", None,
                               "simd_count 1024\ngfx_target_version 90500\n", "simd_count 1216\ngfx_target_version 90402\n")):
        os.makedirs(base / str(i))
        if props is not None:                 # node 2: no readable properties file (what a device cgroup leaves of a GPU that is not ours)
            (base / str(i) / 'properties').write_text(props)
    assert cli.kfd_gpu_count(str(base)) == 2
    # any *_VISIBLE_DEVICES filter: the topology files cannot say what the runtime will show (filters compose, -1 truncates) -> None,
    # the caller asks the runtime
    for var, val in (('HIP_VISIBLE_DEVICES', '0'), ('ROCR_VISIBLE_DEVICES', '0,1,2'), ('CUDA_VISIBLE_DEVICES', '1,-1,0')):
        monkeypatch.setenv(var, val)
        assert cli.kfd_gpu_count(str(base)) is None
        monkeypatch.delenv(var)
    assert cli.kfd_gpu_count(str(base)) == 2
