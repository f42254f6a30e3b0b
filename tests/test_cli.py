"""CPU: CLI surface keeps the reference's `detect` flags and defaults (bin/DeepMod.py:304-338)."""
import importlib.util
import os

import pytest

from conftest import ROOT

spec = importlib.util.spec_from_file_location('deepmod_cli', os.path.join(ROOT, 'bin', 'DeepMod.py'))
cli = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cli)


def test_detect_defaults_match_reference():
    a = cli.build_parser().parse_args(['detect'])
    want = dict(outLevel=2, FileID='mod', outFolder='./mod_output', recursive=1, threads=4, files_per_thread=1000,
                windowsize=21, alignStr='minimap2', SignalGroup='simple', move=False, predDet=1, fnum=7, hidden=100,
                basecall_1d='Basecall_1D_000', basecall_2strand='BaseCalled_template', outputlayer='', Base='C', mod_cluster=0)
    for k, v in want.items():
        assert getattr(a, k) == v, k


def test_bad_inputs_are_reported(tmp_path):
    a = cli.build_parser().parse_args(['detect', '--wrkBase', str(tmp_path / 'nope'), '--modfile', 'nothing'])
    with pytest.raises(SystemExit) as e:
        cli.mDetect(a)
    assert '--wrkBase' in str(e.value) and '--modfile' in str(e.value)
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(['detect', '--Base', 'X'])
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(['train']).func(None)


def test_synthetic_reads_are_self_consistent(tmp_path):
    from deepmod_amd import predstore, synth_reads
    files = synth_reads.write_synthetic_run(str(tmp_path), n_reads=6, reads_per_file=4, genome_len=5000, seed=2,
                                            min_len=200, max_len=600)
    assert len(files) == 2
    seen = set()
    for f in files:
        for rd in predstore.load_feature_container(f):
            bmi, ev = rd['base_map_info'], rd['events']
            n = len(ev) - rd['start_clip'] - rd['end_clip']
            assert (bmi['readbase'] != '-').sum() == n
            assert rd['mfeatures'].shape == (n + 200, 10)
            assert not rd['mfeatures'][:100 - rd['start_clip']].any()      # zero padding rows
            seen.add(rd['strand'])
    assert seen == {'+', '-'}


def test_input_listing_equals_the_glob_patterns_of_the_manager(tmp_path):
    """detect.discover_inputs scans each folder once; what it returns is what the reference's patterns select (`<wrkBase>/*.ext`, then one, two and
    three folders down with --recursive 1, myDetect.py:1143-1158): no hidden entries, folders behind symbolic links followed, sorted; the sizes it
    collects on the way give the same work items as asking the file system again."""
    import glob
    from deepmod_amd import detect, predstore, rawreads
    root = str(tmp_path / 'wrk')
    for d in ['', 'a', 'a/b', 'a/b/c', 'a/b/c/d', 'e']:
        os.makedirs(os.path.join(root, d), exist_ok=True)
        for i, n in enumerate(['x' + predstore.CONTAINER_SUFFIX, '.hidden' + rawreads.RAW_SUFFIX, 'y' + rawreads.RAW_SUFFIX, 'z.txt']):
            with open(os.path.join(root, d, n), 'w') as fh:
                fh.write('q' * (100 * (i + 1) + len(d)))
    os.symlink(os.path.join(root, 'a', 'b'), os.path.join(root, 'lnk'))
    os.symlink(os.path.join(root, 'a', 'b'), os.path.join(root, '.hidden_lnk'))
    os.symlink(os.path.join(root, 'nowhere'), os.path.join(root, 'dangling' + rawreads.RAW_SUFFIX))
    for recursive in (False, True):
        want = []
        for suffix in (predstore.CONTAINER_SUFFIX, rawreads.RAW_SUFFIX):
            for lv in (['', '*/', '*/*/', '*/*/*/'] if recursive else ['']):
                want.extend(glob.glob(os.path.join(root, lv + '*' + suffix)))
        sizes = {}
        got = detect.discover_inputs(root, recursive, sizes)
        dangling = os.path.join(root, 'dangling' + rawreads.RAW_SUFFIX)     # listed by both; the feeder reports it as an unreadable input
        assert got == sorted(want) and dangling in got and dangling not in sizes
        assert all(sizes[f] == os.path.getsize(f) for f in want if f != dangling)
        assert detect.plan_batches_sized(got, 3, 450, sizes) == detect.plan_batches_sized(got, 3, 450)
    assert detect.discover_inputs(str(tmp_path / 'nope'), True) == []
