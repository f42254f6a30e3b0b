"""GPU (-m gpu): the BASELINE.json configurations that earlier rounds only touched in parts.

  * configs[4]'s model on one GPU: `--Base A` (6mA) end to end through `bin/DeepMod.py detect`, with the model directory
    `rnn_conmodA_E1m2wd21_f7ne1u0_4/` rebuilt around its REAL .index and checkpoint files (tests/golden/model_dirs) and a synthetic
    .data shard of the real byte layout; BED bytes of both strands equal to the oracle pipeline's.
  * configs[0] exactly as SURVEY.md 8d writes it (100 reads, 100 kb genome NC_000913.3, seed 1, lengths U[2000, 10000], 6 / 2 / 2 %
    substitutions / insertions / deletions) with `rnn_conmodC_P100wd21_f7ne1u0_4`; plus the metric's second half: site-level
    AUC `roc_curve(label, pct)` at Coverage >= 1 | 5 (DeepMod_tools/cal_EcoliDetPerf.py:255-276) from the GPU BED and the oracle BED.
"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from deepmod_amd import siteperf, synth, synth_reads
from oracle_pipeline import oracle_beds

pytestmark = pytest.mark.gpu


def _model_dir(tmp_path, model, prefix, seed, scale):
    src = os.path.join(GOLDEN, 'model_dirs', model)
    dst = tmp_path / 'train_deepmod' / model
    os.makedirs(dst)
    shutil.copyfile(os.path.join(src, 'checkpoint'), dst / 'checkpoint')
    w = synth.write_synthetic_data_for_index(os.path.join(src, prefix + '.index'), str(dst / prefix), seed=seed, scale=scale)
    return str(dst / prefix), w


def _detect(wrk, modfile, out, fileid, base, extra=()):
    cmd = [sys.executable, os.path.join(ROOT, 'bin', 'DeepMod.py'), 'detect', '--wrkBase', str(wrk), '--modfile', modfile,
           '--outFolder', out, '--FileID', fileid, '--threads', '4', '--files_per_thread', '4', '--Base', base, '--gpus', '1'] + list(extra)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert os.path.exists('%s/%s.done' % (out, fileid))
    return res.stdout


def test_base_A_6mA_model_directory_end_to_end(tmp_path, gpu_device):
    modfile, w = _model_dir(tmp_path, 'rnn_conmodA_E1m2wd21_f7ne1u0_4', 'mod_train_conmodA_E1m2wd21_f3ne1u0', seed=26, scale=4.0)
    wrk = tmp_path / 'reads'
    files = synth_reads.write_synthetic_run(str(wrk), n_reads=30, reads_per_file=5, genome_len=25000, seed=13, chrom='chr6mA',       # min |p1 - 0.5| = 2.0e-4 on this read set
                                            min_len=400, max_len=1600)
    want, margin, nwin = oracle_beds(files, w, 'A')
    assert margin > 1e-4, 'near-tie window in the synthetic set (%.2e): pick another seed' % margin
    out = str(tmp_path / 'out')
    stdout = _detect(wrk, modfile, out, 'a_stream', 'A')                      # the default: streaming detect
    assert 'Streaming detect: 30 reads' in stdout
    _detect(wrk, modfile, out, 'a_stored', 'A', ['--storePred', '1'])        # the reference's file shape
    assert set(want) == {('chr6mA', '+'), ('chr6mA', '-')}
    for (chrom, strand), bed in want.items():
        assert len(bed) > 1000
        for fid in ('a_stream', 'a_stored'):
            got = open('%s/%s/mod_pos.%s%s.A.bed' % (out, fid, chrom, strand), 'rb').read()
            assert got == bed, (fid, strand)
        assert all(ln.split()[3] == 'A' for ln in bed.decode().splitlines())
    assert not [f for f in os.listdir(out + '/a_stream') if f.endswith('.C.bed')]


def test_config1_exact_generator_and_site_level_auc(tmp_path, gpu_device):
    """the generator's defaults ARE the specification (100 reads, 100,000 bp, seed 1, NC_000913.3)."""
    modfile, w = _model_dir(tmp_path, 'rnn_conmodC_P100wd21_f7ne1u0_4', 'mod_train_conmodC_P100wd21_f3ne1u0', seed=7, scale=4.0)
    wrk = tmp_path / 'reads'
    files = synth_reads.write_synthetic_run(str(wrk))                           # n_reads=100, genome_len=100000, seed=1, chrom='NC_000913.3'
    assert len(files) == 20
    want, margin, nwin = oracle_beds(files, w, 'C')
    assert nwin > 400_000
    out = str(tmp_path / 'out')
    stdout = _detect(wrk, modfile, out, 'cfg1', 'C')
    assert 'Streaming detect: 100 reads' in stdout
    # synthetic truth: every third C position of each strand is "methylated"
    for (chrom, strand), bed in want.items():
        assert chrom == 'NC_000913.3'
        got = open('%s/cfg1/mod_pos.%s%s.C.bed' % (out, chrom, strand), 'rb').read()
        if margin > 1e-4:
            assert got == bed                                                   # byte-identical away from near ties
        sites = siteperf.bed_sites(bed)
        truth = sites['pos'][::3]
        auc_gpu = siteperf.site_level_auc(got, truth)
        auc_ref = siteperf.site_level_auc(bed, truth)
        assert set(auc_gpu) == {1, 5}
        for k in (1, 5):
            assert np.isfinite(auc_ref[k]) and abs(auc_gpu[k] - auc_ref[k]) <= (0.0 if margin > 1e-4 else 1e-3), (k, auc_gpu, auc_ref)
        # the module against sklearn's roc_curve / auc on the same sites (the reference's calls, cal_EcoliDetPerf.py:268-270)
        from sklearn.metrics import auc, roc_curve
        sel = sites['cov'] >= 5
        fpr, tpr, _ = roc_curve(np.isin(sites['pos'], truth)[sel], sites['pct'][sel])
        assert abs(auc(fpr, tpr) - auc_ref[5]) < 1e-12


def test_trained_like_model_through_the_command_default_and_auto(tmp_path, gpu_device):
    """the only model with TRAINED weight statistics in the tree (tests/golden/trained_like_weights.npz) as a model
    directory through `bin/DeepMod.py detect`, on reads with read-shaped extremes (3 % of the events: normalised means anywhere in the clip
    range incl. exactly +-5, lengths up to 30,000 samples) - once with the command's defaults (DM_PREC_F16X3) and once with
    DEEPMOD_PRECISION=auto, where the load-time gate lets the int8 cross-term mode in for THIS model.  Both BED files of both runs equal
    the oracle pipeline's (reference myDetect.py:787-834, :1089-1120) byte for byte: the smallest |p1 - 0.5| of the read set is 2.0e-4,
    so a mode that moved any probability by 1e-4 could still not be told from one that flipped a class - the int8 mode's per-window error
    on the same rows is held to 1e-4 in test_gpu_parity.py::test_selected_mode_on_read_shaped_rows."""
    from conftest import trained_like_weights
    from deepmod_amd import tfbundle
    w = trained_like_weights()
    prefix = str(tmp_path / 'model' / 'mod_train_trained_like')
    os.makedirs(os.path.dirname(prefix))
    tfbundle.write_bundle(prefix, w, layout=synth.REAL_LAYOUT, total_size=synth.REAL_DATA_SIZE)
    wrk = tmp_path / 'reads'
    files = synth_reads.write_synthetic_run(str(wrk), n_reads=24, reads_per_file=4, genome_len=25000, seed=33, chrom='chrT', min_len=400,
                                            max_len=1600, p_tail=0.03)
    want, margin, nwin = oracle_beds(files, w, 'C')
    assert margin > 1e-4 and nwin > 20000, (margin, nwin)          # 2.03e-4 on this read set
    out = str(tmp_path / 'out')
    assert os.environ.get('DEEPMOD_PRECISION') is None
    stdout = _detect(wrk, prefix, out, 'dflt', 'C')
    assert 'Streaming detect: 24 reads' in stdout
    os.environ['DEEPMOD_PRECISION'] = 'auto'
    try:
        cmd = [sys.executable, os.path.join(ROOT, 'bin', 'DeepMod.py'), 'detect', '--wrkBase', str(wrk), '--modfile', prefix, '--outFolder', out,
               '--FileID', 'auto', '--threads', '2', '--Base', 'C', '--gpus', '1']
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    finally:
        os.environ.pop('DEEPMOD_PRECISION', None)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert 'DEEPMOD_PRECISION=auto' in res.stderr and '-> f16i8' in res.stderr, res.stderr[-2000:]      # the gate let the int8 mode in, and said so
    for (chrom, strand), bed in want.items():
        assert len(bed) > 100000
        for fid in ('dflt', 'auto'):
            got = open('%s/%s/mod_pos.%s%s.C.bed' % (out, fid, chrom, strand), 'rb').read()
            assert got == bed, (fid, strand)
