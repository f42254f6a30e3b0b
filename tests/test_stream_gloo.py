"""CPU, world_size 2 over gloo: the streaming detect engine (deepmod_amd/stream.py: batch preparation, grouping by
contig x strand, counter bookkeeping, key / length agreement across ranks, rank-0 BED) with a stand-in for the two
things a CPU box lacks - the device calls (oracle classifier + numpy counters instead of libdeepmod_hip) and the
transport (gloo instead of RCCL).  The merged BED files must equal the single-process result and the oracle
pipeline's bytes (the additive merge of DeepMod_tools/sum_chr_mod.py:47-52 / SURVEY.md 8e)."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["DM_ROOT"])
sys.path.insert(0, os.path.join(os.environ["DM_ROOT"], "tests"))
from deepmod_amd import comm, stream, synth
from cpu_backend import OracleBackend

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 1:
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
cfg = json.load(open(os.environ["DM_CFG"]))
mo = {"Base": "C", "outFolder": cfg["out"], "fnum": 7, "hidden": 100, "windowsize": 21}
os.makedirs(cfg["out"], exist_ok=True)
backend = OracleBackend(synth.synthetic_weights(cfg["seed"], cfg["scale"]))
eng = stream.StreamEngine(mo, backend, rank, world)
items = [cfg["files"][i:i + 2] for i in range(0, len(cfg["files"]), 2)]
eng.run(comm.shard(items, rank, world), feeders=2)

def gather(obj):
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out

def scatter_fn(s):                     # transport stand-in for ncclReduceScatter: gloo all-reduce of the host counters, then the slice rule
    t = torch.from_numpy(s.counts)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return s.take_slice(rank, world)

def reduce_fn(s):                      # the rank-0 form of the merge (dm_summary_reduce)
    t = torch.from_numpy(s.counts)
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)

if cfg.get("merge") == "reduce":
    eng.finalize(gather if world > 1 else None, None, reduce_fn=reduce_fn if world > 1 else None)
else:
    eng.finalize(gather if world > 1 else None, scatter_fn if world > 1 else None)
json.dump({"rank": rank, "reads": eng.stats["reads"], "windows": eng.stats["windows"], "errors": dict(eng.errors)},
          open(os.path.join(cfg["out"], "stats.%d.json" % rank), "w"))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _oracle_beds(files, w):
    from deepmod_amd import predstore
    from oracle import detect_oracle, oracle_np
    classify = lambda x: oracle_np.predict_windows_c(w, np.asarray(x, np.float32))[1]
    by = {}
    for f in files:
        for rd in predstore.load_feature_container(f):
            bmi = rd['base_map_info']
            ev_bases = [s[2] for s in rd['events']['model_state']]
            _, _, mod_pred = detect_oracle.mpredict1_oracle(rd['mfeatures'], list(bmi['readbase']), ev_bases,
                                                            rd['start_clip'], rd['end_clip'], classify)
            by.setdefault((rd['chr'], rd['strand']), []).append(
                {'refbase': ''.join(bmi['refbase']), 'readbase': ''.join(bmi['readbase']),
                 'refbasei': [int(v) for v in bmi['refbasei']], 'mod_pred': mod_pred.tolist()})
    return {k: detect_oracle.sum_handler_oracle(k[0], k[1], 'C', v) for k, v in by.items()}


def test_two_rank_streaming_engine_equals_single_process_and_oracle(tmp_path):
    from deepmod_amd import predstore, synth, synth_reads
    # two contigs, both container formats, one read too short to be called (error channel)
    files = synth_reads.write_synthetic_run(str(tmp_path / 'a'), n_reads=7, reads_per_file=2, genome_len=6000, seed=3,
                                            chrom='chrA', min_len=120, max_len=400)
    files_b = synth_reads.write_synthetic_run(str(tmp_path / 'b'), n_reads=5, reads_per_file=2, genome_len=5000, seed=4,
                                              chrom='chrB', min_len=120, max_len=400)
    packed = []
    for i, f in enumerate(files_b):      # re-written in the packed format, contig length in the metadata
        p = str(tmp_path / 'b' / ('packed_%d%s' % (i, predstore.CONTAINER_SUFFIX)))
        predstore.save_packed_container(p, predstore.load_feature_container(f), {'chrB': 5000})
        os.remove(f)
        packed.append(p)
    short = synth_reads.write_synthetic_run(str(tmp_path / 'c'), n_reads=1, reads_per_file=1, genome_len=3000, seed=5,
                                            chrom='chrA', min_len=30, max_len=40)
    all_files = files + packed + short
    w = synth.synthetic_weights(26, 4.0)
    want = _oracle_beds(files + packed, w)
    assert len(want) == 4 and all(len(b) > 100 for b in want.values())

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    outs = {}
    # world 2 and 3 (3: slices of unequal length, 5000 and 6000 positions do not divide) through the reduce-scatter merge
    # (every rank formats its slice of every contig, rank 0 joins the parts), world 2 also through the reduce-to-rank-0 merge
    for tag, world, merge in (("1", 1, "scatter"), ("2", 2, "scatter"), ("3", 3, "scatter"), ("2r", 2, "reduce")):
        out = tmp_path / ("out" + tag)
        cfg = tmp_path / ("cfg%s.json" % tag)
        cfg.write_text(json.dumps({"files": all_files, "out": str(out), "seed": 26, "scale": 4.0, "merge": merge}))
        env = dict(os.environ, DM_ROOT=ROOT, DM_CFG=str(cfg))
        if world == 1:
            cmd = [sys.executable, str(script)]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                   "--master-port", str(29517 + world), str(script)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        outs[tag] = {k: open('%s/mod_pos.%s%s.C.bed' % (out, k[0], k[1]), 'rb').read() for k in want}
        assert not [f for f in os.listdir(out) if '.part' in f]        # the parts were joined and removed
        stats = [json.load(open(out / ("stats.%d.json" % r))) for r in range(world)]
        assert sum(s["reads"] for s in stats) == 12
        assert sum(len(v) for s in stats for v in s["errors"].get("Less Event", [])) > 0
        if world > 1:
            assert all(s["reads"] > 0 for s in stats)          # every rank did part of the work
    for k in want:
        for tag in outs:
            assert outs[tag][k] == want[k], (tag, k)


def test_eight_ranks_six_of_them_idle_uneven_contigs_and_a_contig_only_one_rank_saw(tmp_path):
    """The driver's real world size: 8 ranks over gloo, 2 work items - six ranks never see a read and still take part
    in every agreement and every merge (zeros); contig lengths 6001 and 5003 do not divide by 8 (the last slice is short); chrB reaches
    one rank only.  The joined BED parts equal the oracle pipeline's bytes."""
    from deepmod_amd import synth, synth_reads
    fa = synth_reads.write_synthetic_run(str(tmp_path / 'a'), n_reads=6, reads_per_file=2, genome_len=6001, seed=13, chrom='chrA', min_len=120, max_len=400)
    fb = synth_reads.write_synthetic_run(str(tmp_path / 'b'), n_reads=2, reads_per_file=2, genome_len=5003, seed=14, chrom='chrB', min_len=120, max_len=400)
    all_files = fa + fb                                   # items: [a0, a1] -> rank 0, [a2, b0] -> rank 1; ranks 2..7: nothing
    assert len(all_files) == 4
    w = synth.synthetic_weights(26, 4.0)
    want = _oracle_beds(all_files, w)
    assert len(want) == 4
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "out8"
    cfg = tmp_path / "cfg8.json"
    cfg.write_text(json.dumps({"files": all_files, "out": str(out), "seed": 26, "scale": 4.0, "merge": "scatter"}))
    env = dict(os.environ, DM_ROOT=ROOT, DM_CFG=str(cfg), OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                          "--master-port", "29531", str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    stats = [json.load(open(out / ("stats.%d.json" % r))) for r in range(8)]
    assert [s["reads"] > 0 for s in stats] == [True, True] + [False] * 6
    assert sum(s["reads"] for s in stats) == 8
    for k in want:
        assert open('%s/mod_pos.%s%s.C.bed' % (out, k[0], k[1]), 'rb').read() == want[k], k
    assert not [f for f in os.listdir(out) if '.part' in f]


def test_shard_partitions_everything():
    from deepmod_amd import comm
    items = list(range(23))
    for world in (1, 2, 3, 8):
        parts = [comm.shard(items, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == items
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_file_rendezvous_abort_ends_a_wait_at_once(tmp_path):
    """A rank that fails writes the abort file; the ranks waiting for it raise within milliseconds instead of the timeout."""
    import threading
    import time
    import pytest
    from deepmod_amd import comm
    a = comm.FileRendezvous(str(tmp_path), 0, 2, timeout=60)
    b = comm.FileRendezvous(str(tmp_path), 1, 2, timeout=60)
    threading.Timer(0.2, lambda: b.abort('RuntimeError: no such device')).start()
    t0 = time.time()
    with pytest.raises(RuntimeError, match='aborted by rank 1'):
        a.get('never_written')
    assert time.time() - t0 < 5.0


def test_file_rendezvous_round_trip(tmp_path):
    from deepmod_amd import comm
    a = comm.FileRendezvous(str(tmp_path), 0, 2, timeout=5)
    b = comm.FileRendezvous(str(tmp_path), 1, 2, timeout=5)
    a.broadcast('id', b'x' * 128)
    assert b.broadcast('id', None) == b'x' * 128
    a.put('keys.0', json.dumps({"k": 1}).encode())
    b.put('keys.1', json.dumps({"k": 2}).encode())
    assert a.all_gather_json('keys', {"k": 1}) == [{"k": 1}, {"k": 2}]
