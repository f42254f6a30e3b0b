"""GPU (-m gpu): the N > 1 code of the product through the C ABI's collective calls on a ONE-GPU box.

RCCL refuses a communicator whose ranks share a device, so until round 5 dm_summary_reduce_scatter / dm_summary_reduce / dm_comm_max_f64 had
only ever run with one rank (and the merge logic above them through gloo, tests/test_stream_gloo.py).  DEEPMOD_RCCL_LIBRARY makes
libdeepmod_hip bind another collective library; tests/shim/shmccl.cpp answers the same entry points over shared memory (test
infrastructure: RCCL's documented semantics, host staging).  With it two, three and eight REAL processes on device 0 run every collective
call the product makes: the slice arithmetic, the order of calls over the ranks, the rank-sliced BED, bench.py's merge.  What stays
unmeasured is RCCL itself over xGMI (needs a multi-GPU node)."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from deepmod_amd import synth, synth_reads
from shim import build as shim_build
from shim import rank_worker

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shim_env(gpu_device):
    env = dict(os.environ, DEEPMOD_RCCL_LIBRARY=shim_build.library())
    env.pop("DM_BENCH_FORCE_DIST", None)
    return env


@pytest.mark.parametrize("world,length", [(2, 100001), (3, 70000), (8, 5)])
def test_collectives_of_the_c_abi_with_several_ranks(tmp_path, shim_env, world, length):
    """Every collective form behind include/deepmod_hip.h with `world` processes: max over ranks, barrier, reduce-scatter (slices of
    ceil(length / world) positions: with length 5 over 8 ranks the last ranks own EMPTY slices), reduce to a root, all-reduce - each equal
    to the sum of the ranks' own counters computed here."""
    d = str(tmp_path)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shim", "rank_worker.py"), d, str(r), str(world), str(length)],
                              env=shim_env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    from deepmod_amd.summary import PositionSummary
    want = np.zeros((3, length), np.int64)
    for r in range(world):                          # the ranks' own counters, accumulated one after the other in this process
        s = PositionSummary(length, 0)
        s.add(*rank_worker.rows_of(r, length))
        want += np.stack(s.fetch()).astype(np.int64)
        s.close()
    chunk = -(-length // world)
    got = np.zeros((3, length), np.int64)
    covered = 0
    for r in range(world):
        z = np.load(os.path.join(d, "slice.%d.npz" % r))
        first, count = int(z["first"]), int(z["count"])
        assert first == min(length, r * chunk) and count == min(length, (r + 1) * chunk) - first
        for k, name in enumerate(("touch", "cov", "mod")):
            got[k, first:first + count] = z[name]
        covered += count
        a = np.load(os.path.join(d, "all.%d.npz" % r))
        assert all(np.array_equal(a[name], want[k]) for k, name in enumerate(("touch", "cov", "mod"))), r
        out = json.load(open(os.path.join(d, "out.%d.json" % r)))
        assert out["max"] == 10.0 + world - 1 and out["size"] == world
        assert out["stats"]["collectives"] == 3 and out["stats"]["bytes"] == 3 * 4 * 3 * length
    assert covered == length and np.array_equal(got, want) and want[0].sum() > 0
    root = np.load(os.path.join(d, "root.npz"))
    assert all(np.array_equal(root[name], want[k]) for k, name in enumerate(("touch", "cov", "mod")))


@pytest.mark.parametrize("world,kind", [(2, 'features'), (3, 'features'), (2, 'raw')])
def test_command_with_several_ranks_writes_the_single_rank_bed(tmp_path, shim_env, world, kind):
    """`DeepMod.py detect --gpus N` with N real GPU processes (all on device 0: DEEPMOD_ONE_DEVICE=1) and their feeders: reads sharded over
    the ranks, one reduce-scatter per contig x strand at the end, every rank formats its slice, rank 0 joins them - the BED files
    are byte for byte those of the one-process run, which tests/test_gpu_e2e.py holds to the oracle pipeline.  'raw': raw-signal containers
    with their SAM records - every rank runs its own signal servers and keeps the event statistics of its feeders' batches on the device."""
    wrk = tmp_path / 'reads'
    more = []
    if kind == 'raw':
        synth_reads.write_synthetic_raw_run(str(wrk), n_reads=36, reads_per_file=3, genome_len=30000, seed=5, chrom='chrM2', min_len=300, max_len=1200)
        more = ['--Ref', str(wrk / 'genome.fa'), '--alignStr', 'minimap2']
    else:
        synth_reads.write_synthetic_run(str(wrk), n_reads=40, reads_per_file=3, genome_len=30000, seed=5, chrom='chrM2', min_len=300, max_len=1200)
    prefix = str(tmp_path / 'model' / 'm')
    os.makedirs(os.path.dirname(prefix))
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    beds = {}
    for name, extra, env in (('one', [], dict(os.environ)), ('many', ['--gpus', str(world)], dict(shim_env, DEEPMOD_ONE_DEVICE='1'))):
        out = str(tmp_path / ('out_' + name))
        res = subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'DeepMod.py'), 'detect', '--wrkBase', str(wrk), '--modfile', prefix, '--outFolder', out,
                              '--FileID', 'run', '--threads', '4', '--Base', 'C'] + more + extra, capture_output=True, text=True, timeout=600, env=env)
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
        assert os.path.exists(out + '/run.done')
        beds[name] = {os.path.basename(f): open(f, 'rb').read() for f in sorted(glob.glob(out + '/run/*.bed'))}
        if name == 'many':
            assert 'ncclCommInitRank' not in res.stderr
        if kind == 'raw':
            assert 'event statistics resident on the device' in res.stdout, res.stdout[-1500:]
    assert len(beds['one']) == 2 and all(len(v) > 2000 for v in beds['one'].values())
    assert beds['many'] == beds['one']


def test_bench_eight_ranks_merge_through_the_collective_calls(shim_env):
    """The driver's N = 8 launch, eight real processes on device 0, with the merge RUN: one communicator over eight ranks, the
    per-position counters reduce-scattered inside the timed region, barrier and max over ranks through the same communicator - one JSON
    line whose slice sums add up to the counters every rank accumulated."""
    env = dict(shim_env, DM_BENCH_ONE_DEVICE="1")
    world = 8
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    mg = out["multi_gpu"]
    assert out["n_gpus"] == world and mg["rccl_nranks"] == world and mg["rccl_error"] in (None, "")
    assert "ncclReduceScatter" in mg["collective"] and mg["collectives"] == 2
    assert [r["rank"] for r in mg["per_rank"]] == list(range(world))
    assert out["summary_check"]["touch"] == sum(r["slice_sums"][0] for r in mg["per_rank"]) > 0
    assert mg["measured_on_hardware_with_more_than_one_rank"] is False and "libshmccl.so (ncclGetVersion" in mg["collective_library"]


def test_a_rank_whose_peer_never_joins_gives_up(tmp_path, shim_env):
    """comm.Communicator's watchdog: ncclCommInitRank blocks until every rank has joined; a process whose peer never arrives ends by itself after
    DEEPMOD_COMM_TIMEOUT seconds - exit code 3, the reason on stderr - instead of waiting inside the collective library (the stand-in waits 60 s)."""
    import time
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from deepmod_amd import comm\n"
            "comm.Communicator(0, comm.rccl_unique_id(), 0, 2)\n"
            "print('created')\n" % ROOT)
    t0 = time.time()
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120, env=dict(shim_env, DEEPMOD_COMM_TIMEOUT='3'))
    assert res.returncode == 3 and time.time() - t0 < 100, (res.returncode, res.stderr[-1500:])
    assert 'has not returned after 3 s' in res.stderr and 'created' not in res.stdout
