"""CPU, world_size 2 over gloo: reads sharded over ranks, per-rank dense counters, one integer
all-reduce, rank 0 writes the BED - identical bytes to the single-process result (the additive
merge of DeepMod_tools/sum_chr_mod.py:47-52 / SURVEY.md 8e)."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import GOLDEN, ROOT

WORKER = r'''
import json, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["DM_ROOT"])
from deepmod_amd import detect, dist as dmdist, predstore, summary
from oracle import oracle_np

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
case = json.load(open(os.environ["DM_CASE"]))[0]
tables = [predstore.make_base_map_info(list(r["refbase"]), list(r["readbase"]), r["refbasei"], None, r["mod_pred"])
          for r in case["reads"]]
length = 1 + max(max(r["refbasei"]) for r in case["reads"])
mine = dmdist.shard(list(range(len(tables))), rank, world)
counts = np.zeros((3, length), np.int32)
for i in mine:   # no GPU here: the oracle's C accumulation stands in for dm_summary_add
    oracle_np.summary_add_c(counts[0], counts[1], counts[2], tables[i]["refbasei"].astype(np.int64),
                            detect.base_flags(tables[i], case["Base"]))
total = dmdist.all_reduce_counts(counts, dist)
if rank == 0:
    bed = summary.bed_lines(case["chr"], case["strand"], case["Base"], total[0], total[1], total[2])
    open(os.environ["DM_OUT"], "wb").write(bed)
    json.dump({"mine": mine, "world": world}, open(os.environ["DM_OUT"] + ".meta", "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_sharded_summary_equals_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "merged.bed"
    env = dict(os.environ, DM_ROOT=ROOT, DM_CASE=os.path.join(GOLDEN, "host_sum_handler.json"), DM_OUT=str(out))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    case = json.load(open(os.path.join(GOLDEN, "host_sum_handler.json")))[0]
    assert out.read_bytes() == case["bed"].encode("ascii")
    meta = json.load(open(str(out) + ".meta"))
    assert meta["world"] == 2 and 0 < len(meta["mine"]) < len(case["reads"])


def test_shard_partitions_everything():
    from deepmod_amd import dist as dmdist
    items = list(range(23))
    for world in (1, 2, 3, 8):
        parts = [dmdist.shard(items, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == items
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
