"""One rank of tests/test_gpu_multirank.py::test_collectives_of_the_c_abi_with_several_ranks: python rank_worker.py <dir> <rank> <world> <length>.
Every rank adds its own seeded rows to a summary on device 0, then runs every collective form of the C ABI and writes what it received."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepmod_amd import comm, summary          # noqa: E402


def rows_of(rank: int, length: int):
    rng = np.random.default_rng(1000 + rank)
    n = 20000 + 3000 * rank
    return rng.integers(0, length, n).astype(np.int64), rng.integers(0, 8, n).astype(np.uint8)


def main():
    d, rank, world, length = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rdv = comm.FileRendezvous(d, rank, world, timeout=120.0)
    c = comm.Communicator.from_rendezvous(0, rdv)
    out = {"rank": rank, "max": c.max(10.0 + rank), "size": c.stats()["rccl_nranks"]}
    c.barrier()
    pos, flags = rows_of(rank, length)

    def fresh():
        s = summary.PositionSummary(length, 0)
        s.add(pos, flags)
        return s

    s = fresh()                                     # reduce-scatter: each rank's slice
    first, count = s.reduce_scatter(c)
    np.savez(os.path.join(d, "slice.%d.npz" % rank), first=first, count=count, **dict(zip(("touch", "cov", "mod"), s.fetch_slice())))
    s.close()
    s = fresh()                                     # reduce to the last rank
    s.reduce(c, root=world - 1)
    if rank == world - 1:
        np.savez(os.path.join(d, "root.npz"), **dict(zip(("touch", "cov", "mod"), s.fetch())))
    s.close()
    s = fresh()                                     # all-reduce
    s.reduce(c, root=-1)
    np.savez(os.path.join(d, "all.%d.npz" % rank), **dict(zip(("touch", "cov", "mod"), s.fetch())))
    s.close()
    out["stats"] = c.stats()
    c.close()
    with open(os.path.join(d, "out.%d.json" % rank), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()
