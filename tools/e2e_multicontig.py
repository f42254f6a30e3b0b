"""Dev tool (GPU box): `bin/DeepMod.py detect` on a run over MANY contigs (24 x 10 Mb at 2x by default: 48 contig x strand tables,
~5e8 base-positions) - groups per batch, counters per table, the tables' BED files written side by side.
    python tools/e2e_multicontig.py [feeders] [contigs] [contig length] [coverage]"""
import multiprocessing, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepmod_amd import synth, synth_reads


def _gen_named(args):
    out_dir, contig, length, cov = args
    sub = os.path.join(out_dir, "c%02d" % contig)           # one sub-folder per contig: the generator names files by number
    return synth_reads.write_synthetic_packed_run(sub, length, cov, 100, seed=100 + contig, chrom="chr%d" % (contig + 1))


if __name__ == "__main__":
    feeders = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n_contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    length = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
    cov = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
    tmp = tempfile.mkdtemp()
    wrk = tmp + "/reads"
    t0 = time.time()
    with multiprocessing.get_context("spawn").Pool(min(32, n_contigs, len(os.sched_getaffinity(0)))) as pool:
        files = sum(pool.map(_gen_named, [(wrk, c, length, cov) for c in range(n_contigs)]), [])
    print("generated %d feature containers over %d contigs in %.1f s" % (len(files), n_contigs, time.time() - t0), flush=True)
    prefix = tmp + "/model/m"
    os.makedirs(tmp + "/model")
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    for rep in range(2):                                     # the first run warms the page cache
        cmd = [sys.executable, os.path.join(ROOT, "bin", "DeepMod.py"), "detect", "--wrkBase", wrk, "--recursive", "1", "--modfile", prefix, "--outFolder",
               "%s/out%d" % (tmp, rep), "--Base", "C", "--gpus", "1", "--threads", str(feeders), "--FileID", "s"]
        t0 = time.time()
        res = subprocess.run(cmd, capture_output=True, text=True)
        wall = time.time() - t0
        if res.returncode:
            print(res.stdout[-2000:], res.stderr[-3000:])
            sys.exit(1)
    for ln in res.stdout.splitlines():
        if "Streaming detect" in ln or "host stages" in ln or "timeline" in ln:
            print("   ", ln.strip())
    beds = [f for f in os.listdir("%s/out1/s" % tmp) if f.endswith(".bed")]
    print("%d feeder processes: whole command %.2f s, %d BED files, %.2f GB of text" % (feeders, wall, len(beds), sum(os.path.getsize("%s/out1/s/%s" % (tmp, f)) for f in beds) / 1e9))
