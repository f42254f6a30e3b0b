"""Dev tool (GPU box): end-to-end rate of `bin/DeepMod.py detect` from RAW containers (signal + events + alignments: the
reference's FAST5 shape) - signal statistics on the GPU, dm_map_read, get_Feature, classifier, on-device summary, BED.
    python tools/e2e_detect_raw.py [n_reads] [threads,threads,...]"""
import multiprocessing, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmod_amd import synth, synth_reads

GENOME = 500000


def _gen(args):
    out, part, n = args
    return synth_reads.write_synthetic_raw_run(out, n_reads=n, reads_per_file=10, genome_len=GENOME, seed=3, chrom="chrS", part=part,
                                               min_len=2000, max_len=8000)[0]


if __name__ == "__main__":
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    ncpu = min(32, len(os.sched_getaffinity(0)))
    thread_list = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [ncpu]
    tmp = tempfile.mkdtemp()
    wrk = tmp + "/in"
    per = -(-n_reads // ncpu)
    t0 = time.time()
    with multiprocessing.get_context("spawn").Pool(ncpu) as pool:
        files = sum(pool.map(_gen, [(wrk, p, per) for p in range(ncpu)]), [])
    size = sum(os.path.getsize(f) for f in files)
    print("generated %d raw containers (%d reads, %.2f GB) in %.1f s" % (len(files), per * ncpu, size / 1e9, time.time() - t0), flush=True)
    prefix = tmp + "/model/m"
    os.makedirs(tmp + "/model")
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    for threads in thread_list:
        cmd = [sys.executable, os.path.join(ROOT, "bin", "DeepMod.py"), "detect", "--wrkBase", wrk, "--Ref", wrk + "/genome.fa", "--modfile", prefix,
               "--outFolder", "%s/out%d" % (tmp, threads), "--Base", "C", "--gpus", "1", "--threads", str(threads), "--FileID", "raw", "--alignStr", "minimap2"] + sys.argv[3:]
        t0 = time.time()
        res = subprocess.run(cmd, capture_output=True, text=True)
        wall = time.time() - t0
        if res.returncode:
            print(res.stdout[-2000:], res.stderr[-3000:])
            sys.exit(1)
        for ln in res.stdout.splitlines():
            if "Streaming detect" in ln or "host stages" in ln or "timeline" in ln:
                print(ln.strip())
        print("raw containers -> BED: %d feeder threads, whole command %.1f s" % (threads, wall))
