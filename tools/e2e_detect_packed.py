"""Dev tool (GPU box): end-to-end rate of `bin/DeepMod.py detect` on BASELINE configs[2] (4.64 Mb genome at 30x, feature
containers) as a function of the number of feeder processes - the command, its statistics lines and the wall clock.
    python tools/e2e_detect_packed.py [feeders,feeders,...] [coverage] [extra CLI args...]"""
import multiprocessing, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepmod_amd import synth, synth_reads

GENOME_LEN = int(os.environ.get("DM_E2E_GENOME", 4_641_652))      # DM_E2E_GENOME=248956422: a chr1-sized contig (BASELINE configs[3], one rank's view)
CHROM = "NC_000913.3" if GENOME_LEN == 4_641_652 else "chr1"
READS_PER_FILE = 100


def _gen(args):
    out_dir, first, n, cov = args
    return synth_reads.write_synthetic_packed_run(out_dir, GENOME_LEN, cov, READS_PER_FILE, seed=1, chrom=CHROM, first_file=first, n_files=n)


if __name__ == "__main__":
    feeders = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,4,8").split(",")]
    cov = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    tmp = tempfile.mkdtemp()
    wrk = tmp + "/reads"
    total_files = int(np.ceil(cov * GENOME_LEN / 6000.0 / READS_PER_FILE))
    ncpu = min(32, len(os.sched_getaffinity(0)))
    chunk = int(np.ceil(total_files / ncpu))
    t0 = time.time()
    with multiprocessing.get_context("spawn").Pool(ncpu) as pool:
        files = sum(pool.map(_gen, [(wrk, i, chunk, cov) for i in range(0, total_files, chunk)]), [])
    print("generated %d feature containers in %.1f s" % (len(files), time.time() - t0), flush=True)
    prefix = tmp + "/model/m"
    os.makedirs(tmp + "/model")
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    t0 = time.time()
    subprocess.run([sys.executable, os.path.join(ROOT, "bin", "DeepMod.py"), "detect", "--help"], capture_output=True)
    print("interpreter + argument parser alone: %.2f s" % (time.time() - t0))
    for rep, nf in enumerate([feeders[0]] + feeders):          # the first run also warms the page cache: not reported
        cmd = [sys.executable, os.path.join(ROOT, "bin", "DeepMod.py"), "detect", "--wrkBase", wrk, "--modfile", prefix, "--outFolder",
               "%s/out%d" % (tmp, rep), "--Base", "C", "--gpus", "1", "--threads", str(nf), "--FileID", "s"] + sys.argv[3:]
        if os.environ.get("DM_E2E_PROF") and rep:                # kernel trace of the rank process(es): where the GPU time goes
            cmd = ["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--stats", "--output-format", "csv", "-d", "%s/f%d" % (os.environ["DM_E2E_PROF"], nf), "--"] + cmd
        t0 = time.time()
        res = subprocess.run(cmd, capture_output=True, text=True)
        wall = time.time() - t0
        if res.returncode:
            print(res.stdout[-2000:], res.stderr[-3000:])
            sys.exit(1)
        if rep == 0:
            continue
        for ln in res.stdout.splitlines():
            if "Streaming detect" in ln or "host stages" in ln or "timeline" in ln or "consuming time" in ln:
                print("   ", ln.strip())
        print("%d feeder processes: whole command %.2f s" % (nf, wall), flush=True)
