"""Dev tool (GPU box): the RAW-container command against the oracle chain on every read, at a size the unit test cannot afford.

`tests/test_gpu_e2e.py::test_detect_cli_on_raw_containers_matches_oracle_pipeline` holds 18 reads to the oracle byte for byte.  This runs n_reads (default
3,000: 1.5e7 base-positions, ~3.7e6 classified windows) synthetic raw reads - int16 samples, basecaller events, SAM records with clips, insertions, deletions,
both strands - through `bin/DeepMod.py detect` (streaming, event statistics resident on the device; four feeders) and through the oracle chain of
tests/oracle_pipeline.py (numpy signal normalisation and event statistics, Python alignment walk, loop-level get_Feature / mPredict1 / sum_handler, C
classifier), one process per container, and compares the BED files line by line: lines that differ, and whether each is explained by windows the oracle
itself puts within 1e-4 of a tie at that position.

    python tools/raw_vs_oracle.py [n_reads] > profiles/r06/raw_vs_oracle.json

TEST INFRASTRUCTURE: imports oracle/ and tests/oracle_pipeline.py; never part of the product."""
import json
import multiprocessing
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GENOME = 500000


def _gen(args):
    from deepmod_amd import synth_reads
    out, part, n = args
    return synth_reads.write_synthetic_raw_run(out, n_reads=n, reads_per_file=10, genome_len=GENOME, seed=3, chrom="chrS", part=part, min_len=2000, max_len=8000)[0]


def _oracle(args):
    from deepmod_amd import readmap, synth
    from oracle_pipeline import oracle_raw_container
    path, fasta = args
    genome = readmap.read_fasta(fasta)['chrS']
    w = synth.synthetic_weights(seed=26, scale=4.0)
    return oracle_raw_container(path, genome, w, nthreads=1)


def parse_bed(data):
    rows = {}
    for ln in data.decode().splitlines():
        f = ln.split()
        rows[int(f[1])] = (int(f[9]), int(f[11]), ln)
    return rows


def main():
    from deepmod_amd import synth
    from oracle import detect_oracle
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    ncpu = min(32, len(os.sched_getaffinity(0)))
    tmp = tempfile.mkdtemp()
    wrk = tmp + "/in"
    per = -(-n_reads // ncpu)
    ctx = multiprocessing.get_context("spawn")
    with ctx.Pool(ncpu) as pool:
        files = sum(pool.map(_gen, [(wrk, p, per) for p in range(ncpu)]), [])
    fasta = wrk + "/genome.fa"
    prefix = tmp + "/model/m"
    os.makedirs(tmp + "/model")
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    out = tmp + "/out"
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "DeepMod.py"), "detect", "--wrkBase", wrk, "--Ref", fasta, "--modfile", prefix, "--outFolder", out,
                          "--Base", "C", "--gpus", "1", "--threads", "4", "--FileID", "raw", "--alignStr", "minimap2"], capture_output=True, text=True)
    wall = time.time() - t0
    if res.returncode:
        sys.stderr.write(res.stdout[-2000:] + res.stderr[-3000:])
        raise SystemExit(1)
    resident = [ln.strip() for ln in res.stdout.splitlines() if 'Streaming detect' in ln or 'resident on the device' in ln]
    t0 = time.time()
    from oracle import oracle_np
    nthr = max(1, oracle_np.usable_cores())
    with ctx.Pool(min(nthr, 16)) as pool:          # (each worker's C classifier is itself OpenMP-parallel; the Python loops around it are what is spread out)
        parts = pool.map(_oracle, [(f, fasta) for f in files], chunksize=4)
    t_oracle = time.time() - t0
    by, ties, reads, margin = {'+': [], '-': []}, {'+': {}, '-': {}}, 0, 1.0
    for b, n, m, t in parts:
        reads += n
        margin = min(margin, m)
        for s in '+-':
            by[s].extend(b[s])
            for p, c in t[s].items():
                ties[s][p] = ties[s].get(p, 0) + c
    rep = {"config": "%d synthetic raw reads (10 per container, %d-base genome) -> bin/DeepMod.py detect --threads 4 (streaming, statistics resident on the device) vs the oracle "
                     "chain on every read" % (n_reads, GENOME), "reads_oracle": reads, "oracle_seconds": round(t_oracle, 1), "command_seconds": round(wall, 2),
           "oracle_smallest_margin": margin, "oracle_windows_within_1e-4_of_a_tie": sum(sum(t.values()) for t in ties.values()), "command": resident,
           "bed_lines": 0, "bed_lines_differing_from_oracle": 0, "differing_lines_explained_by_near_tie_windows": 0, "unattributed_lines": 0, "line_set_equal": True,
           "coverage_column_differs": 0, "examples": []}
    for s in '+-':
        want = parse_bed(detect_oracle.sum_handler_oracle('chrS', s, 'C', by[s]))
        path = '%s/raw/mod_pos.chrS%s.C.bed' % (out, s)
        got = parse_bed(open(path, 'rb').read()) if os.path.exists(path) else {}
        rep["bed_lines"] += len(got)
        if got.keys() != want.keys():
            rep["line_set_equal"] = False
            continue
        for pos in got:
            if got[pos][2] != want[pos][2]:
                rep["bed_lines_differing_from_oracle"] += 1
                cov_same = got[pos][0] == want[pos][0]
                rep["coverage_column_differs"] += 0 if cov_same else 1
                if cov_same and abs(got[pos][1] - want[pos][1]) <= ties[s].get(pos, 0):
                    rep["differing_lines_explained_by_near_tie_windows"] += 1
                else:
                    rep["unattributed_lines"] += 1
                if len(rep["examples"]) < 6:
                    rep["examples"].append({"strand": s, "pos": pos, "product": got[pos][:2], "oracle": want[pos][:2], "near_tie_windows_at_pos": ties[s].get(pos, 0)})
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
