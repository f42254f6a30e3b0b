"""Dev tool (GPU box): the raw-container command (`bin/DeepMod.py detect` from signal samples + event tables + alignments to BED) measured and profiled.

    python tools/raw_profile.py [n_reads] [repeat] [feeders,feeders,...] [--rocprof] [--servers=n,n,...]

  * n_reads synthetic raw reads are generated once (10 per container, the shape of tools/e2e_detect_raw.py); `repeat` > 1 multiplies the run WITHOUT
    multiplying the disk: the work folder of the measured command holds `repeat` symbolic links per container (and side-car .sam), so the run is
    repeat x as long - long enough that the 0.5 s of start-up (model load, process spawn) does not set the rate - and the feeders still read every
    byte through the page cache (a real run's containers come from the page cache or a parallel file system at similar rates).
  * per number of feeder processes: the command's own report (rate, host stages, waits, timeline) and the steady-state rate between the first and
    the last batch.
  * --rocprof: the same command once more under `rocprofv3 --kernel-trace --memory-copy-trace --stats` (the GPU process is a child: per-process
    files) -> per kernel: launches, total and average time, and achieved GB/s against its ALGORITHMIC bytes (below) and the 8 TB/s HBM peak;
    then two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) for the HBM traffic of the signal kernels.
    Written to gpurun_out/r06/raw/ (copy what is to be judged into profiles/r06/raw/).

Algorithmic bytes: histogram 2 B/sample; event statistics 2 B/sample + 16 B/event in, 12 B/event out; row assembly 13 B/row in,
28 B/row out; summary 10 B/entry; order statistics and value table: the read's own value range (data dependent, not priced)."""
import csv
import glob
import json
import multiprocessing
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmod_amd import synth, synth_reads          # noqa: E402

GENOME = 500000


def _gen(args):
    out, part, n = args
    return synth_reads.write_synthetic_raw_run(out, n_reads=n, reads_per_file=10, genome_len=GENOME, seed=3, chrom="chrS", part=part,
                                               min_len=2000, max_len=8000)[0]


def run_detect(wrk, prefix, out, threads, extra_env=None, wrapper=None):
    cmd = [sys.executable, os.path.join(ROOT, "bin", "DeepMod.py"), "detect", "--wrkBase", wrk, "--Ref", wrk + "/genome.fa", "--modfile", prefix,
           "--outFolder", out, "--Base", "C", "--gpus", "1", "--threads", str(threads), "--FileID", "raw", "--alignStr", "minimap2"]
    env = dict(os.environ)
    env.update(extra_env or {})
    t0 = time.time()
    res = subprocess.run((wrapper or []) + cmd, capture_output=True, text=True, env=env)
    wall = time.time() - t0
    if res.returncode:
        sys.stderr.write(res.stdout[-3000:] + res.stderr[-3000:])
        raise SystemExit(1)
    return res.stdout, wall


def parse_report(so):
    rep = {}
    m = re.search(r'Streaming detect: (\d+) reads, (\d+) base-positions .* in ([0-9.]+) s = ([0-9.e+]+) base-positions/s', so)
    if m:
        rep.update(reads=int(m.group(1)), base_positions=int(m.group(2)), detect_seconds=float(m.group(3)), base_positions_per_s=float(m.group(4)))
    m = re.search(r'detect wall ([0-9.]+) s, waiting for feeders ([0-9.]+) s', so)
    if m:
        rep.update(detect_wall_s=float(m.group(1)), waiting_for_feeders_s=float(m.group(2)))
    m = re.search(r'waiting for the device ([0-9.]+) s', so)
    if m:
        rep['waiting_for_device_s'] = float(m.group(1))
    m = re.search(r'first batch from a feeder ([0-9.]+), last batch ([0-9.]+), device drained ([0-9.]+)', so)
    if m and 'base_positions' in rep:
        first, last, drained = (float(m.group(i)) for i in (1, 2, 3))
        rep.update(first_batch_s=first, device_drained_s=drained, steady_base_positions_per_s=rep['base_positions'] / max(drained - first, 1e-9))
    m = re.search(r'signal server ([0-9.]+) s \(([0-9.]+) s copying .*?, ([0-9.]+) s inside the signal call\) for (\d+) requests', so)
    if m:
        rep.update(signal_server_s=float(m.group(1)), signal_copy_s=float(m.group(2)), signal_call_s=float(m.group(3)), signal_requests=int(m.group(4)))
    m = re.search(r'signal stage: (\d+) samples, (\d+) merged events', so)
    if m:
        rep.update(samples=int(m.group(1)), merged_events=int(m.group(2)))
    rep['lines'] = [ln.strip() for ln in so.splitlines() if any(k in ln for k in ('Streaming detect', 'host stages', 'timeline', 'windows run through', 'signal stage:'))]
    return rep


def _short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*', '', name).split('::')[-1]
    return re.sub(r'<.*', '', name).replace('void ', '').strip()


def signal_only(wrk, n_files, reps):
    """--signal-only leg (run under rocprofv3 by main()): the signal stage of ONE worker batch - the request a feeder posts for its first n_files
    containers - through dm_signal_event_stats_device `reps` times with nothing else on the device: what the four kernels take when they do not
    wait for the classifier's compute units."""
    import numpy as np
    from deepmod_amd import _lib, npzmap, signal as dmsignal
    from deepmod_amd.model import DeviceArray
    lib = _lib.load()
    files = sorted(glob.glob(wrk + '/*.dmraw.npz'))[:n_files]
    raws, raw_off, ev_off, starts, lens = [], [0], [0], [], []
    for f in files:
        z = npzmap.load(f)
        n = len(z['raw_off']) - 1
        eo = np.ascontiguousarray(z['ev_off'], np.int64)
        ne = int(eo[-1])
        mev_off = np.empty(n + 1, np.int64)
        ms = np.ascontiguousarray(z['ev_model_state'])
        m_start, m_len, m_base = np.empty(ne, np.uint64), np.empty(ne, np.uint64), np.empty(ne, 'S1')
        got = lib.dm_events_merge(n, ne, eo.ctypes.data, None, None, np.ascontiguousarray(z['ev_start'], np.uint64).ctypes.data,
                                  np.ascontiguousarray(z['ev_length'], np.uint64).ctypes.data, ms.ctypes.data, ms.dtype.itemsize // 4,
                                  np.ascontiguousarray(z['ev_move'], np.int64).ctypes.data, mev_off.ctypes.data, None, None, m_start.ctypes.data,
                                  m_len.ctypes.data, m_base.ctypes.data)
        assert got > 0, _lib.last_error()
        ro = np.asarray(z['raw_off'], np.int64)
        raws.append(np.asarray(z['raw'][:int(ro[-1])]))
        raw_off.extend((raw_off[-1] + ro[1:]).tolist())
        ev_off.extend((ev_off[-1] + mev_off[1:]).tolist())
        starts.append(m_start[:got]); lens.append(m_len[:got])
    raw, st, ln = np.concatenate(raws), np.concatenate(starts), np.concatenate(lens)
    raw_off, ev_off = np.array(raw_off, np.int64), np.array(ev_off, np.int64)
    nz = dmsignal.SignalNormalizer(0)
    blk = DeviceArray((len(st), 3), np.float32, 0)
    nz.event_stats_device(raw, raw_off, st, ln, ev_off, blk.ptr)
    t0 = time.time()
    for _ in range(reps):
        nz.event_stats_device(raw, raw_off, st, ln, ev_off, blk.ptr)
    dt = (time.time() - t0) / reps
    print("signal-only: %d reads, %d samples, %d merged events per request: %.3f ms per call (pageable host arrays) = %.3g samples/s"
          % (len(raw_off) - 1, len(raw), len(st), dt * 1e3, len(raw) / dt))
    print("SIGNAL_ONLY_WORK %d %d %d %d" % (len(raw_off) - 1, len(raw), len(st), reps + 1))


def kernel_table(prof_dir, work):
    """Sum the per-process kernel traces; attach algorithmic bytes and GB/s."""
    per = {}
    for f in glob.glob(prof_dir + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d = per.setdefault(r["Kernel_Name"], [0, 0])
            d[0] += 1
            d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    copies = {}
    for f in glob.glob(prof_dir + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d = copies.setdefault(r.get("Direction", "?"), [0, 0])
            d[0] += 1
            d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    S, E, NR, R, W = work['samples'], work['merged_events'], work['reads'], work['rows'], work['classified']
    # (order statistics and value table work on each read's own value range since round 6 - a few thousand of the 65,536 bins, data dependent: no
    #  fixed algorithmic byte count; their time and measured HBM traffic are reported as they are)
    alg = {'signal_hist_batch_kernel': 2 * S,
           'event_ev3_batch_kernel': 2 * S + 28 * E, 'event_stats_batch_kernel': 2 * S + 24 * E, 'rows_assemble_kernel': 41 * R,
           'summary_add_kernel': 10 * (W + 0.3 * W), 'head_finish_kernel': 25 * W}
    rows = []
    for name, (calls, ns) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        short = _short(name)
        b = next((v for k, v in alg.items() if k in name), None)
        rows.append({"kernel": short, "launches": calls, "total_ms": round(ns / 1e6, 3), "avg_us": round(ns / 1e3 / calls, 2),
                     "algorithmic_GB": None if b is None else round(b / 1e9, 3),
                     "achieved_GB_per_s": None if b is None else round(b / max(ns, 1), 1), "frac_of_8TBps": None if b is None else round(b / max(ns, 1) / 8000.0, 4)})
    return rows, {k: {"copies": v[0], "total_ms": round(v[1] / 1e6, 3)} for k, v in copies.items()}


def pmc_bytes(prof_dir, counter):
    """per kernel: sum of a TCC counter over the run's launches (KB as rocprofv3 reports FETCH_SIZE / WRITE_SIZE)"""
    per = {}
    for f in glob.glob(prof_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            short = _short(r["Kernel_Name"])
            per[short] = per.get(short, 0.0) + float(r["Counter_Value"])
    return per


def main():
    if '--signal-only' in sys.argv:
        i = sys.argv.index('--signal-only')
        return signal_only(sys.argv[i + 1], int(sys.argv[i + 2]), int(sys.argv[i + 3]))
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n_reads = int(args[0]) if len(args) > 0 else 20000
    repeat = int(args[1]) if len(args) > 1 else 1
    feeders = [int(v) for v in args[2].split(",")] if len(args) > 2 else [4]
    rocprof = '--rocprof' in sys.argv
    ncpu = min(32, len(os.sched_getaffinity(0)))
    tmp = tempfile.mkdtemp()
    src = tmp + "/src"
    per = -(-n_reads // ncpu)
    t0 = time.time()
    with multiprocessing.get_context("spawn").Pool(ncpu) as pool:
        files = sum(pool.map(_gen, [(src, p, per) for p in range(ncpu)]), [])
    size = sum(os.path.getsize(f) for f in files)
    wrk = src
    if repeat > 1:
        wrk = tmp + "/in"
        os.makedirs(wrk)
        os.symlink(src + "/genome.fa", wrk + "/genome.fa")
        for k in range(repeat):
            for f in files:
                b = os.path.basename(f)
                os.symlink(f, "%s/c%02d_%s" % (wrk, k, b))
                sam = f[:-len('.dmraw.npz')] + '.sam'
                os.symlink(sam, "%s/c%02d_%s" % (wrk, k, os.path.basename(sam)))
    report = {"generated": {"containers": len(files), "reads": per * ncpu, "GB": round(size / 1e9, 2), "seconds": round(time.time() - t0, 1)}, "repeat": repeat,
              "runs": []}
    print("generated %d raw containers (%d reads, %.2f GB) in %.1f s; the measured folder holds them %d x" % (len(files), per * ncpu, size / 1e9, time.time() - t0, repeat),
          flush=True)
    prefix = tmp + "/model/m"
    os.makedirs(tmp + "/model")
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    run_detect(wrk, prefix, tmp + "/warm", feeders[0])          # page cache, library load
    last = None
    servers = [None]
    for a in sys.argv[1:]:
        if a.startswith('--servers='):          # signal-stage threads of the GPU process (DEEPMOD_SIGNAL_SERVERS), one run per value and number of feeders
            servers = [int(v) for v in a.split('=')[1].split(',')]
    digests = set()
    for th in feeders:
        for sv in servers:
            out = "%s/out%d_%s" % (tmp, th, sv)
            so, wall = run_detect(wrk, prefix, out, th, None if sv is None else {'DEEPMOD_SIGNAL_SERVERS': str(sv)})
            rep = parse_report(so)
            import hashlib
            rep.update(feeders=th, whole_command_s=round(wall, 2), signal_servers=sv,
                       bed_sha256={os.path.basename(f): hashlib.sha256(open(f, 'rb').read()).hexdigest() for f in sorted(glob.glob(out + '/raw/*.bed'))})
            digests.add(json.dumps(rep['bed_sha256'], sort_keys=True))
            report["runs"].append(rep)
            last = rep
            for ln in rep['lines']:
                print(ln)
            print("%d feeders%s: whole command %.2f s; steady state (first batch -> device drained) %.3g base-positions/s; waiting for feeders %.0f %% of the detect wall"
                  % (th, '' if sv is None else ', %d signal server thread(s)' % sv, wall, rep.get('steady_base_positions_per_s', 0),
                     100 * rep.get('waiting_for_feeders_s', 0) / max(rep.get('detect_wall_s', 1), 1e-9)), flush=True)
    print("BED files of the %d runs: %s" % (len(report["runs"]), "byte-identical (%s)" % ', '.join('%s %s' % (k, v[:16]) for k, v in sorted(last['bed_sha256'].items()))
                                             if len(digests) == 1 else "DIFFERENT between runs: %d distinct digests" % len(digests)), flush=True)
    out_dir = os.path.join(ROOT, 'gpurun_out', 'r06', 'raw')
    os.makedirs(out_dir, exist_ok=True)
    if rocprof:
        th = feeders[-1]
        os.environ['TMPDIR'] = '/tmp'
        prof = tmp + "/prof_kt"
        so, wall = run_detect(wrk, prefix, tmp + "/outp", th, wrapper=["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--stats", "--output-format", "csv", "-d", prof, "--"])
        rep = parse_report(so)
        m = re.search(r'windows run through the classifier: (\d+) of', so)
        work = {"samples": rep.get('samples', 0), "merged_events": rep.get('merged_events', 0), "reads": rep.get('reads', 0), "rows": rep.get('base_positions', 0) + 200 * rep.get('reads', 0),
                "classified": int(m.group(1)) if m else 0}
        rows, copies = kernel_table(prof, work)
        traffic = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            pdir = "%s/prof_%s" % (tmp, counter)
            try:
                run_detect(wrk, prefix, tmp + "/outc_" + counter, th, wrapper=["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", pdir, "--"])
                traffic[counter] = pmc_bytes(pdir, counter)
            except SystemExit:
                traffic[counter] = None
        for r in rows:
            f, w = (traffic.get("FETCH_SIZE") or {}).get(r['kernel']), (traffic.get("WRITE_SIZE") or {}).get(r['kernel'])
            # FETCH_SIZE of a wide coalesced stream reads half the bytes on gfx950 (MI355X_MICROARCH.md, HBM): doubled, as the guide prescribes
            r['hbm_traffic_GB'] = None if f is None or w is None else round((2 * f + w) * 1024 / 1e9, 3)
        # the signal stage of one worker batch alone on the device (no classifier holding the compute units)
        sdir = tmp + "/prof_signal"
        res = subprocess.run(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--stats", "--output-format", "csv", "-d", sdir, "--", sys.executable, os.path.abspath(__file__),
                              "--signal-only", src, "22", "20"], capture_output=True, text=True)
        alone = None
        m2 = re.search(r'SIGNAL_ONLY_WORK (\d+) (\d+) (\d+) (\d+)', res.stdout)
        if res.returncode == 0 and m2:
            nr, ns_, ne_, calls = (int(v) for v in m2.groups())
            arows, acopies = kernel_table(sdir, {"samples": ns_ * calls, "merged_events": ne_ * calls, "reads": nr * calls, "rows": 0, "classified": 0})
            # timeline of one request, averaged over the calls: uploads (samples, event tables, per-read tables), fills (histograms, range, flag), the four
            # kernels, the few bytes that come back (fallback flags, range flag) - everything on the handle's stream, ONE host wait at the end
            h2d = acopies.get('MEMORY_COPY_HOST_TO_DEVICE', {"copies": 0, "total_ms": 0.0})
            d2h = acopies.get('MEMORY_COPY_DEVICE_TO_HOST', {"copies": 0, "total_ms": 0.0})
            per_call = {"upload_ms": round(h2d["total_ms"] / calls, 4), "uploads": h2d["copies"] // calls,
                        "upload_bytes": 2 * ns_ + 16 * ne_ + 40 * nr, "download_ms": round(d2h["total_ms"] / calls, 4), "downloads": d2h["copies"] // calls}
            for r in arows:
                per_call[r['kernel'] + "_ms"] = round(r['total_ms'] / calls, 4)
            alone = {"request": {"reads": nr, "samples": ns_, "merged_events": ne_, "calls": calls}, "kernels": arows, "memory_copies": acopies, "timeline_of_one_request_ms": per_call,
                     "line": [ln for ln in res.stdout.splitlines() if ln.startswith('signal-only')]}
            print("signal stage alone on the device (one worker batch, %d calls):" % calls)
            for r in arows:
                print("%-34s %8d %10.2f %9.1f %9s %9s %8s" % (r['kernel'][:34], r['launches'], r['total_ms'], r['avg_us'], r['algorithmic_GB'], r['achieved_GB_per_s'], r['frac_of_8TBps']))
            print("timeline of one request (ms, averaged):", json.dumps(per_call))
            for ln in alone["line"]:
                print(ln)
        else:
            sys.stderr.write(res.stdout[-1500:] + res.stderr[-1500:])
        report["signal_stage_alone"] = alone
        report["rocprof"] = {"feeders": th, "under_profiler": {k: rep.get(k) for k in ('base_positions_per_s', 'steady_base_positions_per_s', 'detect_wall_s')}, "work": work,
                             "kernels": rows, "memory_copies": copies}
        with open(out_dir + '/kernel_stats.csv', 'w') as fh:
            wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            wr.writeheader()
            wr.writerows(rows)
        print("%-34s %8s %10s %9s %9s %9s %8s %9s" % ("kernel", "launches", "total ms", "avg us", "alg GB", "GB/s", "of 8TB/s", "HBM GB"))
        for r in rows:
            print("%-34s %8d %10.2f %9.1f %9s %9s %8s %9s" % (r['kernel'][:34], r['launches'], r['total_ms'], r['avg_us'], r['algorithmic_GB'], r['achieved_GB_per_s'],
                                                          r['frac_of_8TBps'], r['hbm_traffic_GB']))
        print("memory copies:", copies)
    json.dump(report, open(out_dir + '/summary.json', 'w'), indent=1)


if __name__ == "__main__":
    main()
