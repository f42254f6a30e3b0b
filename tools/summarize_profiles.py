"""Dev tool: condense gpurun_out/prof_<tag>/ (from tools/profile_round.sh) into profiles/<round>/."""
import collections, csv, glob, json, os, shutil, sys
tag, dest = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs(dest, exist_ok=True)
stats = sorted(glob.glob(src + "/kt/*/*_kernel_stats.csv"), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(dest, "kernel_stats.csv"))
out = {}
for d in sorted(glob.glob(src + "/pmc_*")):
    files = sorted(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime, reverse=True)   # newest run first
    if not files:
        continue
    acc = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(files[0])):
        if "bilstm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in acc.items():
        out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v), "mean_kernel_ns_in_this_pass": sum(dur) / len(dur)}
# the kernel-trace pass: per-launch durations of the classifier kernel.  The first 32 + 1 launches of the bench command are its untimed
# set-up and warm-up launches (bench.py SETUP_LAUNCHES, --warmup 1): the GPU leaves its idle power state during them (max 1.9 ms against
# 1.5 steady).  `steady` = what bench.py's timed region sees, so that roofline.frac can be recomputed from profiles/ alone:
# frac = 8.924e6 FLOP x 65,536 windows / steady_mean_ns / peak.
traces = sorted(glob.glob(src + "/kt/*/*_kernel_trace.csv"), key=os.path.getmtime)
if traces:
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(traces[-1])) if "bilstm" in r["Kernel_Name"]]
    dur = [d for _, d in sorted(rows)]
    if dur:
        steady = dur[33:] if len(dur) > 33 else dur
        med = lambda v: sorted(v)[len(v) // 2]
        out["kernel_trace"] = {"launches": len(dur), "mean_ns_all_launches": sum(dur) / len(dur), "median_ns_all_launches": med(dur),
                               "skipped_setup_and_warmup_launches": len(dur) - len(steady), "steady_launches": len(steady),
                               "steady_mean_ns": sum(steady) / len(steady), "steady_median_ns": med(steady), "steady_min_ns": min(steady), "steady_max_ns": max(steady)}
        out["kernel_ms_rocprof_steady"] = sum(steady) / len(steady) * 1e-6
out["windows_per_launch"] = 65536
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_sha: bench.py marks this summary stale once the kernel sources change)
out["kernel_src_sha"] = bench.kernel_source_sha(tag.split("_", 1)[1] if "_" in tag else "f16x3")
json.dump(out, open(os.path.join(dest, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: (v.get("mean_per_launch", v) if isinstance(v, dict) else v) for k, v in out.items()}, indent=1))
