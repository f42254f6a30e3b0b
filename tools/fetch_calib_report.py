"""Dev tool (GPU box): known byte counts of tools/ubench/fetch_calib against what rocprofv3's FETCH_SIZE / WRITE_SIZE counted (KB per launch)."""
import collections, csv, glob
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob("/tmp/fc_%s/*/*_counter_collection.csv" % C))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(C, k, "mean per launch %.1f KB = %.0f bytes" % (sum(v) / len(v), sum(v) / len(v) * 1024), v)
