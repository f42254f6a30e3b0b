"""Dev tool (GPU box): average rocprofv3 --pmc counters per kernel from the counter_collection CSVs under a directory."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-34s n=%-5d mean %.6g" % (c, len(v), sum(v) / len(v)))
