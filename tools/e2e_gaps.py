import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows))
    big = [k for k in ks if "bilstm" in k[2]]
    print(f, len(ks), "kernels,", len(big), "bilstm")
    t0 = big[0][0]
    busy = sum(e - s for s, e, _ in ks)
    print("first bilstm start -> last end: %.3f s, all kernels busy %.3f s" % ((big[-1][1] - t0) / 1e9, busy / 1e9))
    prev = None
    for s, e, n in big:
        print("  start %.3f dur %.1f ms gap before %.1f ms" % ((s - t0) / 1e9, (e - s) / 1e6, 0 if prev is None else (s - prev) / 1e6))
        prev = e
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), "copies; columns", list(rows[0].keys()) if rows else None)
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    print("copy time total %.3f s" % (tot / 1e9))
