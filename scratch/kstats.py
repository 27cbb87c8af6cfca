import csv, glob, statistics, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0][-40:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) > 5:
        v = v[5:]
    print("%-42s n=%5d min=%7d med=%8.0f max=%8d sum_ms=%.3f" % (k, len(v), min(v), statistics.median(v), max(v), sum(v) / 1e6))
